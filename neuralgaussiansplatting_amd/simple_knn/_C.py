"""Binding of the nearest-neighbour routine to ``libngs_rasterizer.so``: plays the role of the reference's compiled
``simple_knn._C`` (pybind name at submodules/simple-knn/ext.cpp:15-17, marshalling at spatial.cu:15-26).  GPU only."""
import ctypes

import torch

from ..diff_gaussian_rasterization._lib import _check, _f32, _ptr, _stream, lib

_configured = False


def _lib():
    global _configured
    L = lib()
    if not _configured:
        L.ngs_knn_workspace_bytes.restype = ctypes.c_size_t
        L.ngs_knn_mean_dist2.restype = ctypes.c_int
        _configured = True
    return L


def distCUDA2(points):
    """[P,3] float tensor on the GPU -> [P] mean squared distance to the three nearest other points
    (spatial.cu:15-26).  The name is the reference's; the device is an MI355X."""
    if not points.is_cuda:
        raise RuntimeError("distCUDA2 runs on the GPU only: points is on %s" % points.device)
    dev = points.device
    if points.dim() != 2 or points.size(1) != 3:
        raise RuntimeError("points must have dimensions (num_points, 3)")
    P = points.size(0)
    points = _f32(points, dev)
    means = torch.empty((P,), dtype=torch.float32, device=dev)
    if P == 0:
        return means
    L = _lib()
    workspace = torch.empty(L.ngs_knn_workspace_bytes(P), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _check(L.ngs_knn_mean_dist2(P, _ptr(points), _ptr(means), _ptr(workspace), 0, _stream(dev)))
    return means
