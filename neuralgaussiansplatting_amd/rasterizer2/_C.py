"""Binding of the index-map operators to ``libngs_rasterizer.so``: plays the role of the reference's compiled
``rasterizer2._C`` (pybind names at submodules/rasterizer2/ext.cpp:18-19, marshalling at
rasterize_points.cu:183-295).  Same function names, argument order and return tuples; GPU only."""
import ctypes

import torch

from ..diff_gaussian_rasterization._lib import _check, _f32, _ptr, _require_gpu, _stream, lib

NUM_CHANNELS = 3   # submodules/rasterizer2/cuda_rasterizer/config.h:2
NUM_FEATURES = 64  # config.h:3

_configured = False


def _lib():
    global _configured
    L = lib()
    if not _configured:
        L.ngs_idxmap_workspace_bytes.restype = ctypes.c_size_t
        L.ngs_idxmap_forward.restype = ctypes.c_int
        L.ngs_idxmap_backward.restype = ctypes.c_int
        _configured = True
    return L


def get_idxmap2(background, means3D, normal, scales, rotations, viewmatrix, projmatrix, tan_fovx, tan_fovy,
                image_height, image_width, feature_vector, sh, campos, debug, count_entries=True):
    """-> (num_rendered, idxmap int32 [H,W], colmap [H,W,3], depthmap [H,W], featuremap [H,W,64])
    (rasterize_points.cu:183-259).  ``count_entries=False`` skips the host synchronisation that reading the
    entry count needs and returns -1 in its place (the Python layer above never uses the count)."""
    _require_gpu(means3D)
    dev = means3D.device
    P, H, W = means3D.size(0), int(image_height), int(image_width)
    means3D, feature_vector, viewmatrix, projmatrix, campos = (
        _f32(x, dev) for x in (means3D, feature_vector, viewmatrix, projmatrix, campos))
    if P != 0 and (feature_vector.dim() != 2 or feature_vector.size(0) != P or feature_vector.size(1) != NUM_FEATURES):
        raise RuntimeError("feature_vector must have dimensions (num_points, %d)" % NUM_FEATURES)
    L = _lib()
    if P == 0:  # the reference skips the rasterizer and returns its zero-filled tensors (rasterize_points.cu:210-225)
        z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
        return 0, torch.zeros((H, W), dtype=torch.int32, device=dev), z(H, W, NUM_CHANNELS), z(H, W), \
            z(H, W, NUM_FEATURES)
    out_idx = torch.empty((H, W), dtype=torch.int32, device=dev)
    out_color = torch.empty((H, W, NUM_CHANNELS), dtype=torch.float32, device=dev)
    out_depth = torch.empty((H, W), dtype=torch.float32, device=dev)
    out_feature = torch.empty((H, W, NUM_FEATURES), dtype=torch.float32, device=dev)
    workspace = torch.empty(L.ngs_idxmap_workspace_bytes(W, H), dtype=torch.uint8, device=dev)
    rendered = ctypes.c_int(0)
    M = sh.size(1) if sh is not None and sh.numel() != 0 else 0
    with torch.cuda.device(dev):
        _check(L.ngs_idxmap_forward(
            P, int(M), W, H, _ptr(means3D), None, None, None, _ptr(viewmatrix), _ptr(projmatrix),
            ctypes.c_float(tan_fovx), ctypes.c_float(tan_fovy), _ptr(feature_vector), None, _ptr(campos),
            _ptr(out_idx), _ptr(out_color), _ptr(out_depth), _ptr(out_feature), None, _ptr(workspace),
            int(bool(debug)), _stream(dev), ctypes.byref(rendered) if count_entries else None))
    return (rendered.value if count_entries else -1), out_idx, out_color, out_depth, out_feature


def get_idxmap_backward2(means3D, image_height, image_width, campos, out_idx, out_feature, debug, viewmatrix=None,
                         projmatrix=None):
    """-> (dL_dfeature_vector [P,64],) from the gradient wrt the feature map (rasterize_points.cu:261-295).
    With the camera matrices of the forward call (not part of the reference's argument list) the library gathers
    per Gaussian instead of scattering per pixel with atomics: faster and with a fixed addition order."""
    _require_gpu(means3D)
    dev = means3D.device
    P, H, W = means3D.size(0), int(image_height), int(image_width)
    out_feature = _f32(out_feature, dev)
    if out_idx.dtype != torch.int32 or out_idx.device != dev or out_idx.numel() != H * W:
        raise RuntimeError("out_idx must be the int32 [H, W] index map on %s" % dev)
    if out_feature.numel() != H * W * NUM_FEATURES:
        raise RuntimeError("the feature-map gradient must have dimensions (H, W, %d)" % NUM_FEATURES)
    grad = torch.empty((P, NUM_FEATURES), dtype=torch.float32, device=dev)
    if P != 0:
        with torch.cuda.device(dev):
            gather = viewmatrix is not None and projmatrix is not None
            _check(_lib().ngs_idxmap_backward(
                P, W, H, _ptr(_f32(means3D, dev)) if gather else None, None, _ptr(out_idx.contiguous()),
                _ptr(out_feature), _ptr(grad), _ptr(_f32(viewmatrix, dev)) if gather else None,
                _ptr(_f32(projmatrix, dev)) if gather else None, int(bool(debug)), _stream(dev)))
    return (grad,)
