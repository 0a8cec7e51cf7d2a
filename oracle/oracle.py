"""numpy front-end of the CPU checker in ``ngs_oracle.c``.

TEST INFRASTRUCTURE ONLY.  Importable from ``tests/``, ``__graft_entry__.smoke``
and the ``cpu_baseline`` leg of ``bench.py``; the shipped package never imports
it.  ``forward`` / ``backward`` run the stages in the order of the reference's
``Rasterizer::forward`` / ``::backward``
(submodules/diff-gaussian-rasterization/cuda_rasterizer/rasterizer_impl.cu:198-434)
and hand back every intermediate so that the HIP path can be compared stage by
stage.  Parity status of each stage: see the header of ``ngs_oracle.c``.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libngs_oracle.so")
_lib = None


def build(force=False):
    newest = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("ngs_oracle.c", "ngs_oracle_idxmap.c", "ngs_oracle_knn.c"))
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < newest:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libngs_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.ngs_oracle_higher_msb.restype = ctypes.c_uint32
        _lib.ngs_oracle_threads.restype = ctypes.c_int
        _lib.ngs_oracle_preprocess.restype = ctypes.c_int
        _lib.ngs_oracle_render_backward.restype = ctypes.c_int
        _lib.ngs_oracle_render_backward_wide.restype = ctypes.c_int
        _lib.ngs_oracle_idxmap_forward.restype = ctypes.c_longlong
    return _lib


def _p(a):
    if a is None:
        return ctypes.c_void_p(0)
    assert a.flags["C_CONTIGUOUS"]
    return ctypes.c_void_p(a.ctypes.data)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def set_threads(n):
    lib().ngs_oracle_set_threads(int(n))


def threads():
    return int(lib().ngs_oracle_threads())


def higher_msb(n):
    return int(lib().ngs_oracle_higher_msb(ctypes.c_uint32(n)))


def sh_to_rgb(deg, means3D, campos, shs):
    """(rgb [P,3], clamped [P,3]) of the SH->RGB stage alone."""
    means3D, campos, shs = _f32(means3D), _f32(campos), _f32(shs)
    P, M = shs.shape[0], shs.shape[1]
    rgb = np.zeros((P, 3), np.float32)
    cl = np.zeros((P, 3), np.uint8)
    lib().ngs_oracle_sh_to_rgb(P, int(deg), M, _p(means3D), _p(campos), _p(shs), _p(rgb), _p(cl))
    return rgb, cl


def sh_backward(deg, means3D, campos, shs, clamped, dL_dcolor):
    """(dL_dsh [P,M,3], dL_dmeans [P,3]) of the SH stage of the backward pass alone."""
    means3D, campos, shs, dL_dcolor = _f32(means3D), _f32(campos), _f32(shs), _f32(dL_dcolor)
    clamped = np.ascontiguousarray(clamped, dtype=np.uint8)
    P, M = shs.shape[0], shs.shape[1]
    d_sh = np.zeros((P, M, 3), np.float32)
    d_means = np.zeros((P, 3), np.float32)
    lib().ngs_oracle_sh_backward(P, int(deg), M, _p(means3D), _p(campos), _p(shs), _p(clamped), _p(dL_dcolor),
                                 _p(d_means), _p(d_sh))
    return d_sh, d_means


def cov3d(scales, rotations, scale_modifier=1.0):
    """[P,6] upper triangle of the 3D covariance, the covariance stage of the forward pass alone."""
    scales, rotations = _f32(scales), _f32(rotations)
    P = scales.shape[0]
    out = np.zeros((P, 6), np.float32)
    lib().ngs_oracle_cov3d(P, _p(scales), ctypes.c_float(scale_modifier), _p(rotations), _p(out))
    return out


def cov3d_backward(scales, rotations, dL_dcov3D, scale_modifier=1.0):
    """(dL_dscales [P,3], dL_drotations [P,4]) of the covariance stage of the backward pass alone."""
    scales, rotations, dL_dcov3D = _f32(scales), _f32(rotations), _f32(dL_dcov3D)
    P = scales.shape[0]
    d_s = np.zeros((P, 3), np.float32)
    d_r = np.zeros((P, 4), np.float32)
    lib().ngs_oracle_cov3d_backward(P, _p(scales), ctypes.c_float(scale_modifier), _p(rotations), _p(dL_dcov3D),
                                    _p(d_s), _p(d_r))
    return d_s, d_r


def mark_visible(means3D, viewmatrix, projmatrix):
    means3D = _f32(means3D)
    P = means3D.shape[0]
    out = np.zeros(P, dtype=np.uint8)
    lib().ngs_oracle_mark_visible(P, _p(means3D), _p(_f32(viewmatrix)), _p(_f32(projmatrix)), _p(out))
    return out.astype(bool)


def forward(means3D, opacities, viewmatrix, projmatrix, campos, bg, W, H, tanfovx, tanfovy, sh_degree=0, shs=None,
            colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, scale_modifier=1.0,
            prefiltered=False):
    """Returns a dict with every intermediate of the reference's forward pass."""
    L = lib()
    means3D = _f32(means3D)
    P = means3D.shape[0]
    shs, colors_precomp, scales, rotations, cov3D_precomp = map(_f32, (shs, colors_precomp, scales, rotations,
                                                                       cov3D_precomp))
    opacities, viewmatrix, projmatrix, campos, bg = map(_f32, (opacities, viewmatrix, projmatrix, campos, bg))
    M = 0 if shs is None or shs.size == 0 else shs.shape[1]
    if shs is not None and shs.size == 0:
        shs = None
    gx, gy = (W + 15) // 16, (H + 15) // 16
    s = dict(P=P, W=W, H=H, M=M, D=sh_degree, grid=(gx, gy))
    s["radii"] = np.zeros(P, np.int32)
    s["means2D"] = np.zeros((P, 2), np.float32)
    s["depths"] = np.zeros(P, np.float32)
    s["cov3D"] = np.zeros((P, 6), np.float32)
    s["rgb"] = np.zeros((P, 3), np.float32)
    s["conic_opacity"] = np.zeros((P, 4), np.float32)
    s["tiles_touched"] = np.zeros(P, np.uint32)
    s["clamped"] = np.zeros((P, 3), np.uint8)
    s["out_color"] = np.zeros((3, H, W), np.float32)
    s["final_T"] = np.zeros((H, W), np.float32)
    s["n_contrib"] = np.zeros((H, W), np.uint32)
    s["ranges"] = np.zeros((gx * gy, 2), np.uint32)
    s["num_rendered"] = 0
    if P == 0:
        return s
    bad = L.ngs_oracle_preprocess(
        P, int(sh_degree), M, _p(means3D), _p(scales), ctypes.c_float(scale_modifier), _p(rotations), _p(opacities),
        _p(shs), _p(cov3D_precomp), _p(colors_precomp), _p(viewmatrix), _p(projmatrix), _p(campos), W, H,
        ctypes.c_float(tanfovx), ctypes.c_float(tanfovy), int(bool(prefiltered)), _p(s["radii"]), _p(s["means2D"]),
        _p(s["depths"]), _p(s["cov3D"]), _p(s["rgb"]), _p(s["conic_opacity"]), _p(s["tiles_touched"]),
        _p(s["clamped"]))
    if bad:
        raise RuntimeError("Point is filtered although prefiltered is set.")
    s["point_offsets"] = np.zeros(P, np.uint32)
    L.ngs_oracle_scan(P, _p(s["tiles_touched"]), _p(s["point_offsets"]))
    R = int(s["point_offsets"][-1])
    s["num_rendered"] = R
    s["keys_unsorted"] = np.zeros(R, np.uint64)
    s["values_unsorted"] = np.zeros(R, np.uint32)
    L.ngs_oracle_duplicate(P, _p(s["means2D"]), _p(s["depths"]), _p(s["point_offsets"]), _p(s["radii"]), W, H,
                           _p(s["keys_unsorted"]), _p(s["values_unsorted"]))
    s["sort_bits"] = 32 + higher_msb(gx * gy)
    s["keys_sorted"] = np.zeros(R, np.uint64)
    s["point_list"] = np.zeros(R, np.uint32)
    L.ngs_oracle_sort_pairs(ctypes.c_int64(R), _p(s["keys_unsorted"]), _p(s["values_unsorted"]), _p(s["keys_sorted"]),
                            _p(s["point_list"]), s["sort_bits"])
    L.ngs_oracle_tile_ranges(ctypes.c_int64(R), _p(s["keys_sorted"]), gx * gy, _p(s["ranges"]))
    s["features"] = colors_precomp if colors_precomp is not None and colors_precomp.size else s["rgb"]
    L.ngs_oracle_render_forward(W, H, _p(s["ranges"]), _p(s["point_list"]), _p(s["means2D"]), _p(s["features"]),
                                _p(s["conic_opacity"]), _p(bg), _p(s["final_T"]), _p(s["n_contrib"]),
                                _p(s["out_color"]))
    return s


def backward(s, dL_dpix, means3D, viewmatrix, projmatrix, campos, bg, tanfovx, tanfovy, shs=None,
             colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, scale_modifier=1.0, stage1=None,
             wide_sums=False):
    """Gradients in the layout of ``RasterizeGaussiansBackwardCUDA``
    (submodules/diff-gaussian-rasterization/rasterize_points.cu:151-159).
    ``stage1`` = dict(dL_dmeans2D, dL_dconic, dL_dopacity, dL_dcolors) replaces the compositing stage's sums by the
    given ones, so that the per-Gaussian chain after it can be compared on identical inputs.  ``wide_sums`` holds the
    compositing stage's running sums in double precision (same float32 terms): not the reference's arithmetic, a
    yardstick for the rounding of long float32 sums."""
    L = lib()
    P, W, H, M, D = s["P"], s["W"], s["H"], s["M"], s["D"]
    means3D = _f32(means3D)
    shs, colors_precomp, scales, rotations, cov3D_precomp = map(_f32, (shs, colors_precomp, scales, rotations,
                                                                       cov3D_precomp))
    if shs is not None and shs.size == 0:
        shs = None
    viewmatrix, projmatrix, campos, bg, dL_dpix = map(_f32, (viewmatrix, projmatrix, campos, bg, dL_dpix))
    g = dict(
        dL_dmeans3D=np.zeros((P, 3), np.float32), dL_dmeans2D=np.zeros((P, 3), np.float32),
        dL_dcolors=np.zeros((P, 3), np.float32), dL_dconic=np.zeros((P, 2, 2), np.float32),
        dL_dopacity=np.zeros((P, 1), np.float32), dL_dcov3D=np.zeros((P, 6), np.float32),
        dL_dsh=np.zeros((P, M, 3), np.float32), dL_dscales=np.zeros((P, 3), np.float32),
        dL_drotations=np.zeros((P, 4), np.float32))
    if P == 0:
        return g
    if stage1 is not None:
        for k in ("dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors"):
            g[k][...] = np.asarray(stage1[k], np.float32).reshape(g[k].shape)
    else:
        fn = L.ngs_oracle_render_backward_wide if wide_sums else L.ngs_oracle_render_backward
        if fn(P, W, H, _p(s["ranges"]), _p(s["point_list"]), _p(bg), _p(s["means2D"]),
                                        _p(s["conic_opacity"]), _p(s["features"]), _p(s["final_T"]),
                                        _p(s["n_contrib"]), _p(dL_dpix), _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]),
                                        _p(g["dL_dopacity"]), _p(g["dL_dcolors"])) != 0:
            raise MemoryError("oracle: per-thread gradient accumulators could not be allocated")
    fy = np.float32(H) / (np.float32(2.0) * np.float32(tanfovy))
    fx = np.float32(W) / (np.float32(2.0) * np.float32(tanfovx))
    cov3d = cov3D_precomp if cov3D_precomp is not None and cov3D_precomp.size else s["cov3D"]
    L.ngs_oracle_cov2d_backward(P, _p(means3D), _p(s["radii"]), _p(cov3d), ctypes.c_float(fx), ctypes.c_float(fy),
                                ctypes.c_float(tanfovx), ctypes.c_float(tanfovy), _p(viewmatrix), _p(g["dL_dconic"]),
                                _p(g["dL_dmeans3D"]), _p(g["dL_dcov3D"]))
    L.ngs_oracle_preprocess_backward(P, D, M, _p(means3D), _p(s["radii"]), _p(shs), _p(s["clamped"]), _p(scales),
                                     _p(rotations), ctypes.c_float(scale_modifier), _p(projmatrix), _p(campos),
                                     _p(g["dL_dmeans2D"]), _p(g["dL_dmeans3D"]), _p(g["dL_dcolors"]),
                                     _p(g["dL_dcov3D"]), _p(g["dL_dsh"]), _p(g["dL_dscales"]), _p(g["dL_drotations"]))
    return g


IDXMAP_FEATURES = 64  # submodules/rasterizer2/cuda_rasterizer/config.h:3


def idxmap_forward(means3D, viewmatrix, projmatrix, feature_vector, campos, W, H):
    """The reference's index-map path (ngs_oracle_idxmap.c).  Returns a dict with the four maps, the entry count
    and the per-Gaussian intermediates."""
    means3D, viewmatrix, projmatrix, feature_vector, campos = map(_f32, (means3D, viewmatrix, projmatrix,
                                                                        feature_vector, campos))
    P = means3D.shape[0]
    out = dict(idx=np.zeros((H, W), np.int32), color=np.zeros((H, W, 3), np.float32),
               depth=np.zeros((H, W), np.float32), feature=np.zeros((H, W, IDXMAP_FEATURES), np.float32),
               radii=np.zeros(P, np.int32), depths=np.zeros(P, np.float32), pixels=np.zeros((P, 2), np.int32),
               touched=np.zeros(P, np.uint32))
    R = lib().ngs_oracle_idxmap_forward(P, int(W), int(H), _p(means3D), _p(viewmatrix), _p(projmatrix),
                                        _p(feature_vector), _p(campos), _p(out["idx"]), _p(out["color"]),
                                        _p(out["depth"]), _p(out["feature"]), _p(out["radii"]), _p(out["depths"]),
                                        _p(out["pixels"]), _p(out["touched"]))
    if R < 0:
        raise MemoryError("idxmap oracle: allocation failed")
    out["num_rendered"] = int(R)
    return out


def idxmap_backward(P, idx, dL_dfeature):
    """-> dL_dfeature_vector [P, 64]"""
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    dL_dfeature = _f32(dL_dfeature)
    H, W = idx.shape
    out = np.zeros((P, IDXMAP_FEATURES), np.float32)
    lib().ngs_oracle_idxmap_backward(int(P), int(W), int(H), _p(idx), _p(dL_dfeature), _p(out))
    return out


def knn_mean_dist2(points, return_order=False):
    """The reference's simple-knn routine (ngs_oracle_knn.c): [P,3] -> mean squared distance to the three nearest
    other points [P] (and, on request, the Morton order of the points)."""
    points = _f32(points)
    P = points.shape[0]
    out = np.zeros(P, np.float32)
    order = np.zeros(P, np.uint32)
    if lib().ngs_oracle_knn(int(P), _p(points), _p(out), _p(order)) != 0:
        raise MemoryError("knn oracle: allocation failed")
    return (out, order) if return_order else out
