"""Decode the rasterizer's three opaque state buffers into reference-shaped
numpy arrays (inspection, parity tests and benchmark statistics only; the
layout comes from ``ngs_state_layout`` in include/ngs_rasterizer.h)."""
import numpy as np


def _view(buf, off, dtype, count):
    """numpy view of `count` items of `dtype` at byte offset `off` past the 256-aligned base of a CUDA byte tensor."""
    base = buf.data_ptr()
    aligned = (base + 255) // 256 * 256
    start = aligned - base + off
    nbytes = count * np.dtype(dtype).itemsize
    return buf[start:start + nbytes].cpu().numpy().view(dtype)


def decode_state(_C, P, R, W, H, geom, binning, img):
    """Unpack the HIP library's three state buffers into reference-shaped numpy arrays."""
    gl, bl, il = _C.state_layout(P, R, W, H)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    T = gx * gy
    rec = _view(geom, gl["splat_record"], np.float32, 16 * P).reshape(P, 16)
    out = dict(
        means2D=rec[:, 0:2].copy(),
        conic_opacity=rec[:, 2:6].copy(),
        rgb=rec[:, 6:9].copy(),
        depths=_view(geom, gl["depth"], np.float32, P),
        tile_rect=_view(geom, gl["tile_rect"], np.uint32, 2 * P).reshape(P, 2),
        tiles_touched=_view(geom, gl["tiles_touched"], np.uint32, P),
        clamped_bits=_view(geom, gl["clamped_bits"], np.uint8, P),
        point_list=_view(binning, bl["point_list"], np.uint32, R),
        ranges=_view(img, il["ranges"], np.uint32, 2 * T).reshape(T, 2),
    )
    # The sorted 64-bit keys of the reference's BinningState (tile << 32 | depth bits, rasterizer_impl.cu:88-109) are
    # not stored by the library; what they would hold follows from the list, the ranges and the depth array.  An
    # entry no range covers (there is none when the binning is right) keeps the all-ones key, which never matches.
    tile_of = np.full(R, 0xFFFFFFFF, np.uint64)
    rg = out["ranges"].astype(np.int64)
    for t in np.nonzero(rg[:, 1] > rg[:, 0])[0]:
        tile_of[rg[t, 0]:rg[t, 1]] = t
    depth_bits = out["depths"].view(np.uint32)[out["point_list"].astype(np.int64)].astype(np.uint64) if R else \
        np.zeros(0, np.uint64)
    out["keys_sorted"] = (tile_of << np.uint64(32)) | depth_bits
    # per tile in thread order: 2x2 blocks of 8x8 pixels (one per wave), each 2x2 quadrants of 4x4 pixels (one per
    # 16-lane row of the wave), row-major inside a quadrant
    ft = _view(img, il["final_T"], np.float32, T * 256).reshape(gy, gx, 2, 2, 2, 2, 4, 4)
    nc = _view(img, il["n_contrib"], np.uint32, T * 256).reshape(gy, gx, 2, 2, 2, 2, 4, 4)
    order = (0, 2, 4, 6, 1, 3, 5, 7)  # -> [tile row][block row][quadrant row][y][tile column][block column][quadrant column][x]
    out["final_T"] = ft.transpose(order).reshape(gy * 16, gx * 16)[:H, :W]
    out["n_contrib"] = nc.transpose(order).reshape(gy * 16, gx * 16)[:H, :W]
    return out
