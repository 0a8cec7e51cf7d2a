// Helpers shared by the forward and backward compositing kernels.
#pragma once
#include "ngs_common.h"

// Conservative screen-space half extents of the region where a splat can pass
// the reference's per-pixel tests (forward.cu:335-345): a pixel contributes
// only if power <= 0 and opacity * exp(power) >= 1/255, i.e. power >= -tau with
// tau = ln(255 * opacity).  {d : 0.5 d^T Q d <= tau} is an ellipse whose
// bounding box has half widths sqrt(2 tau inv(Q)_xx), sqrt(2 tau inv(Q)_yy),
// Q = [[a, b], [b, c]] being the conic.  The box is inflated slightly so that
// rounding can never drop a pair the exact test would keep; a conic that is
// not positive definite (or not finite) gets an infinite box.  Culling with
// this box therefore never changes a result, it only skips pairs that would
// have failed the alpha test.
__device__ __forceinline__ float2 splat_half_extent(float a, float b, float c, float opacity)
{
    const float det = a * c - b * b;
    float hx = __builtin_inff(), hy = __builtin_inff();
    if (det > 0.0f && a > 0.0f && c > 0.0f) {
        const float s = 2.0f * logf(255.0f * opacity) / det;
        // 1 % + half a pixel: far above the float error of the per-pixel power (<= ~0.2 % of the extent even
        // for needle-shaped splats whose quadratic-form terms cancel heavily)
        hx = sqrtf(fmaxf(s * c, 0.0f)) * 1.01f + 0.5f;
        hy = sqrtf(fmaxf(s * a, 0.0f)) * 1.01f + 0.5f;
    }
    return make_float2(hx, hy);
}

// True unless the splat's box provably misses the pixel rectangle [x0,x1] x [y0,y1].
__device__ __forceinline__ bool splat_touches(float sx, float sy, float2 h, float x0, float x1, float y0, float y1)
{
    const bool outside = (sx + h.x < x0) || (sx - h.x > x1) || (sy + h.y < y0) || (sy - h.y > y1);
    return !outside;
}

// Pixel layout of a tile's 256 threads: each wave64 covers one 8x8 pixel block (2x2 blocks per 16x16 tile), lanes
// row-major inside the block.  A square block is touched by fewer splats than a 16x4 strip of the same area, and a
// touching splat covers more of its pixels.  final_T / n_contrib are stored per tile in thread order, so the
// forward and backward kernels (and the state decoder) share this mapping.
#define NGS_WAVE_W 8
#define NGS_WAVE_H 8
__device__ __forceinline__ int tile_local_x(int tid) { return ((tid >> 6) & 1) * NGS_WAVE_W + (tid & 7); }
__device__ __forceinline__ int tile_local_y(int tid) { return (tid >> 7) * NGS_WAVE_H + ((tid >> 3) & 7); }

// Index of the lowest set bit of a non-zero wave mask.
__device__ __forceinline__ int lowest_lane(unsigned long long m) { return __builtin_ctzll(m); }
