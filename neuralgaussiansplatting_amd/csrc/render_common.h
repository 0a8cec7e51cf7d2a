// Helpers shared by the forward and backward compositing kernels.
#pragma once
#include "ngs_common.h"

// Wave-level culling of (splat, pixel block) pairs.  A pixel passes the reference's per-pixel tests
// (forward.cu:335-345) only if power <= 0 and opacity * exp(power) >= 1/255, i.e. q(d) <= 2 tau with
// q(d) = a dx^2 + 2 b dx dy + c dy^2 (the conic's quadratic form at the pixel's offset d from the centre) and
// tau = ln(255 * opacity).  splat_reaches() minimises q over the continuous rectangle spanned by a wave's
// pixel block (inside -> 0, else the best of the four edges, each a clamped 1-D parabola) and keeps the pair
// unless even that minimum fails the test by a margin: 0.01 in tau plus 1e-5 of the magnitude of the form's
// terms over the block, about thirty times the rounding error of the per-pixel evaluation, however badly the
// terms cancel (needle-shaped splats).  A conic that is not positive definite (or not finite) is never culled.
// Culling therefore never changes a result; it only skips pairs every pixel of the block would have rejected.
__device__ __forceinline__ float splat_tau(float opacity) { return logf(255.0f * opacity); }

__device__ __forceinline__ bool splat_reaches(float sx, float sy, float a, float b, float c, float tau, float x0,
                                              float x1, float y0, float y1)
{
    if (!(a > 0.0f && c > 0.0f && a * c - b * b > 0.0f)) return true;
    const float xl = x0 - sx, xh = x1 - sx, yl = y0 - sy, yh = y1 - sy;
    const bool inside = xl <= 0.0f && xh >= 0.0f && yl <= 0.0f && yh >= 0.0f;
    const float tb = 2.0f * b;
    // edges x = xl, x = xh: q is minimised over y at -b x / c, clamped to the edge
    const float ya = fminf(fmaxf(-b * xl / c, yl), yh), yb = fminf(fmaxf(-b * xh / c, yl), yh);
    const float q0 = a * xl * xl + tb * xl * ya + c * ya * ya;
    const float q1 = a * xh * xh + tb * xh * yb + c * yb * yb;
    // edges y = yl, y = yh
    const float xa = fminf(fmaxf(-b * yl / a, xl), xh), xb = fminf(fmaxf(-b * yh / a, xl), xh);
    const float q2 = a * xa * xa + tb * xa * yl + c * yl * yl;
    const float q3 = a * xb * xb + tb * xb * yh + c * yh * yh;
    const float qmin = inside ? 0.0f : fminf(fminf(q0, q1), fminf(q2, q3));
    const float mx = fmaxf(fabsf(xl), fabsf(xh)), my = fmaxf(fabsf(yl), fabsf(yh));
    const float mag = a * mx * mx + fabsf(tb) * mx * my + c * my * my;
    return 0.5f * qmin <= tau + 0.01f + 1e-5f * mag;
}

// Pixel layout of a tile's 256 threads: each wave64 covers one 8x8 pixel block (2x2 blocks per 16x16 tile), and
// inside a block every 16-lane row of the wave (the unit the DPP row operations work on) covers one 4x4 quadrant,
// row-major inside the quadrant.  final_T / n_contrib are stored per tile in thread order, so the forward and
// backward kernels (and the state decoder) share this mapping.
#define NGS_WAVE_W 8
#define NGS_WAVE_H 8
#define NGS_QUAD 4
__device__ __forceinline__ int tile_local_x(int tid)
{
    return ((tid >> 6) & 1) * NGS_WAVE_W + ((tid >> 4) & 1) * NGS_QUAD + (tid & 3);
}
__device__ __forceinline__ int tile_local_y(int tid)
{
    return (tid >> 7) * NGS_WAVE_H + ((tid >> 5) & 1) * NGS_QUAD + ((tid >> 2) & 3);
}

// exp(x) for the compositing kernels, whose argument is the Gaussian exponent: never positive when it is used.
// Same operation sequence as the device library's expf (argument reduction by log2(e) in two parts, v_exp_f32,
// ldexp), so the result is bit-identical for x <= 0, but without expf's two range guards: for x below -103.3
// expf returns exactly 0 while this returns at most a denormal (alpha then fails the 1/255 test either way),
// and positive arguments are rejected by the caller before the value is used.
__device__ __forceinline__ float exp_nonpos(float x)
{
    const float log2e_hi = __builtin_bit_cast(float, 0x3fb8aa3bu), log2e_lo = __builtin_bit_cast(float, 0x32a5705fu);
    const float ph = x * log2e_hi;
    const float pl = fmaf(x, log2e_lo, fmaf(x, log2e_hi, -ph));
    const float n = rintf(ph);
    return ldexpf(__builtin_amdgcn_exp2f((ph - n) + pl), (int)n);
}

// Index of the lowest set bit of a non-zero wave mask.
__device__ __forceinline__ int lowest_lane(unsigned long long m) { return __builtin_ctzll(m); }
