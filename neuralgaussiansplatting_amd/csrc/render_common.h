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
    // reciprocals instead of divisions: an ulp of error in the minimiser's position moves the minimum by a
    // second-order amount, far inside the margin below
    const float mbc = -b * __builtin_amdgcn_rcpf(c), mba = -b * __builtin_amdgcn_rcpf(a);
    // edges x = xl, x = xh: q is minimised over y at -b x / c, clamped to the edge
    const float ya = fminf(fmaxf(mbc * xl, yl), yh), yb = fminf(fmaxf(mbc * xh, yl), yh);
    const float q0 = a * xl * xl + tb * xl * ya + c * ya * ya;
    const float q1 = a * xh * xh + tb * xh * yb + c * yb * yb;
    // edges y = yl, y = yh
    const float xa = fminf(fmaxf(mba * yl, xl), xh), xb = fminf(fmaxf(mba * yh, xl), xh);
    const float q2 = a * xa * xa + tb * xa * yl + c * yl * yl;
    const float q3 = a * xb * xb + tb * xb * yh + c * yh * yh;
    const float qmin = inside ? 0.0f : fminf(fminf(q0, q1), fminf(q2, q3));
    const float mx = fmaxf(fabsf(xl), fabsf(xh)), my = fmaxf(fabsf(yl), fabsf(yh));
    const float mag = a * mx * mx + fabsf(tb) * mx * my + c * my * my;
    return 0.5f * qmin <= tau + 0.01f + 1e-5f * mag;
}

// What a splat record carries for the compositing kernels' first, cheap test of a (splat, pixel block) pair:
// x = tau = ln(255 opacity); y = a - b^2 / c and z = c - b^2 / a, the factors by which the quadratic form grows with
// dx^2 alone and with dy^2 alone whatever the other coordinate does (the minimum of q over dy for a given dx is
// dx^2 (a - b^2 / c)).  Zero factors for a conic that is not positive definite: such a splat is never culled.
__device__ __forceinline__ float3 record_cull_terms(float a, float b, float c, float opacity)
{
    const bool pd = a > 0.0f && c > 0.0f && a * c - b * b > 0.0f;
    return make_float3(splat_tau(opacity), pd ? a - b * b / c : 0.0f, pd ? c - b * b / a : 0.0f);
}

// The same test for the four 4x4 quadrants of an 8x8 block (corner pixel (bx0, by0); bit k of the
// result is quadrant (k & 1, k >> 1)), from the edges that FACE the splat's centre only: the form is convex with its
// minimum at the centre, so from the point of a rectangle where it is smallest the way to the centre leaves the
// rectangle at once -- that point lies on an edge whose outward side the centre is on.  Per axis a quadrant has at
// most one such edge (none when the centre lies within the quadrant's extent along that axis), and the quadrants of
// a column (row) share it: four edge lines, each against two extents, instead of eight.  The unclamped minimisers use reciprocals instead of
// divisions, and the margin uses the block's extents, which are at least the quadrant's.
__device__ __forceinline__ unsigned splat_reaches_quadrants(float sx, float sy, float a, float b, float c, float tau,
                                                          float bx0, float by0)
{
    if (!(a > 0.0f && c > 0.0f && a * c - b * b > 0.0f)) return 0xfu;
    const float X[4] = {bx0 - sx, (bx0 + 3.0f) - sx, (bx0 + 4.0f) - sx, (bx0 + 7.0f) - sx};
    const float Y[4] = {by0 - sy, (by0 + 3.0f) - sy, (by0 + 4.0f) - sy, (by0 + 7.0f) - sy};
    const float tb = 2.0f * b;
    const float mbc = -b * __builtin_amdgcn_rcpf(c), mba = -b * __builtin_amdgcn_rcpf(a);
    const float inf = __builtin_huge_valf();
    bool inx[2], iny[2];
    float v[2][2], h[2][2];  // [column][row] minima over the facing vertical / horizontal edge
#pragma unroll
    for (int k = 0; k < 2; k++) {
        inx[k] = X[2 * k] <= 0.0f && X[2 * k + 1] >= 0.0f;
        const float x = X[2 * k] > 0.0f ? X[2 * k] : X[2 * k + 1];
        const float ax2 = a * x * x, tbx = tb * x, yu = mbc * x;
        iny[k] = Y[2 * k] <= 0.0f && Y[2 * k + 1] >= 0.0f;
        const float y = Y[2 * k] > 0.0f ? Y[2 * k] : Y[2 * k + 1];
        const float cy2 = c * y * y, tby = tb * y, xu = mba * y;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const float ys = fminf(fmaxf(yu, Y[2 * j]), Y[2 * j + 1]);
            v[k][j] = inx[k] ? inf : ax2 + tbx * ys + c * ys * ys;
            const float xs = fminf(fmaxf(xu, X[2 * j]), X[2 * j + 1]);
            h[j][k] = iny[k] ? inf : cy2 + tby * xs + a * xs * xs;
        }
    }
    const float mx = fmaxf(fabsf(X[0]), fabsf(X[3])), my = fmaxf(fabsf(Y[0]), fabsf(Y[3]));
    const float bound = 2.0f * (tau + 0.01f + 1e-5f * (a * mx * mx + fabsf(tb) * mx * my + c * my * my));
    unsigned bits = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int kx = k & 1, ky = k >> 1;
        if ((inx[kx] && iny[ky]) || fminf(v[kx][ky], h[kx][ky]) <= bound) bits |= 1u << k;
    }
    return bits;
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
// One hardware exp2 of the rounded product x * log2(e), then a first-order correction for what the rounding dropped
// (the product's error, recovered exactly by an fma, plus the low word of log2(e)): exp2(p + d) = exp2(p) (1 + d ln 2)
// for |d| < 1e-6.  Six instructions against the nine of the device library's expf (which reduces the argument to
// [-0.5, 0.5], takes exp2 of that and scales by ldexp); over (-16, 0] the result is within 1.26 ulp of the exact value
// (expf: 0.86) and equals expf in three cases out of four, one ulp off in the fourth (tools/probes/exp_short_probe.hip).
// Forward and backward pass use this same function, so the backward pass recomputes the forward's alpha bit for bit.
// Arguments far below the range are harmless: exp2 underflows to zero and alpha fails the 1/255 test.
__device__ __forceinline__ float exp_nonpos(float x)
{
#ifdef NGS_LIBRARY_EXP  // measurements only: the device library's expf in both compositing kernels
    return expf(x);
#endif
    const float log2e_hi = __builtin_bit_cast(float, 0x3fb8aa3bu), log2e_lo = __builtin_bit_cast(float, 0x32a5705fu);
    const float ph = x * log2e_hi;
    const float d = fmaf(x, log2e_lo, fmaf(x, log2e_hi, -ph));
    const float e = __builtin_amdgcn_exp2f(ph);
    return fmaf(e * d, 0.69314718f, e);
}

// Wave mask of a lane predicate, straight from the comparison that produced it (the generic __ballot first turns
// the predicate into an integer register and compares that again).
__device__ __forceinline__ unsigned long long wave_mask(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// Index of the lowest set bit of a non-zero wave mask.
__device__ __forceinline__ int lowest_lane(unsigned long long m) { return __builtin_ctzll(m); }
