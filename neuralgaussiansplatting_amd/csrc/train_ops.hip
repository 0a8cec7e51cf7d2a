// The two per-iteration costs next to the rasterizer in the reference's training loop, as fused kernels:
//
//  * the image loss  (1 - lambda) * L1 + lambda * (1 - SSIM)  of train.py:90-92 with utils/loss_utils.py:17-63
//    (11x11 Gaussian window, sigma 1.5, zero padding, per channel), forward value AND its gradient wrt the rendered
//    image in two passes over the image, instead of five depthwise convolutions plus their autograd graph;
//  * the optimiser step of scene/gaussian_model.py:257-283 (torch.optim.Adam, eps 1e-15, one learning rate per
//    parameter group) fused with the parameter activations of gaussian_model.py:113-141 (sigmoid, exp, normalise) and
//    their backward: the kernels take the gradients the rasterizer produces (wrt the ACTIVATED values), update the raw
//    parameters and write the activated values the next forward pass reads.
//
// Everything here streams through HBM once; there is no reuse to exploit beyond the LDS tiles of the blur.
#include "ngs_common.h"

#define SSIM_RADIUS 5
#define SSIM_TAPS (2 * SSIM_RADIUS + 1)
#define SSIM_TW 32                            // output tile of one workgroup: 32 x 32 pixels of one channel
#define SSIM_TH 32
#define SSIM_EW (SSIM_TW + 2 * SSIM_RADIUS)   // 42: tile plus halo
#define SSIM_EH (SSIM_TH + 2 * SSIM_RADIUS)
#define SSIM_SP (SSIM_EW + 1)                 // odd LDS pitches: lanes that differ in the row hit different banks
#define SSIM_HP (SSIM_TW + 1)
#define SSIM_HSPAN 8                          // horizontal pass: consecutive outputs per thread
#define SSIM_VSPAN 4                          // vertical pass: consecutive output rows per thread
#define SSIM_C1 (0.01f * 0.01f)
#define SSIM_C2 (0.03f * 0.03f)

struct SsimWindow {
    float w[SSIM_TAPS];  // normalised 1-D Gaussian; the reference's 2-D window is its outer product
};

// The halo tile whose first output pixel is (x0, y0) travels from one image plane to LDS in two steps, so that a
// kernel can request the tiles of all its planes before it waits for any of them: fetch_halo issues this thread's
// global loads into registers (zeros outside the image: the reference pads its convolutions with zeros),
// store_halo parks them in LDS.
#define SSIM_HALO_PER_THREAD ((SSIM_EH * SSIM_EW + NGS_TILE_PIX - 1) / NGS_TILE_PIX)

__device__ __forceinline__ void fetch_halo(const float* __restrict__ plane, int W, int H, int x0, int y0, int tid,
                                           float (&v)[SSIM_HALO_PER_THREAD])
{
#pragma unroll
    for (int k = 0; k < SSIM_HALO_PER_THREAD; k++) {
        const int i = tid + k * NGS_TILE_PIX;
        const int r = i / SSIM_EW, c = i - r * SSIM_EW;
        const int gx = x0 + c - SSIM_RADIUS, gy = y0 + r - SSIM_RADIUS;
        v[k] = (i < SSIM_EH * SSIM_EW && gx >= 0 && gx < W && gy >= 0 && gy < H) ? plane[(size_t)gy * W + gx] : 0.0f;
    }
}

__device__ __forceinline__ void store_halo(float (*dst)[SSIM_SP], int tid, const float (&v)[SSIM_HALO_PER_THREAD])
{
#pragma unroll
    for (int k = 0; k < SSIM_HALO_PER_THREAD; k++) {
        const int i = tid + k * NGS_TILE_PIX;
        const int r = i / SSIM_EW, c = i - r * SSIM_EW;
        if (i < SSIM_EH * SSIM_EW) dst[r][c] = v[k];
    }
}

// Both blur passes keep a sliding window in registers: a thread of the horizontal pass reads 18 neighbours once and
// produces 8 consecutive outputs of its row, a thread of the vertical pass reads 14 rows of its column and produces
// 4, so each staged value is read from LDS 2-3 times instead of 11.
//
// Horizontal pass over the rows of the halo tile: work item (row r, group g) -> outputs hb[.][r][8g .. 8g+7].
#define SSIM_HITEMS (SSIM_EH * (SSIM_TW / SSIM_HSPAN))

// Vertical pass: K blurred values for the 4 output rows starting at row `ry` of column `lx`.
template <int K>
__device__ __forceinline__ void blur_columns(const float (*hb)[SSIM_EH][SSIM_HP], const SsimWindow& win, int lx, int ry,
                                             float (*out)[SSIM_VSPAN])
{
#pragma clang fp contract(fast)
#pragma unroll
    for (int m = 0; m < K; m++) {
        float col[SSIM_VSPAN + SSIM_TAPS - 1];
#pragma unroll
        for (int j = 0; j < SSIM_VSPAN + SSIM_TAPS - 1; j++) col[j] = hb[m][ry + j][lx];
#pragma unroll
        for (int o = 0; o < SSIM_VSPAN; o++) {
            float a = 0.0f;
#pragma unroll
            for (int k = 0; k < SSIM_TAPS; k++) a += win.w[k] * col[o + k];
            out[m][o] = a;
        }
    }
}

__device__ __forceinline__ float fast_recip(float q)  // reciprocal refined by one Newton step (error below 1 ulp)
{
#pragma clang fp contract(fast)
    const float x = __builtin_amdgcn_rcpf(q);
    return x * (2.0f - q * x);
}

// Pass 1: SSIM map, L1 term and the three per-pixel partial derivatives of the SSIM value wrt the blurred maps
// mu1 = w*x, E1 = w*x^2, E12 = w*x*y (x the rendered image, y the ground truth).
__global__ __launch_bounds__(NGS_TILE_PIX) void ssim_l1_forward_kernel(
    int W, int H, SsimWindow win, const float* __restrict__ img, const float* __restrict__ gt,
    float* __restrict__ d_mu1, float* __restrict__ d_e1, float* __restrict__ d_e12, float* __restrict__ partial_sums)
{
#pragma clang fp contract(fast)  // the blurs are FMA chains; nothing here has to match another evaluation bit for bit
    __shared__ float sx[SSIM_EH][SSIM_SP], sy[SSIM_EH][SSIM_SP];
    __shared__ float hb[5][SSIM_EH][SSIM_HP];
    __shared__ float red[2][NGS_TILE_PIX / 64];
    const int tid = threadIdx.x;
    const int tiles_x = (W + SSIM_TW - 1) / SSIM_TW;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, ch = blockIdx.y;
    const int x0 = tx * SSIM_TW, y0 = ty * SSIM_TH;
    const size_t plane = (size_t)W * H;
    {
        float vx[SSIM_HALO_PER_THREAD], vy[SSIM_HALO_PER_THREAD];
        fetch_halo(img + ch * plane, W, H, x0, y0, tid, vx);
        fetch_halo(gt + ch * plane, W, H, x0, y0, tid, vy);
        store_halo(sx, tid, vx);
        store_halo(sy, tid, vy);
    }
    __syncthreads();
    // horizontal blur of x, y, x^2, y^2, xy for every row of the halo tile
    if (tid < SSIM_HITEMS) {
        const int g = tid / SSIM_EH, r = tid - g * SSIM_EH, c0 = g * SSIM_HSPAN;
        float xs[SSIM_HSPAN + SSIM_TAPS - 1], ys[SSIM_HSPAN + SSIM_TAPS - 1], xx[SSIM_HSPAN + SSIM_TAPS - 1],
            yy[SSIM_HSPAN + SSIM_TAPS - 1], xy[SSIM_HSPAN + SSIM_TAPS - 1];
#pragma unroll
        for (int j = 0; j < SSIM_HSPAN + SSIM_TAPS - 1; j++) {
            xs[j] = sx[r][c0 + j];
            ys[j] = sy[r][c0 + j];
            xx[j] = xs[j] * xs[j];
            yy[j] = ys[j] * ys[j];
            xy[j] = xs[j] * ys[j];
        }
#pragma unroll
        for (int o = 0; o < SSIM_HSPAN; o++) {
            float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
#pragma unroll
            for (int k = 0; k < SSIM_TAPS; k++) {
                const float w = win.w[k];
                a0 += w * xs[o + k];
                a1 += w * ys[o + k];
                a2 += w * xx[o + k];
                a3 += w * yy[o + k];
                a4 += w * xy[o + k];
            }
            hb[0][r][c0 + o] = a0; hb[1][r][c0 + o] = a1; hb[2][r][c0 + o] = a2; hb[3][r][c0 + o] = a3;
            hb[4][r][c0 + o] = a4;
        }
    }
    __syncthreads();
    const int lx = tid & (SSIM_TW - 1), ry = (tid / SSIM_TW) * SSIM_VSPAN;
    float b[5][SSIM_VSPAN];
    blur_columns<5>(hb, win, lx, ry, b);
    const int gx = x0 + lx;
    float s_sum = 0.0f, l1_sum = 0.0f;
#pragma unroll
    for (int o = 0; o < SSIM_VSPAN; o++) {
        const int gy = y0 + ry + o;
        if (gx < W && gy < H) {
            const float mu1 = b[0][o], mu2 = b[1][o], e1 = b[2][o], e2 = b[3][o], e12 = b[4][o];
            const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
            const float sigma1 = e1 - mu1_sq, sigma2 = e2 - mu2_sq, sigma12 = e12 - mu12;
            const float n1 = 2.0f * mu12 + SSIM_C1, n2 = 2.0f * sigma12 + SSIM_C2;
            const float q1 = mu1_sq + mu2_sq + SSIM_C1, q2 = sigma1 + sigma2 + SSIM_C2;
            const float inv2 = fast_recip(q2);
            const float inv = fast_recip(q1) * inv2;
            const float s_val = (n1 * n2) * inv;
            // d/d(mu1) with E1, E12 fixed: n1' = 2 mu2, n2' = -2 mu2, q1' = 2 mu1, q2' = -2 mu1
            const float dnum = 2.0f * mu2 * (n2 - n1);
            const float dden = 2.0f * mu1 * (q2 - q1);
            const size_t p = ch * plane + (size_t)gy * W + gx;
            d_mu1[p] = (dnum - s_val * dden) * inv;
            d_e1[p] = -s_val * inv2;
            d_e12[p] = 2.0f * n1 * inv;
            s_sum += s_val;
            l1_sum += fabsf(sx[ry + o + SSIM_RADIUS][lx + SSIM_RADIUS] - sy[ry + o + SSIM_RADIUS][lx + SSIM_RADIUS]);
        }
    }
    // block sums, in a fixed order (the final reduction is a single block as well: the loss value is deterministic)
    const float ws = wave_sum_lane63(s_sum), wl = wave_sum_lane63(l1_sum);
    if ((tid & 63) == 63) {
        red[0][tid >> 6] = ws;
        red[1][tid >> 6] = wl;
    }
    __syncthreads();
    if (tid == 0) {
        const size_t bi = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        partial_sums[2 * bi] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partial_sums[2 * bi + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// loss_terms = {L1 mean, SSIM mean, (1 - lambda) * L1 + lambda * (1 - SSIM)}
#define LOSS_REDUCE_THREADS 1024
__global__ __launch_bounds__(LOSS_REDUCE_THREADS) void loss_reduce_kernel(int nblocks,
                                                                        const float* __restrict__ partial_sums,
                                                                        float inv_count, float lambda_dssim,
                                                                        float* __restrict__ loss_terms)
{
    __shared__ double red[2][LOSS_REDUCE_THREADS];
    double s = 0.0, l = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += LOSS_REDUCE_THREADS) {
        const float2 v = ((const float2*)partial_sums)[i];
        s += (double)v.x;
        l += (double)v.y;
    }
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = l;
    __syncthreads();
    for (int o = LOSS_REDUCE_THREADS / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            red[0][threadIdx.x] += red[0][threadIdx.x + o];
            red[1][threadIdx.x] += red[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float ssim = (float)(red[0][0] * (double)inv_count), l1 = (float)(red[1][0] * (double)inv_count);
        loss_terms[0] = l1;
        loss_terms[1] = ssim;
        loss_terms[2] = (1.0f - lambda_dssim) * l1 + lambda_dssim * (1.0f - ssim);
    }
}

// Pass 2: dLoss/dx = c_l1 * sign(x - y) - c_ssim * (w*d_mu1 + 2 x (w*d_e1) + y (w*d_e12)), the transposed
// convolutions being the same symmetric blur with zeros outside the image.
__global__ __launch_bounds__(NGS_TILE_PIX) void ssim_l1_backward_kernel(
    int W, int H, SsimWindow win, const float* __restrict__ img, const float* __restrict__ gt,
    const float* __restrict__ d_mu1, const float* __restrict__ d_e1, const float* __restrict__ d_e12, float c_l1,
    float c_ssim, float* __restrict__ dL_dimg)
{
#pragma clang fp contract(fast)
    __shared__ float sp[3][SSIM_EH][SSIM_SP];
    __shared__ float hb[3][SSIM_EH][SSIM_HP];
    const int tid = threadIdx.x;
    const int tiles_x = (W + SSIM_TW - 1) / SSIM_TW;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, ch = blockIdx.y;
    const int x0 = tx * SSIM_TW, y0 = ty * SSIM_TH;
    const size_t plane = (size_t)W * H;
    {
        float v0[SSIM_HALO_PER_THREAD], v1[SSIM_HALO_PER_THREAD], v2[SSIM_HALO_PER_THREAD];
        fetch_halo(d_mu1 + ch * plane, W, H, x0, y0, tid, v0);
        fetch_halo(d_e1 + ch * plane, W, H, x0, y0, tid, v1);
        fetch_halo(d_e12 + ch * plane, W, H, x0, y0, tid, v2);
        store_halo(sp[0], tid, v0);
        store_halo(sp[1], tid, v1);
        store_halo(sp[2], tid, v2);
    }
    __syncthreads();
    if (tid < SSIM_HITEMS) {
        const int g = tid / SSIM_EH, r = tid - g * SSIM_EH, c0 = g * SSIM_HSPAN;
#pragma unroll
        for (int m = 0; m < 3; m++) {
            float v[SSIM_HSPAN + SSIM_TAPS - 1];
#pragma unroll
            for (int j = 0; j < SSIM_HSPAN + SSIM_TAPS - 1; j++) v[j] = sp[m][r][c0 + j];
#pragma unroll
            for (int o = 0; o < SSIM_HSPAN; o++) {
                float a = 0.0f;
#pragma unroll
                for (int k = 0; k < SSIM_TAPS; k++) a += win.w[k] * v[o + k];
                hb[m][r][c0 + o] = a;
            }
        }
    }
    __syncthreads();
    const int lx = tid & (SSIM_TW - 1), ry = (tid / SSIM_TW) * SSIM_VSPAN;
    float b[3][SSIM_VSPAN];
    blur_columns<3>(hb, win, lx, ry, b);
    const int gx = x0 + lx;
#pragma unroll
    for (int o = 0; o < SSIM_VSPAN; o++) {
        const int gy = y0 + ry + o;
        if (gx < W && gy < H) {
            const size_t p = ch * plane + (size_t)gy * W + gx;
            const float x = img[p], y = gt[p], d = x - y;
            const float sgn = d > 0.0f ? 1.0f : d < 0.0f ? -1.0f : 0.0f;
            dL_dimg[p] = c_l1 * sgn - c_ssim * (b[0][o] + 2.0f * x * b[1][o] + y * b[2][o]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Adam.  Operation order of torch.optim.Adam's single-tensor path: exp_avg.lerp_(g, 1-b1); exp_avg_sq = b2*v +
// (1-b2)*g*g; denom = sqrt(v)/sqrt(1-b2^t) + eps; p -= (lr/(1-b1^t)) * m/denom.
// ---------------------------------------------------------------------------------------------------------------
typedef float v4f __attribute__((ext_vector_type(4)));

// Flat tensor whose activation is the identity (positions, SH coefficients).  Elements whose index modulo `period`
// is below `head` use lr_head, the others lr_tail (SH rows [P][M][3]: the DC coefficient and the higher bands are
// separate parameter groups in the reference, gaussian_model.py:262-263).
__global__ __launch_bounds__(256) void adam_flat_kernel(long long n, float* __restrict__ param,
                                                        const float* __restrict__ grad, float* __restrict__ exp_avg,
                                                        float* __restrict__ exp_avg_sq, float step_head,
                                                        float step_tail, int period, int head, AdamConst k)
{
    // Each thread owns two float4 groups a whole workgroup-stride apart, so all eight 16-byte loads of a lane are
    // in flight together and every wave instruction still covers 1 KiB of contiguous memory.
    const long long base = (long long)blockIdx.x * 2048 + threadIdx.x * 4;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const long long i4 = base + half * 1024;
        if (i4 >= n) return;
        // position inside the period: 32-bit arithmetic whenever the index fits (a 64-bit modulo is ~100 instructions)
        unsigned r = i4 < 0x7fffffffLL ? (unsigned)i4 % (unsigned)period : (unsigned)(i4 % period);
        if (i4 + 4 <= n) {
            // every byte is touched once per step: non-temporal accesses keep the streams from evicting each other
            v4f p = __builtin_nontemporal_load((const v4f*)(param + i4));
            v4f g = __builtin_nontemporal_load((const v4f*)(grad + i4));
            v4f m = __builtin_nontemporal_load((const v4f*)(exp_avg + i4));
            v4f v = __builtin_nontemporal_load((const v4f*)(exp_avg_sq + i4));
            float pe[4] = {p.x, p.y, p.z, p.w}, ge[4] = {g.x, g.y, g.z, g.w}, me[4] = {m.x, m.y, m.z, m.w},
                  ve[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                pe[e] = adam_update(pe[e], ge[e] * k.grad_scale, me[e], ve[e], (int)r < head ? step_head : step_tail, k);
                r = r + 1 == (unsigned)period ? 0u : r + 1;
            }
            __builtin_nontemporal_store((v4f){pe[0], pe[1], pe[2], pe[3]}, (v4f*)(param + i4));
            __builtin_nontemporal_store((v4f){me[0], me[1], me[2], me[3]}, (v4f*)(exp_avg + i4));
            __builtin_nontemporal_store((v4f){ve[0], ve[1], ve[2], ve[3]}, (v4f*)(exp_avg_sq + i4));
        } else {
            for (long long i = i4; i < n; i++) {
                float m = exp_avg[i], v = exp_avg_sq[i];
                param[i] = adam_update(param[i], grad[i] * k.grad_scale, m, v, (int)r < head ? step_head : step_tail, k);
                exp_avg[i] = m;
                exp_avg_sq[i] = v;
                r = r + 1 == (unsigned)period ? 0u : r + 1;
            }
        }
    }
}

struct ActivatedGroup {
    float* raw;        // raw parameter, updated in place
    const float* grad; // gradient wrt the ACTIVATED value (what the rasterizer's backward returns)
    float* exp_avg;
    float* exp_avg_sq;
    float* activated;  // activated value for the next forward pass, written here
    float step;        // float(lr / (1 - beta1^t)), t = the number of steps THIS group has taken
    float bc2_sqrt;    // float(sqrt(1 - beta2^t))
};

// One lane per Gaussian: opacity (sigmoid), scale (exp) and rotation (normalise), each with the backward of its
// activation, Adam, and the new activated value; plus the densification statistics of train.py:113-116 /
// gaussian_model.py:550-552 when their arrays are given.
__global__ __launch_bounds__(256) void adam_activated_kernel(int P, ActivatedGroup op, ActivatedGroup sc,
                                                             ActivatedGroup ro, AdamConst k,
                                                             const float* __restrict__ dL_dmean2D,
                                                             const int* __restrict__ radii,
                                                             float* __restrict__ grad_accum,
                                                             float* __restrict__ denom, int* __restrict__ max_radii)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const size_t i = (size_t)idx;
    {   // opacity = sigmoid(raw): d raw = d act * act * (1 - act)
        const float raw = op.raw[i];
        const float act = sigmoid_act(raw);
        const float g = (op.grad[i] * k.grad_scale) * ((1.0f - act) * act);
        float m = op.exp_avg[i], v = op.exp_avg_sq[i];
        const float nr = adam_update(raw, g, m, v, op.step, op.bc2_sqrt, k);
        op.raw[i] = nr;
        op.exp_avg[i] = m;
        op.exp_avg_sq[i] = v;
        op.activated[i] = sigmoid_act(nr);
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {  // scale = exp(raw): d raw = d act * act
        const size_t j = 3 * i + c;
        const float raw = sc.raw[j];
        const float g = (sc.grad[j] * k.grad_scale) * expf(raw);
        float m = sc.exp_avg[j], v = sc.exp_avg_sq[j];
        const float nr = adam_update(raw, g, m, v, sc.step, sc.bc2_sqrt, k);
        sc.raw[j] = nr;
        sc.exp_avg[j] = m;
        sc.exp_avg_sq[j] = v;
        sc.activated[j] = expf(nr);
    }
    {   // rotation = raw / max(|raw|, 1e-12): d raw = (d act - act (act . d act)) / |raw|
        const float4 raw = ((const float4*)ro.raw)[i];
        float4 g = ((const float4*)ro.grad)[i];
        g.x *= k.grad_scale; g.y *= k.grad_scale; g.z *= k.grad_scale; g.w *= k.grad_scale;
        float len;
        const float4 a = normalize_quat(raw, len);
        const float4 gr = normalize_quat_backward(a, len, g);
        float4 m = ((const float4*)ro.exp_avg)[i], v = ((const float4*)ro.exp_avg_sq)[i];
        float4 nr;
        nr.x = adam_update(raw.x, gr.x, m.x, v.x, ro.step, ro.bc2_sqrt, k);
        nr.y = adam_update(raw.y, gr.y, m.y, v.y, ro.step, ro.bc2_sqrt, k);
        nr.z = adam_update(raw.z, gr.z, m.z, v.z, ro.step, ro.bc2_sqrt, k);
        nr.w = adam_update(raw.w, gr.w, m.w, v.w, ro.step, ro.bc2_sqrt, k);
        ((float4*)ro.raw)[i] = nr;
        ((float4*)ro.exp_avg)[i] = m;
        ((float4*)ro.exp_avg_sq)[i] = v;
        float nl;
        ((float4*)ro.activated)[i] = normalize_quat(nr, nl);
    }
    if (grad_accum) {
        const int r = radii[i];
        if (r > 0) {
            const float gx = dL_dmean2D[3 * i], gy = dL_dmean2D[3 * i + 1];
            grad_accum[i] += sqrtf(gx * gx + gy * gy);
            denom[i] += 1.0f;
            max_radii[i] = max(max_radii[i], r);
        }
    }
}

// The same update with every array walked as float4: thread t takes quaternion t, elements 4t..4t+3 of the flat
// scale arrays (the activation is elementwise, so groups of four need not coincide with Gaussians) and elements
// 4t..4t+3 of the opacity arrays.  One 16-byte access per array and thread instead of strided scalar ones.  Needs
// 16-byte aligned arrays; the arithmetic per element is that of adam_activated_kernel.
__device__ __forceinline__ void adam_opacity_elem(float& raw, float g, float& m, float& v, float& act,
                                                  const ActivatedGroup& op, const AdamConst& k)
{
    const float a = sigmoid_act(raw);
    const float gr = (g * k.grad_scale) * ((1.0f - a) * a);
    raw = adam_update(raw, gr, m, v, op.step, op.bc2_sqrt, k);
    act = sigmoid_act(raw);
}
__device__ __forceinline__ void adam_scale_elem(float& raw, float g, float& m, float& v, float& act,
                                                const ActivatedGroup& sc, const AdamConst& k)
{
    const float gr = (g * k.grad_scale) * expf(raw);
    raw = adam_update(raw, gr, m, v, sc.step, sc.bc2_sqrt, k);
    act = expf(raw);
}

template <bool SCALE>
__device__ __forceinline__ void adam_elementwise_group(size_t first, size_t n, const ActivatedGroup& grp,
                                                       const AdamConst& k)
{
    if (first >= n) return;
    if (first + 4 <= n) {
        float4 raw = ((const float4*)grp.raw)[first / 4], g = ((const float4*)grp.grad)[first / 4];
        float4 m = ((const float4*)grp.exp_avg)[first / 4], v = ((const float4*)grp.exp_avg_sq)[first / 4], act;
        if (SCALE) {
            adam_scale_elem(raw.x, g.x, m.x, v.x, act.x, grp, k); adam_scale_elem(raw.y, g.y, m.y, v.y, act.y, grp, k);
            adam_scale_elem(raw.z, g.z, m.z, v.z, act.z, grp, k); adam_scale_elem(raw.w, g.w, m.w, v.w, act.w, grp, k);
        } else {
            adam_opacity_elem(raw.x, g.x, m.x, v.x, act.x, grp, k); adam_opacity_elem(raw.y, g.y, m.y, v.y, act.y, grp, k);
            adam_opacity_elem(raw.z, g.z, m.z, v.z, act.z, grp, k); adam_opacity_elem(raw.w, g.w, m.w, v.w, act.w, grp, k);
        }
        ((float4*)grp.raw)[first / 4] = raw;
        ((float4*)grp.exp_avg)[first / 4] = m;
        ((float4*)grp.exp_avg_sq)[first / 4] = v;
        ((float4*)grp.activated)[first / 4] = act;
    } else {
        for (size_t j = first; j < n; j++) {  // the last one to three elements
            float raw = grp.raw[j], m = grp.exp_avg[j], v = grp.exp_avg_sq[j], act;
            if (SCALE) adam_scale_elem(raw, grp.grad[j], m, v, act, grp, k);
            else adam_opacity_elem(raw, grp.grad[j], m, v, act, grp, k);
            grp.raw[j] = raw; grp.exp_avg[j] = m; grp.exp_avg_sq[j] = v; grp.activated[j] = act;
        }
    }
}

__global__ __launch_bounds__(256) void adam_activated_vec_kernel(int P, ActivatedGroup op, ActivatedGroup sc,
                                                                 ActivatedGroup ro, AdamConst k,
                                                                 const float* __restrict__ dL_dmean2D,
                                                                 const int* __restrict__ radii,
                                                                 float* __restrict__ grad_accum,
                                                                 float* __restrict__ denom, int* __restrict__ max_radii)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const size_t i = (size_t)idx;
    adam_elementwise_group<false>(4 * i, (size_t)P, op, k);
    adam_elementwise_group<true>(4 * i, 3 * (size_t)P, sc, k);
    {   // rotation = raw / max(|raw|, 1e-12): d raw = (d act - act (act . d act)) / |raw|
        const float4 raw = ((const float4*)ro.raw)[i];
        float4 g = ((const float4*)ro.grad)[i];
        g.x *= k.grad_scale; g.y *= k.grad_scale; g.z *= k.grad_scale; g.w *= k.grad_scale;
        float len;
        const float4 a = normalize_quat(raw, len);
        const float4 gr = normalize_quat_backward(a, len, g);
        float4 m = ((const float4*)ro.exp_avg)[i], v = ((const float4*)ro.exp_avg_sq)[i];
        float4 nr;
        nr.x = adam_update(raw.x, gr.x, m.x, v.x, ro.step, ro.bc2_sqrt, k);
        nr.y = adam_update(raw.y, gr.y, m.y, v.y, ro.step, ro.bc2_sqrt, k);
        nr.z = adam_update(raw.z, gr.z, m.z, v.z, ro.step, ro.bc2_sqrt, k);
        nr.w = adam_update(raw.w, gr.w, m.w, v.w, ro.step, ro.bc2_sqrt, k);
        ((float4*)ro.raw)[i] = nr;
        ((float4*)ro.exp_avg)[i] = m;
        ((float4*)ro.exp_avg_sq)[i] = v;
        float nl;
        ((float4*)ro.activated)[i] = normalize_quat(nr, nl);
    }
    if (grad_accum) {
        const int r = radii[i];
        if (r > 0) {
            const float gx = dL_dmean2D[3 * i], gy = dL_dmean2D[3 * i + 1];
            grad_accum[i] += sqrtf(gx * gx + gy * gy);
            denom[i] += 1.0f;
            max_radii[i] = max(max_radii[i], r);
        }
    }
}

// One view's densification statistics as two summable rows (scene/gaussian_model.py:550-552): the view-space gradient
// norm and the visibility flag of every Gaussian, zeros where it was not rendered; the running maximum of the screen
// radius (train.py:115) is updated in place when given.  Ranks sum the rows together with their gradients.
__global__ __launch_bounds__(256) void view_statistics_kernel(int P, const float* __restrict__ dL_dmean2D,
                                                              const int* __restrict__ radii,
                                                              float* __restrict__ norm_out, float* __restrict__ vis_out,
                                                              int* __restrict__ max_radii)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int r = radii[i];
    float n = 0.0f, v = 0.0f;
    if (r > 0) {
        const float gx = dL_dmean2D[3 * (size_t)i], gy = dL_dmean2D[3 * (size_t)i + 1];
        n = sqrtf(gx * gx + gy * gy);
        v = 1.0f;
        if (max_radii) max_radii[i] = max(max_radii[i], r);
    }
    norm_out[i] = n;
    vis_out[i] = v;
}

// The three activations on their own (any output may be NULL); element accesses, no alignment needed.
__global__ __launch_bounds__(256) void activate_kernel(int P, const float* __restrict__ op_raw,
                                                       const float* __restrict__ sc_raw,
                                                       const float* __restrict__ ro_raw, float* __restrict__ op,
                                                       float* __restrict__ sc, float* __restrict__ ro)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const size_t i = (size_t)idx;
    if (op) op[i] = sigmoid_act(op_raw[i]);
    if (sc) {
#pragma unroll
        for (int c = 0; c < 3; c++) sc[3 * i + c] = expf(sc_raw[3 * i + c]);
    }
    if (ro) {
        float len;
        const float4 a = normalize_quat(make_float4(ro_raw[4 * i], ro_raw[4 * i + 1], ro_raw[4 * i + 2],
                                                    ro_raw[4 * i + 3]), len);
        ro[4 * i] = a.x; ro[4 * i + 1] = a.y; ro[4 * i + 2] = a.z; ro[4 * i + 3] = a.w;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Host side.
// ---------------------------------------------------------------------------------------------------------------
static SsimWindow make_window()
{
    // utils/loss_utils.py:23-25: exp() in double, stored as float32, normalised in float32
    SsimWindow w;
    float sum = 0.0f;
    for (int i = 0; i < SSIM_TAPS; i++) {
        const double d = (double)(i - SSIM_RADIUS);
        w.w[i] = (float)exp(-(d * d) / (2.0 * 1.5 * 1.5));
        sum += w.w[i];
    }
    for (int i = 0; i < SSIM_TAPS; i++) w.w[i] /= sum;
    return w;
}

size_t ngs_loss_workspace_floats(int W, int H)
{
    const size_t tiles = (size_t)((W + SSIM_TW - 1) / SSIM_TW) * ((H + SSIM_TH - 1) / SSIM_TH);
    return 9 * (size_t)W * H + 2 + 2 * 3 * tiles;  // +2: the block sums start on an 8-byte boundary
}

void ngs_launch_l1_ssim(hipStream_t s, int W, int H, float lambda_dssim, const float* img, const float* gt,
                        float* workspace, float* loss_terms, float* dL_dimg)
{
    const SsimWindow win = make_window();
    const size_t n = 3 * (size_t)W * H;
    float* d_mu1 = workspace;
    float* d_e1 = workspace + n;
    float* d_e12 = workspace + 2 * n;
    float* partial = workspace + ((3 * n + 1) & ~(size_t)1);  // read back as float2
    const int tiles = ((W + SSIM_TW - 1) / SSIM_TW) * ((H + SSIM_TH - 1) / SSIM_TH);
    hipLaunchKernelGGL(ssim_l1_forward_kernel, dim3(tiles, 3), dim3(NGS_TILE_PIX), 0, s, W, H, win, img, gt, d_mu1,
                       d_e1, d_e12, partial);
    const float inv_count = (float)(1.0 / (double)n);
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(LOSS_REDUCE_THREADS), 0, s, tiles * 3, partial, inv_count,
                       lambda_dssim, loss_terms);
    hipLaunchKernelGGL(ssim_l1_backward_kernel, dim3(tiles, 3), dim3(NGS_TILE_PIX), 0, s, W, H, win, img, gt, d_mu1,
                       d_e1, d_e12, (1.0f - lambda_dssim) * inv_count, lambda_dssim * inv_count, dL_dimg);
}

void ngs_launch_adam_flat(hipStream_t s, long long n, float* param, const float* grad, float* exp_avg,
                          float* exp_avg_sq, double lr_head, double lr_tail, int period, int head, double beta1,
                          double beta2, double eps, int step, float grad_scale)
{
    hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, s, n, param, grad,
                       exp_avg, exp_avg_sq, ngs_adam_step_size(lr_head, beta1, step), ngs_adam_step_size(lr_tail, beta1, step),
                       period, head, ngs_adam_const(beta1, beta2, eps, step, grad_scale));
}

void ngs_launch_adam_activated(hipStream_t s, int P, float* const raw[3], const float* const grad[3],
                               float* const exp_avg[3], float* const exp_avg_sq[3], float* const activated[3],
                               const double lr[3], double beta1, double beta2, double eps, const int step[3],
                               float grad_scale,
                               const float* dL_dmean2D, const int* radii, float* grad_accum, float* denom,
                               int* max_radii)
{
    ActivatedGroup g[3];
    for (int i = 0; i < 3; i++) {
        g[i].raw = raw[i];
        g[i].grad = grad[i];
        g[i].exp_avg = exp_avg[i];
        g[i].exp_avg_sq = exp_avg_sq[i];
        g[i].activated = activated[i];
        g[i].step = ngs_adam_step_size(lr[i], beta1, step[i]);
        g[i].bc2_sqrt = ngs_adam_const(beta1, beta2, eps, step[i], grad_scale).bc2_sqrt;
    }
    uintptr_t bits = 0;
    for (int i = 0; i < 2; i++)
        bits |= (uintptr_t)raw[i] | (uintptr_t)grad[i] | (uintptr_t)exp_avg[i] | (uintptr_t)exp_avg_sq[i] |
                (uintptr_t)activated[i];
    if (bits % 16 == 0)
        hipLaunchKernelGGL(adam_activated_vec_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, g[0], g[1], g[2],
                           ngs_adam_const(beta1, beta2, eps, step[0], grad_scale), dL_dmean2D, radii, grad_accum,
                           denom, max_radii);
    else
        hipLaunchKernelGGL(adam_activated_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, g[0], g[1], g[2],
                           ngs_adam_const(beta1, beta2, eps, step[0], grad_scale), dL_dmean2D, radii, grad_accum,
                           denom, max_radii);
}

void ngs_launch_activate(hipStream_t s, int P, const float* op_raw, const float* sc_raw, const float* ro_raw,
                         float* op, float* sc, float* ro)
{
    hipLaunchKernelGGL(activate_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, op_raw, sc_raw, ro_raw, op, sc, ro);
}

void ngs_launch_view_statistics(hipStream_t s, int P, const float* dL_dmean2D, const int* radii, float* norm_out,
                                float* vis_out, int* max_radii)
{
    hipLaunchKernelGGL(view_statistics_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, dL_dmean2D, radii, norm_out,
                       vis_out, max_radii);
}
