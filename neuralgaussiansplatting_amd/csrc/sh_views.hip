// View-factored SH gradient for view-sharded data parallelism.
//
// For one view the gradient of SH coefficient k of a Gaussian is
//     dL_dsh[k] = basis_k(direction from that view's camera) * dL_dRGB        (backward.cu:49-137)
// with dL_dRGB the colour gradient after the clamp mask (backward.cu:42-47).
// Summed over V views this does not have to travel as 3*M floats per Gaussian
// and view: every rank can rebuild the sum from the V masked colour gradients
// (3 floats per Gaussian and view) and the V camera positions, which is what
// sh_from_views_kernel does.  All ranks add the views in the same order, so
// they end up with bit-identical sums.  The products are formed by the same
// expression as the per-view backward (sh_basis_weights), so for V == 1 the
// result equals the per-view dL_dsh bit for bit.
#include <string.h>
#include "ngs_common.h"

#define SH_ROW 48      // floats per Gaussian when M == 16
#define SH_ROW_LDS 52  // padded LDS row stride (conflict-free 128-bit accesses)

// out[i] = dL_dcolor[i] with the channels that were clamped in the forward pass zeroed.
__global__ __launch_bounds__(NGS_GAUSS_BLOCK) void masked_color_kernel(int P, const uint8_t* __restrict__ clamped,
                                                                      const float* __restrict__ dL_dcolor,
                                                                      float* __restrict__ out)
{
    const int idx = blockIdx.x * NGS_GAUSS_BLOCK + threadIdx.x;
    if (idx >= P) return;
    const size_t i = (size_t)idx;
    const uint32_t bits = clamped[i];
    out[3 * i] = dL_dcolor[3 * i] * ((bits & 1u) ? 0.0f : 1.0f);
    out[3 * i + 1] = dL_dcolor[3 * i + 1] * ((bits & 2u) ? 0.0f : 1.0f);
    out[3 * i + 2] = dL_dcolor[3 * i + 2] * ((bits & 4u) ? 0.0f : 1.0f);
}

// The same, read straight from the compositing backward's accumulator rows (floats 6..8 of a row are the colour
// gradient), so that it is available before the per-Gaussian backward has run.
__global__ __launch_bounds__(NGS_GAUSS_BLOCK) void masked_color_from_acc_kernel(int P, const uint8_t* __restrict__ clamped,
                                                                               const float* __restrict__ acc,
                                                                               float* __restrict__ out)
{
    const int idx = blockIdx.x * NGS_GAUSS_BLOCK + threadIdx.x;
    if (idx >= P) return;
    const size_t i = (size_t)idx;
    const uint32_t bits = clamped[i];
    const float* row = acc + i * NGS_ACC_STRIDE;
    out[3 * i] = row[6] * ((bits & 1u) ? 0.0f : 1.0f);
    out[3 * i + 1] = row[7] * ((bits & 2u) ? 0.0f : 1.0f);
    out[3 * i + 2] = row[8] * ((bits & 4u) ? 0.0f : 1.0f);
}

// One lane per Gaussian.  LDS: M == 16, the 64 rows of a wave (12 KiB contiguous in memory) leave through LDS as
// coalesced 16-byte stores instead of 64 scattered 192-byte rows.
// TURNS: in how many turns a wave's 64 rows pass through LDS (1 or 2).
template <bool LDS, int TURNS>
__global__ __launch_bounds__(NGS_GAUSS_BLOCK) void sh_from_views_kernel(
    int P, int D, int M, int V, const float* __restrict__ means, const float* __restrict__ campos,
    long long campos_stride, const float* __restrict__ colors, long long colors_stride, float* __restrict__ dL_dsh,
    const ShAdamArgs adam)
{
    __shared__ float s_sh[LDS ? (NGS_GAUSS_BLOCK / TURNS) * SH_ROW_LDS : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int idx = blockIdx.x * NGS_GAUSS_BLOCK + tid;
    const size_t i = (size_t)idx;
    const int coeffs = (D + 1) * (D + 1);
    float3 acc[16];
#pragma unroll
    for (int k = 0; k < 16; k++) acc[k] = make_float3(0.0f, 0.0f, 0.0f);
    if (idx < P) {
        const float3 pos = make_float3(means[3 * i], means[3 * i + 1], means[3 * i + 2]);
        for (int v = 0; v < V; v++) {
            const float* g_ptr = colors + (size_t)v * colors_stride + 3 * i;
            const float3 g = make_float3(g_ptr[0], g_ptr[1], g_ptr[2]);
            if (g.x == 0.0f && g.y == 0.0f && g.z == 0.0f) continue;  // culled or unseen in this view: adds zeros
            const float* cp = campos + (size_t)v * campos_stride;
            // view direction exactly as in the forward and backward SH code (forward.cu:27-29)
            const float3 d0 = make_float3(pos.x - cp[0], pos.y - cp[1], pos.z - cp[2]);
            const float len = sqrtf(f3_dot(d0, d0));
            float w[16];
            sh_basis_weights(D, d0.x / len, d0.y / len, d0.z / len, w);
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (k < coeffs) acc[k] = f3_add(acc[k], f3_scale(w[k], g));
        }
    }
    if (LDS) {
        // With TURNS == 2 the wave's rows leave in two turns of 32 (lanes 0..31, then 32..63): half the LDS, so that
        // it does not cap the waves per SIMD; every store instruction still writes 1 KiB of consecutive memory.
        // (Worth 7 % for the plain gradient; the fused optimiser step is bound by its 3.5 GB of traffic and keeps one
        // turn, which needs fewer registers.)
        constexpr int ROWS = 64 / TURNS;
        float* const wrows = s_sh + wave * ROWS * SH_ROW_LDS;
        const int wave_first = blockIdx.x * NGS_GAUSS_BLOCK + wave * 64;
#pragma unroll
        for (int half = 0; half < TURNS; half++) {
            if (lane / ROWS == half) {
                float3* const row = (float3*)(wrows + (lane % ROWS) * SH_ROW_LDS);
#pragma unroll
                for (int k = 0; k < 16; k++) row[k] = acc[k];
            }
            __builtin_amdgcn_wave_barrier();  // rows are wave-private and a wave's LDS operations complete in order
#pragma unroll
            for (int t = 0; t < SH_ROW * ROWS / 4 / 64; t++) {
                const int f = t * 64 + lane, g = f / (SH_ROW / 4), c = f - g * (SH_ROW / 4);
                const int gi = wave_first + half * ROWS + g;
                if (gi >= P) continue;
                const float4 gr = *(const float4*)(wrows + g * SH_ROW_LDS + c * 4);
                const size_t o = (size_t)gi * SH_ROW + 4 * c;
                if (adam.param) {
                    // optimiser step fused in: the summed gradient is consumed here instead of being written out
                    float4 p = stream_load4(adam.param + o), m = stream_load4(adam.exp_avg + o),
                           v = stream_load4(adam.exp_avg_sq + o);
                    const float s3 = c == 0 ? adam.step_head : adam.step_tail;  // elements 0..2 of a row: the DC term
                    p.x = adam_update(p.x, gr.x * adam.k.grad_scale, m.x, v.x, s3, adam.k);
                    p.y = adam_update(p.y, gr.y * adam.k.grad_scale, m.y, v.y, s3, adam.k);
                    p.z = adam_update(p.z, gr.z * adam.k.grad_scale, m.z, v.z, s3, adam.k);
                    p.w = adam_update(p.w, gr.w * adam.k.grad_scale, m.w, v.w, adam.step_tail, adam.k);
                    stream_store4(adam.param + o, p);
                    stream_store4(adam.exp_avg + o, m);
                    stream_store4(adam.exp_avg_sq + o, v);
                } else {
                    stream_store4(dL_dsh + o, gr);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    } else if (idx < P) {
        float3* out = (float3*)(dL_dsh + (size_t)3 * M * i);
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (k < coeffs && k < M) out[k] = acc[k];
        for (int k = coeffs; k < M; k++) out[k] = make_float3(0.0f, 0.0f, 0.0f);
    }
}

void ngs_launch_masked_color(hipStream_t s, int P, const uint8_t* clamped, const float* dL_dcolor, float* out)
{
    const int nb = (P + NGS_GAUSS_BLOCK - 1) / NGS_GAUSS_BLOCK;
    hipLaunchKernelGGL(masked_color_kernel, dim3(nb), dim3(NGS_GAUSS_BLOCK), 0, s, P, clamped, dL_dcolor, out);
}

void ngs_launch_masked_color_from_acc(hipStream_t s, int P, const uint8_t* clamped, const float* acc, float* out)
{
    const int nb = (P + NGS_GAUSS_BLOCK - 1) / NGS_GAUSS_BLOCK;
    hipLaunchKernelGGL(masked_color_from_acc_kernel, dim3(nb), dim3(NGS_GAUSS_BLOCK), 0, s, P, clamped, acc, out);
}

void ngs_launch_sh_from_views(hipStream_t s, int P, int D, int M, int V, const float* means, const float* campos,
                              long long campos_stride, const float* colors, long long colors_stride, float* dL_dsh,
                              const ShAdamArgs* sh_adam)
{
    const int nb = (P + NGS_GAUSS_BLOCK - 1) / NGS_GAUSS_BLOCK;
    ShAdamArgs adam;
    memset(&adam, 0, sizeof adam);
    if (sh_adam) adam = *sh_adam;  // the caller has checked M == 16 and the alignment
    if (M == 16 && (uintptr_t)dL_dsh % 16 == 0 && adam.param)
        hipLaunchKernelGGL((sh_from_views_kernel<true, 1>), dim3(nb), dim3(NGS_GAUSS_BLOCK), 0, s, P, D, M, V, means,
                           campos, campos_stride, colors, colors_stride, dL_dsh, adam);
    else if (M == 16 && (uintptr_t)dL_dsh % 16 == 0)
        hipLaunchKernelGGL((sh_from_views_kernel<true, 2>), dim3(nb), dim3(NGS_GAUSS_BLOCK), 0, s, P, D, M, V, means,
                           campos, campos_stride, colors, colors_stride, dL_dsh, adam);
    else
        hipLaunchKernelGGL((sh_from_views_kernel<false, 1>), dim3(nb), dim3(NGS_GAUSS_BLOCK), 0, s, P, D, M, V, means,
                           campos, campos_stride, colors, colors_stride, dL_dsh, adam);
}
