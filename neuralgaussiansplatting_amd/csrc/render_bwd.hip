// Per-tile back-to-front compositing gradient.  Same tiling and LDS staging
// as the forward kernel, traversed in reverse; per-pixel arithmetic follows
// cuda_rasterizer/backward.cu:399-557 of the reference.  What differs is the
// reduction: the reference issues nine global float atomics per (pixel,
// splat) pair.  Here all 64 lanes of a wave handle the same splat in
// lock-step, so the nine partials are first summed across the wave by DPP,
// then across the tile's four waves in an LDS table holding one row per
// staged splat, and each splat's row leaves the workgroup once per batch as
// ONE 64-byte-row group of global atomics into the [P][16] accumulator.
#include "ngs_common.h"

__global__ __launch_bounds__(NGS_TILE_PIX) void render_bwd_kernel(
    int W, int H, int grid_x, const uint2* __restrict__ ranges, const uint32_t* __restrict__ plist,
    const float4* __restrict__ g0, const float4* __restrict__ g1, const float* __restrict__ g2,
    const float* __restrict__ bg, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dpix, float* __restrict__ acc)
{
    __shared__ float4 s_a[NGS_TILE_PIX];  // x, y, conic a, conic b
    __shared__ float4 s_b[NGS_TILE_PIX];  // conic c, opacity, r, g
    __shared__ float s_c[NGS_TILE_PIX];   // b
    __shared__ uint32_t s_id[NGS_TILE_PIX];
    __shared__ float s_acc[NGS_TILE_PIX][ACC_USED];
    const int tid = threadIdx.x, lane = tid & 63;
    const int tile = blockIdx.x;
    const int tx = tile % grid_x, ty = tile / grid_x;
    const int px = tx * NGS_TILE_W + (tid & 15), py = ty * NGS_TILE_H + (tid >> 4);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];
    int todo = (int)(range.y - range.x);
    const int rounds = (todo + NGS_TILE_PIX - 1) / NGS_TILE_PIX;

    const size_t tp = (size_t)tile * NGS_TILE_PIX + tid;
    const float T_final = inside ? final_T[tp] : 0.0f;
    float T = T_final;
    uint32_t contributor = (uint32_t)todo;
    const uint32_t last_contributor = inside ? n_contrib[tp] : 0u;
    float dpr = 0.0f, dpg = 0.0f, dpb = 0.0f;
    if (inside) {
        const size_t pid = (size_t)W * py + px, plane = (size_t)W * H;
        dpr = dL_dpix[pid];
        dpg = dL_dpix[plane + pid];
        dpb = dL_dpix[2 * plane + pid];
    }
    float bgdot = 0.0f;
    bgdot += bg[0] * dpr;
    bgdot += bg[1] * dpg;
    bgdot += bg[2] * dpb;
    float rec_r = 0.0f, rec_g = 0.0f, rec_b = 0.0f, last_r = 0.0f, last_g = 0.0f, last_b = 0.0f, last_alpha = 0.0f;
    const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H);

    for (int r = 0; r < rounds; r++, todo -= NGS_TILE_PIX) {
        __syncthreads();  // previous batch fully flushed before its LDS is reused
        const uint32_t prog = (uint32_t)(r * NGS_TILE_PIX + tid);
        if (range.x + prog < range.y) {
            const uint32_t id = plist[range.y - prog - 1];
            s_id[tid] = id;
            s_a[tid] = g0[id];
            s_b[tid] = g1[id];
            s_c[tid] = g2[id];
        }
#pragma unroll
        for (int c = 0; c < ACC_USED; c++) s_acc[tid][c] = 0.0f;
        __syncthreads();
        const int nb = min(NGS_TILE_PIX, todo);
        // Every lane of the wave walks the whole batch (no per-lane early exit:
        // the wave-level reductions below need all 64 lanes present).
        for (int j = 0; j < nb; j++) {
            contributor--;
            bool live = inside && contributor < last_contributor;
            const float4 a = s_a[j];
            const float4 b = s_b[j];
            const float dx = a.x - pxf, dy = a.y - pyf;
            const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
            live = live && !(power > 0.0f);
            const float G = expf(power);
            const float alpha = fminf(0.99f, b.y * G);
            live = live && !(alpha < 1.0f / 255.0f);
            if (__ballot(live) == 0ull) continue;  // wave-uniform: nobody in this 16x4 strip is touched

            float v_dx = 0.0f, v_dy = 0.0f, v_ca = 0.0f, v_cb = 0.0f, v_cc = 0.0f, v_op = 0.0f, v_r = 0.0f,
                  v_g = 0.0f, v_b = 0.0f;
            if (live) {
                T = T / (1.f - alpha);
                const float w = alpha * T;
                const float cr = b.z, cg = b.w, cb = s_c[j];
                float dL_dalpha = 0.0f;
                rec_r = last_alpha * last_r + (1.f - last_alpha) * rec_r;
                last_r = cr;
                dL_dalpha += (cr - rec_r) * dpr;
                v_r = w * dpr;
                rec_g = last_alpha * last_g + (1.f - last_alpha) * rec_g;
                last_g = cg;
                dL_dalpha += (cg - rec_g) * dpg;
                v_g = w * dpg;
                rec_b = last_alpha * last_b + (1.f - last_alpha) * rec_b;
                last_b = cb;
                dL_dalpha += (cb - rec_b) * dpb;
                v_b = w * dpb;
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final / (1.f - alpha)) * bgdot;
                const float dL_dG = b.y * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * a.z - gdy * a.w;
                const float dG_ddely = -gdy * b.x - gdx * a.w;
                v_dx = dL_dG * dG_ddelx * ddelx_dx;
                v_dy = dL_dG * dG_ddely * ddely_dy;
                v_ca = -0.5f * gdx * dx * dL_dG;
                v_cb = -0.5f * gdx * dy * dL_dG;
                v_cc = -0.5f * gdy * dy * dL_dG;
                v_op = G * dL_dalpha;
            }
            v_dx = wave_sum_lane63(v_dx);
            v_dy = wave_sum_lane63(v_dy);
            v_ca = wave_sum_lane63(v_ca);
            v_cb = wave_sum_lane63(v_cb);
            v_cc = wave_sum_lane63(v_cc);
            v_op = wave_sum_lane63(v_op);
            v_r = wave_sum_lane63(v_r);
            v_g = wave_sum_lane63(v_g);
            v_b = wave_sum_lane63(v_b);
            if (lane == 63) {
                atomicAdd(&s_acc[j][ACC_DX], v_dx);
                atomicAdd(&s_acc[j][ACC_DY], v_dy);
                atomicAdd(&s_acc[j][ACC_CON_A], v_ca);
                atomicAdd(&s_acc[j][ACC_CON_B], v_cb);
                atomicAdd(&s_acc[j][ACC_CON_C], v_cc);
                atomicAdd(&s_acc[j][ACC_OPAC], v_op);
                atomicAdd(&s_acc[j][ACC_R], v_r);
                atomicAdd(&s_acc[j][ACC_G], v_g);
                atomicAdd(&s_acc[j][ACC_B], v_b);
            }
        }
        __syncthreads();
        // Flush: 16 lanes per splat, lane c adds slot c of the splat's 64-byte accumulator row.
        const int c = tid & 15;
        for (int j = tid >> 4; j < nb; j += NGS_TILE_PIX / 16) {
            if (c < ACC_USED) {
                const float v = s_acc[j][c];
                if (v != 0.0f) atomicAdd(acc + (size_t)s_id[j] * NGS_ACC_STRIDE + c, v);
            }
        }
    }
}

void ngs_launch_render_bwd(hipStream_t s, const NgsCamera& cam, const uint2* ranges, const uint32_t* plist,
                           const float4* g0, const float4* g1, const float* g2, const float* bg, const float* final_T,
                           const uint32_t* n_contrib, const float* dL_dpix, float* acc)
{
    hipLaunchKernelGGL(render_bwd_kernel, dim3(cam.grid_x * cam.grid_y), dim3(NGS_TILE_PIX), 0, s, cam.W, cam.H,
                       cam.grid_x, ranges, plist, g0, g1, g2, bg, final_T, n_contrib, dL_dpix, acc);
}
