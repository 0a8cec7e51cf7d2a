// Per-tile back-to-front compositing gradient.  Same tiling and LDS staging
// as the forward kernel, traversed in reverse; per-pixel arithmetic follows
// cuda_rasterizer/backward.cu:399-557 of the reference.  What differs:
//
//  * The traversal starts at the tile's deepest contributor (the maximum of
//    the per-pixel last-contributor indices the forward pass stored) instead
//    of at the end of the tile's list: entries behind it are skipped by every
//    pixel in the reference too (backward.cu:486-488), so they are never staged.
//  * Each wave ballots which staged splats can reach its 8x8 block AND lie in
//    front of the block's own deepest contributor, and walks only those.
//  * Reduction: the reference issues nine global float atomics per (pixel,
//    splat) pair.  Here all 64 lanes of a wave handle the same splat in
//    lock-step, so the nine partials are summed across the wave in registers
//    (lane-half / row swaps + DPP row shifts, a reduce-scatter), then
//    across the tile's four waves in an LDS table holding one row per staged
//    splat, and each splat's row leaves the workgroup once per batch as ONE
//    64-byte-row group of global atomics into the [P][16] accumulator.
#include "render_common.h"

typedef float v2f __attribute__((ext_vector_type(2)));  // pairs of channels go through the packed-FP32 instructions

// [a summed over lane pairs (l, l+32) in lanes 0-31 | b likewise in lanes 32-63]
__device__ __forceinline__ float fold32(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b),
                                                    false, false);
    unsigned x = r[0], y = r[1];
    // hipcc (ROCm 7.2) folds "r[0] + r[1]" of a lane swap into "r[0] + r[0]"; an opaque use of the second result
    // keeps the two registers apart.
    asm volatile("" : "+v"(y));
    return __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y);
}

// rows 0,2: a summed over row pairs (0,1) / (2,3); rows 1,3: b likewise
__device__ __forceinline__ float fold16(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b),
                                                    false, false);
    unsigned x = r[0], y = r[1];
    // hipcc (ROCm 7.2) folds "r[0] + r[1]" of a lane swap into "r[0] + r[0]"; an opaque use of the second result
    // keeps the two registers apart.
    asm volatile("" : "+v"(y));
    return __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y);
}

// Sum over each 16-lane row; valid in lane 15 of the row.
__device__ __forceinline__ float row_sum_lane15(float v)
{
#define NGS_ROW_ADD(ctrl) \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
    NGS_ROW_ADD(0x111);  // row_shr:1
    NGS_ROW_ADD(0x112);  // row_shr:2
    NGS_ROW_ADD(0x114);  // row_shr:4
    NGS_ROW_ADD(0x118);  // row_shr:8
#undef NGS_ROW_ADD
    return v;
}

__global__ __launch_bounds__(NGS_TILE_PIX) void render_bwd_kernel(
    int W, int H, int grid_x, const uint2* __restrict__ ranges, const uint32_t* __restrict__ plist,
    const float4* __restrict__ splat_rec, const float* __restrict__ bg, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dpix, float* __restrict__ acc)
{
    // record k: [0] = {x, y, ln(255 opacity), -}, [1] = {conic a, b, c, opacity}, [2] = {r, g, b, Gaussian id bits}
    __shared__ float4 s_rec[NGS_TILE_PIX][3];
    __shared__ float s_acc[NGS_TILE_PIX][ACC_USED];
    __shared__ uint32_t s_wmax[NGS_TILE_PIX / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const int tx = tile % grid_x, ty = tile / grid_x;
    const int px = tx * NGS_TILE_W + tile_local_x(tid), py = ty * NGS_TILE_H + tile_local_y(tid);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float sx0 = (float)(tx * NGS_TILE_W + (wave & 1) * NGS_WAVE_W), sx1 = sx0 + (NGS_WAVE_W - 1);
    const float sy0 = (float)(ty * NGS_TILE_H + (wave >> 1) * NGS_WAVE_H), sy1 = sy0 + (NGS_WAVE_H - 1);
    const uint2 range = ranges[tile];

    const size_t tp = (size_t)tile * NGS_TILE_PIX + tid;
    const float T_final = inside ? final_T[tp] : 0.0f;
    float T = T_final;
    const uint32_t last_contributor = inside ? n_contrib[tp] : 0u;
    // deepest contributor of this block (wave) and of the tile
    uint32_t wmax = last_contributor;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, o, 64));
    if (lane == 0) s_wmax[wave] = wmax;
    __syncthreads();
    const uint32_t tmax = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));
    if (tmax == 0) return;  // nothing in this tile was blended (block-uniform)

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
    v2f rec_rg = {0.0f, 0.0f};
    float rec_b = 0.0f;
    const v2f dp_rg = {dpr, dpg};
    const v2f ddel = {(float)(0.5 * W), (float)(0.5 * H)};  // d(pixel offset)/d(ndc)
    const v2f pix = {pxf, pyf};
    // accumulator slots this lane's 16-lane row ends up holding after the reduce-scatter below
    const int lrow = lane >> 4;
    const int slot0 = lrow == 0 ? ACC_DX : lrow == 1 ? ACC_CON_B : lrow == 2 ? ACC_OPAC : ACC_B;
    const int slot1 = lrow == 0 ? ACC_DY : lrow == 1 ? ACC_CON_C : lrow == 2 ? ACC_R : -1;
    const int slot2 = lrow == 0 ? ACC_CON_A : lrow == 2 ? ACC_G : -1;

    // Zero-based list positions tmax-1 ... 0 are walked in that order, 256 per batch;
    // staged slot s of batch r holds list position tmax-1 - r*256 - s.
    const int rounds = (int)((tmax + NGS_TILE_PIX - 1) / NGS_TILE_PIX);
    for (int r = 0; r < rounds; r++) {
        __syncthreads();  // previous batch fully flushed before its LDS is reused
        const int first = (int)tmax - 1 - r * NGS_TILE_PIX;  // list position of slot 0
        const int nb = min(NGS_TILE_PIX, first + 1);
        if (tid < nb) {
            const uint32_t id = plist[range.x + (uint32_t)(first - tid)];
            const float4* gr = splat_rec + 4 * (size_t)id;  // one 64-byte line per splat
            const float4 a = gr[0];
            const float4 b = gr[1];
            const float c = gr[2].x;
            s_rec[tid][0] = make_float4(a.x, a.y, splat_tau(b.y), 0.0f);
            s_rec[tid][1] = make_float4(a.z, a.w, b.x, b.y);
            s_rec[tid][2] = make_float4(b.z, b.w, c, __uint_as_float(id));
        }
#pragma unroll
        for (int c = 0; c < ACC_USED; c++) s_acc[tid][c] = 0.0f;
        __syncthreads();
        for (int chunk = 0; chunk < nb; chunk += 64) {
            const int mine = chunk + lane;
            bool reach = false;
            if (mine < nb && (uint32_t)(first - mine) < wmax) {
                const float4 p = s_rec[mine][0];
                const float4 q = s_rec[mine][1];
                reach = splat_reaches(p.x, p.y, q.x, q.y, q.z, p.z, sx0, sx1, sy0, sy1);
            }
            unsigned long long mask = __ballot(reach);
            while (mask) {
                const int j = chunk + lowest_lane(mask);
                mask &= mask - 1ull;
                const uint32_t pos = (uint32_t)(first - j);  // the reference's `contributor` after its decrement
                const float4 p = s_rec[j][0];
                const float4 q = s_rec[j][1];
                const float4 col = s_rec[j][2];
                const v2f d = (v2f){p.x, p.y} - pix;  // {dx, dy}
                const float dx = d.x, dy = d.y;
                // same evaluation order as the forward pass (and the reference): alpha must come out bit-identical
                const float power = -0.5f * (q.x * dx * dx + q.z * dy * dy) - q.y * dx * dy;
                const float G = exp_nonpos(power);
                const float alpha = fminf(0.99f, q.w * G);
                const bool live = pos < last_contributor && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
                if (__ballot(live) == 0ull) continue;  // wave-uniform

                // Every lane evaluates the body.  A dead lane (pixel that does not blend this splat) runs it with
                // alpha = 0, which leaves its state untouched without a single select: T / (1 - 0) = T, the
                // colour behind it becomes 0 * colour + 1 * rec = rec, and every partial below is an exact zero
                // (G and dL/dalpha are forced to 0 as well).
                const float alpha_l = live ? alpha : 0.0f;
                const float om = 1.f - alpha_l;
                const float inv = __builtin_amdgcn_rcpf(om);  // v_rcp_f32, 1 ulp; exactly 1 for a dead lane
                const float Tn = T * inv;
                const float w = alpha_l * Tn;
                const v2f col_rg = {col.x, col.y};
                // `rec` is the colour accumulated behind this splat (the reference's accum_rec after its update,
                // backward.cu:503-513)
                const v2f d_rg = (col_rg - rec_rg) * dp_rg;
                float dL_dalpha = d_rg.x + d_rg.y;
                dL_dalpha += (col.z - rec_b) * dpb;
                dL_dalpha *= Tn;
                dL_dalpha += (-T_final * inv) * bgdot;
                dL_dalpha = live ? dL_dalpha : 0.0f;
                const float Gl = live ? G : 0.0f;
                const float dL_dG = q.w * dL_dalpha;
                const v2f gd = Gl * d;  // {G dx, G dy}
                const v2f gx = {gd.x, gd.x}, gy = {gd.y, gd.y};
                // {dG/ddelx, dG/ddely} = -gdx * {a, b} - gdy * {b, c}
                const v2f dG = -gx * (v2f){q.x, q.y} - gy * (v2f){q.y, q.z};
                const v2f v01 = (dL_dG * dG) * ddel;            // ACC_DX, ACC_DY
                const float v0 = v01.x, v1 = v01.y;
                const v2f h = -0.5f * gd;
                const v2f v23 = ((v2f){h.x, h.x} * d) * dL_dG;  // ACC_CON_A, ACC_CON_B
                const float v2 = v23.x, v3 = v23.y;
                const float v4 = h.y * dy * dL_dG;              // ACC_CON_C
                const float v5 = Gl * dL_dalpha;                // ACC_OPAC
                const v2f v67 = w * dp_rg;                      // ACC_R, ACC_G
                const float v6 = v67.x, v7 = v67.y;
                const float v8 = w * dpb;                       // ACC_B
                // state for the next (nearer) splat: transmittance in front of this one, and the colour behind
                // the next one = this splat blended over what was behind it (same expression the reference
                // evaluates one step later from its last_alpha / last_color)
                T = Tn;
                rec_rg = alpha_l * col_rg + om * rec_rg;
                rec_b = alpha_l * col.z + om * rec_b;

                // Wave reduction of the nine partials as a reduce-scatter: two half/row swaps fold the 64 lanes
                // into 16 columns while spreading the nine values over the four 16-lane rows of three registers,
                // then one 4-step DPP row reduction per register leaves each value's total in lane 15 of its row.
                //   s0 rows: [v0, v3, v5, v8]   s1 rows: [v1, v4, v6, -]   s2 rows: [v2, -, v7, -]
                const float r0 = fold32(v0, v5), r1 = fold32(v1, v6), r2 = fold32(v2, v7), r3 = fold32(v3, v8),
                            r4 = fold32(v4, 0.0f);
                float s0 = fold16(r0, r3), s1 = fold16(r1, r4), s2 = fold16(r2, 0.0f);
                s0 = row_sum_lane15(s0);
                s1 = row_sum_lane15(s1);
                s2 = row_sum_lane15(s2);
                if ((lane & 15) == 15) {
                    float* row = &s_acc[j][0];
                    atomicAdd(row + slot0, s0);
                    if (slot1 >= 0) atomicAdd(row + slot1, s1);
                    if (slot2 >= 0) atomicAdd(row + slot2, s2);
                }
            }
        }
        __syncthreads();
        // Flush: 16 lanes per splat, lane c adds slot c of the splat's 64-byte accumulator row.
        const int c = tid & 15;
        for (int j = tid >> 4; j < nb; j += NGS_TILE_PIX / 16) {
            if (c < ACC_USED) {
                const float v = s_acc[j][c];
                if (v != 0.0f)
                    atomicAdd(acc + (size_t)__float_as_uint(s_rec[j][2].w) * NGS_ACC_STRIDE + c, v);
            }
        }
    }
}

void ngs_launch_render_bwd(hipStream_t s, const NgsCamera& cam, const uint2* ranges, const uint32_t* plist,
                           const float4* rec, const float* bg, const float* final_T,
                           const uint32_t* n_contrib, const float* dL_dpix, float* acc)
{
    hipLaunchKernelGGL(render_bwd_kernel, dim3(cam.grid_x * cam.grid_y), dim3(NGS_TILE_PIX), 0, s, cam.W, cam.H,
                       cam.grid_x, ranges, plist, rec, bg, final_T, n_contrib, dL_dpix, acc);
}
