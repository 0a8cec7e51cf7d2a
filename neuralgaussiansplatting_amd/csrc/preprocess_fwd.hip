// Per-Gaussian forward stage: near cull, projection, EWA screen-space
// covariance, conic + radius, tile rectangle, SH -> RGB.  One lane per
// Gaussian; outputs go to the packed SoA records the compositing kernels
// stage into LDS.  Follows cuda_rasterizer/forward.cu:155-256 and
// auxiliary.h:41-56,139-164 of the reference.  Built with -ffp-contract=off:
// radii, rectangles and depth bits must equal the CPU checker's exactly.
#include "render_common.h"

// auxiliary.h:41-44 (double arithmetic, rounded once to float)
__device__ __forceinline__ float ndc_to_pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

// forward.cu:20-71.  `sh` points at this Gaussian's M x 3 coefficients.
__device__ __forceinline__ float3 sh_to_rgb(int deg, const float* __restrict__ sh, float3 dir, uint32_t& clamp_bits)
{
    const float3* c = (const float3*)sh;
    float3 res = f3_scale(NGS_SH_C0, c[0]);
    if (deg > 0) {
        const float x = dir.x, y = dir.y, z = dir.z;
        res = f3_sub(f3_add(f3_sub(res, f3_scale(NGS_SH_C1 * y, c[1])), f3_scale(NGS_SH_C1 * z, c[2])),
                     f3_scale(NGS_SH_C1 * x, c[3]));
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            res = f3_add(res, f3_scale(NGS_SH_C2_0 * xy, c[4]));
            res = f3_add(res, f3_scale(NGS_SH_C2_1 * yz, c[5]));
            res = f3_add(res, f3_scale(NGS_SH_C2_2 * (2.0f * zz - xx - yy), c[6]));
            res = f3_add(res, f3_scale(NGS_SH_C2_3 * xz, c[7]));
            res = f3_add(res, f3_scale(NGS_SH_C2_4 * (xx - yy), c[8]));
            if (deg > 2) {
                res = f3_add(res, f3_scale(NGS_SH_C3_0 * y * (3.0f * xx - yy), c[9]));
                res = f3_add(res, f3_scale(NGS_SH_C3_1 * xy * z, c[10]));
                res = f3_add(res, f3_scale(NGS_SH_C3_2 * y * (4.0f * zz - xx - yy), c[11]));
                res = f3_add(res, f3_scale(NGS_SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy), c[12]));
                res = f3_add(res, f3_scale(NGS_SH_C3_4 * x * (4.0f * zz - xx - yy), c[13]));
                res = f3_add(res, f3_scale(NGS_SH_C3_5 * z * (xx - yy), c[14]));
                res = f3_add(res, f3_scale(NGS_SH_C3_6 * x * (xx - 3.0f * yy), c[15]));
            }
        }
    }
    res.x += 0.5f; res.y += 0.5f; res.z += 0.5f;
    clamp_bits = (res.x < 0 ? 1u : 0u) | (res.y < 0 ? 2u : 0u) | (res.z < 0 ? 4u : 0u);
    return make_float3(fmaxf(res.x, 0.0f), fmaxf(res.y, 0.0f), fmaxf(res.z, 0.0f));
}

// The same sum taken in two parts, coefficients 0..7 then 8..15 of a degree-3 row, so that a caller can hold half a
// row at a time; the additions happen in the order above.
__device__ __forceinline__ float3 sh_to_rgb_first8(const float3* c, float3 dir)
{
    const float x = dir.x, y = dir.y, z = dir.z;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    float3 res = f3_scale(NGS_SH_C0, c[0]);
    res = f3_sub(f3_add(f3_sub(res, f3_scale(NGS_SH_C1 * y, c[1])), f3_scale(NGS_SH_C1 * z, c[2])),
                 f3_scale(NGS_SH_C1 * x, c[3]));
    res = f3_add(res, f3_scale(NGS_SH_C2_0 * xy, c[4]));
    res = f3_add(res, f3_scale(NGS_SH_C2_1 * yz, c[5]));
    res = f3_add(res, f3_scale(NGS_SH_C2_2 * (2.0f * zz - xx - yy), c[6]));
    res = f3_add(res, f3_scale(NGS_SH_C2_3 * xz, c[7]));
    return res;
}
__device__ __forceinline__ float3 sh_to_rgb_last8(float3 res, const float3* c8, float3 dir, uint32_t& clamp_bits)
{
    const float x = dir.x, y = dir.y, z = dir.z;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y;
    res = f3_add(res, f3_scale(NGS_SH_C2_4 * (xx - yy), c8[0]));
    res = f3_add(res, f3_scale(NGS_SH_C3_0 * y * (3.0f * xx - yy), c8[1]));
    res = f3_add(res, f3_scale(NGS_SH_C3_1 * xy * z, c8[2]));
    res = f3_add(res, f3_scale(NGS_SH_C3_2 * y * (4.0f * zz - xx - yy), c8[3]));
    res = f3_add(res, f3_scale(NGS_SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy), c8[4]));
    res = f3_add(res, f3_scale(NGS_SH_C3_4 * x * (4.0f * zz - xx - yy), c8[5]));
    res = f3_add(res, f3_scale(NGS_SH_C3_5 * z * (xx - yy), c8[6]));
    res = f3_add(res, f3_scale(NGS_SH_C3_6 * x * (xx - 3.0f * yy), c8[7]));
    res.x += 0.5f; res.y += 0.5f; res.z += 0.5f;
    clamp_bits = (res.x < 0 ? 1u : 0u) | (res.y < 0 ? 2u : 0u) | (res.z < 0 ? 4u : 0u);
    return make_float3(fmaxf(res.x, 0.0f), fmaxf(res.y, 0.0f), fmaxf(res.z, 0.0f));
}

// LDS staging of the EAGER kernel: a wave holds HALF of its 64 SH rows at a time, as six slots of 64 float4 (slot c of
// row g at float4 index c * SH_SLOT + g; the odd slot stride keeps both the row-major writes and the lane-major reads
// free of bank conflicts).  6.1 KiB per wave instead of 13: LDS no longer caps the waves per SIMD.
#define SH_SLOT 65
#define SH_WAVE_F4 (6 * SH_SLOT)

// EAGER: the common training shape (scales + rotations, full SH rows of 16 coefficients at degree 3).  All of a
// Gaussian's inputs are requested before any arithmetic: the culling tests that decide whether they are needed
// depend on one another (position -> covariance -> footprint -> colour), and waiting for each answer before asking
// for the next operand costs a memory round trip per stage.  A culled Gaussian's few extra bytes are cheaper.
template <bool EAGER>
__global__ __launch_bounds__(NGS_GAUSS_BLOCK) void preprocess_fwd_kernel(
    int P, int D, int M, const float* __restrict__ means, const float* __restrict__ scales, float mod,
    const float* __restrict__ rots, const float* __restrict__ opac, int raw, const float* __restrict__ shs,
    const float* __restrict__ cov_pre, const float* __restrict__ col_pre, const NgsCamera cam,
    const float* __restrict__ view, const float* __restrict__ proj, const float* __restrict__ campos, int prefiltered,
    int* __restrict__ radii, float4* __restrict__ rec, float* __restrict__ depth, uint2* __restrict__ rect, uint32_t* __restrict__ tiles, uint8_t* __restrict__ clamped,
    uint32_t* __restrict__ block_sums, uint32_t* __restrict__ block_depths, uint32_t* __restrict__ misc,
    uint32_t* __restrict__ dkeys, uint32_t* __restrict__ order)
{
    __shared__ uint32_t wave_tot[NGS_GAUSS_BLOCK / 64];
    __shared__ uint32_t wave_dmin[NGS_GAUSS_BLOCK / 64], wave_dmax[NGS_GAUSS_BLOCK / 64];
    __shared__ float4 s_sh[EAGER ? (NGS_GAUSS_BLOCK / 64) * SH_WAVE_F4 : 1];
    const int idx = blockIdx.x * NGS_GAUSS_BLOCK + threadIdx.x;
    uint32_t my_tiles = 0;
    bool filtered_anyway = false;  // culled by the depth test although the caller said the points were prefiltered
    int my_radius = 0;
    float4 ld[EAGER ? 12 : 1];  // this lane's share of the wave's SH rows on their way to LDS
    float4* const wslots = s_sh + (EAGER ? (threadIdx.x >> 6) * SH_WAVE_F4 : 0);
    if (EAGER) {
        // Each wave moves its 64 SH rows (12 KiB, contiguous in memory) through LDS with coalesced streaming loads:
        // a lane reading its own 192-byte row would make every load instruction touch 64 different cache lines.
        // All twelve loads are issued here; the first half of every row goes to LDS now, the second half waits in
        // registers until the first has been consumed.  Rows are written and read by the same wave only: no
        // workgroup barrier is needed, LDS operations of a wave complete in order.
        const int lane = threadIdx.x & 63;
        const int wave_first = blockIdx.x * NGS_GAUSS_BLOCK + (threadIdx.x >> 6) * 64;
#pragma unroll
        for (int t = 0; t < 12; t++) {
            const int f = t * 64 + lane, g = f / 12, c = f - g * 12;
            ld[t] = make_float4(0, 0, 0, 0);
            if (wave_first + g < P) ld[t] = stream_load4(shs + (size_t)(wave_first + g) * 48 + 4 * c);
        }
#pragma unroll
        for (int t = 0; t < 12; t++) {
            const int f = t * 64 + lane, g = f / 12, c = f - g * 12;
            if (c < 6) wslots[c * SH_SLOT + g] = ld[t];
        }
    }
    // Geometry first (projection, footprint, tile rectangle); `live` = the Gaussian produces instances.
    bool live = false;
    float px = 0, py = 0, pz = 0, e_op = 0, view_z = 0, ix = 0, iy = 0, con_x = 0, con_y = 0, con_z = 0;
    uint32_t x0 = 0, x1 = 0, y0 = 0, y1 = 0;
    if (idx < P) {
        px = means[3 * idx]; py = means[3 * idx + 1]; pz = means[3 * idx + 2];
        float4 e_q = make_float4(0, 0, 0, 0);
        float e_s0 = 0, e_s1 = 0, e_s2 = 0;
        if (EAGER) {
            e_q = ((const float4*)rots)[idx];
            e_s0 = scales[3 * idx]; e_s1 = scales[3 * idx + 1]; e_s2 = scales[3 * idx + 2];
            e_op = opac[idx];
            if (raw) {  // the optimiser's raw parameters: activations of scene/gaussian_model.py:36-43
                float len;
                e_q = normalize_quat(e_q, len);
                e_s0 = expf(e_s0); e_s1 = expf(e_s1); e_s2 = expf(e_s2);
                e_op = sigmoid_act(e_op);
            }
        }
        const float3 pv = xform43(px, py, pz, view);
        view_z = pv.z;
        if (pv.z <= 0.2f) {
            // The reference aborts the kernel here when `prefiltered` is set
            // (auxiliary.h:156-160); this library raises an error flag that the
            // host turns into an error return instead of faulting the GPU.
            filtered_anyway = prefiltered != 0;
        } else {
            const float4 ph = xform44(px, py, pz, proj);
            const float pw = 1.0f / (ph.w + 0.0000001f);
            const float projx = ph.x * pw, projy = ph.y * pw;
            float c6[6];
            if (EAGER) {
                cov3d_from_scale_rot(e_s0, e_s1, e_s2, mod, e_q.x, e_q.y, e_q.z, e_q.w, c6);
            } else if (cov_pre) {
#pragma unroll
                for (int k = 0; k < 6; k++) c6[k] = cov_pre[6 * (size_t)idx + k];
            } else {
                // element loads: this kernel also serves quaternion arrays that are not 16-byte aligned
                const float* q = rots + 4 * (size_t)idx;
                float4 qv = make_float4(q[0], q[1], q[2], q[3]);
                float s0 = scales[3 * idx], s1 = scales[3 * idx + 1], s2 = scales[3 * idx + 2];
                if (raw) {
                    float len;
                    qv = normalize_quat(qv, len);
                    s0 = expf(s0); s1 = expf(s1); s2 = expf(s2);
                }
                cov3d_from_scale_rot(s0, s1, s2, mod, qv.x, qv.y, qv.z, qv.w, c6);
            }
            const Cov2D k = cov2d_eval(px, py, pz, cam, view, c6);
            const float det = k.a * k.c - k.b * k.b;
            if (det != 0.0f) {
                const float dinv = 1.f / det;
                const float mid = 0.5f * (k.a + k.c);
                const float l1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
                const float l2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
                const float rad = ceilf(3.f * sqrtf(fmaxf(l1, l2)));
                ix = ndc_to_pix(projx, cam.W); iy = ndc_to_pix(projy, cam.H);
                const int ri = (int)rad;  // v_cvt_i32_f32 saturates like the reference's conversion
                int a;
                a = max(0, (int)((ix - ri) / NGS_TILE_W)); x0 = min((uint32_t)cam.grid_x, (uint32_t)a);
                a = max(0, (int)((iy - ri) / NGS_TILE_H)); y0 = min((uint32_t)cam.grid_y, (uint32_t)a);
                a = max(0, (int)((ix + ri + NGS_TILE_W - 1) / NGS_TILE_W)); x1 = min((uint32_t)cam.grid_x, (uint32_t)a);
                a = max(0, (int)((iy + ri + NGS_TILE_H - 1) / NGS_TILE_H)); y1 = min((uint32_t)cam.grid_y, (uint32_t)a);
                const uint32_t area = (x1 - x0) * (y1 - y0);
                if (area != 0) {
                    live = true;
                    con_x = k.c * dinv; con_y = -k.b * dinv; con_z = k.a * dinv;
                    my_radius = ri;
                    my_tiles = area;
                }
            }
        }
    }
    // Colour, then the record.
    float3 col = make_float3(0, 0, 0);
    uint32_t cb = 0;
    if (EAGER) {
        const int lane = threadIdx.x & 63;
        float3 d = make_float3(px - campos[0], py - campos[1], pz - campos[2]);
        const float len = sqrtf(f3_dot(d, d));
        d = make_float3(d.x / len, d.y / len, d.z / len);
        float4 e_sh[6];
        float3 part = make_float3(0, 0, 0);
        if (live) {
#pragma unroll
            for (int k = 0; k < 6; k++) e_sh[k] = wslots[k * SH_SLOT + lane];
            part = sh_to_rgb_first8((const float3*)e_sh, d);
        }
        // second half of every row: stored by the whole wave, whichever of its lanes are live
#pragma unroll
        for (int t = 0; t < 12; t++) {
            const int f = t * 64 + lane, g = f / 12, c = f - g * 12;
            if (c >= 6) wslots[(c - 6) * SH_SLOT + g] = ld[t];
        }
        if (live) {
#pragma unroll
            for (int k = 0; k < 6; k++) e_sh[k] = wslots[k * SH_SLOT + lane];
            col = sh_to_rgb_last8(part, (const float3*)e_sh, d, cb);
            // the record is parked in this lane's (now consumed) LDS slots and leaves with the wave's other records
            // as coalesced 1-KiB stores below
            const float3 cull = record_cull_terms(con_x, con_y, con_z, e_op);
            wslots[0 * SH_SLOT + lane] = make_float4(ix, iy, con_x, con_y);
            wslots[1 * SH_SLOT + lane] = make_float4(con_z, e_op, col.x, col.y);
            wslots[2 * SH_SLOT + lane] = make_float4(col.z, cull.x, cull.y, cull.z);
            wslots[3 * SH_SLOT + lane] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    } else if (live) {
        if (col_pre) {
            col = make_float3(col_pre[3 * (size_t)idx], col_pre[3 * (size_t)idx + 1], col_pre[3 * (size_t)idx + 2]);
        } else {
            float3 d = make_float3(px - campos[0], py - campos[1], pz - campos[2]);
            const float len = sqrtf(f3_dot(d, d));
            d = make_float3(d.x / len, d.y / len, d.z / len);
            col = sh_to_rgb(D, shs + (size_t)3 * M * idx, d, cb);
        }
        float4* r = rec + 4 * (size_t)idx;
        r[0] = make_float4(ix, iy, con_x, con_y);
        const float op = raw ? sigmoid_act(opac[idx]) : opac[idx];
        const float3 cull = record_cull_terms(con_x, con_y, con_z, op);
        r[1] = make_float4(con_z, op, col.x, col.y);
        r[2] = make_float4(col.z, cull.x, cull.y, cull.z);
    }
    if (live) {
        depth[idx] = view_z;
        rect[idx] = make_uint2(x0 | (x1 << 16), y0 | (y1 << 16));
        clamped[idx] = (uint8_t)cb;
    }
    if (idx < P) {
        radii[idx] = my_radius;
        tiles[idx] = my_tiles;
        // input of the depth sort: the bits of a positive float order like the float; a culled Gaussian emits no
        // instance, so its place in the order does not matter -- it goes to the end
        if (dkeys) {
            dkeys[idx] = live ? __float_as_uint(view_z) : 0xffffffffu;
            order[idx] = (uint32_t)idx;
        }
    }
    if (EAGER) {
        // records of the wave's visible Gaussians: 4 KiB contiguous per wave (a culled Gaussian's record is never
        // read, so its slot is simply skipped)
        const int lane = threadIdx.x & 63;
        const int wave_first = blockIdx.x * NGS_GAUSS_BLOCK + (threadIdx.x >> 6) * 64;
        const unsigned long long vis = __ballot(my_tiles != 0);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int f = t * 64 + lane, g = f >> 2, c = f & 3;
            if ((vis >> g) & 1ull) rec[4 * (size_t)(wave_first + g) + c] = wslots[c * SH_SLOT + g];
        }
    }
    // Tiles touched by this block of 256 Gaussians: the first level of the instance-offset scan.
    // (the top bit of a wave's or a block's count says that one of its points was culled although prefiltered was
    // set; the scan kernel turns it into the error flag, so that no word has to be cleared before this kernel runs)
    const uint32_t ws = min(wave_sum_u32(my_tiles), 0x7fffffffu) | (__ballot(filtered_anyway) != 0ull ? 0x80000000u : 0u);
    // Depth extremes of the block's visible Gaussians: the frame's extremes fix the scale of the depth rank that
    // rides in the instances' sort keys.
    uint32_t dlo = live ? __float_as_uint(view_z) : 0xffffffffu, dhi = live ? __float_as_uint(view_z) : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        dlo = min(dlo, (uint32_t)__shfl_xor((int)dlo, o, 64));
        dhi = max(dhi, (uint32_t)__shfl_xor((int)dhi, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        wave_tot[threadIdx.x >> 6] = ws;
        wave_dmin[threadIdx.x >> 6] = dlo;
        wave_dmax[threadIdx.x >> 6] = dhi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t sum = 0, flag = 0, blo = 0xffffffffu, bhi = 0u;
#pragma unroll
        for (int w = 0; w < NGS_GAUSS_BLOCK / 64; w++) {
            sum += wave_tot[w] & 0x7fffffffu;
            flag |= wave_tot[w] & 0x80000000u;
            blo = min(blo, wave_dmin[w]);
            bhi = max(bhi, wave_dmax[w]);
        }
        block_sums[blockIdx.x] = min(sum, 0x7fffffffu) | flag;
        // bounds, 16 bits each: the smallest depth's high half, and the largest's plus one (a visible depth is a
        // positive finite float, so that still fits); 0xffff and 0 for a block without a visible Gaussian
        block_depths[blockIdx.x] = (blo >> 16) | (bhi ? min((bhi >> 16) + 1u, 0xffffu) << 16 : 0u);
    }
}

// rasterizer_impl.cu:54-66: present[i] = view-space z > 0.2
__global__ __launch_bounds__(256) void mark_visible_kernel(int P, const float* __restrict__ means,
                                                            const float* __restrict__ view,
                                                            uint8_t* __restrict__ present)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const float3 pv = xform43(means[3 * idx], means[3 * idx + 1], means[3 * idx + 2], view);
    present[idx] = pv.z > 0.2f ? 1 : 0;
}

void ngs_launch_preprocess_fwd(hipStream_t s, int P, int D, int M, const float* means, const float* scales, float mod,
                               const float* rots, const float* opac, int raw, const float* shs, const float* cov_pre,
                               const float* col_pre, const NgsCamera& cam, int prefiltered, int* radii, float4* rec,
                               float* depth, uint2* rect, uint32_t* tiles, uint8_t* clamped,
                               uint32_t* block_sums, uint32_t* block_depths, uint32_t* misc, uint32_t* dkeys,
                               uint32_t* order)
{
    const int nb = (P + NGS_GAUSS_BLOCK - 1) / NGS_GAUSS_BLOCK;
    const bool eager = !cov_pre && !col_pre && shs && scales && rots && D == 3 && M == 16 && (uintptr_t)shs % 16 == 0 &&
                       (uintptr_t)rots % 16 == 0;
    if (eager)
        hipLaunchKernelGGL(preprocess_fwd_kernel<true>, dim3(nb), dim3(NGS_GAUSS_BLOCK), 0, s, P, D, M, means, scales, mod, rots,
                       opac, raw, shs, cov_pre, col_pre, cam, cam.view, cam.proj, cam.campos, prefiltered, radii, rec, depth, rect, tiles, clamped,
                       block_sums, block_depths, misc, dkeys, order);
    else
        hipLaunchKernelGGL(preprocess_fwd_kernel<false>, dim3(nb), dim3(NGS_GAUSS_BLOCK), 0, s, P, D, M, means, scales,
                           mod, rots, opac, raw, shs, cov_pre, col_pre, cam, cam.view, cam.proj, cam.campos, prefiltered,
                           radii, rec, depth, rect, tiles, clamped, block_sums, block_depths, misc, dkeys, order);
}

void ngs_launch_mark_visible(hipStream_t s, int P, const float* means, const float* view, uint8_t* present)
{
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means, view, present);
}
