// Per-Gaussian backward stage, fused: screen-space covariance backward,
// perspective-projection backward, SH backward and scale/rotation backward in
// one pass over the Gaussians (the reference runs two kernels,
// cuda_rasterizer/backward.cu:144-274 and :346-396, with the world covariance
// and its gradient making a round trip through memory between them).  The
// kernel also converts the 64-byte accumulator rows written by the compositing
// backward into the caller's gradient tensors and writes zeros for culled
// Gaussians, so none of the outputs needs a separate zero fill.
//
// The SH coefficients and their gradients are 192-byte rows (M = 16), two
// thirds of this kernel's traffic.  A lane touching its own row makes every
// wave instruction hit 64 different cache lines, so for M = 16 each wave moves
// its 64 rows (12 KiB, contiguous in memory) through LDS with fully coalesced
// 16-byte accesses and the lanes work on rows in LDS (row stride padded to
// 52 floats: conflict-free 128-bit reads).
// Built with -ffp-contract=off so the recomputed covariance equals the
// forward's bit for bit.
#include <string.h>
#include "ngs_common.h"

#define SH_ROW 48      // floats per Gaussian when M == 16
#define SH_ROW_LDS 52  // padded LDS row stride

// backward.cu:20-139.  Writes dL_dsh[0..M) for this Gaussian (zeros above the
// active degree) unless `dsh_out` is NULL, and returns the gradient wrt the mean
// through the view direction.  `shs` and `dsh_out` may be the SAME row (LDS
// staging): every read of a coefficient happens before the first write.
__device__ __forceinline__ float3 sh_backward(int deg, int M, float3 pos, const float* __restrict__ campos,
                                              const float* shs, uint32_t clamp_bits, float3 dRGB, float* dsh_out)
{
    const float3 d0 = make_float3(pos.x - campos[0], pos.y - campos[1], pos.z - campos[2]);
    const float len = sqrtf(f3_dot(d0, d0));
    const float x = d0.x / len, y = d0.y / len, z = d0.z / len;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    const float3* sh = (const float3*)shs;
    float3* g = (float3*)dsh_out;
    dRGB.x *= (clamp_bits & 1u) ? 0.0f : 1.0f;
    dRGB.y *= (clamp_bits & 2u) ? 0.0f : 1.0f;
    dRGB.z *= (clamp_bits & 4u) ? 0.0f : 1.0f;

    // ---- phase 1 (reads): derivative of the colour wrt the view direction ----
    float3 ddx = make_float3(0, 0, 0), ddy = ddx, ddz = ddx;
    if (deg > 0) {
        ddx = f3_scale(-NGS_SH_C1, sh[3]);
        ddy = f3_scale(-NGS_SH_C1, sh[1]);
        ddz = f3_scale(NGS_SH_C1, sh[2]);
        if (deg > 1) {
            float3 s;
            s = f3_scale(NGS_SH_C2_0 * y, sh[4]);
            s = f3_add(s, f3_scale(NGS_SH_C2_2 * 2.f * -x, sh[6]));
            s = f3_add(s, f3_scale(NGS_SH_C2_3 * z, sh[7]));
            s = f3_add(s, f3_scale(NGS_SH_C2_4 * 2.f * x, sh[8]));
            ddx = f3_add(ddx, s);
            s = f3_scale(NGS_SH_C2_0 * x, sh[4]);
            s = f3_add(s, f3_scale(NGS_SH_C2_1 * z, sh[5]));
            s = f3_add(s, f3_scale(NGS_SH_C2_2 * 2.f * -y, sh[6]));
            s = f3_add(s, f3_scale(NGS_SH_C2_4 * 2.f * -y, sh[8]));
            ddy = f3_add(ddy, s);
            s = f3_scale(NGS_SH_C2_1 * y, sh[5]);
            s = f3_add(s, f3_scale(NGS_SH_C2_2 * 2.f * 2.f * z, sh[6]));
            s = f3_add(s, f3_scale(NGS_SH_C2_3 * x, sh[7]));
            ddz = f3_add(ddz, s);
            if (deg > 2) {
                // every term is (constant * coefficient) scaled by the remaining factors left to right
                s = f3_scale(xy, f3_scale(2.f, f3_scale(3.f, f3_scale(NGS_SH_C3_0, sh[9]))));
                s = f3_add(s, f3_scale(yz, f3_scale(NGS_SH_C3_1, sh[10])));
                s = f3_add(s, f3_scale(xy, f3_scale(-2.f, f3_scale(NGS_SH_C3_2, sh[11]))));
                s = f3_add(s, f3_scale(xz, f3_scale(2.f, f3_scale(-3.f, f3_scale(NGS_SH_C3_3, sh[12])))));
                s = f3_add(s, f3_scale(-3.f * xx + 4.f * zz - yy, f3_scale(NGS_SH_C3_4, sh[13])));
                s = f3_add(s, f3_scale(xz, f3_scale(2.f, f3_scale(NGS_SH_C3_5, sh[14]))));
                s = f3_add(s, f3_scale(xx - yy, f3_scale(3.f, f3_scale(NGS_SH_C3_6, sh[15]))));
                ddx = f3_add(ddx, s);
                s = f3_scale(xx - yy, f3_scale(3.f, f3_scale(NGS_SH_C3_0, sh[9])));
                s = f3_add(s, f3_scale(xz, f3_scale(NGS_SH_C3_1, sh[10])));
                s = f3_add(s, f3_scale(-3.f * yy + 4.f * zz - xx, f3_scale(NGS_SH_C3_2, sh[11])));
                s = f3_add(s, f3_scale(yz, f3_scale(2.f, f3_scale(-3.f, f3_scale(NGS_SH_C3_3, sh[12])))));
                s = f3_add(s, f3_scale(xy, f3_scale(-2.f, f3_scale(NGS_SH_C3_4, sh[13]))));
                s = f3_add(s, f3_scale(yz, f3_scale(-2.f, f3_scale(NGS_SH_C3_5, sh[14]))));
                s = f3_add(s, f3_scale(xy, f3_scale(2.f, f3_scale(-3.f, f3_scale(NGS_SH_C3_6, sh[15])))));
                ddy = f3_add(ddy, s);
                s = f3_scale(xy, f3_scale(NGS_SH_C3_1, sh[10]));
                s = f3_add(s, f3_scale(yz, f3_scale(2.f, f3_scale(4.f, f3_scale(NGS_SH_C3_2, sh[11])))));
                s = f3_add(s, f3_scale(2.f * zz - xx - yy, f3_scale(3.f, f3_scale(NGS_SH_C3_3, sh[12]))));
                s = f3_add(s, f3_scale(xz, f3_scale(2.f, f3_scale(4.f, f3_scale(NGS_SH_C3_4, sh[13])))));
                s = f3_add(s, f3_scale(xx - yy, f3_scale(NGS_SH_C3_5, sh[14])));
                ddz = f3_add(ddz, s);
            }
        }
    }
    const float3 dl = make_float3(f3_dot(ddx, dRGB), f3_dot(ddy, dRGB), f3_dot(ddz, dRGB));
    // Jacobian of v/|v| (auxiliary.h:107-117)
    const float sum2 = d0.x * d0.x + d0.y * d0.y + d0.z * d0.z;
    const float inv32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    const float3 dmean =
        make_float3(((+sum2 - d0.x * d0.x) * dl.x - d0.y * d0.x * dl.y - d0.z * d0.x * dl.z) * inv32,
                    (-d0.x * d0.y * dl.x + (sum2 - d0.y * d0.y) * dl.y - d0.z * d0.y * dl.z) * inv32,
                    (-d0.x * d0.z * dl.x - d0.y * d0.z * dl.y + (sum2 - d0.z * d0.z) * dl.z) * inv32);
    // every coefficient has been read; a compiler-level fence keeps the writes below after the reads above
    __builtin_amdgcn_sched_barrier(0);

    // ---- phase 2 (writes): gradient of each coefficient; skipped when the caller does not want it ----
    if (g) {
        float w[16];
        sh_basis_weights(deg, x, y, z, w);
        const int coeffs = (deg + 1) * (deg + 1);
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (k < coeffs) g[k] = f3_scale(w[k], dRGB);
        for (int k = coeffs; k < M; k++) g[k] = make_float3(0.0f, 0.0f, 0.0f);
    }
    return dmean;
}

// SH_LDS: M == 16 and the SH rows travel through LDS (see the file header).
// RAW: scales / rots / opac_raw are the optimiser's raw parameters; only this variant carries the activations, their
// backward and the fused optimiser step (the plain one keeps the registers and kernel arguments of the drop-in path).
template <bool SH_LDS, bool RAW>
__global__ __launch_bounds__(NGS_GAUSS_BLOCK) void preprocess_bwd_kernel(
    int P, int D, int M, const float* means /* may alias geo.xyz.param */, const int* __restrict__ radii,
    const float* shs /* may alias adam.param */, const uint8_t* __restrict__ clamped,
    const float* scales /* may alias geo.scaling.param */, const float* rots /* may alias geo.rotation.param */,
    const float* opac_raw /* may alias geo.opacity.param */, float mod, const float* __restrict__ cov_pre,
    const NgsCamera cam, const float* __restrict__ view, const float* __restrict__ proj,
    const float* __restrict__ campos, const float* __restrict__ acc, float* __restrict__ dL_dmean2D,
    float* __restrict__ dL_dconic, float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolor,
    float* __restrict__ dL_dmean3D, float* __restrict__ dL_dcov3D, float* __restrict__ dL_dsh,
    float* __restrict__ dL_dscale, float* __restrict__ dL_drot, const ShAdamArgs adam, const GeoAdamArgs geo)
{
    __shared__ float s_sh[SH_LDS ? NGS_GAUSS_BLOCK * SH_ROW_LDS : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int idx = blockIdx.x * NGS_GAUSS_BLOCK + tid;
    const size_t i = (size_t)idx;
    const bool in_range = idx < P;
    const bool vis = in_range && radii[idx] > 0;
    // this wave's 64 rows: Gaussians wave_first .. wave_first+63
    const int wave_first = blockIdx.x * NGS_GAUSS_BLOCK + wave * 64;
    float* const wrows = s_sh + wave * 64 * SH_ROW_LDS;
    float* const my_row = wrows + lane * SH_ROW_LDS;
    unsigned long long vis_mask = 0ull;
    if (SH_LDS) {
        vis_mask = __ballot(vis);
        if (D > 0) {  // degree 0 never reads a coefficient
#pragma unroll
            for (int t = 0; t < SH_ROW * 64 / 4 / 64; t++) {
                const int f = t * 64 + lane, g = f / (SH_ROW / 4), c = f - g * (SH_ROW / 4);
                if (wave_first + g < P && ((vis_mask >> g) & 1ull))
                    *(float4*)(wrows + g * SH_ROW_LDS + c * 4) =
                        stream_load4(shs + (size_t)(wave_first + g) * SH_ROW + 4 * c);
            }
        }
        __builtin_amdgcn_wave_barrier();  // rows are wave-private and a wave's LDS operations complete in order
    }

    // Everything this Gaussian hands on: zero unless it was rendered.
    float g_dx = 0, g_dy = 0, g_ca = 0, g_cb = 0, g_cc = 0, g_op = 0;
    float3 g_col = make_float3(0, 0, 0), dmean = make_float3(0, 0, 0);
    float dc[6] = {0, 0, 0, 0, 0, 0};
    float dsc[3] = {0, 0, 0};
    float4 dq = make_float4(0, 0, 0, 0);
    // raw parameters (read by every Gaussian that takes an optimiser step, rendered or not)
    const bool fused = RAW && geo.xyz.param != nullptr;
    float mx = 0, my = 0, mz = 0, sx = 0, sy = 0, sz = 0, op_in = 0;
    float4 q = make_float4(0, 0, 0, 0);
    if (vis || (fused && in_range)) {
        mx = means[3 * i]; my = means[3 * i + 1]; mz = means[3 * i + 2];
        if (!cov_pre) {
            sx = scales[3 * i]; sy = scales[3 * i + 1]; sz = scales[3 * i + 2];
            q = ((const float4*)rots)[i];
        }
        if (RAW) op_in = opac_raw[i];
    }
    // raw -> activated (scene/gaussian_model.py:36-43); the raw values stay in *_in for the chain rule below
    const float sx_in = sx, sy_in = sy, sz_in = sz;
    const float4 q_in = q;
    float q_len = 1.0f, op_act = 0.0f;
    if (RAW) {
        sx = expf(sx); sy = expf(sy); sz = expf(sz);
        q = normalize_quat(q, q_len);
        op_act = sigmoid_act(op_in);
    }
    if (vis) {
        const float4* row = (const float4*)(acc + i * NGS_ACC_STRIDE);
        const float4 r0 = row[0], r1 = row[1];
        const float r2 = acc[i * NGS_ACC_STRIDE + 8];
        // The compositing pass leaves, per Gaussian, the moments of dL/dG over the pixel offsets d it was blended at:
        // m1 = sum dL/dG G d (r0.x, r0.y) and m2 = sum dL/dG G (dx^2, dx dy, dy^2) (r0.z, r0.w, r1.x); the conic that
        // turns them into the gradients of backward.cu:545-551 is the same for every pixel and is applied below.
        const float m1x = r0.x, m1y = r0.y;
        g_ca = -0.5f * r0.z; g_cb = -0.5f * r0.w; g_cc = -0.5f * r1.x; g_op = r1.y;
        g_col = make_float3(r1.z, r1.w, r2);

        float c6[6];
        if (cov_pre) {
#pragma unroll
            for (int k = 0; k < 6; k++) c6[k] = cov_pre[6 * i + k];
        } else {
            cov3d_from_scale_rot(sx, sy, sz, mod, q.x, q.y, q.z, q.w, c6);
        }

        // ---- screen-space covariance backward (backward.cu:144-274) ----
        const Cov2D k = cov2d_eval(mx, my, mz, cam, view, c6);
        const float xmul = (k.txtz < -k.limx || k.txtz > k.limx) ? 0.f : 1.f;
        const float ymul = (k.tytz < -k.limy || k.tytz > k.limy) ? 0.f : 1.f;
        const float a = k.a, b = k.b, c = k.c;
        const float denom = a * c - b * b;
        {   // the conic as the forward pass formed it (forward.cu:219-222); a visible Gaussian has denom != 0
            const float dinv = 1.f / denom;
            const float con_x = c * dinv, con_y = -b * dinv, con_z = a * dinv;
            g_dx = -(con_x * m1x + con_y * m1y) * (float)(0.5 * cam.W);
            g_dy = -(con_y * m1x + con_z * m1y) * (float)(0.5 * cam.H);
        }
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        const float d2inv = 1.0f / ((denom * denom) + 0.0000001f);
        const Mat3& T = k.T;
        const Mat3& V = k.V;
        const Mat3& Wm = k.W;
        if (d2inv != 0) {
            dL_da = d2inv * (-c * c * g_ca + 2 * b * c * g_cb + (denom - a * c) * g_cc);
            dL_dc = d2inv * (-a * a * g_cc + 2 * a * b * g_cb + (denom - a * c) * g_ca);
            dL_db = d2inv * 2 * (b * c * g_ca - (denom + 2 * b * b) * g_cb + a * b * g_cc);
            dc[0] = (T.c[0][0] * T.c[0][0] * dL_da + T.c[0][0] * T.c[1][0] * dL_db + T.c[1][0] * T.c[1][0] * dL_dc);
            dc[3] = (T.c[0][1] * T.c[0][1] * dL_da + T.c[0][1] * T.c[1][1] * dL_db + T.c[1][1] * T.c[1][1] * dL_dc);
            dc[5] = (T.c[0][2] * T.c[0][2] * dL_da + T.c[0][2] * T.c[1][2] * dL_db + T.c[1][2] * T.c[1][2] * dL_dc);
            dc[1] = 2 * T.c[0][0] * T.c[0][1] * dL_da + (T.c[0][0] * T.c[1][1] + T.c[0][1] * T.c[1][0]) * dL_db +
                    2 * T.c[1][0] * T.c[1][1] * dL_dc;
            dc[2] = 2 * T.c[0][0] * T.c[0][2] * dL_da + (T.c[0][0] * T.c[1][2] + T.c[0][2] * T.c[1][0]) * dL_db +
                    2 * T.c[1][0] * T.c[1][2] * dL_dc;
            dc[4] = 2 * T.c[0][2] * T.c[0][1] * dL_da + (T.c[0][1] * T.c[1][2] + T.c[0][2] * T.c[1][1]) * dL_db +
                    2 * T.c[1][1] * T.c[1][2] * dL_dc;
        }
        float t0[3], t1[3];  // rows 0 and 1 of T * Vrk
#pragma unroll
        for (int j = 0; j < 3; j++) {
            t0[j] = T.c[0][0] * V.c[j][0] + T.c[0][1] * V.c[j][1] + T.c[0][2] * V.c[j][2];
            t1[j] = T.c[1][0] * V.c[j][0] + T.c[1][1] * V.c[j][1] + T.c[1][2] * V.c[j][2];
        }
        const float dT00 = 2 * t0[0] * dL_da + t1[0] * dL_db;
        const float dT01 = 2 * t0[1] * dL_da + t1[1] * dL_db;
        const float dT02 = 2 * t0[2] * dL_da + t1[2] * dL_db;
        const float dT10 = 2 * t1[0] * dL_dc + t0[0] * dL_db;
        const float dT11 = 2 * t1[1] * dL_dc + t0[1] * dL_db;
        const float dT12 = 2 * t1[2] * dL_dc + t0[2] * dL_db;
        const float dJ00 = Wm.c[0][0] * dT00 + Wm.c[0][1] * dT01 + Wm.c[0][2] * dT02;
        const float dJ02 = Wm.c[2][0] * dT00 + Wm.c[2][1] * dT01 + Wm.c[2][2] * dT02;
        const float dJ11 = Wm.c[1][0] * dT10 + Wm.c[1][1] * dT11 + Wm.c[1][2] * dT12;
        const float dJ12 = Wm.c[2][0] * dT10 + Wm.c[2][1] * dT11 + Wm.c[2][2] * dT12;
        const float fx = cam.focal_x, fy = cam.focal_y;
        const float tz = 1.f / k.t.z, tz2 = tz * tz, tz3 = tz2 * tz;
        const float dtx = xmul * -fx * tz2 * dJ02;
        const float dty = ymul * -fy * tz2 * dJ12;
        const float dtz =
            -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2 * fx * k.t.x) * tz3 * dJ02 + (2 * fy * k.t.y) * tz3 * dJ12;
        dmean = make_float3(view[0] * dtx + view[1] * dty + view[2] * dtz,
                            view[4] * dtx + view[5] * dty + view[6] * dtz,
                            view[8] * dtx + view[9] * dty + view[10] * dtz);

        // ---- perspective projection backward (backward.cu:372-387) ----
        const float4 mh = xform44(mx, my, mz, proj);
        const float mw = 1.0f / (mh.w + 0.0000001f);
        const float mul1 = (proj[0] * mx + proj[4] * my + proj[8] * mz + proj[12]) * mw * mw;
        const float mul2 = (proj[1] * mx + proj[5] * my + proj[9] * mz + proj[13]) * mw * mw;
        const float px_ = (proj[0] * mw - proj[3] * mul1) * g_dx + (proj[1] * mw - proj[3] * mul2) * g_dy;
        const float py_ = (proj[4] * mw - proj[7] * mul1) * g_dx + (proj[5] * mw - proj[7] * mul2) * g_dy;
        const float pz_ = (proj[8] * mw - proj[11] * mul1) * g_dx + (proj[9] * mw - proj[11] * mul2) * g_dy;
        dmean.x += px_; dmean.y += py_; dmean.z += pz_;

        // ---- SH backward ----
        if (shs) {
            const float* sh_in = SH_LDS ? my_row : shs + (size_t)3 * M * i;
            float* sh_out = SH_LDS ? ((dL_dsh || adam.param) ? my_row : nullptr)
                                   : (dL_dsh ? dL_dsh + (size_t)3 * M * i : nullptr);
            const float3 ds = sh_backward(D, M, make_float3(mx, my, mz), campos, sh_in, clamped[i], g_col, sh_out);
            dmean.x += ds.x; dmean.y += ds.y; dmean.z += ds.z;
        }

        // ---- scale / rotation backward (backward.cu:278-341) ----
        if (scales) {
            const float r = q.x, x = q.y, y = q.z, z = q.w;
            const Mat3 R = m3_from_quat(r, x, y, z);
            const float s0 = mod * sx, s1 = mod * sy, s2 = mod * sz;
            const Mat3 Mm = m3_mul(m3_diag(s0, s1, s2), R);
            Mat3 dSig;
            dSig.c[0][0] = dc[0]; dSig.c[0][1] = 0.5f * dc[1]; dSig.c[0][2] = 0.5f * dc[2];
            dSig.c[1][0] = 0.5f * dc[1]; dSig.c[1][1] = dc[3]; dSig.c[1][2] = 0.5f * dc[4];
            dSig.c[2][0] = 0.5f * dc[2]; dSig.c[2][1] = 0.5f * dc[4]; dSig.c[2][2] = dc[5];
            Mat3 M2;
#pragma unroll
            for (int j = 0; j < 3; j++)
#pragma unroll
                for (int e = 0; e < 3; e++) M2.c[j][e] = Mm.c[j][e] * 2.0f;
            const Mat3 dM = m3_mul(M2, dSig);
            const Mat3 Rt = m3_tr(R);
            Mat3 dMt = m3_tr(dM);
            dsc[0] = Rt.c[0][0] * dMt.c[0][0] + Rt.c[0][1] * dMt.c[0][1] + Rt.c[0][2] * dMt.c[0][2];
            dsc[1] = Rt.c[1][0] * dMt.c[1][0] + Rt.c[1][1] * dMt.c[1][1] + Rt.c[1][2] * dMt.c[1][2];
            dsc[2] = Rt.c[2][0] * dMt.c[2][0] + Rt.c[2][1] * dMt.c[2][1] + Rt.c[2][2] * dMt.c[2][2];
#pragma unroll
            for (int e = 0; e < 3; e++) { dMt.c[0][e] *= s0; dMt.c[1][e] *= s1; dMt.c[2][e] *= s2; }
            dq.x = 2 * z * (dMt.c[0][1] - dMt.c[1][0]) + 2 * y * (dMt.c[2][0] - dMt.c[0][2]) + 2 * x * (dMt.c[1][2] - dMt.c[2][1]);
            dq.y = 2 * y * (dMt.c[1][0] + dMt.c[0][1]) + 2 * z * (dMt.c[2][0] + dMt.c[0][2]) + 2 * r * (dMt.c[1][2] - dMt.c[2][1]) - 4 * x * (dMt.c[2][2] + dMt.c[1][1]);
            dq.z = 2 * x * (dMt.c[1][0] + dMt.c[0][1]) + 2 * r * (dMt.c[2][0] - dMt.c[0][2]) + 2 * z * (dMt.c[1][2] + dMt.c[2][1]) - 4 * y * (dMt.c[2][2] + dMt.c[0][0]);
            dq.w = 2 * r * (dMt.c[0][1] - dMt.c[1][0]) + 2 * x * (dMt.c[2][0] + dMt.c[0][2]) + 2 * y * (dMt.c[1][2] + dMt.c[2][1]) - 4 * z * (dMt.c[1][1] + dMt.c[0][0]);
        }
    }

    if (fused && in_range) {
        // Fused optimiser step of the positions, raw opacities, raw scales and raw rotations: the activations'
        // backward (sigmoid / exp / normalise) and Adam, with the gradients still in registers.  A Gaussian that was
        // not rendered steps on a zero gradient like every other (its moments decay, its parameters coast).
        const AdamConst& ak = geo.k;
        {
            float p[3] = {mx, my, mz}, g[3] = {dmean.x, dmean.y, dmean.z};
#pragma unroll
            for (int e = 0; e < 3; e++) {
                float m = geo.xyz.exp_avg[3 * i + e], v = geo.xyz.exp_avg_sq[3 * i + e];
                geo.xyz.param[3 * i + e] = adam_update(p[e], g[e] * ak.grad_scale, m, v, geo.xyz.step, geo.xyz.bc2_sqrt, ak);
                geo.xyz.exp_avg[3 * i + e] = m;
                geo.xyz.exp_avg_sq[3 * i + e] = v;
            }
        }
        {
            const float g = (g_op * ak.grad_scale) * ((1.0f - op_act) * op_act);
            float m = geo.opacity.exp_avg[i], v = geo.opacity.exp_avg_sq[i];
            geo.opacity.param[i] = adam_update(op_in, g, m, v, geo.opacity.step, geo.opacity.bc2_sqrt, ak);
            geo.opacity.exp_avg[i] = m;
            geo.opacity.exp_avg_sq[i] = v;
        }
        {
            const float p[3] = {sx_in, sy_in, sz_in}, a[3] = {sx, sy, sz};
#pragma unroll
            for (int e = 0; e < 3; e++) {
                const float g = (dsc[e] * ak.grad_scale) * a[e];
                float m = geo.scaling.exp_avg[3 * i + e], v = geo.scaling.exp_avg_sq[3 * i + e];
                geo.scaling.param[3 * i + e] = adam_update(p[e], g, m, v, geo.scaling.step, geo.scaling.bc2_sqrt, ak);
                geo.scaling.exp_avg[3 * i + e] = m;
                geo.scaling.exp_avg_sq[3 * i + e] = v;
            }
        }
        {
            const float4 g = make_float4(dq.x * ak.grad_scale, dq.y * ak.grad_scale, dq.z * ak.grad_scale,
                                         dq.w * ak.grad_scale);
            const float4 gr = normalize_quat_backward(q, q_len, g);
            float4 m = ((const float4*)geo.rotation.exp_avg)[i], v = ((const float4*)geo.rotation.exp_avg_sq)[i];
            float4 nr;
            nr.x = adam_update(q_in.x, gr.x, m.x, v.x, geo.rotation.step, geo.rotation.bc2_sqrt, ak);
            nr.y = adam_update(q_in.y, gr.y, m.y, v.y, geo.rotation.step, geo.rotation.bc2_sqrt, ak);
            nr.z = adam_update(q_in.z, gr.z, m.z, v.z, geo.rotation.step, geo.rotation.bc2_sqrt, ak);
            nr.w = adam_update(q_in.w, gr.w, m.w, v.w, geo.rotation.step, geo.rotation.bc2_sqrt, ak);
            ((float4*)geo.rotation.param)[i] = nr;
            ((float4*)geo.rotation.exp_avg)[i] = m;
            ((float4*)geo.rotation.exp_avg_sq)[i] = v;
        }
        if (geo.grad_accum && vis) {  // densification statistics (train.py:115, scene/gaussian_model.py:550-552)
            geo.grad_accum[i] += sqrtf(g_dx * g_dx + g_dy * g_dy);
            geo.denom[i] += 1.0f;
            geo.max_radii[i] = max(geo.max_radii[i], radii[idx]);
        }
    } else if (in_range) {
        if (RAW) {  // gradients wrt the raw parameters: the activations' backward
            g_op = g_op * ((1.0f - op_act) * op_act);
            dsc[0] = dsc[0] * sx; dsc[1] = dsc[1] * sy; dsc[2] = dsc[2] * sz;
            dq = normalize_quat_backward(q, q_len, dq);
        }
        dL_dmean2D[3 * i] = g_dx; dL_dmean2D[3 * i + 1] = g_dy; dL_dmean2D[3 * i + 2] = 0.0f;
        if (dL_dconic) ((float4*)dL_dconic)[i] = make_float4(g_ca, g_cb, 0.0f, g_cc);
        dL_dopacity[i] = g_op;
        dL_dcolor[3 * i] = g_col.x; dL_dcolor[3 * i + 1] = g_col.y; dL_dcolor[3 * i + 2] = g_col.z;
        dL_dmean3D[3 * i] = dmean.x; dL_dmean3D[3 * i + 1] = dmean.y; dL_dmean3D[3 * i + 2] = dmean.z;
#pragma unroll
        for (int k = 0; k < 6; k++) dL_dcov3D[6 * i + k] = dc[k];
        if (!SH_LDS && dL_dsh && !vis)
            for (int k = 0; k < 3 * M; k++) dL_dsh[(size_t)3 * M * i + k] = 0.0f;
        dL_dscale[3 * i] = dsc[0]; dL_dscale[3 * i + 1] = dsc[1]; dL_dscale[3 * i + 2] = dsc[2];
        ((float4*)dL_drot)[i] = dq;
    }

    if (SH_LDS && adam.param) {
        __builtin_amdgcn_wave_barrier();  // rows are wave-private and a wave's LDS operations complete in order
        // Fused optimiser step: the gradient rows never leave the chip.  Each 16-byte group of the wave's 64 rows
        // takes its gradient from LDS (zero for a culled Gaussian, whose moments still decay), its coefficient
        // and Adam moments from memory, and writes all three back; coalesced and streamed like the gradient
        // store it replaces.
#pragma unroll
        for (int t = 0; t < SH_ROW * 64 / 4 / 64; t++) {
            const int f = t * 64 + lane, g = f / (SH_ROW / 4), c = f - g * (SH_ROW / 4);
            if (wave_first + g < P) {
                float4 gr = make_float4(0, 0, 0, 0);
                if ((vis_mask >> g) & 1ull) gr = *(const float4*)(wrows + g * SH_ROW_LDS + c * 4);
                const size_t o = (size_t)(wave_first + g) * SH_ROW + 4 * c;
                float4 p = stream_load4(adam.param + o), m = stream_load4(adam.exp_avg + o),
                       v = stream_load4(adam.exp_avg_sq + o);
                const float s3 = c == 0 ? adam.step_head : adam.step_tail;  // elements 0..2 of a row are the DC term
                p.x = adam_update(p.x, gr.x * adam.k.grad_scale, m.x, v.x, s3, adam.k);
                p.y = adam_update(p.y, gr.y * adam.k.grad_scale, m.y, v.y, s3, adam.k);
                p.z = adam_update(p.z, gr.z * adam.k.grad_scale, m.z, v.z, s3, adam.k);
                p.w = adam_update(p.w, gr.w * adam.k.grad_scale, m.w, v.w, adam.step_tail, adam.k);
                stream_store4(adam.param + o, p);
                stream_store4(adam.exp_avg + o, m);
                stream_store4(adam.exp_avg_sq + o, v);
            }
        }
    } else if (SH_LDS && dL_dsh) {
        __builtin_amdgcn_wave_barrier();  // rows are wave-private and a wave's LDS operations complete in order
        // the wave's 64 gradient rows leave as 12 fully coalesced 1-KiB stores; culled Gaussians get zeros
#pragma unroll
        for (int t = 0; t < SH_ROW * 64 / 4 / 64; t++) {
            const int f = t * 64 + lane, g = f / (SH_ROW / 4), c = f - g * (SH_ROW / 4);
            if (wave_first + g < P) {
                float4 v = make_float4(0, 0, 0, 0);
                if ((vis_mask >> g) & 1ull) v = *(const float4*)(wrows + g * SH_ROW_LDS + c * 4);
                stream_store4(dL_dsh + (size_t)(wave_first + g) * SH_ROW + 4 * c, v);
            }
        }
    }
}

void ngs_launch_preprocess_bwd(hipStream_t s, int P, int D, int M, const float* means, const int* radii,
                               const float* shs, const uint8_t* clamped, const float* scales, const float* rots,
                               const float* opac_raw, float mod, const float* cov_pre, const NgsCamera& cam,
                               const float* acc, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                               float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                               float* dL_drot, const ShAdamArgs* sh_adam, const GeoAdamArgs* geo_adam)
{
    const int nb = (P + NGS_GAUSS_BLOCK - 1) / NGS_GAUSS_BLOCK;
    ShAdamArgs adam;
    memset(&adam, 0, sizeof adam);
    if (sh_adam) adam = *sh_adam;  // the caller has checked that the LDS path below applies
    GeoAdamArgs geo;
    memset(&geo, 0, sizeof geo);
    if (geo_adam) geo = *geo_adam;  // the caller has checked that the parameters are the raw ones
    // LDS staging needs 16-byte aligned 192-byte rows
    const bool lds = shs && M == 16 && ((uintptr_t)shs % 16 == 0) && ((uintptr_t)dL_dsh % 16 == 0);
#define NGS_PRE_BWD(LDS, RAWP)                                                                                        \
    hipLaunchKernelGGL((preprocess_bwd_kernel<LDS, RAWP>), dim3(nb), dim3(NGS_GAUSS_BLOCK), 0, s, P, D, M, means,      \
                       radii, shs, clamped, scales, rots, opac_raw, mod, cov_pre, cam, cam.view, cam.proj, cam.campos, \
                       acc, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale,   \
                       dL_drot, adam, geo)
    if (opac_raw) {
        if (lds) NGS_PRE_BWD(true, true); else NGS_PRE_BWD(false, true);
    } else {
        if (lds) NGS_PRE_BWD(true, false); else NGS_PRE_BWD(false, false);
    }
#undef NGS_PRE_BWD
}
