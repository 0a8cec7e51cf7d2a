/*
 * ngs_oracle.c -- CPU restatement of the tile-based differentiable Gaussian
 * rasterizer.  TEST INFRASTRUCTURE ONLY: this file is the checker the HIP
 * library is compared against (tests/, the smoke check and the benchmark's
 * cpu_baseline leg).  Nothing in the shipped package may call it.
 *
 * PARITY STATUS: pinned by fixtures generated from the reference's own Python:
 * the SH->RGB stage and its backward (utils/sh_utils.py:57-112), the 3D
 * covariance from scale and rotation and its backward
 * (scene/gaussian_model.py:29-33 over utils/general_utils.py:64-110), and the
 * camera matrices (utils/graphics_utils.py:38-71).  Every other stage --
 * screen-space covariance, conic and radius, binning, compositing and their
 * backward -- is "parity unpinned": the
 * reference ships no tests or golden vectors for it and its CUDA sources
 * cannot be compiled here (they need the CUDA toolkit headers and CUB).  Those
 * stages are cross-checked by an independent PyTorch-autograd formulation and
 * finite differences in tests/.
 *
 * Each function names the reference lines it follows
 * (dgr = submodules/diff-gaussian-rasterization).  Arithmetic is written in
 * the reference's evaluation order and must be built with -ffp-contract=off so
 * that integer results derived from floats (radii, tile rectangles, depth key
 * bits) are reproducible.
 *
 * Matrix convention: m3 stores c[col][row] and m3_mul / m3_tr reproduce the
 * column-major product order of the vendored GLM
 * (dgr/third_party/glm/glm/detail/type_mat3x3.inl:486-519).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TILE_W 16
#define TILE_H 16
#define TILE_PIX (TILE_W * TILE_H)

/* dgr/cuda_rasterizer/auxiliary.h:22-39 */
static const float K0 = 0.28209479177387814f;
static const float K1 = 0.4886025119029199f;
static const float K2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                            -1.0925484305920792f, 0.5462742152960396f};
static const float K3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                            -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

typedef struct { float c[3][3]; } m3;
typedef struct { float x, y, z; } v3;

static m3 m3_mul(m3 a, m3 b)
{
    m3 r;
    for (int j = 0; j < 3; j++)
        for (int i = 0; i < 3; i++)
            r.c[j][i] = a.c[0][i] * b.c[j][0] + a.c[1][i] * b.c[j][1] + a.c[2][i] * b.c[j][2];
    return r;
}

static m3 m3_tr(m3 a)
{
    m3 r;
    for (int j = 0; j < 3; j++)
        for (int i = 0; i < 3; i++)
            r.c[j][i] = a.c[i][j];
    return r;
}

static v3 v3_scale(float s, v3 v) { v3 r = {s * v.x, s * v.y, s * v.z}; return r; }
static v3 v3_add(v3 a, v3 b) { v3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
static v3 v3_sub(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static float v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

/* float -> int conversion as the GPU does it: saturating, NaN -> 0 */
static int f2i(float f)
{
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
}

/* dgr/cuda_rasterizer/auxiliary.h:58-77 */
static v3 xform43(const float* p, const float* m)
{
    v3 r = {m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12],
            m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13],
            m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14]};
    return r;
}

static void xform44(const float* p, const float* m, float out[4])
{
    out[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    out[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    out[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    out[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* dgr/cuda_rasterizer/auxiliary.h:41-44 (double arithmetic, rounded to float on return) */
static float ndc_to_pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

/* dgr/cuda_rasterizer/auxiliary.h:46-56 */
static void tile_rect(float px, float py, int radius, int gx, int gy, uint32_t rmin[2], uint32_t rmax[2])
{
    int a;
    a = f2i((px - radius) / TILE_W); if (a < 0) a = 0; rmin[0] = (uint32_t)a < (uint32_t)gx ? (uint32_t)a : (uint32_t)gx;
    a = f2i((py - radius) / TILE_H); if (a < 0) a = 0; rmin[1] = (uint32_t)a < (uint32_t)gy ? (uint32_t)a : (uint32_t)gy;
    a = f2i((px + radius + TILE_W - 1) / TILE_W); if (a < 0) a = 0; rmax[0] = (uint32_t)a < (uint32_t)gx ? (uint32_t)a : (uint32_t)gx;
    a = f2i((py + radius + TILE_H - 1) / TILE_H); if (a < 0) a = 0; rmax[1] = (uint32_t)a < (uint32_t)gy ? (uint32_t)a : (uint32_t)gy;
}

/* dgr/cuda_rasterizer/forward.cu:118-152 (quaternion used as given, no normalisation) */
static m3 rot_from_quat(const float* q)
{
    float r = q[0], x = q[1], y = q[2], z = q[3];
    m3 R;
    R.c[0][0] = 1.f - 2.f * (y * y + z * z); R.c[0][1] = 2.f * (x * y - r * z); R.c[0][2] = 2.f * (x * z + r * y);
    R.c[1][0] = 2.f * (x * y + r * z); R.c[1][1] = 1.f - 2.f * (x * x + z * z); R.c[1][2] = 2.f * (y * z - r * x);
    R.c[2][0] = 2.f * (x * z - r * y); R.c[2][1] = 2.f * (y * z + r * x); R.c[2][2] = 1.f - 2.f * (x * x + y * y);
    return R;
}

static m3 diag3(float a, float b, float c)
{
    m3 S;
    memset(&S, 0, sizeof S);
    S.c[0][0] = a; S.c[1][1] = b; S.c[2][2] = c;
    return S;
}

static void cov3d_from_scale_rot(const float* s, float mod, const float* q, float* out6)
{
    m3 S = diag3(mod * s[0], mod * s[1], mod * s[2]);
    m3 M = m3_mul(S, rot_from_quat(q));
    m3 Sig = m3_mul(m3_tr(M), M);
    out6[0] = Sig.c[0][0]; out6[1] = Sig.c[0][1]; out6[2] = Sig.c[0][2];
    out6[3] = Sig.c[1][1]; out6[4] = Sig.c[1][2]; out6[5] = Sig.c[2][2];
}

/* shared by forward (forward.cu:74-113) and backward (backward.cu:166-199) */
typedef struct { m3 T, W, V; v3 t; float txtz, tytz, limx, limy; float a, b, c; } cov2d_ctx;

static cov2d_ctx cov2d_eval(const float* mean, float fx, float fy, float tanx, float tany, const float* cov6, const float* vm)
{
    cov2d_ctx k;
    k.t = xform43(mean, vm);
    k.limx = 1.3f * tanx;
    k.limy = 1.3f * tany;
    k.txtz = k.t.x / k.t.z;
    k.tytz = k.t.y / k.t.z;
    k.t.x = fminf(k.limx, fmaxf(-k.limx, k.txtz)) * k.t.z;
    k.t.y = fminf(k.limy, fmaxf(-k.limy, k.tytz)) * k.t.z;
    m3 J;
    J.c[0][0] = fx / k.t.z; J.c[0][1] = 0.0f; J.c[0][2] = -(fx * k.t.x) / (k.t.z * k.t.z);
    J.c[1][0] = 0.0f; J.c[1][1] = fy / k.t.z; J.c[1][2] = -(fy * k.t.y) / (k.t.z * k.t.z);
    J.c[2][0] = 0.0f; J.c[2][1] = 0.0f; J.c[2][2] = 0.0f;
    k.W.c[0][0] = vm[0]; k.W.c[0][1] = vm[4]; k.W.c[0][2] = vm[8];
    k.W.c[1][0] = vm[1]; k.W.c[1][1] = vm[5]; k.W.c[1][2] = vm[9];
    k.W.c[2][0] = vm[2]; k.W.c[2][1] = vm[6]; k.W.c[2][2] = vm[10];
    k.T = m3_mul(k.W, J);
    k.V.c[0][0] = cov6[0]; k.V.c[0][1] = cov6[1]; k.V.c[0][2] = cov6[2];
    k.V.c[1][0] = cov6[1]; k.V.c[1][1] = cov6[3]; k.V.c[1][2] = cov6[4];
    k.V.c[2][0] = cov6[2]; k.V.c[2][1] = cov6[4]; k.V.c[2][2] = cov6[5];
    m3 cov = m3_mul(m3_mul(m3_tr(k.T), m3_tr(k.V)), k.T);
    k.a = cov.c[0][0] + 0.3f;
    k.b = cov.c[0][1];
    k.c = cov.c[1][1] + 0.3f;
    return k;
}

/* dgr/cuda_rasterizer/forward.cu:20-71 */
static v3 sh_to_rgb(int deg, const float* sh, v3 dir, uint8_t* clamped3)
{
    const v3* c = (const v3*)sh;
    v3 res = v3_scale(K0, c[0]);
    if (deg > 0) {
        float x = dir.x, y = dir.y, z = dir.z;
        res = v3_sub(v3_add(v3_sub(res, v3_scale(K1 * y, c[1])), v3_scale(K1 * z, c[2])), v3_scale(K1 * x, c[3]));
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            res = v3_add(res, v3_scale(K2[0] * xy, c[4]));
            res = v3_add(res, v3_scale(K2[1] * yz, c[5]));
            res = v3_add(res, v3_scale(K2[2] * (2.0f * zz - xx - yy), c[6]));
            res = v3_add(res, v3_scale(K2[3] * xz, c[7]));
            res = v3_add(res, v3_scale(K2[4] * (xx - yy), c[8]));
            if (deg > 2) {
                res = v3_add(res, v3_scale(K3[0] * y * (3.0f * xx - yy), c[9]));
                res = v3_add(res, v3_scale(K3[1] * xy * z, c[10]));
                res = v3_add(res, v3_scale(K3[2] * y * (4.0f * zz - xx - yy), c[11]));
                res = v3_add(res, v3_scale(K3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy), c[12]));
                res = v3_add(res, v3_scale(K3[4] * x * (4.0f * zz - xx - yy), c[13]));
                res = v3_add(res, v3_scale(K3[5] * z * (xx - yy), c[14]));
                res = v3_add(res, v3_scale(K3[6] * x * (xx - 3.0f * yy), c[15]));
            }
        }
    }
    res.x += 0.5f; res.y += 0.5f; res.z += 0.5f;
    clamped3[0] = res.x < 0; clamped3[1] = res.y < 0; clamped3[2] = res.z < 0;
    res.x = fmaxf(res.x, 0.0f); res.y = fmaxf(res.y, 0.0f); res.z = fmaxf(res.z, 0.0f);
    return res;
}

/*
 * Per-Gaussian forward stage.  dgr/cuda_rasterizer/forward.cu:155-256 with
 * in_frustum from auxiliary.h:139-164.  Returns 0, or 1 when `prefiltered` is
 * set and a point fails the near test (the reference traps the kernel there).
 * Arrays for culled Gaussians keep whatever the caller put in them except
 * radii / tiles_touched, which are always written.
 */
int ngs_oracle_preprocess(int P, int D, int M, const float* means, const float* scales, float mod, const float* rots,
                          const float* opac, const float* shs, const float* cov_pre, const float* col_pre,
                          const float* vm, const float* pm, const float* campos, int W, int H, float tanx, float tany,
                          int prefiltered, int* radii, float* xy, float* depths, float* cov3d, float* rgb,
                          float* conic_o, uint32_t* tiles, uint8_t* clamped)
{
    const float fy = H / (2.0f * tany);
    const float fx = W / (2.0f * tanx);
    const int gx = (W + TILE_W - 1) / TILE_W, gy = (H + TILE_H - 1) / TILE_H;
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int i = 0; i < P; i++) {
        radii[i] = 0;
        tiles[i] = 0;
        const float* p = means + 3 * i;
        v3 pv = xform43(p, vm);
        if (pv.z <= 0.2f) {
            if (prefiltered) bad |= 1;
            continue;
        }
        float ph[4];
        xform44(p, pm, ph);
        float pw = 1.0f / (ph[3] + 0.0000001f);
        float projx = ph[0] * pw, projy = ph[1] * pw;
        const float* c6;
        if (cov_pre) {
            c6 = cov_pre + 6 * i;
        } else {
            cov3d_from_scale_rot(scales + 3 * i, mod, rots + 4 * i, cov3d + 6 * i);
            c6 = cov3d + 6 * i;
        }
        cov2d_ctx k = cov2d_eval(p, fx, fy, tanx, tany, c6, vm);
        float det = k.a * k.c - k.b * k.b;
        if (det == 0.0f) continue;
        float dinv = 1.f / det;
        float ca = k.c * dinv, cb = -k.b * dinv, cc = k.a * dinv;
        float mid = 0.5f * (k.a + k.c);
        float l1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        float l2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        float rad = ceilf(3.f * sqrtf(fmaxf(l1, l2)));
        float ix = ndc_to_pix(projx, W), iy = ndc_to_pix(projy, H);
        uint32_t rmin[2], rmax[2];
        tile_rect(ix, iy, f2i(rad), gx, gy, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;
        if (!col_pre) {
            v3 pos = {p[0], p[1], p[2]}, cam = {campos[0], campos[1], campos[2]};
            v3 d = v3_sub(pos, cam);
            float len = sqrtf(v3_dot(d, d));
            d.x = d.x / len; d.y = d.y / len; d.z = d.z / len;
            v3 col = sh_to_rgb(D, shs + (size_t)3 * M * i, d, clamped + 3 * i);
            rgb[3 * i] = col.x; rgb[3 * i + 1] = col.y; rgb[3 * i + 2] = col.z;
        }
        depths[i] = pv.z;
        radii[i] = f2i(rad);
        xy[2 * i] = ix; xy[2 * i + 1] = iy;
        conic_o[4 * i] = ca; conic_o[4 * i + 1] = cb; conic_o[4 * i + 2] = cc; conic_o[4 * i + 3] = opac[i];
        tiles[i] = (rmax[1] - rmin[1]) * (rmax[0] - rmin[0]);
    }
    return bad;
}

/* SH -> RGB alone, for every Gaussian (same code path as inside ngs_oracle_preprocess): lets the stage be
 * checked against vectors produced by the reference's Python eval_sh (utils/sh_utils.py:57-112). */
void ngs_oracle_sh_to_rgb(int P, int D, int M, const float* means, const float* campos, const float* shs, float* rgb,
                          uint8_t* clamped)
{
    for (int i = 0; i < P; i++) {
        const float* p = means + 3 * i;
        v3 pos = {p[0], p[1], p[2]}, cam = {campos[0], campos[1], campos[2]};
        v3 d = v3_sub(pos, cam);
        float len = sqrtf(v3_dot(d, d));
        d.x = d.x / len; d.y = d.y / len; d.z = d.z / len;
        v3 col = sh_to_rgb(D, shs + (size_t)3 * M * i, d, clamped + 3 * i);
        rgb[3 * i] = col.x; rgb[3 * i + 1] = col.y; rgb[3 * i + 2] = col.z;
    }
}

/* dgr/cuda_rasterizer/rasterizer_impl.cu:54-66 */
void ngs_oracle_mark_visible(int P, const float* means, const float* vm, const float* pm, uint8_t* present)
{
    (void)pm;
    for (int i = 0; i < P; i++) present[i] = xform43(means + 3 * i, vm).z > 0.2f;
}

/* uint32 inclusive prefix sum: the result CUB's InclusiveSum defines (rasterizer_impl.cu:277) */
void ngs_oracle_scan(int P, const uint32_t* in, uint32_t* out)
{
    uint32_t s = 0;
    for (int i = 0; i < P; i++) { s += in[i]; out[i] = s; }
}

/* dgr/cuda_rasterizer/rasterizer_impl.cu:70-111 */
void ngs_oracle_duplicate(int P, const float* xy, const float* depths, const uint32_t* offs, const int* radii, int W,
                          int H, uint64_t* keys, uint32_t* vals)
{
    const int gx = (W + TILE_W - 1) / TILE_W, gy = (H + TILE_H - 1) / TILE_H;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        if (radii[i] <= 0) continue;
        uint32_t off = i == 0 ? 0 : offs[i - 1];
        uint32_t rmin[2], rmax[2];
        tile_rect(xy[2 * i], xy[2 * i + 1], radii[i], gx, gy, rmin, rmax);
        uint32_t dbits;
        memcpy(&dbits, depths + i, 4);
        for (uint32_t y = rmin[1]; y < rmax[1]; y++)
            for (uint32_t x = rmin[0]; x < rmax[0]; x++) {
                keys[off] = ((uint64_t)(y * (uint32_t)gx + x) << 32) | dbits;
                vals[off] = (uint32_t)i;
                off++;
            }
    }
}

/* dgr/cuda_rasterizer/rasterizer_impl.cu:35-50 */
uint32_t ngs_oracle_higher_msb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

/*
 * Stable sort of (key,value) pairs on key bits [0,end_bit): the result
 * cub::DeviceRadixSort::SortPairs defines (rasterizer_impl.cu:303-308).
 * LSD radix, 8-bit digits, thread-chunked so the benchmark's CPU leg can use
 * every core; the output does not depend on the thread count.
 */
void ngs_oracle_sort_pairs(int64_t n, const uint64_t* kin, const uint32_t* vin, uint64_t* kout, uint32_t* vout,
                           int end_bit)
{
    if (n <= 0) return;
    uint64_t* kb[2];
    uint32_t* vb[2];
    kb[0] = (uint64_t*)malloc(sizeof(uint64_t) * n); kb[1] = (uint64_t*)malloc(sizeof(uint64_t) * n);
    vb[0] = (uint32_t*)malloc(sizeof(uint32_t) * n); vb[1] = (uint32_t*)malloc(sizeof(uint32_t) * n);
    memcpy(kb[0], kin, sizeof(uint64_t) * n);
    memcpy(vb[0], vin, sizeof(uint32_t) * n);
    int nt = 1;
#ifdef _OPENMP
    nt = omp_get_max_threads();
#endif
    if (n < 65536) nt = 1;
    int64_t* hist = (int64_t*)malloc(sizeof(int64_t) * 256 * nt);
    int cur = 0;
    for (int shift = 0; shift < end_bit; shift += 8) {
        int bits = end_bit - shift < 8 ? end_bit - shift : 8;
        uint64_t mask = (1u << bits) - 1;
        const uint64_t* ks = kb[cur]; const uint32_t* vs = vb[cur];
        uint64_t* kd = kb[cur ^ 1]; uint32_t* vd = vb[cur ^ 1];
        memset(hist, 0, sizeof(int64_t) * 256 * nt);
#pragma omp parallel num_threads(nt)
        {
            int t = 0;
#ifdef _OPENMP
            t = omp_get_thread_num();
#endif
            int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
            int64_t* h = hist + 256 * t;
            for (int64_t i = lo; i < hi; i++) h[(ks[i] >> shift) & mask]++;
#pragma omp barrier
#pragma omp single
            {
                int64_t run = 0;
                for (int d = 0; d < 256; d++)
                    for (int u = 0; u < nt; u++) { int64_t c = hist[256 * u + d]; hist[256 * u + d] = run; run += c; }
            }
            for (int64_t i = lo; i < hi; i++) {
                int64_t dst = h[(ks[i] >> shift) & mask]++;
                kd[dst] = ks[i]; vd[dst] = vs[i];
            }
        }
        cur ^= 1;
    }
    memcpy(kout, kb[cur], sizeof(uint64_t) * n);
    memcpy(vout, vb[cur], sizeof(uint32_t) * n);
    free(kb[0]); free(kb[1]); free(vb[0]); free(vb[1]); free(hist);
}

/* dgr/cuda_rasterizer/rasterizer_impl.cu:116-138 after the memset at :310; ranges is [T][2] */
void ngs_oracle_tile_ranges(int64_t L, const uint64_t* keys, int T, uint32_t* ranges)
{
    memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)T);
    for (int64_t i = 0; i < L; i++) {
        uint32_t cur = (uint32_t)(keys[i] >> 32);
        if (i == 0) ranges[2 * cur] = 0;
        else {
            uint32_t prev = (uint32_t)(keys[i - 1] >> 32);
            if (cur != prev) { ranges[2 * prev + 1] = (uint32_t)i; ranges[2 * cur] = (uint32_t)i; }
        }
        if (i == L - 1) ranges[2 * cur + 1] = (uint32_t)L;
    }
}

/* dgr/cuda_rasterizer/forward.cu:261-374.  final_T / n_contrib are row-major [H*W], out is [3][H][W]. */
void ngs_oracle_render_forward(int W, int H, const uint32_t* ranges, const uint32_t* plist, const float* xy,
                               const float* feat, const float* conic_o, const float* bg, float* final_T,
                               uint32_t* n_contrib, float* out)
{
    const int gx = (W + TILE_W - 1) / TILE_W, gy = (H + TILE_H - 1) / TILE_H;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        int tx = tile % gx, ty = tile / gx;
        uint32_t lo = ranges[2 * tile], hi = ranges[2 * tile + 1];
        for (int ly = 0; ly < TILE_H; ly++)
            for (int lx = 0; lx < TILE_W; lx++) {
                int px = tx * TILE_W + lx, py = ty * TILE_H + ly;
                if (px >= W || py >= H) continue;
                float T = 1.0f, C[3] = {0, 0, 0};
                uint32_t contributor = 0, last = 0;
                for (uint32_t k = lo; k < hi; k++) {
                    contributor++;
                    uint32_t g = plist[k];
                    float dx = xy[2 * g] - (float)px, dy = xy[2 * g + 1] - (float)py;
                    const float* co = conic_o + 4 * g;
                    float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0.0f) continue;
                    float alpha = fminf(0.99f, co[3] * expf(power));
                    if (alpha < 1.0f / 255.0f) continue;
                    float test_T = T * (1 - alpha);
                    if (test_T < 0.0001f) break;
                    for (int ch = 0; ch < 3; ch++) C[ch] += feat[3 * g + ch] * alpha * T;
                    T = test_T;
                    last = contributor;
                }
                size_t pid = (size_t)W * py + px;
                final_T[pid] = T;
                n_contrib[pid] = last;
                for (int ch = 0; ch < 3; ch++) out[(size_t)ch * H * W + pid] = C[ch] + T * bg[ch];
            }
    }
}

/*
 * One tile of dgr/cuda_rasterizer/backward.cu:399-557: every pixel of the tile walks the tile's list back to front
 * and adds its terms to d_xy [P][3] (x,y used), d_conic [P][4] (slots 0,1,3), d_opac [P], d_col [P][3] with plain
 * additions, pixel after pixel (the reference's float-atomic order is unspecified).
 * Generated twice: with float sums (the reference's arithmetic) and with double sums (every per-pixel term is
 * still the reference's float32 value; only the running sums are wide), which tells the rounding of a long float32
 * sum apart from an error in the terms.
 */
#define RENDER_BACKWARD_TILE(NAME, ACC) \
static void NAME(int tile, int W, int H, int gx, const uint32_t* ranges, const uint32_t* plist, \
                 const float* bg, const float* xy, const float* conic_o, const float* feat, \
                 const float* final_T, const uint32_t* n_contrib, const float* dpix, ACC* d_xy, \
                 ACC* d_conic, ACC* d_opac, ACC* d_col) \
{ \
    const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H); \
    int tx = tile % gx, ty = tile / gx; \
    uint32_t lo = ranges[2 * tile], hi = ranges[2 * tile + 1]; \
    for (int ly = 0; ly < TILE_H; ly++) \
        for (int lx = 0; lx < TILE_W; lx++) { \
            int px = tx * TILE_W + lx, py = ty * TILE_H + ly; \
            if (px >= W || py >= H) continue; \
            size_t pid = (size_t)W * py + px; \
            const float T_final = final_T[pid]; \
            float T = T_final; \
            uint32_t contributor = hi - lo; \
            const uint32_t last = n_contrib[pid]; \
            float rec[3] = {0, 0, 0}, last_c[3] = {0, 0, 0}, last_a = 0, dp[3]; \
            for (int ch = 0; ch < 3; ch++) dp[ch] = dpix[(size_t)ch * H * W + pid]; \
            for (uint32_t k = hi; k-- > lo;) { \
                contributor--; \
                if (contributor >= last) continue; \
                uint32_t g = plist[k]; \
                float dx = xy[2 * g] - (float)px, dy = xy[2 * g + 1] - (float)py; \
                const float* co = conic_o + 4 * g; \
                float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy; \
                if (power > 0.0f) continue; \
                float G = expf(power); \
                float alpha = fminf(0.99f, co[3] * G); \
                if (alpha < 1.0f / 255.0f) continue; \
                T = T / (1.f - alpha); \
                float w = alpha * T; \
                float dL_da = 0.0f; \
                for (int ch = 0; ch < 3; ch++) { \
                    float c = feat[3 * g + ch]; \
                    rec[ch] = last_a * last_c[ch] + (1.f - last_a) * rec[ch]; \
                    last_c[ch] = c; \
                    dL_da += (c - rec[ch]) * dp[ch]; \
                    d_col[3 * g + ch] += w * dp[ch]; \
                } \
                dL_da *= T; \
                last_a = alpha; \
                float bgdot = 0; \
                for (int ch = 0; ch < 3; ch++) bgdot += bg[ch] * dp[ch]; \
                dL_da += (-T_final / (1.f - alpha)) * bgdot; \
                float dL_dG = co[3] * dL_da; \
                float gdx = G * dx, gdy = G * dy; \
                float dG_dx = -gdx * co[0] - gdy * co[1]; \
                float dG_dy = -gdy * co[2] - gdx * co[1]; \
                d_xy[3 * g] += dL_dG * dG_dx * ddelx_dx; \
                d_xy[3 * g + 1] += dL_dG * dG_dy * ddely_dy; \
                d_conic[4 * g] += -0.5f * gdx * dx * dL_dG; \
                d_conic[4 * g + 1] += -0.5f * gdx * dy * dL_dG; \
                d_conic[4 * g + 3] += -0.5f * gdy * dy * dL_dG; \
                d_opac[g] += G * dL_da; \
            } \
        } \
}
RENDER_BACKWARD_TILE(render_backward_tile, float)
RENDER_BACKWARD_TILE(render_backward_tile_wide, double)

/*
 * dgr/cuda_rasterizer/backward.cu:399-557 over all tiles.  Accumulates into the outputs (callers zero them).
 * With one thread the additions happen in a fixed order (tile, pixel, back to front) -- the mode the parity tests
 * use.  With several threads every thread adds into private copies of the four arrays (11 floats per Gaussian
 * each), which are summed afterwards: no atomics inside the pixel loop, at the price of an addition order that
 * follows the schedule.  Returns 0, or 1 if the private copies could not be allocated.
 */
int ngs_oracle_render_backward(int P, int W, int H, const uint32_t* ranges, const uint32_t* plist, const float* bg,
                               const float* xy, const float* conic_o, const float* feat, const float* final_T,
                               const uint32_t* n_contrib, const float* dpix, float* d_xy, float* d_conic,
                               float* d_opac, float* d_col)
{
    const int gx = (W + TILE_W - 1) / TILE_W, gy = (H + TILE_H - 1) / TILE_H;
    int threads = 1;
#ifdef _OPENMP
    threads = omp_get_max_threads();
#endif
    if (threads <= 1) {
        for (int tile = 0; tile < gx * gy; tile++)
            render_backward_tile(tile, W, H, gx, ranges, plist, bg, xy, conic_o, feat, final_T, n_contrib, dpix,
                                 d_xy, d_conic, d_opac, d_col);
        return 0;
    }
    const size_t per = (size_t)11 * (size_t)P; /* 3 + 4 + 1 + 3 floats per Gaussian */
    float* priv = (float*)calloc((size_t)threads * per, sizeof(float));
    if (!priv) return 1;
#pragma omp parallel num_threads(threads)
    {
        int me = 0;
#ifdef _OPENMP
        me = omp_get_thread_num();
#endif
        float* base = priv + (size_t)me * per;
        float *p_xy = base, *p_conic = base + (size_t)3 * P, *p_opac = base + (size_t)7 * P,
              *p_col = base + (size_t)8 * P;
#pragma omp for schedule(dynamic, 1)
        for (int tile = 0; tile < gx * gy; tile++)
            render_backward_tile(tile, W, H, gx, ranges, plist, bg, xy, conic_o, feat, final_T, n_contrib, dpix,
                                 p_xy, p_conic, p_opac, p_col);
    }
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++)
        for (int t = 0; t < threads; t++) {
            const float* base = priv + (size_t)t * per;
            for (int k = 0; k < 3; k++) d_xy[3 * i + k] += base[3 * (size_t)i + k];
            for (int k = 0; k < 4; k++) d_conic[4 * i + k] += base[(size_t)3 * P + 4 * (size_t)i + k];
            d_opac[i] += base[(size_t)7 * P + i];
            for (int k = 0; k < 3; k++) d_col[3 * i + k] += base[(size_t)8 * P + 3 * (size_t)i + k];
        }
    free(priv);
    return 0;
}

/*
 * The same sums held in double precision (per-thread private copies, summed in thread order, rounded to float at the
 * end).  Not the reference's arithmetic: a yardstick for how far a float32 sum of up to a million terms -- in the
 * reference's order or any other -- sits from the exact sum of the same terms.
 */
int ngs_oracle_render_backward_wide(int P, int W, int H, const uint32_t* ranges, const uint32_t* plist,
                                    const float* bg, const float* xy, const float* conic_o, const float* feat,
                                    const float* final_T, const uint32_t* n_contrib, const float* dpix, float* d_xy,
                                    float* d_conic, float* d_opac, float* d_col)
{
    const int gx = (W + TILE_W - 1) / TILE_W, gy = (H + TILE_H - 1) / TILE_H;
    int threads = 1;
#ifdef _OPENMP
    threads = omp_get_max_threads();
#endif
    const size_t per = (size_t)11 * (size_t)P;
    double* priv = (double*)calloc((size_t)threads * per, sizeof(double));
    if (!priv) return 1;
#pragma omp parallel num_threads(threads)
    {
        int me = 0;
#ifdef _OPENMP
        me = omp_get_thread_num();
#endif
        double* base = priv + (size_t)me * per;
#pragma omp for schedule(dynamic, 1)
        for (int tile = 0; tile < gx * gy; tile++)
            render_backward_tile_wide(tile, W, H, gx, ranges, plist, bg, xy, conic_o, feat, final_T, n_contrib, dpix,
                                      base, base + (size_t)3 * P, base + (size_t)7 * P, base + (size_t)8 * P);
    }
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        double acc[11] = {0};
        for (int t = 0; t < threads; t++) {
            const double* base = priv + (size_t)t * per;
            for (int k = 0; k < 3; k++) acc[k] += base[3 * (size_t)i + k];
            for (int k = 0; k < 4; k++) acc[3 + k] += base[(size_t)3 * P + 4 * (size_t)i + k];
            acc[7] += base[(size_t)7 * P + i];
            for (int k = 0; k < 3; k++) acc[8 + k] += base[(size_t)8 * P + 3 * (size_t)i + k];
        }
        for (int k = 0; k < 3; k++) d_xy[3 * i + k] += (float)acc[k];
        for (int k = 0; k < 4; k++) d_conic[4 * i + k] += (float)acc[3 + k];
        d_opac[i] += (float)acc[7];
        for (int k = 0; k < 3; k++) d_col[3 * i + k] += (float)acc[8 + k];
    }
    free(priv);
    return 0;
}

/* dgr/cuda_rasterizer/backward.cu:144-274.  Writes d_means[i] and d_cov[6i..] for radii>0. */
void ngs_oracle_cov2d_backward(int P, const float* means, const int* radii, const float* cov3d, float fx, float fy,
                               float tanx, float tany, const float* vm, const float* d_conic, float* d_means,
                               float* d_cov)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        if (!(radii[i] > 0)) continue;
        cov2d_ctx k = cov2d_eval(means + 3 * i, fx, fy, tanx, tany, cov3d + 6 * i, vm);
        float gx_ = d_conic[4 * i], gy_ = d_conic[4 * i + 1], gz_ = d_conic[4 * i + 3];
        float xmul = (k.txtz < -k.limx || k.txtz > k.limx) ? 0.f : 1.f;
        float ymul = (k.tytz < -k.limy || k.tytz > k.limy) ? 0.f : 1.f;
        float a = k.a, b = k.b, c = k.c;
        float denom = a * c - b * b;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        float d2inv = 1.0f / ((denom * denom) + 0.0000001f);
        const m3 T = k.T, V = k.V, Wm = k.W;
        float* dc = d_cov + 6 * i;
        if (d2inv != 0) {
            dL_da = d2inv * (-c * c * gx_ + 2 * b * c * gy_ + (denom - a * c) * gz_);
            dL_dc = d2inv * (-a * a * gz_ + 2 * a * b * gy_ + (denom - a * c) * gx_);
            dL_db = d2inv * 2 * (b * c * gx_ - (denom + 2 * b * b) * gy_ + a * b * gz_);
            dc[0] = (T.c[0][0] * T.c[0][0] * dL_da + T.c[0][0] * T.c[1][0] * dL_db + T.c[1][0] * T.c[1][0] * dL_dc);
            dc[3] = (T.c[0][1] * T.c[0][1] * dL_da + T.c[0][1] * T.c[1][1] * dL_db + T.c[1][1] * T.c[1][1] * dL_dc);
            dc[5] = (T.c[0][2] * T.c[0][2] * dL_da + T.c[0][2] * T.c[1][2] * dL_db + T.c[1][2] * T.c[1][2] * dL_dc);
            dc[1] = 2 * T.c[0][0] * T.c[0][1] * dL_da + (T.c[0][0] * T.c[1][1] + T.c[0][1] * T.c[1][0]) * dL_db +
                    2 * T.c[1][0] * T.c[1][1] * dL_dc;
            dc[2] = 2 * T.c[0][0] * T.c[0][2] * dL_da + (T.c[0][0] * T.c[1][2] + T.c[0][2] * T.c[1][0]) * dL_db +
                    2 * T.c[1][0] * T.c[1][2] * dL_dc;
            dc[4] = 2 * T.c[0][2] * T.c[0][1] * dL_da + (T.c[0][1] * T.c[1][2] + T.c[0][2] * T.c[1][1]) * dL_db +
                    2 * T.c[1][1] * T.c[1][2] * dL_dc;
        } else {
            for (int j = 0; j < 6; j++) dc[j] = 0;
        }
        float r0[3], r1[3]; /* rows of T*Vrk: r0[j] = sum_m T[0][m] V[j][m] */
        for (int j = 0; j < 3; j++) {
            r0[j] = T.c[0][0] * V.c[j][0] + T.c[0][1] * V.c[j][1] + T.c[0][2] * V.c[j][2];
            r1[j] = T.c[1][0] * V.c[j][0] + T.c[1][1] * V.c[j][1] + T.c[1][2] * V.c[j][2];
        }
        float dT00 = 2 * r0[0] * dL_da + r1[0] * dL_db;
        float dT01 = 2 * r0[1] * dL_da + r1[1] * dL_db;
        float dT02 = 2 * r0[2] * dL_da + r1[2] * dL_db;
        float dT10 = 2 * r1[0] * dL_dc + r0[0] * dL_db;
        float dT11 = 2 * r1[1] * dL_dc + r0[1] * dL_db;
        float dT12 = 2 * r1[2] * dL_dc + r0[2] * dL_db;
        float dJ00 = Wm.c[0][0] * dT00 + Wm.c[0][1] * dT01 + Wm.c[0][2] * dT02;
        float dJ02 = Wm.c[2][0] * dT00 + Wm.c[2][1] * dT01 + Wm.c[2][2] * dT02;
        float dJ11 = Wm.c[1][0] * dT10 + Wm.c[1][1] * dT11 + Wm.c[1][2] * dT12;
        float dJ12 = Wm.c[2][0] * dT10 + Wm.c[2][1] * dT11 + Wm.c[2][2] * dT12;
        float tz = 1.f / k.t.z, tz2 = tz * tz, tz3 = tz2 * tz;
        float dtx = xmul * -fx * tz2 * dJ02;
        float dty = ymul * -fy * tz2 * dJ12;
        float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2 * fx * k.t.x) * tz3 * dJ02 + (2 * fy * k.t.y) * tz3 * dJ12;
        d_means[3 * i] = vm[0] * dtx + vm[1] * dty + vm[2] * dtz;
        d_means[3 * i + 1] = vm[4] * dtx + vm[5] * dty + vm[6] * dtz;
        d_means[3 * i + 2] = vm[8] * dtx + vm[9] * dty + vm[10] * dtz;
    }
}

/* dgr/cuda_rasterizer/backward.cu:20-139 */
static void sh_backward(int deg, int M, const float* mean, const float* campos, const float* shs, const uint8_t* cl,
                        const float* d_col, float* d_mean_acc, float* d_sh)
{
    v3 pos = {mean[0], mean[1], mean[2]}, cam = {campos[0], campos[1], campos[2]};
    v3 d0 = v3_sub(pos, cam);
    float len = sqrtf(v3_dot(d0, d0));
    v3 dir = {d0.x / len, d0.y / len, d0.z / len};
    const v3* sh = (const v3*)shs;
    v3* g = (v3*)d_sh;
    (void)M;
    v3 dRGB = {d_col[0], d_col[1], d_col[2]};
    dRGB.x *= cl[0] ? 0 : 1; dRGB.y *= cl[1] ? 0 : 1; dRGB.z *= cl[2] ? 0 : 1;
    v3 ddx = {0, 0, 0}, ddy = {0, 0, 0}, ddz = {0, 0, 0};
    float x = dir.x, y = dir.y, z = dir.z;
    g[0] = v3_scale(K0, dRGB);
    if (deg > 0) {
        g[1] = v3_scale(-K1 * y, dRGB);
        g[2] = v3_scale(K1 * z, dRGB);
        g[3] = v3_scale(-K1 * x, dRGB);
        ddx = v3_scale(-K1, sh[3]);
        ddy = v3_scale(-K1, sh[1]);
        ddz = v3_scale(K1, sh[2]);
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            g[4] = v3_scale(K2[0] * xy, dRGB);
            g[5] = v3_scale(K2[1] * yz, dRGB);
            g[6] = v3_scale(K2[2] * (2.f * zz - xx - yy), dRGB);
            g[7] = v3_scale(K2[3] * xz, dRGB);
            g[8] = v3_scale(K2[4] * (xx - yy), dRGB);
            v3 s;
            s = v3_scale(K2[0] * y, sh[4]);
            s = v3_add(s, v3_scale(K2[2] * 2.f * -x, sh[6]));
            s = v3_add(s, v3_scale(K2[3] * z, sh[7]));
            s = v3_add(s, v3_scale(K2[4] * 2.f * x, sh[8]));
            ddx = v3_add(ddx, s);
            s = v3_scale(K2[0] * x, sh[4]);
            s = v3_add(s, v3_scale(K2[1] * z, sh[5]));
            s = v3_add(s, v3_scale(K2[2] * 2.f * -y, sh[6]));
            s = v3_add(s, v3_scale(K2[4] * 2.f * -y, sh[8]));
            ddy = v3_add(ddy, s);
            s = v3_scale(K2[1] * y, sh[5]);
            s = v3_add(s, v3_scale(K2[2] * 2.f * 2.f * z, sh[6]));
            s = v3_add(s, v3_scale(K2[3] * x, sh[7]));
            ddz = v3_add(ddz, s);
            if (deg > 2) {
                g[9] = v3_scale(K3[0] * y * (3.f * xx - yy), dRGB);
                g[10] = v3_scale(K3[1] * xy * z, dRGB);
                g[11] = v3_scale(K3[2] * y * (4.f * zz - xx - yy), dRGB);
                g[12] = v3_scale(K3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy), dRGB);
                g[13] = v3_scale(K3[4] * x * (4.f * zz - xx - yy), dRGB);
                g[14] = v3_scale(K3[5] * z * (xx - yy), dRGB);
                g[15] = v3_scale(K3[6] * x * (xx - 3.f * yy), dRGB);
                /* each term is (K * sh) * scalars..., left to right as written in the reference */
                s = v3_scale(xy, v3_scale(2.f, v3_scale(3.f, v3_scale(K3[0], sh[9]))));
                s = v3_add(s, v3_scale(yz, v3_scale(K3[1], sh[10])));
                s = v3_add(s, v3_scale(xy, v3_scale(-2.f, v3_scale(K3[2], sh[11]))));
                s = v3_add(s, v3_scale(xz, v3_scale(2.f, v3_scale(-3.f, v3_scale(K3[3], sh[12])))));
                s = v3_add(s, v3_scale(-3.f * xx + 4.f * zz - yy, v3_scale(K3[4], sh[13])));
                s = v3_add(s, v3_scale(xz, v3_scale(2.f, v3_scale(K3[5], sh[14]))));
                s = v3_add(s, v3_scale(xx - yy, v3_scale(3.f, v3_scale(K3[6], sh[15]))));
                ddx = v3_add(ddx, s);
                s = v3_scale(xx - yy, v3_scale(3.f, v3_scale(K3[0], sh[9])));
                s = v3_add(s, v3_scale(xz, v3_scale(K3[1], sh[10])));
                s = v3_add(s, v3_scale(-3.f * yy + 4.f * zz - xx, v3_scale(K3[2], sh[11])));
                s = v3_add(s, v3_scale(yz, v3_scale(2.f, v3_scale(-3.f, v3_scale(K3[3], sh[12])))));
                s = v3_add(s, v3_scale(xy, v3_scale(-2.f, v3_scale(K3[4], sh[13]))));
                s = v3_add(s, v3_scale(yz, v3_scale(-2.f, v3_scale(K3[5], sh[14]))));
                s = v3_add(s, v3_scale(xy, v3_scale(2.f, v3_scale(-3.f, v3_scale(K3[6], sh[15])))));
                ddy = v3_add(ddy, s);
                s = v3_scale(xy, v3_scale(K3[1], sh[10]));
                s = v3_add(s, v3_scale(yz, v3_scale(2.f, v3_scale(4.f, v3_scale(K3[2], sh[11])))));
                s = v3_add(s, v3_scale(2.f * zz - xx - yy, v3_scale(3.f, v3_scale(K3[3], sh[12]))));
                s = v3_add(s, v3_scale(xz, v3_scale(2.f, v3_scale(4.f, v3_scale(K3[4], sh[13])))));
                s = v3_add(s, v3_scale(xx - yy, v3_scale(K3[5], sh[14])));
                ddz = v3_add(ddz, s);
            }
        }
    }
    v3 dl = {v3_dot(ddx, dRGB), v3_dot(ddy, dRGB), v3_dot(ddz, dRGB)};
    /* auxiliary.h:107-117 */
    float sum2 = d0.x * d0.x + d0.y * d0.y + d0.z * d0.z;
    float inv32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    d_mean_acc[0] += ((+sum2 - d0.x * d0.x) * dl.x - d0.y * d0.x * dl.y - d0.z * d0.x * dl.z) * inv32;
    d_mean_acc[1] += (-d0.x * d0.y * dl.x + (sum2 - d0.y * d0.y) * dl.y - d0.z * d0.y * dl.z) * inv32;
    d_mean_acc[2] += (-d0.x * d0.z * dl.x - d0.y * d0.z * dl.y + (sum2 - d0.z * d0.z) * dl.z) * inv32;
}

/*
 * The SH stage of the backward pass on its own (dgr/cuda_rasterizer/backward.cu:20-139 as called from :390-391):
 * d_sh [P][M][3] is written for coefficients below (D+1)^2, d_means [P][3] receives (+=) the share that reaches
 * the position through the view direction.  Exists so that this stage can be compared with gradients the
 * reference's Python eval_sh yields under autograd.
 */
void ngs_oracle_sh_backward(int P, int D, int M, const float* means, const float* campos, const float* shs,
                            const uint8_t* clamped, const float* d_col, float* d_means, float* d_sh)
{
    for (int i = 0; i < P; i++)
        sh_backward(D, M, means + 3 * i, campos, shs + (size_t)3 * M * i, clamped + 3 * i, d_col + 3 * i,
                    d_means + 3 * i, d_sh + (size_t)3 * M * i);
}

/* dgr/cuda_rasterizer/backward.cu:278-341 */
static void cov3d_backward(const float* s_in, float mod, const float* q, const float* dcov, float* d_scale,
                           float* d_rot)
{
    float r = q[0], x = q[1], y = q[2], z = q[3];
    m3 R = rot_from_quat(q);
    float sx = mod * s_in[0], sy = mod * s_in[1], sz = mod * s_in[2];
    m3 M = m3_mul(diag3(sx, sy, sz), R);
    m3 dSig;
    dSig.c[0][0] = dcov[0]; dSig.c[0][1] = 0.5f * dcov[1]; dSig.c[0][2] = 0.5f * dcov[2];
    dSig.c[1][0] = 0.5f * dcov[1]; dSig.c[1][1] = dcov[3]; dSig.c[1][2] = 0.5f * dcov[4];
    dSig.c[2][0] = 0.5f * dcov[2]; dSig.c[2][1] = 0.5f * dcov[4]; dSig.c[2][2] = dcov[5];
    m3 M2;
    for (int j = 0; j < 3; j++)
        for (int i = 0; i < 3; i++) M2.c[j][i] = M.c[j][i] * 2.0f;
    m3 dM = m3_mul(M2, dSig);
    m3 Rt = m3_tr(R), dMt = m3_tr(dM);
    d_scale[0] = Rt.c[0][0] * dMt.c[0][0] + Rt.c[0][1] * dMt.c[0][1] + Rt.c[0][2] * dMt.c[0][2];
    d_scale[1] = Rt.c[1][0] * dMt.c[1][0] + Rt.c[1][1] * dMt.c[1][1] + Rt.c[1][2] * dMt.c[1][2];
    d_scale[2] = Rt.c[2][0] * dMt.c[2][0] + Rt.c[2][1] * dMt.c[2][1] + Rt.c[2][2] * dMt.c[2][2];
    for (int i = 0; i < 3; i++) { dMt.c[0][i] *= sx; dMt.c[1][i] *= sy; dMt.c[2][i] *= sz; }
    d_rot[0] = 2 * z * (dMt.c[0][1] - dMt.c[1][0]) + 2 * y * (dMt.c[2][0] - dMt.c[0][2]) + 2 * x * (dMt.c[1][2] - dMt.c[2][1]);
    d_rot[1] = 2 * y * (dMt.c[1][0] + dMt.c[0][1]) + 2 * z * (dMt.c[2][0] + dMt.c[0][2]) + 2 * r * (dMt.c[1][2] - dMt.c[2][1]) - 4 * x * (dMt.c[2][2] + dMt.c[1][1]);
    d_rot[2] = 2 * x * (dMt.c[1][0] + dMt.c[0][1]) + 2 * r * (dMt.c[2][0] - dMt.c[0][2]) + 2 * z * (dMt.c[1][2] + dMt.c[2][1]) - 4 * y * (dMt.c[2][2] + dMt.c[0][0]);
    d_rot[3] = 2 * r * (dMt.c[0][1] - dMt.c[1][0]) + 2 * x * (dMt.c[2][0] + dMt.c[0][2]) + 2 * y * (dMt.c[1][2] + dMt.c[2][1]) - 4 * z * (dMt.c[1][1] + dMt.c[0][0]);
}

/*
 * The 3D-covariance stage on its own, forward (dgr/cuda_rasterizer/forward.cu:118-152) and backward
 * (backward.cu:278-341).  Exist so that this stage can be compared with what the reference's Python
 * build_covariance_from_scaling_rotation (scene/gaussian_model.py:29-33) yields, also under autograd.
 */
void ngs_oracle_cov3d(int P, const float* scales, float mod, const float* rots, float* cov6)
{
    for (int i = 0; i < P; i++) cov3d_from_scale_rot(scales + 3 * i, mod, rots + 4 * i, cov6 + 6 * i);
}

void ngs_oracle_cov3d_backward(int P, const float* scales, float mod, const float* rots, const float* d_cov,
                               float* d_scale, float* d_rot)
{
    for (int i = 0; i < P; i++)
        cov3d_backward(scales + 3 * i, mod, rots + 4 * i, d_cov + 6 * i, d_scale + 3 * i, d_rot + 4 * i);
}

/*
 * dgr/cuda_rasterizer/backward.cu:346-396.  d_means must already hold the
 * cov2D part (ngs_oracle_cov2d_backward); this adds the projection and SH
 * parts.  shs / scales may be NULL exactly as in the reference.
 */
void ngs_oracle_preprocess_backward(int P, int D, int M, const float* means, const int* radii, const float* shs,
                                    const uint8_t* clamped, const float* scales, const float* rots, float mod,
                                    const float* pm, const float* campos, const float* d_xy, float* d_means,
                                    const float* d_col, const float* d_cov, float* d_sh, float* d_scale, float* d_rot)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        if (!(radii[i] > 0)) continue;
        const float* m = means + 3 * i;
        float mh[4];
        xform44(m, pm, mh);
        float mw = 1.0f / (mh[3] + 0.0000001f);
        float mul1 = (pm[0] * m[0] + pm[4] * m[1] + pm[8] * m[2] + pm[12]) * mw * mw;
        float mul2 = (pm[1] * m[0] + pm[5] * m[1] + pm[9] * m[2] + pm[13]) * mw * mw;
        float gx_ = d_xy[3 * i], gy_ = d_xy[3 * i + 1];
        float dmx = (pm[0] * mw - pm[3] * mul1) * gx_ + (pm[1] * mw - pm[3] * mul2) * gy_;
        float dmy = (pm[4] * mw - pm[7] * mul1) * gx_ + (pm[5] * mw - pm[7] * mul2) * gy_;
        float dmz = (pm[8] * mw - pm[11] * mul1) * gx_ + (pm[9] * mw - pm[11] * mul2) * gy_;
        d_means[3 * i] += dmx; d_means[3 * i + 1] += dmy; d_means[3 * i + 2] += dmz;
        if (shs)
            sh_backward(D, M, m, campos, shs + (size_t)3 * M * i, clamped + 3 * i, d_col + 3 * i, d_means + 3 * i,
                        d_sh + (size_t)3 * M * i);
        if (scales) cov3d_backward(scales + 3 * i, mod, rots + 4 * i, d_cov + 6 * i, d_scale + 3 * i, d_rot + 4 * i);
    }
}

int ngs_oracle_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void ngs_oracle_set_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}
