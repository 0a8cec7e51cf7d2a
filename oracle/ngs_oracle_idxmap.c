/*
 * ngs_oracle_idxmap.c -- CPU restatement of the reference's index-map path
 * (r2 = submodules/rasterizer2): every Gaussian centre is splatted as a square
 * of half-width 3/depth pixels, the entries are sorted by (pixel, depth), and
 * each pixel takes the first entry of its range, i.e. the nearest centre.
 * TEST INFRASTRUCTURE ONLY, like ngs_oracle.c.
 *
 * PARITY STATUS: "parity unpinned".  The reference holds no tests, fixtures or
 * outputs for this path and its CUDA sources cannot be built here; the
 * restatement below follows r2/cuda_rasterizer/raster.cu:10-66 (per-Gaussian
 * stage), rasterizer_impl.cu:578-615 (key emission), :680-726 (pixel ranges),
 * :731-817 (map construction), :820-847 (backward) and the pipeline order of
 * :851-984 line by line, including the sort (a stable sort on the whole key,
 * which is what the reference's radix sort over bits 0..62 produces).  It is
 * cross-checked in tests/ by a brute-force nearest-centre search.
 *
 * The backward restatement adds the contributions of the pixels in pixel
 * order; the reference's kernel does the same additions without atomics from
 * concurrent threads (a data race), so its result is one of the orderings at
 * best.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define R2_FEATURES 64      /* r2/cuda_rasterizer/config.h:3 */
#define R2_FREQS 4          /* r2/cuda_rasterizer/auxiliary.h:33 */
#define R2_PE_DIMS 24       /* r2/cuda_rasterizer/auxiliary.h:34 */
#define R2_PI 3.14159265358979323846f
#define R2_FIRST_COPIED 25  /* features 25..63 are copied from the Gaussian (rasterizer_impl.cu:797-799) */

static float ndc_to_pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); } /* r2 auxiliary.h:38-41 */

static int f2i(float f) /* float -> int as the GPU converts: saturating, NaN -> 0 */
{
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
}

typedef struct { uint64_t key; uint32_t val; uint32_t order; } entry_t;

static int entry_cmp(const void* a, const void* b)
{
    const entry_t* x = (const entry_t*)a;
    const entry_t* y = (const entry_t*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->order < y->order ? -1 : (x->order > y->order ? 1 : 0); /* stable */
}

/*
 * Forward.  Outputs (caller-zeroed, as the reference's binding creates them, rasterize_points.cu:210-214):
 * out_idx [H][W], out_color [H][W][3], out_depth [H][W], out_feature [H][W][64], radii [P].
 * Also returns the per-Gaussian intermediates for the tests: depths [P], pixels [P][2], touched [P].
 * Return value: number of (Gaussian, pixel) entries, the reference's num_rendered; -1 on allocation failure.
 */
long long ngs_oracle_idxmap_forward(int P, int W, int H, const float* means, const float* vm, const float* pm,
                                    const float* feature_vector, const float* campos, int* out_idx,
                                    float* out_color, float* out_depth, float* out_feature, int* radii,
                                    float* depths, int* pixels, uint32_t* touched)
{
    const float S = 3.0f; /* raster.cu:85 */
    float* img_xy = (float*)calloc((size_t)P * 2 + 1, sizeof(float));
    float* radii_f = (float*)calloc((size_t)P + 1, sizeof(float));
    uint32_t* offs = (uint32_t*)calloc((size_t)P + 1, sizeof(uint32_t));
    if (!img_xy || !radii_f || !offs) return -1;
    /* per-Gaussian stage, raster.cu:10-66 */
    for (int i = 0; i < P; i++) {
        radii[i] = 0;
        touched[i] = 0;
        const float* p = means + 3 * i;
        const float hx = pm[0] * p[0] + pm[4] * p[1] + pm[8] * p[2] + pm[12];
        const float hy = pm[1] * p[0] + pm[5] * p[1] + pm[9] * p[2] + pm[13];
        const float hw = pm[3] * p[0] + pm[7] * p[1] + pm[11] * p[2] + pm[15];
        const float pw = 1.0f / (hw + 0.0000001f);
        const float projx = hx * pw, projy = hy * pw;
        const float vz = vm[2] * p[0] + vm[6] * p[1] + vm[10] * p[2] + vm[14];
        if (vz <= 0.2f) continue;
        const int pix = f2i(ndc_to_pix(projx, W)), piy = f2i(ndc_to_pix(projy, H));
        radii_f[i] = S / vz;
        img_xy[2 * i] = ndc_to_pix(projx, W);
        img_xy[2 * i + 1] = ndc_to_pix(projy, H);
        const int tlx = f2i(fmaxf(0.0f, img_xy[2 * i] - radii_f[i])), tly = f2i(fmaxf(0.0f, img_xy[2 * i + 1] - radii_f[i]));
        const int brx = f2i(fminf((float)W, img_xy[2 * i] + radii_f[i] + 1.0f));
        const int bry = f2i(fminf((float)H, img_xy[2 * i + 1] + radii_f[i] + 1.0f));
        if (pix < 0 || pix >= W || piy < 0 || piy >= H) continue;
        depths[i] = vz;
        pixels[2 * i] = pix;
        pixels[2 * i + 1] = piy;
        radii[i] = 1;
        touched[i] = (uint32_t)((brx - tlx) * (bry - tly));
    }
    /* inclusive scan, rasterizer_impl.cu:914-919 */
    uint32_t run = 0;
    for (int i = 0; i < P; i++) { run += touched[i]; offs[i] = run; }
    const long long R = P > 0 ? (long long)offs[P - 1] : 0;
    entry_t* e = (entry_t*)malloc(((size_t)R + 1) * sizeof(entry_t));
    uint32_t* ranges = (uint32_t*)calloc((size_t)W * H * 2 + 2, sizeof(uint32_t));
    if (!e || !ranges) return -1;
    /* key emission, rasterizer_impl.cu:578-615: key = x << 48 | y << 32 | depth bits */
    for (int i = 0; i < P; i++) {
        if (radii[i] == 0) continue;
        uint32_t off = i == 0 ? 0 : offs[i - 1];
        const int tlx = f2i(fmaxf(0.0f, img_xy[2 * i] - radii_f[i])), tly = f2i(fmaxf(0.0f, img_xy[2 * i + 1] - radii_f[i]));
        const int brx = f2i(fminf((float)W, img_xy[2 * i] + radii_f[i] + 1.0f));
        const int bry = f2i(fminf((float)H, img_xy[2 * i + 1] + radii_f[i] + 1.0f));
        uint32_t dbits;
        memcpy(&dbits, &depths[i], 4);
        for (short x = (short)tlx; x < brx; x++)
            for (short y = (short)tly; y < bry; y++) {
                uint64_t key = (uint64_t)(uint16_t)x;
                key <<= 16;
                key |= (uint64_t)(uint16_t)y;
                key <<= 32;
                key |= dbits;
                e[off].key = key;
                e[off].val = (uint32_t)i;
                e[off].order = off;
                off++;
            }
    }
    /* stable sort on the key, rasterizer_impl.cu:947-952 */
    qsort(e, (size_t)R, sizeof(entry_t), entry_cmp);
    /* pixel ranges, rasterizer_impl.cu:680-726 */
    for (long long k = 0; k < R; k++) {
        const uint32_t hi = (uint32_t)(e[k].key >> 32);
        const uint32_t cur = (hi & 0xFFFFu) * (uint32_t)W + (hi >> 16);
        if (k == 0) ranges[2 * cur] = 0;
        else {
            const uint32_t ph = (uint32_t)(e[k - 1].key >> 32);
            const uint32_t prev = (ph & 0xFFFFu) * (uint32_t)W + (ph >> 16);
            if (cur != prev) {
                ranges[2 * prev + 1] = (uint32_t)k;
                ranges[2 * cur] = (uint32_t)k;
            }
        }
        if (k == R - 1) ranges[2 * cur + 1] = (uint32_t)R;
    }
    /* map construction, rasterizer_impl.cu:731-817 */
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t pix = (size_t)y * W + x;
            if (ranges[2 * pix + 1] == ranges[2 * pix]) { out_idx[pix] = -1; continue; }
            const int g = (int)e[ranges[2 * pix]].val;
            float dir[3] = {means[3 * g] - campos[0], means[3 * g + 1] - campos[1], means[3 * g + 2] - campos[2]};
            const float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
            dir[0] /= len; dir[1] /= len; dir[2] /= len;
            out_idx[pix] = g;
            float* f = out_feature + pix * R2_FEATURES;
            f[0] = depths[g];
            for (int dim = 0; dim < 3; dim++)
                for (int fr = 0; fr < R2_FREQS; fr++) {
                    const float frequency = powf(2.0f, (float)fr);
                    f[1 + dim * R2_FREQS * 2 + fr * 2] = sinf(frequency * R2_PI * dir[dim]);
                    f[1 + dim * R2_FREQS * 2 + fr * 2 + 1] = cosf(frequency * R2_PI * dir[dim]);
                }
            out_color[3 * pix] = dir[0];
            out_color[3 * pix + 1] = dir[1];
            out_color[3 * pix + 2] = dir[2];
            for (int k = R2_FIRST_COPIED; k < R2_FEATURES; k++) f[k] = feature_vector[(size_t)g * R2_FEATURES + k];
            out_depth[pix] = depths[g];
        }
    free(e); free(ranges); free(img_xy); free(radii_f); free(offs);
    return R;
}

/* Backward, rasterizer_impl.cu:820-847: dL_dfeature_vector [P][64] (caller-zeroed, rasterize_points.cu:279). */
void ngs_oracle_idxmap_backward(int P, int W, int H, const int* out_idx, const float* dL_dfeature,
                                float* dL_dfeature_vector)
{
    for (size_t pix = 0; pix < (size_t)W * H; pix++) {
        const int g = out_idx[pix];
        if (g < 0 || g >= P) continue;
        for (int k = R2_FIRST_COPIED; k < R2_FEATURES; k++)
            dL_dfeature_vector[(size_t)g * R2_FEATURES + k] += dL_dfeature[pix * R2_FEATURES + k];
    }
}
