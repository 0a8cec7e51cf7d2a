/*
 * ngs_oracle_knn.c -- CPU restatement of the reference's simple-knn module
 * (submodules/simple-knn): mean squared distance of every point to its three
 * nearest neighbours.  TEST INFRASTRUCTURE ONLY, like ngs_oracle.c.
 *
 * PARITY STATUS: "parity unpinned".  The reference holds no tests, fixtures or
 * outputs for this routine and its CUDA sources cannot be built here.  The
 * restatement follows simple_knn.cu:185-221 step by step: bounds reduced from
 * the initial value {0,0,0} (:191-200), 30-bit Morton codes (:45-70), a stable
 * sort of the point indices by code (:210-213), bounding boxes of 1024
 * consecutive sorted points (:78-117) and, per point, the walk over all boxes
 * with the rejection test and the three-best insertion (:131-183).  tests/
 * cross-checks it against a brute-force search and a k-d tree.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define KNN_BOX 1024 /* simple_knn.cu:12 */

static uint32_t spread_bits(uint32_t x) /* simple_knn.cu:45-52 */
{
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}

static uint32_t f2u(float f) /* float -> uint32 as the GPU converts: saturating, NaN -> 0 */
{
    if (f != f || f <= 0.0f) return 0;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}

static uint32_t morton_of(const float* c, const float* lo, const float* hi) /* simple_knn.cu:54-61 */
{
    const uint32_t x = spread_bits(f2u(((c[0] - lo[0]) / (hi[0] - lo[0])) * 1023.0f));
    const uint32_t y = spread_bits(f2u(((c[1] - lo[1]) / (hi[1] - lo[1])) * 1023.0f));
    const uint32_t z = spread_bits(f2u(((c[2] - lo[2]) / (hi[2] - lo[2])) * 1023.0f));
    return x | (y << 1) | (z << 2);
}

typedef struct { uint32_t code, index; } coded_t;

static int coded_cmp(const void* a, const void* b)
{
    const coded_t* x = (const coded_t*)a;
    const coded_t* y = (const coded_t*)b;
    if (x->code != y->code) return x->code < y->code ? -1 : 1;
    return x->index < y->index ? -1 : (x->index > y->index ? 1 : 0); /* stable: input order = index order */
}

static float box_point(const float* box, const float* p) /* simple_knn.cu:119-129; box = min xyz, max xyz */
{
    float d[3] = {0.0f, 0.0f, 0.0f};
    for (int a = 0; a < 3; a++)
        if (p[a] < box[a] || p[a] > box[3 + a]) d[a] = fminf(fabsf(p[a] - box[a]), fabsf(p[a] - box[3 + a]));
    return d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
}

static void update_best(const float* ref, const float* point, float* knn) /* simple_knn.cu:131-145 */
{
    const float dx = point[0] - ref[0], dy = point[1] - ref[1], dz = point[2] - ref[2];
    float dist = dx * dx + dy * dy + dz * dz;
    for (int j = 0; j < 3; j++)
        if (knn[j] > dist) {
            const float t = knn[j];
            knn[j] = dist;
            dist = t;
        }
}

/* points [P][3] -> mean_dist2 [P]; sorted_index [P] (may be NULL) receives the Morton order.  0 on success. */
int ngs_oracle_knn(int P, const float* points, float* mean_dist2, uint32_t* sorted_index)
{
    if (P <= 0) return 0;
    float lo[3] = {0.0f, 0.0f, 0.0f}, hi[3] = {0.0f, 0.0f, 0.0f}; /* simple_knn.cu:191 */
    for (int i = 0; i < P; i++)
        for (int a = 0; a < 3; a++) {
            lo[a] = fminf(lo[a], points[3 * (size_t)i + a]);
            hi[a] = fmaxf(hi[a], points[3 * (size_t)i + a]);
        }
    coded_t* order = (coded_t*)malloc((size_t)P * sizeof(coded_t));
    const int nboxes = (P + KNN_BOX - 1) / KNN_BOX;
    float* boxes = (float*)malloc((size_t)nboxes * 6 * sizeof(float));
    if (!order || !boxes) return -1;
    for (int i = 0; i < P; i++) {
        order[i].code = morton_of(points + 3 * (size_t)i, lo, hi);
        order[i].index = (uint32_t)i;
    }
    qsort(order, (size_t)P, sizeof(coded_t), coded_cmp);
    if (sorted_index)
        for (int i = 0; i < P; i++) sorted_index[i] = order[i].index;
    /* boxes of 1024 sorted points, simple_knn.cu:78-117 */
    for (int b = 0; b < nboxes; b++) {
        float* box = boxes + 6 * (size_t)b;
        for (int a = 0; a < 3; a++) { box[a] = FLT_MAX; box[3 + a] = -FLT_MAX; }
        const int end = (b + 1) * KNN_BOX < P ? (b + 1) * KNN_BOX : P;
        for (int i = b * KNN_BOX; i < end; i++)
            for (int a = 0; a < 3; a++) {
                const float c = points[3 * (size_t)order[i].index + a];
                box[a] = fminf(box[a], c);
                box[3 + a] = fmaxf(box[3 + a], c);
            }
    }
    /* per-point search, simple_knn.cu:147-183 */
#pragma omp parallel for schedule(dynamic, 256)
    for (int idx = 0; idx < P; idx++) {
        const float* point = points + 3 * (size_t)order[idx].index;
        float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
        const int from = idx - 3 > 0 ? idx - 3 : 0, to = idx + 3 < P - 1 ? idx + 3 : P - 1;
        for (int i = from; i <= to; i++) {
            if (i == idx) continue;
            update_best(point, points + 3 * (size_t)order[i].index, best);
        }
        const float reject = best[2];
        best[0] = best[1] = best[2] = FLT_MAX;
        for (int b = 0; b < nboxes; b++) {
            const float dist = box_point(boxes + 6 * (size_t)b, point);
            if (dist > reject || dist > best[2]) continue;
            const int end = (b + 1) * KNN_BOX < P ? (b + 1) * KNN_BOX : P;
            for (int i = b * KNN_BOX; i < end; i++) {
                if (i == idx) continue;
                update_best(point, points + 3 * (size_t)order[i].index, best);
            }
        }
        mean_dist2[order[idx].index] = (best[0] + best[1] + best[2]) / 3.0f;
    }
    free(order);
    free(boxes);
    return 0;
}
