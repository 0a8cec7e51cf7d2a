// Mean squared distance of every point to its three nearest neighbours: the
// quantity the reference's simple-knn module computes once per scene to size
// the initial Gaussians (simple_knn.cu:185-221, called from spatial.cu:15-26).
//
// The reference sorts the points by a 30-bit Morton code, cuts the sorted list
// into boxes of 1024, and lets every point (one thread each) walk ALL boxes,
// skipping those whose bounding box is farther than its current third-best
// distance.  The search is exact, so the result is defined by the inputs alone:
// the three smallest values of dx*dx + dy*dy + dz*dz over the other points
// (an identical point elsewhere in the list counts, with distance 0), averaged.
//
// Here the same Morton order is used differently:
//   * the sorted points are gathered into one contiguous float4 array
//     (x, y, z, original index), so a box of points is one coalesced load
//     instead of 64 dependent gathers;
//   * boxes are 64 points (one wave-wide load) with a second level of 64 boxes;
//   * a WAVE owns 64 consecutive queries (one leaf).  Its lanes test 64 boxes at
//     a time against the bounding box of the wave's queries, so the walk over
//     all boxes costs P/4096 + 64 lane-parallel tests per visited super box
//     instead of P/1024 serial tests per query; control flow stays wave-uniform;
//   * a surviving leaf is taken only if some lane still needs it (the per-point
//     test of the reference), and its 64 points are then broadcast lane by
//     lane from registers (v_readlane), no shared memory.
// The three best distances are kept sorted with min/max pairs, which equals the
// reference's insertion loop (simple_knn.cu:131-145).
#include <float.h>
#include "ngs_common.h"

#define KNN_LEAF 64

struct KnnBox { float4 lo, hi; };  // .w unused

// Order-preserving map float -> uint32 (for atomicMin/atomicMax on floats of either sign).
__device__ __forceinline__ uint32_t knn_encode(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float knn_decode(uint32_t u)
{
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__device__ __forceinline__ float wave_min_f(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max_f(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// bounds[0..2] = min, bounds[3..5] = max over all points AND the origin (the reference reduces with the initial
// value {0,0,0}, simple_knn.cu:191-200); the caller presets all six words to encode(0).
__global__ __launch_bounds__(256) void knn_bounds_kernel(int P, const float* __restrict__ points,
                                                         uint32_t* __restrict__ bounds)
{
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float c = points[(size_t)3 * i + a];
            if (isfinite(c)) {  // an infinite coordinate would flatten every Morton code of its axis to zero
                lo[a] = fminf(lo[a], c);
                hi[a] = fmaxf(hi[a], c);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
        lo[a] = wave_min_f(lo[a]);
        hi[a] = wave_max_f(hi[a]);
    }
    // one set of atomics per workgroup: thousands of waves hitting the same six words would serialise
    __shared__ float s_lo[4][3], s_hi[4][3];
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            s_lo[threadIdx.x >> 6][a] = lo[a];
            s_hi[threadIdx.x >> 6][a] = hi[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        const float l = fminf(fminf(s_lo[0][a], s_lo[1][a]), fminf(s_lo[2][a], s_lo[3][a]));
        const float h = fmaxf(fmaxf(s_hi[0][a], s_hi[1][a]), fmaxf(s_hi[2][a], s_hi[3][a]));
        if (l != FLT_MAX) atomicMin(&bounds[a], knn_encode(l));
        if (h != -FLT_MAX) atomicMax(&bounds[3 + a], knn_encode(h));
    }
}

__device__ __forceinline__ uint32_t knn_spread_bits(uint32_t x)  // simple_knn.cu:45-52
{
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

// key = 30-bit Morton code, value = point index (simple_knn.cu:54-70).
__global__ __launch_bounds__(256) void knn_morton_kernel(int P, const float* __restrict__ points,
                                                         const uint32_t* __restrict__ bounds,
                                                         uint32_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    uint32_t code = 0;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float lo = knn_decode(bounds[a]), hi = knn_decode(bounds[3 + a]);
        const float t = ((points[(size_t)3 * i + a] - lo) / (hi - lo)) * 1023.0f;
        code |= knn_spread_bits((uint32_t)t) << a;  // the conversion saturates and maps NaN to 0, as on the reference's GPU
    }
    keys[i] = code;
    vals[i] = (uint32_t)i;
}

// sorted[i] = (point of rank i, its original index); one bounding box per 64 ranks.
__global__ __launch_bounds__(256) void knn_gather_kernel(int P, const float* __restrict__ points,
                                                         const uint32_t* __restrict__ order,
                                                         float4* __restrict__ sorted, KnnBox* __restrict__ leaves)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    float3 lo = make_float3(FLT_MAX, FLT_MAX, FLT_MAX), hi = make_float3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    if (i < P) {
        const uint32_t src = order[i];
        float3 p = make_float3(points[(size_t)3 * src], points[(size_t)3 * src + 1], points[(size_t)3 * src + 2]);
        if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
            lo = hi = p;
        } else {
            // The reference's insertion ignores a NaN distance (simple_knn.cu:138), i.e. such a point is nobody's
            // neighbour: moved to infinity it yields infinite distances, which are never inserted either, and it
            // stays out of the bounding boxes.
            p = make_float3(INFINITY, INFINITY, INFINITY);
        }
        sorted[i] = make_float4(p.x, p.y, p.z, __uint_as_float(src));
    }
    lo.x = wave_min_f(lo.x); lo.y = wave_min_f(lo.y); lo.z = wave_min_f(lo.z);
    hi.x = wave_max_f(hi.x); hi.y = wave_max_f(hi.y); hi.z = wave_max_f(hi.z);
    if ((threadIdx.x & 63) == 0 && (i < P)) {
        KnnBox b;
        b.lo = make_float4(lo.x, lo.y, lo.z, 0.0f);
        b.hi = make_float4(hi.x, hi.y, hi.z, 0.0f);
        leaves[i / KNN_LEAF] = b;
    }
}

// One wave per super box: the union of up to 64 leaf boxes.
__global__ __launch_bounds__(64) void knn_super_kernel(int L, const KnnBox* __restrict__ leaves,
                                                       KnnBox* __restrict__ supers)
{
    const int l = blockIdx.x * 64 + threadIdx.x;
    float3 lo = make_float3(FLT_MAX, FLT_MAX, FLT_MAX), hi = make_float3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    if (l < L) {
        const KnnBox b = leaves[l];
        lo = make_float3(b.lo.x, b.lo.y, b.lo.z);
        hi = make_float3(b.hi.x, b.hi.y, b.hi.z);
    }
    lo.x = wave_min_f(lo.x); lo.y = wave_min_f(lo.y); lo.z = wave_min_f(lo.z);
    hi.x = wave_max_f(hi.x); hi.y = wave_max_f(hi.y); hi.z = wave_max_f(hi.z);
    if (threadIdx.x == 0) {
        KnnBox b;
        b.lo = make_float4(lo.x, lo.y, lo.z, 0.0f);
        b.hi = make_float4(hi.x, hi.y, hi.z, 0.0f);
        supers[blockIdx.x] = b;
    }
}

// Squared distance from p to the box, evaluated as the reference does (simple_knn.cu:119-129).  In floating point it
// never exceeds the squared distance from p to any point inside the box (subtraction, product and sum are monotonic
// and the sum is formed in the same order as for a point), so a skipped box cannot hold a better neighbour.
__device__ __forceinline__ float knn_box_point(const KnnBox& b, float3 p)
{
    float3 d = make_float3(0.0f, 0.0f, 0.0f);
    if (p.x < b.lo.x || p.x > b.hi.x) d.x = fminf(fabsf(p.x - b.lo.x), fabsf(p.x - b.hi.x));
    if (p.y < b.lo.y || p.y > b.hi.y) d.y = fminf(fabsf(p.y - b.lo.y), fabsf(p.y - b.hi.y));
    if (p.z < b.lo.z || p.z > b.hi.z) d.z = fminf(fabsf(p.z - b.lo.z), fabsf(p.z - b.hi.z));
    return d.x * d.x + d.y * d.y + d.z * d.z;
}

// Squared gap between two boxes: a lower bound of knn_box_point(b, p) for every p inside q, by the same argument.
__device__ __forceinline__ float knn_box_box(const KnnBox& b, const KnnBox& q)
{
    const float gx = fmaxf(0.0f, fmaxf(b.lo.x - q.hi.x, q.lo.x - b.hi.x));
    const float gy = fmaxf(0.0f, fmaxf(b.lo.y - q.hi.y, q.lo.y - b.hi.y));
    const float gz = fmaxf(0.0f, fmaxf(b.lo.z - q.hi.z, q.lo.z - b.hi.z));
    return gx * gx + gy * gy + gz * gz;
}

__device__ __forceinline__ float lane_value(float v, int l)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// Every lane offers its point to every lane's best-three list; `skip` is the lane whose own point must not count.
__device__ __forceinline__ void knn_scan_leaf(float4 mine, int n, float3 q, int skip, float& b0, float& b1, float& b2)
{
    for (int j = 0; j < n; j++) {
        const float px = lane_value(mine.x, j), py = lane_value(mine.y, j), pz = lane_value(mine.z, j);
        const float dx = px - q.x, dy = py - q.y, dz = pz - q.z;
        float d = dx * dx + dy * dy + dz * dz;
        d = j == skip ? FLT_MAX : d;
        const float c1 = fmaxf(b0, d);
        b0 = fminf(b0, d);
        const float c2 = fmaxf(b1, c1);
        b1 = fminf(b1, c1);
        b2 = fminf(b2, c2);
    }
}

__global__ __launch_bounds__(256) void knn_search_kernel(int P, int L, int S, const float4* __restrict__ sorted,
                                                         const KnnBox* __restrict__ leaves,
                                                         const KnnBox* __restrict__ supers,
                                                         float* __restrict__ mean_dist2)
{
    const int lane = threadIdx.x & 63;
    const int own = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));  // wave-uniform
    if (own >= L) return;
    const int n_own = min(KNN_LEAF, P - own * KNN_LEAF);
    const bool present = lane < n_own;
    float4 me = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (present) me = sorted[(size_t)own * KNN_LEAF + lane];
    const float3 q = make_float3(me.x, me.y, me.z);
    // A point with a non-finite coordinate has no defined neighbours; it must not drive the search either (its
    // rejection radius would never shrink and its wave would visit every leaf).  It keeps what its own leaf gave.
    const bool valid = present && isfinite(q.x) && isfinite(q.y) && isfinite(q.z);
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
    // the wave's own leaf first: it yields the rejection radius (the reference looks at the 3 neighbours either side
    // in Morton order for that, simple_knn.cu:156-163)
    knn_scan_leaf(me, n_own, q, lane, b0, b1, b2);
    const KnnBox qbox = leaves[own];
    float reach = wave_max_f(valid ? b2 : 0.0f);

    // One candidate leaf: taken only if some lane still needs it (the per-point test of simple_knn.cu:172).
    auto visit = [&](const int l) {
        const KnnBox box = leaves[l];
        const bool need = valid && !(knn_box_point(box, q) > b2);
        if (__ballot(need) == 0) return;
        const int n = min(KNN_LEAF, P - l * KNN_LEAF);
        float4 pt = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (lane < n) pt = sorted[(size_t)l * KNN_LEAF + lane];
        knn_scan_leaf(pt, n, q, -1, b0, b1, b2);
        reach = wave_max_f(valid ? b2 : 0.0f);
    };
    // Near things first, so that the rejection radius is small before the sweep over everything else: the leaves
    // next to this one in Morton order, then the super boxes outwards from the one that holds this leaf.
    constexpr int NEAR = 2;
    for (int d = 1; d <= NEAR; d++) {
        if (own - d >= 0) visit(own - d);
        if (own + d < L) visit(own + d);
    }
    const int chunks = (S + 63) / 64, home = (own / 64) / 64;
    for (int k = 0; k < chunks; k++) {
        for (int side = 0; side < 2; side++) {
            const int c = side ? home - k : home + k;
            if ((side && k == 0) || c < 0 || c >= chunks) continue;  // wave-uniform
            const int s0 = c * 64;
            bool pass = false;
            if (s0 + lane < S) pass = knn_box_box(supers[s0 + lane], qbox) <= reach;
            uint64_t smask = __ballot(pass);
            while (smask) {
                const int sb = s0 + __builtin_ctzll(smask);
                smask &= smask - 1;
                const int lf = sb * 64 + lane;
                bool pass2 = false;
                if (lf < L && abs(lf - own) > NEAR) pass2 = knn_box_box(leaves[lf], qbox) <= reach;
                uint64_t lmask = __ballot(pass2);
                while (lmask) {
                    visit(sb * 64 + __builtin_ctzll(lmask));
                    lmask &= lmask - 1;
                }
            }
        }
    }
    if (present) mean_dist2[__float_as_uint(me.w)] = (b0 + b1 + b2) / 3.0f;  // simple_knn.cu:182
}

struct KnnLayout {
    BinLayout bin;
    size_t bin_off, sorted, leaves, supers, bounds, total;
};

static KnnLayout knn_layout(size_t P)
{
    KnnLayout k;
    k.bin = ngs_bin_layout(P);
    const size_t L = (P + KNN_LEAF - 1) / KNN_LEAF, S = (L + 63) / 64;
    size_t o = 0;
    k.bin_off = o; o = ngs_align_up(o + k.bin.total);
    k.sorted = o; o = ngs_align_up(o + P * sizeof(float4));
    k.leaves = o; o = ngs_align_up(o + L * sizeof(KnnBox));
    k.supers = o; o = ngs_align_up(o + S * sizeof(KnnBox));
    k.bounds = o; o = ngs_align_up(o + 8 * 4);
    k.total = o + NGS_ALIGN;
    return k;
}

size_t ngs_knn_workspace_size(int P) { return knn_layout((size_t)(P > 0 ? P : 0)).total; }

void ngs_launch_knn(hipStream_t s, int P, const float* points, char* workspace, float* mean_dist2)
{
    if (P <= 0) return;
    const KnnLayout k = knn_layout((size_t)P);
    char* const base = ngs_align_ptr(workspace);
    char* const bb = base + k.bin_off;
    uint32_t* const keys[2] = {(uint32_t*)(bb + k.bin.keys[0]), (uint32_t*)(bb + k.bin.keys[1])};
    uint32_t* const vals[2] = {(uint32_t*)(bb + k.bin.vals[0]), (uint32_t*)(bb + k.bin.vals[1])};
    uint32_t* const hist = (uint32_t*)(bb + k.bin.hist);
    uint32_t* const totals = (uint32_t*)(bb + k.bin.totals);
    float4* const sorted = (float4*)(base + k.sorted);
    KnnBox* const leaves = (KnnBox*)(base + k.leaves);
    KnnBox* const supers = (KnnBox*)(base + k.supers);
    uint32_t* const bounds = (uint32_t*)(base + k.bounds);
    const int nb = (P + 255) / 256, L = (P + KNN_LEAF - 1) / KNN_LEAF, S = (L + 63) / 64;
    const int end_bit = 30;  // Morton bits
    const int in_buf = ngs_sort_input_buffer(end_bit);

    (void)hipMemsetD32Async((hipDeviceptr_t)bounds, (int)0x80000000u, 6, s);  // encode(0.0f)
    hipLaunchKernelGGL(knn_bounds_kernel, dim3(nb < 512 ? nb : 512), dim3(256), 0, s, P, points, bounds);
    hipLaunchKernelGGL(knn_morton_kernel, dim3(nb), dim3(256), 0, s, P, points, bounds, keys[in_buf], vals[in_buf]);
    ngs_launch_sort_tile_bits(s, nullptr, P, keys, vals, hist, totals, end_bit);  // stable: equal codes keep index order
    hipLaunchKernelGGL(knn_gather_kernel, dim3(nb), dim3(256), 0, s, P, points, vals[0], sorted, leaves);
    hipLaunchKernelGGL(knn_super_kernel, dim3(S), dim3(64), 0, s, L, leaves, supers);
    hipLaunchKernelGGL(knn_search_kernel, dim3((L + 3) / 4), dim3(256), 0, s, P, L, S, sorted, leaves, supers,
                       mean_dist2);
}
