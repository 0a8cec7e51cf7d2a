// Stable LSD radix sort of (32-bit key, 32-bit value) pairs, written for wave64 without any library.  Together
// with the order in which the instances are emitted it produces the result the reference obtains from
// cub::DeviceRadixSort::SortPairs over 64-bit keys (tile << 32) | depth bits
// (cuda_rasterizer/rasterizer_impl.cu:303-308): keys ascending, ties in input order.
//
// The key's structure is exploited instead of running ceil(end_bit/8) passes over all R instances with 12-byte
// pairs: only the tile id (one or two 8-bit passes at image sizes up to 4096x4096, on 8-byte (tile id, Gaussian
// index) pairs) is sorted over the R instances, and the depth order comes from one of two places.  Either the P
// Gaussians are sorted by their 32 depth bits first (four 8-bit passes over P pairs) and their instances are
// emitted in that order: sorting stably by tile a sequence that is ordered by (depth, index) gives exactly the
// order of a stable sort on (tile, depth) of index-ordered instances.  Or instances are emitted in index order and
// every tile's list is sorted by depth afterwards by one workgroup (ngs_launch_sort_segments).  The 64-bit keys themselves are never stored: nothing downstream reads them (the compositing kernels
// need the list and the tile ranges), and (tile << 32) | depth bits of each list entry can be re-derived from the
// ranges and the depth array whenever someone wants to look at them.
//
// Per pass: per-workgroup digit histogram (stored digit-major), one workgroup per digit turning its histogram row
// into exclusive prefixes, and a scatter that ranks pairs inside the workgroup in input order with wave ballots,
// reorders them through LDS so each digit's run leaves as contiguous stores, and writes them at
// digit base + workgroup prefix + rank.
//
// The number of pairs comes from device memory when the host does not know it yet (the instance count the scan
// kernel wrote), capped by the capacity the buffers were sized for; launches are sized by the capacity, so the
// passes can be queued before the host has seen the count.  The same passes sort the Morton codes of the
// nearest-neighbour search (knn.hip).
#include "ngs_common.h"

#define ST NGS_SORT_THREADS
#define SI NGS_SORT_ITEMS
#define STILE NGS_SORT_TILE
#define SWAVES (ST / 64)

__device__ __forceinline__ uint32_t digit_of(uint32_t key, int shift, uint32_t mask)
{
    return (key >> shift) & mask;
}

// Rank of this lane's item among the wave's items with the same digit that come earlier in input order, plus
// the number of such items in earlier rounds (kept in cnt[digit]); updates cnt.  All 64 lanes must call it.
template <int BITS = 8>
__device__ __forceinline__ uint32_t wave_rank(uint32_t d, bool valid, uint32_t* cnt, uint64_t lanes_below)
{
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < BITS; b++) {
        const bool bit = (d >> b) & 1u;
        const uint64_t m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    const uint32_t before = cnt[d];
    const uint32_t rank = before + (uint32_t)__popcll(peers & lanes_below);
    __builtin_amdgcn_wave_barrier();
    if (valid && (peers & lanes_below) == 0) cnt[d] = before + (uint32_t)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
    return rank;
}

__device__ __forceinline__ int64_t pair_count(const uint32_t* count, int64_t capacity)
{
    return count ? min((int64_t)*count, capacity) : capacity;
}

template <int ITEMS>
__global__ __launch_bounds__(ST) void sort_hist_kernel(const uint32_t* __restrict__ count, int64_t capacity,
                                                       const uint32_t* __restrict__ keys, int shift,
                                                       uint32_t mask, int nblocks, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t s_cnt[SWAVES][256];
    constexpr int TILE = ST * ITEMS;
    const int tid = threadIdx.x, wave = tid >> 6;
    const int64_t n = pair_count(count, capacity);
    if ((int64_t)blockIdx.x * TILE >= n) {  // a block past the end (launches are sized by the capacity)
        hist[(size_t)tid * nblocks + blockIdx.x] = 0;
        return;
    }
    for (int w = 0; w < SWAVES; w++) s_cnt[w][tid] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * TILE;
    if (base + TILE <= n && (uintptr_t)keys % 16 == 0) {
        // a full tile: four 16-byte loads per thread, all in flight before the first key is counted (the order in
        // which a histogram meets its keys does not matter)
        uint4 k4[ITEMS / 4];
#pragma unroll
        for (int i = 0; i < ITEMS / 4; i++) k4[i] = ((const uint4*)(keys + base))[i * ST + tid];
#pragma unroll
        for (int i = 0; i < ITEMS / 4; i++) {
            atomicAdd(&s_cnt[wave][digit_of(k4[i].x, shift, mask)], 1u);
            atomicAdd(&s_cnt[wave][digit_of(k4[i].y, shift, mask)], 1u);
            atomicAdd(&s_cnt[wave][digit_of(k4[i].z, shift, mask)], 1u);
            atomicAdd(&s_cnt[wave][digit_of(k4[i].w, shift, mask)], 1u);
        }
    } else {
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const int64_t p = base + i * ST + tid;
            if (p < n) atomicAdd(&s_cnt[wave][digit_of(keys[p], shift, mask)], 1u);
        }
    }
    __syncthreads();
    uint32_t c = 0;
    for (int w = 0; w < SWAVES; w++) c += s_cnt[w][tid];
    hist[(size_t)tid * nblocks + blockIdx.x] = c;
}

// Workgroup d: exclusive prefix over hist[d][0..nblocks) in place, total to totals[d].
// Thread t owns a contiguous run of ceil(nblocks/1024) entries.
#define SCAN_T 1024
__global__ __launch_bounds__(SCAN_T) void sort_scan_kernel(int nblocks, uint32_t* __restrict__ hist,
                                                           uint32_t* __restrict__ totals)
{
    __shared__ uint32_t wave_tot[SCAN_T / 64];
    uint32_t* row = hist + (size_t)blockIdx.x * nblocks;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (nblocks + SCAN_T - 1) / SCAN_T;
    const int lo = tid * per, hi = min(nblocks, lo + per);
    uint32_t s = 0;
    for (int i = lo; i < hi; i++) s += row[i];
    const uint32_t incl = wave_incl_scan_u32(s, lane);
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (int w = 0; w < SCAN_T / 64; w++) {
        const uint32_t t = wave_tot[w];
        if (w < wave) base += t;
        total += t;
    }
    uint32_t run = base + incl - s;
    for (int i = lo; i < hi; i++) {
        const uint32_t v = row[i];
        row[i] = run;
        run += v;
    }
    if (tid == 0) totals[blockIdx.x] = total;
}

template <int BITS, int ITEMS>
__global__ __launch_bounds__(ST) void sort_scatter_kernel(const uint32_t* __restrict__ count, int64_t capacity,
                                                          const uint32_t* __restrict__ kin,
                                                          const uint32_t* __restrict__ vin,
                                                          uint32_t* __restrict__ kout, uint32_t* __restrict__ vout,
                                                          int shift, uint32_t mask, int nblocks,
                                                          const uint32_t* __restrict__ hist,
                                                          const uint32_t* __restrict__ totals)
{
    __shared__ uint32_t s_cnt[SWAVES][256];  // per-wave digit counts, then per-wave exclusive bases
    __shared__ uint32_t s_start[256];        // first local slot of each digit
    __shared__ uint32_t s_gofs[256];         // global position of local slot 0 of each digit (minus s_start)
    __shared__ uint32_t s_wtot[SWAVES];
    __shared__ uint32_t s_gwtot[SWAVES];
    constexpr int TILE = ST * ITEMS;
    __shared__ uint32_t s_key[TILE];
    __shared__ uint32_t s_val[TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n = pair_count(count, capacity);
    const int64_t base = (int64_t)blockIdx.x * TILE;
    if (base >= n) return;  // block-uniform
    const int n_valid = (int)min((int64_t)TILE, n - base);
    for (int w = 0; w < SWAVES; w++) s_cnt[w][tid] = 0;
    __syncthreads();

    // Wave-striped ownership keeps input order: wave w owns slots [w*64*ITEMS, (w+1)*64*ITEMS), item i of lane l is
    // slot w*64*ITEMS + i*64 + l.
    uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
    const uint64_t lanes_below = (1ull << lane) - 1ull;
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const int slot = wave * (64 * ITEMS) + i * 64 + lane;
        const bool valid = slot < n_valid;
        key[i] = valid ? kin[base + slot] : ~0u;
        val[i] = valid ? vin[base + slot] : 0u;
    }
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const int slot = wave * (64 * ITEMS) + i * 64 + lane;
        rank[i] = wave_rank<BITS>(digit_of(key[i], shift, mask), slot < n_valid, s_cnt[wave], lanes_below);
    }
    __syncthreads();

    // Digit `tid`: per-wave exclusive bases, workgroup total, local start (exclusive scan over digits).
    uint32_t tot = 0;
    for (int w = 0; w < SWAVES; w++) {
        const uint32_t c = s_cnt[w][tid];
        s_cnt[w][tid] = tot;
        tot += c;
    }
    const uint32_t incl = wave_incl_scan_u32(tot, lane);
    // The same pattern scans the global digit totals to get each digit's base in the output.
    const uint32_t gtot = totals[tid];
    const uint32_t gincl = wave_incl_scan_u32(gtot, lane);
    if (lane == 63) { s_wtot[wave] = incl; s_gwtot[wave] = gincl; }
    __syncthreads();
    uint32_t lbase = 0, gbase = 0;
    for (int w = 0; w < wave; w++) { lbase += s_wtot[w]; gbase += s_gwtot[w]; }
    const uint32_t lstart = lbase + incl - tot;
    s_start[tid] = lstart;
    s_gofs[tid] = (gbase + gincl - gtot) + hist[(size_t)tid * nblocks + blockIdx.x] - lstart;
    __syncthreads();

#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const int slot = wave * (64 * ITEMS) + i * 64 + lane;
        if (slot < n_valid) {
            const uint32_t d = digit_of(key[i], shift, mask);
            const uint32_t pos = s_start[d] + s_cnt[wave][d] + rank[i];
            s_key[pos] = key[i];
            s_val[pos] = val[i];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const int pos = i * ST + tid;
        if (pos < n_valid) {
            const uint32_t k = s_key[pos];
            const uint32_t g = s_gofs[digit_of(k, shift, mask)] + (uint32_t)pos;
            kout[g] = k;
            vout[g] = s_val[pos];
        }
    }
}

// ---------------------------------------------------------------------------
// Per-tile stable sort by the 32 depth bits: the other way to the depth order (see ngs_launch_sort_segments).
// ---------------------------------------------------------------------------
// One 8-bit pass over the (up to 256*ITEMS) items a workgroup holds in registers: ranks them stably by digit and
// returns each item's position among the workgroup's items; tot_out = items with digit `tid` in the workgroup.
// `quota` = slots owned by each wave (multiple of 64); item i of lane l of wave w is slot w*quota + i*64 + l.
template <int ITEMS, int WAVES = SWAVES>
__device__ __forceinline__ void block_rank_pass(const uint32_t (&key)[ITEMS], int shift, int n_valid, int quota,
                                                uint32_t (*s_cnt)[256], uint32_t* s_start, uint32_t* s_wtot,
                                                uint32_t (&pos)[ITEMS], uint32_t& tot_out)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool digit_thread = tid < 256;  // threads 0..255 also act for digit `tid` (all of them when WAVES == 4)
    const uint64_t lanes_below = (1ull << lane) - 1ull;
    if (digit_thread)
        for (int w = 0; w < WAVES; w++) s_cnt[w][tid] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        if (i * 64 < quota) {  // wave-uniform
            const int slot = wave * quota + i * 64 + lane;
            pos[i] = wave_rank((key[i] >> shift) & 0xffu, slot < n_valid, s_cnt[wave], lanes_below);
        }
    }
    __syncthreads();
    uint32_t tot = 0;
    if (digit_thread)
        for (int w = 0; w < WAVES; w++) {
            const uint32_t c = s_cnt[w][tid];
            s_cnt[w][tid] = tot;
            tot += c;
        }
    const uint32_t incl = wave_incl_scan_u32(tot, lane);
    if (lane == 63 && digit_thread) s_wtot[wave] = incl;
    __syncthreads();
    if (digit_thread) {
        uint32_t lbase = 0;
        for (int w = 0; w < wave; w++) lbase += s_wtot[w];
        s_start[tid] = lbase + incl - tot;
    }
    tot_out = tot;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        if (i * 64 < quota) {
            const uint32_t d = (key[i] >> shift) & 0xffu;
            pos[i] += s_start[d] + s_cnt[wave][d];
        }
    }
}

// Per-tile sort, common case: lists whose depths are spread out.  One distribution step instead of four radix
// passes: every entry goes to bucket (depth bits - smallest depth bits) >> s, with s chosen so that the tile's depth
// range spans between NB/2 and NB buckets (the bit pattern of a positive float grows with its value, so buckets are
// ordered by depth); buckets then hold a handful of entries each, and an entry's final place is its bucket's start
// plus the number of bucket mates that precede it in (depth, position in the incoming list) order -- the stable
// order by depth.  Five barriers per tile, 6 bytes of LDS per entry.
// One workgroup per tile takes lists of up to SEG_ITEMS * THREADS entries (in < 0), or workgroups share out the
// tiles of work list `in`.  A tile whose depths are all equal is sorted already.  Tiles a launch cannot take (longer
// lists, or a bucket with more than SEG_BUCKET_LIMIT entries: many equal or nearly equal depths) are appended to
// work list `out` for the next kernel.
// work lists (see ImageLayout::hard): word 0 / 1 = length of list 0 / 1, list k at words 4 + k T.
//
// RANKED: the entries bring their depth along instead of having it gathered, one scattered word per entry, from the
// per-Gaussian depth array: a rank r = (depth bits - smallest depth bits of the frame) >> shift (see DepthRank) is
// split over the low bits of the entry's sort key and the bits of its list word above the Gaussian index, both of
// which this kernel reads as contiguous streams.  When the frame's depth span fits the rank's bits (shift == 0, the
// usual case) the rank orders exactly like the depth bits and is the sort key here; when it does not, the kernel
// gathers the depths after all.  Either way the list leaves as plain Gaussian indices.
#define SEG_ITEMS 16
#define SEG_BUCKET_LIMIT 128
template <int THREADS, bool RANKED>
__global__ __launch_bounds__(THREADS, THREADS == 256 ? 5 : 2) void segment_sort_bucket_kernel(int T, const uint2* __restrict__ ranges,
                                                                      const float* __restrict__ depth,
                                                                      const uint32_t* __restrict__ keys0,
                                                                      const DepthRank dr,
                                                                      const uint32_t* __restrict__ misc,
                                                                      uint32_t* __restrict__ vals0,
                                                                      uint32_t* __restrict__ hard, int in, int out)
{
    constexpr int CAP = SEG_ITEMS * THREADS, NB = 8 * THREADS, WAVES = THREADS / 64;
    const uint32_t idx_mask = RANKED ? dr.idx_mask() : 0xffffffffu;
    // block-uniform: the rank in the keys is as good as the depth
    const bool exact = RANKED && dr.shift(misc[NGS_MISC_DMIN], misc[NGS_MISC_DMAX]) == 0;
    __shared__ uint32_t s_cnt[NB];   // bucket sizes, then each bucket's cursor (its end once all entries sit)
    __shared__ uint32_t s_key[CAP];  // depth bits grouped by bucket
    __shared__ uint16_t s_slot[CAP]; // and where in the incoming list each came from
    // the three rows of per-wave partial results live in the first words of s_key, which is not written before the last
    // of them has been read (a barrier lies between): on their own they would push the kernel past a fifth of the LDS
    uint32_t(*const s_red)[WAVES] = (uint32_t(*)[WAVES])s_key;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n_work = in < 0 ? (uint32_t)T : hard[in];
    for (uint32_t w = blockIdx.x; w < n_work; w += gridDim.x) {  // one trip when every tile has its own workgroup
        const uint32_t tile = in < 0 ? w : hard[4 + (size_t)in * T + w];
        const uint2 range = ranges[tile];
        const int n = (int)(range.y - range.x);
        const size_t lo = range.x;
        if (n < 2) {  // block-uniform, like every branch on n below
            if (RANKED && n == 1 && tid == 0) vals0[lo] &= idx_mask;
            continue;
        }
        if (n > CAP) {
            if (tid == 0) hard[4 + (size_t)out * T + atomicAdd(&hard[out], 1u)] = tile;
            continue;
        }
        __syncthreads();  // a previous trip's readers of the LDS tables are done
#pragma unroll
        for (int j = 0; j < NB / THREADS; j++) s_cnt[j * THREADS + tid] = 0;
        uint32_t key[SEG_ITEMS], val[SEG_ITEMS];
#pragma unroll
        for (int i = 0; i < SEG_ITEMS; i++) {
            const int slot = i * THREADS + tid;
            val[i] = (i * THREADS < n && slot < n) ? vals0[lo + slot] : 0u;
            if (RANKED) key[i] = (exact && i * THREADS < n && slot < n) ? keys0[lo + slot] : 0u;
        }
        // all list entries are on their way before the first dependent depth gather is issued
        uint32_t kmin = ~0u, kmax = 0u;
#pragma unroll
        for (int i = 0; i < SEG_ITEMS; i++) {
            const int slot = i * THREADS + tid;
            if (i * THREADS < n && slot < n) {
                if (RANKED) {
                    const uint32_t word = val[i];
                    val[i] = word & idx_mask;
                    key[i] = exact ? dr.join(key[i], word) : __float_as_uint(depth[val[i]]);
                } else {
                    key[i] = __float_as_uint(depth[val[i]]);
                }
                kmin = min(kmin, key[i]);
                kmax = max(kmax, key[i]);
            } else {
                key[i] = 0u;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, o, 64));
            kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, o, 64));
        }
        if (lane == 0) { s_red[0][wave] = kmin; s_red[1][wave] = kmax; }
        __syncthreads();
#pragma unroll
        for (int v = 0; v < WAVES; v++) { kmin = min(kmin, s_red[0][v]); kmax = max(kmax, s_red[1][v]); }
        const uint32_t span = kmax - kmin;
        if (span == 0u) {  // one depth for the whole list: the incoming order is the answer
            if (RANKED) {
#pragma unroll
                for (int i = 0; i < SEG_ITEMS; i++) {
                    const int slot = i * THREADS + tid;
                    if (i * THREADS < n && slot < n) vals0[lo + slot] = val[i];
                }
            }
            continue;
        }
        int sh = 32 - __clz((int)span);  // bit length of the span
        sh = max(0, sh - (THREADS == 256 ? 11 : 12));  // span >> sh < NB
        static_assert(THREADS == 256 || THREADS == 512, "NB = 8 THREADS is 2^11 or 2^12");

#pragma unroll
        for (int i = 0; i < SEG_ITEMS; i++) {
            const int slot = i * THREADS + tid;
            if (i * THREADS < n && slot < n) atomicAdd(&s_cnt[(key[i] - kmin) >> sh], 1u);
        }
        __syncthreads();
        bool too_full;
        {   // exclusive scan over the buckets (thread t: buckets 8t .. 8t+7), and the largest bucket
            uint32_t c[8], sum = 0, big = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) { c[j] = s_cnt[8 * tid + j]; sum += c[j]; big = max(big, c[j]); }
            const uint32_t incl = wave_incl_scan_u32(sum, lane);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) big = max(big, (uint32_t)__shfl_xor((int)big, o, 64));
            if (lane == 63) s_red[0][wave] = incl;
            if (lane == 0) s_red[2][wave] = big;
            __syncthreads();
            uint32_t run = incl - sum;
#pragma unroll
            for (int v = 0; v < WAVES; v++) {
                if (v < wave) run += s_red[0][v];
                big = max(big, s_red[2][v]);
            }
            too_full = big > SEG_BUCKET_LIMIT;
            if (!too_full) {
#pragma unroll
                for (int j = 0; j < 8; j++) { s_cnt[8 * tid + j] = run; run += c[j]; }
            }
        }
        if (too_full) {
            if (tid == 0) hard[4 + (size_t)out * T + atomicAdd(&hard[out], 1u)] = tile;
            continue;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SEG_ITEMS; i++) {
            const int slot = i * THREADS + tid;
            if (i * THREADS < n && slot < n) {
                const uint32_t p = atomicAdd(&s_cnt[(key[i] - kmin) >> sh], 1u);
                s_key[p] = key[i];
                s_slot[p] = (uint16_t)slot;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SEG_ITEMS; i++) {
            const int slot = i * THREADS + tid;
            if (i * THREADS < n && slot < n) {
                const uint32_t b = (key[i] - kmin) >> sh;
                const uint32_t begin = b ? s_cnt[b - 1] : 0u, end = s_cnt[b];
                uint32_t r = begin;
                for (uint32_t e = begin; e < end; e++) {
                    const uint32_t k = s_key[e];
                    r += (k < key[i] || (k == key[i] && s_slot[e] < (uint16_t)slot)) ? 1u : 0u;
                }
                vals0[lo + r] = val[i];
            }
        }
    }
}

// Lists the kernel above cannot take, first step: split by depth.  One workgroup streams a tile's list three times
// (depth extremes; bucket counts; scatter) and leaves it in the spare buffers grouped into SPLIT_NB buckets by
// (depth key - the list's smallest) >> s, buckets in order of depth, entries inside a bucket in no particular order.
// Runs of whole buckets then form sub-lists: a sub-list starts at the first bucket that starts inside each block of
// SPLIT_GROUP entries, so with no bucket above SPLIT_GROUP entries every sub-list is shorter than 2 SPLIT_GROUP = what
// one workgroup of segment_sort_sub_kernel holds.  Sub-lists are appended to `subs` ((first, end) positions in the
// frame's list; hard[2] counts them).  Lists with a fuller bucket (many equal or nearly equal depths), with more than
// SPLIT_MAX entries, or that find `subs` full go to work list `out` for the general kernel; a list whose depths are
// all equal is in order already.  Inside a tile the incoming list is in ascending order of the Gaussian index (the
// instances were emitted in index order and the tile-level passes are stable), so "ties in incoming order" is
// "ties in index order" and the incoming positions need not be kept.
#define SPLIT_NB 2048
#define SPLIT_GROUP 2048
#define SPLIT_MAX_GROUPS 256
#define SPLIT_MAX (SPLIT_GROUP * SPLIT_MAX_GROUPS)
template <bool RANKED>
__global__ __launch_bounds__(ST) void segment_split_kernel(int T, const uint2* __restrict__ ranges,
                                                           const float* __restrict__ depth,
                                                           const uint32_t* __restrict__ keys0, uint32_t* __restrict__ vals0,
                                                           uint32_t* __restrict__ keys1, uint32_t* __restrict__ vals1,
                                                           const DepthRank dr, const uint32_t* __restrict__ misc,
                                                           uint32_t* __restrict__ hard, int in, int out,
                                                           uint2* __restrict__ subs, uint32_t subs_cap)
{
    static_assert(SPLIT_NB == 8 * ST, "thread t scans buckets 8t .. 8t+7");
    const uint32_t idx_mask = RANKED ? dr.idx_mask() : 0xffffffffu;
    const bool exact = RANKED && dr.shift(misc[NGS_MISC_DMIN], misc[NGS_MISC_DMAX]) == 0;
    __shared__ uint32_t s_cnt[SPLIT_NB];
    __shared__ uint32_t s_first[SPLIT_MAX_GROUPS + 1];  // smallest bucket start inside each block of SPLIT_GROUP entries
    __shared__ uint32_t s_red[3][SWAVES];
    __shared__ uint32_t s_word[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n_work = hard[in];
    for (uint32_t w = blockIdx.x; w < n_work; w += gridDim.x) {
        const uint32_t tile = hard[4 + (size_t)in * T + w];
        const uint2 range = ranges[tile];
        const uint32_t n = range.y - range.x;
        const size_t lo = range.x;
        if (n > SPLIT_MAX) {  // block-uniform, like every branch below that leaves the trip
            if (tid == 0) hard[4 + (size_t)out * T + atomicAdd(&hard[out], 1u)] = tile;
            continue;
        }
        auto key_of = [&](uint32_t i, uint32_t& index) {
            const uint32_t word = vals0[lo + i];
            index = word & idx_mask;
            return exact ? dr.join(keys0[lo + i], word) : __float_as_uint(depth[index]);
        };
        __syncthreads();  // the previous trip is done with the tables
        uint32_t kmin = ~0u, kmax = 0u;
        for (uint32_t i = tid; i < n; i += ST) {
            uint32_t g;
            const uint32_t k = key_of(i, g);
            kmin = min(kmin, k);
            kmax = max(kmax, k);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, o, 64));
            kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, o, 64));
        }
        if (lane == 0) { s_red[0][wave] = kmin; s_red[1][wave] = kmax; }
#pragma unroll
        for (int j = 0; j < SPLIT_NB / ST; j++) s_cnt[j * ST + tid] = 0;
        for (int j = tid; j <= SPLIT_MAX_GROUPS; j += ST) s_first[j] = ~0u;
        __syncthreads();
#pragma unroll
        for (int v = 0; v < SWAVES; v++) { kmin = min(kmin, s_red[0][v]); kmax = max(kmax, s_red[1][v]); }
        const uint32_t span = kmax - kmin;
        if (span == 0u) {  // one depth for the whole list: the incoming order is the answer
            if (RANKED)
                for (uint32_t i = tid; i < n; i += ST) vals0[lo + i] &= idx_mask;
            continue;
        }
        const int sh = max(0, 32 - __clz((int)span) - 11);  // span >> sh < SPLIT_NB
        for (uint32_t i = tid; i < n; i += ST) {
            uint32_t g;
            atomicAdd(&s_cnt[(key_of(i, g) - kmin) >> sh], 1u);
        }
        __syncthreads();
        uint32_t c[8], sum = 0, big = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) { c[j] = s_cnt[8 * tid + j]; sum += c[j]; big = max(big, c[j]); }
        const uint32_t incl = wave_incl_scan_u32(sum, lane);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) big = max(big, (uint32_t)__shfl_xor((int)big, o, 64));
        if (lane == 63) s_red[0][wave] = incl;
        if (lane == 0) s_red[2][wave] = big;
        __syncthreads();
        uint32_t run = incl - sum;
#pragma unroll
        for (int v = 0; v < SWAVES; v++) {
            if (v < wave) run += s_red[0][v];
            big = max(big, s_red[2][v]);
        }
        // blocks of SPLIT_GROUP entries the list covers: one slot of `subs` each (a block in which no bucket starts --
        // only the last can be one while no bucket is above SPLIT_GROUP entries -- leaves its slot an empty sub-list)
        const uint32_t groups = (n - 1u) / SPLIT_GROUP + 1u;
        if (tid == 0) s_word[0] = big <= SPLIT_GROUP ? atomicAdd(&hard[2], groups) : ~0u;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            s_cnt[8 * tid + j] = run;  // the bucket's cursor
            if (c[j]) atomicMin(&s_first[run / SPLIT_GROUP], run);
            run += c[j];
        }
        __syncthreads();
        const uint32_t base = s_word[0];
        const bool room = base != ~0u && base + groups <= subs_cap;
        if (base != ~0u) {
            for (uint32_t j = tid; j < groups; j += ST) {
                if (base + j >= subs_cap) break;
                const uint32_t first = s_first[j];
                uint32_t end = n;
                for (uint32_t k = j + 1; k < groups; k++)
                    if (s_first[k] != ~0u) { end = s_first[k]; break; }
                subs[base + j] = (!room || first == ~0u) ? make_uint2(0u, 0u)
                                                         : make_uint2((uint32_t)lo + first, (uint32_t)lo + end);
            }
        }
        if (!room) {
            if (tid == 0) hard[4 + (size_t)out * T + atomicAdd(&hard[out], 1u)] = tile;
            continue;
        }
        for (uint32_t i = tid; i < n; i += ST) {
            uint32_t g;
            const uint32_t k = key_of(i, g);
            const uint32_t p = atomicAdd(&s_cnt[(k - kmin) >> sh], 1u);
            keys1[lo + p] = k;
            vals1[lo + p] = g;
        }
    }
}

// Second step: every sub-list the split left in the spare buffers (keys1: depth keys, vals1: Gaussian indices;
// at most SUB_CAP entries, unordered inside) is put in (depth key, index) order by one workgroup and written to its
// place in the list: the distribution step of segment_sort_bucket_kernel with the tie broken by the index.  A
// sub-list whose keys are all equal is bucketed by the index instead.  Crowded buckets are simply walked (a
// sub-list of a few thousand entries bounds the cost), so this kernel hands nothing on.
#define SUB_CAP (2 * SPLIT_GROUP)
__global__ __launch_bounds__(ST, 3) void segment_sort_sub_kernel(const uint2* __restrict__ subs, uint32_t subs_cap,
                                                                 const uint32_t* __restrict__ hard,
                                                                 const uint32_t* __restrict__ keys1,
                                                                 const uint32_t* __restrict__ vals1,
                                                                 uint32_t* __restrict__ vals0)
{
    constexpr int ITEMS = SUB_CAP / ST, NB = 8 * ST;
    __shared__ uint32_t s_cnt[NB];
    __shared__ uint32_t s_key[SUB_CAP];
    __shared__ uint32_t s_val[SUB_CAP];
    __shared__ uint32_t s_red[2][SWAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n_work = min(hard[2], subs_cap);
    for (uint32_t w = blockIdx.x; w < n_work; w += gridDim.x) {
        const uint2 range = subs[w];
        const int n = (int)(range.y - range.x);
        const size_t lo = range.x;
        if (n < 1 || n > SUB_CAP) continue;  // block-uniform (longer ones do not occur: see the split kernel)
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NB / ST; j++) s_cnt[j * ST + tid] = 0;
        uint32_t key[ITEMS], val[ITEMS];
        uint32_t kmin = ~0u, kmax = 0u, vmin = ~0u, vmax = 0u;
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const int slot = i * ST + tid;
            const bool valid = i * ST < n && slot < n;
            key[i] = valid ? keys1[lo + slot] : 0u;
            val[i] = valid ? vals1[lo + slot] : 0u;
            if (valid) {
                kmin = min(kmin, key[i]); kmax = max(kmax, key[i]);
                vmin = min(vmin, val[i]); vmax = max(vmax, val[i]);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, o, 64));
            kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, o, 64));
            vmin = min(vmin, (uint32_t)__shfl_xor((int)vmin, o, 64));
            vmax = max(vmax, (uint32_t)__shfl_xor((int)vmax, o, 64));
        }
        if (lane == 0) { s_key[wave] = kmin; s_key[SWAVES + wave] = kmax; s_val[wave] = vmin; s_val[SWAVES + wave] = vmax; }
        __syncthreads();
#pragma unroll
        for (int v = 0; v < SWAVES; v++) {
            kmin = min(kmin, s_key[v]); kmax = max(kmax, s_key[SWAVES + v]);
            vmin = min(vmin, s_val[v]); vmax = max(vmax, s_val[SWAVES + v]);
        }
        const bool by_index = kmax == kmin;  // one depth: the order is the index order
        const uint32_t bmin = by_index ? vmin : kmin, span = by_index ? vmax - vmin : kmax - kmin;
        const int sh = span ? max(0, 32 - __clz((int)span) - 11) : 0;  // span >> sh < NB
        static_assert(NB == 2048, "11 bits of bucket number");
        __syncthreads();  // the extremes have been read out of s_key / s_val
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const int slot = i * ST + tid;
            if (i * ST < n && slot < n) atomicAdd(&s_cnt[((by_index ? val[i] : key[i]) - bmin) >> sh], 1u);
        }
        __syncthreads();
        {
            uint32_t c[8], sum = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) { c[j] = s_cnt[8 * tid + j]; sum += c[j]; }
            const uint32_t incl = wave_incl_scan_u32(sum, lane);
            if (lane == 63) s_red[0][wave] = incl;
            __syncthreads();
            uint32_t run = incl - sum;
#pragma unroll
            for (int v = 0; v < SWAVES; v++)
                if (v < wave) run += s_red[0][v];
#pragma unroll
            for (int j = 0; j < 8; j++) { s_cnt[8 * tid + j] = run; run += c[j]; }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const int slot = i * ST + tid;
            if (i * ST < n && slot < n) {
                const uint32_t p = atomicAdd(&s_cnt[((by_index ? val[i] : key[i]) - bmin) >> sh], 1u);
                s_key[p] = key[i];
                s_val[p] = val[i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const int slot = i * ST + tid;
            if (i * ST < n && slot < n) {
                const uint32_t b = ((by_index ? val[i] : key[i]) - bmin) >> sh;
                const uint32_t begin = b ? s_cnt[b - 1] : 0u, end = s_cnt[b];
                uint32_t r = begin;
                for (uint32_t e = begin; e < end; e++) {
                    const uint32_t k = s_key[e];
                    r += (k < key[i] || (k == key[i] && s_val[e] < val[i])) ? 1u : 0u;
                }
                vals0[lo + r] = val[i];
            }
        }
    }
}

// The general per-tile sort: four stable 8-bit passes over the depth bits, chunk by chunk, ping-ponging through the
// spare buffers (depth bits in keys0 / keys1, the list in vals0 / vals1; four passes end where they started, in
// vals0).  Any list length, any distribution of depths; workgroups share out the tiles of work list `in`.
__global__ __launch_bounds__(ST) void segment_sort_long_kernel(int T, const uint2* __restrict__ ranges,
                                                               const float* __restrict__ depth, uint32_t* keys0,
                                                               uint32_t* vals0, uint32_t* keys1, uint32_t* vals1,
                                                               const uint32_t* __restrict__ hard, int in,
                                                               uint32_t idx_mask)
{
    __shared__ uint32_t s_cnt[SWAVES][256];
    __shared__ uint32_t s_start[256];
    __shared__ uint32_t s_base[256];
    __shared__ uint32_t s_wtot[SWAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n_work = hard[in];
    for (uint32_t h = blockIdx.x; h < n_work; h += gridDim.x) {
        const uint2 range = ranges[hard[4 + (size_t)in * T + h]];
        const int n = (int)(range.y - range.x);
        const size_t lo = range.x;
        const int quota = 64 * SI;
        for (int pass = 0; pass < 4; pass++) {
            const int shift = pass * 8;
            const uint32_t* ksrc = (pass & 1) ? keys1 : keys0;
            const uint32_t* vsrc = (pass & 1) ? vals1 : vals0;
            uint32_t* kdst = (pass & 1) ? keys0 : keys1;
            uint32_t* vdst = (pass & 1) ? vals0 : vals1;
            __syncthreads();  // the previous pass's stores are visible to the whole workgroup
            // digit histogram of the whole segment -> running digit bases
            s_cnt[0][tid] = 0;
            __syncthreads();
            // the first pass has no keys yet: it builds them from the depth of each listed Gaussian
            for (int i = tid; i < n; i += ST) {
                // incoming list words may carry rank bits above the Gaussian index; the first pass strips them
                const uint32_t k = pass == 0 ? __float_as_uint(depth[vsrc[lo + i] & idx_mask]) : ksrc[lo + i];
                atomicAdd(&s_cnt[0][(k >> shift) & 0xffu], 1u);
            }
            __syncthreads();
            {
                const uint32_t c = s_cnt[0][tid];
                const uint32_t incl = wave_incl_scan_u32(c, lane);
                if (lane == 63) s_wtot[wave] = incl;
                __syncthreads();
                uint32_t b = 0;
                for (int w = 0; w < wave; w++) b += s_wtot[w];
                s_base[tid] = b + incl - c;
            }
            __syncthreads();
            for (int c0 = 0; c0 < n; c0 += STILE) {
                const int m = min(STILE, n - c0);
                uint32_t key[SI], pos[SI], val[SI];
#pragma unroll
                for (int i = 0; i < SI; i++) {
                    const int slot = wave * quota + i * 64 + lane;
                    const bool valid = slot < m;
                    val[i] = valid ? vsrc[lo + c0 + slot] & (pass == 0 ? idx_mask : 0xffffffffu) : 0u;
                    key[i] = !valid ? ~0u : pass == 0 ? __float_as_uint(depth[val[i]]) : ksrc[lo + c0 + slot];
                }
                uint32_t tot;
                block_rank_pass<SI>(key, shift, m, quota, s_cnt, s_start, s_wtot, pos, tot);
                // pos = rank inside the chunk ordered by digit; convert to the segment-wide position
#pragma unroll
                for (int i = 0; i < SI; i++) {
                    const int slot = wave * quota + i * 64 + lane;
                    if (slot < m) {
                        const uint32_t d = (key[i] >> shift) & 0xffu;
                        const uint32_t g = s_base[d] + (pos[i] - s_start[d]);
                        kdst[lo + g] = key[i];
                        vdst[lo + g] = val[i];
                    }
                }
                __syncthreads();
                s_base[tid] += tot;
                __syncthreads();
            }
        }
        __syncthreads();
    }
}

static int tile_passes(int bits) { return (bits + 7) / 8; }

int ngs_sort_input_buffer(int bits) { return tile_passes(bits) & 1; }

template <int ITEMS>
static void launch_sort_passes(hipStream_t s, const uint32_t* count, int64_t capacity, uint32_t* const keys[2],
                               uint32_t* const vals[2], uint32_t* hist, uint32_t* totals, int bits, int first_bit)
{
    const int nblocks = (int)((capacity + ST * ITEMS - 1) / (ST * ITEMS));
    int cur = ngs_sort_input_buffer(bits);
    // the bits are spread evenly over the passes (13 bits: 7 + 6, not 8 + 5): fewer buckets in a pass mean longer
    // runs per bucket in the scatter, i.e. fewer partially written memory lines
    int passes_left = tile_passes(bits);
    for (int shift = 0, width; shift < bits; shift += width, passes_left--) {
        width = (bits - shift + passes_left - 1) / passes_left;
        const uint32_t mask = (1u << width) - 1u;
        hipLaunchKernelGGL(sort_hist_kernel<ITEMS>, dim3(nblocks), dim3(ST), 0, s, count, capacity, keys[cur],
                           first_bit + shift, mask, nblocks, hist);
        hipLaunchKernelGGL(sort_scan_kernel, dim3(256), dim3(SCAN_T), 0, s, nblocks, hist, totals);
        // the ranking inside a wave costs one ballot per digit bit: the common widths get a kernel of their own
        auto scatter = width <= 6   ? sort_scatter_kernel<6, ITEMS>
                       : width == 7 ? sort_scatter_kernel<7, ITEMS>
                                    : sort_scatter_kernel<8, ITEMS>;
        hipLaunchKernelGGL(scatter, dim3(nblocks), dim3(ST), 0, s, count, capacity, keys[cur], vals[cur],
                           keys[cur ^ 1], vals[cur ^ 1], first_bit + shift, mask, nblocks, hist, totals);
        cur ^= 1;
    }
}

// hist: u32[256 * ngs_sort_blocks(capacity)]
void ngs_launch_sort_tile_bits(hipStream_t s, const uint32_t* count, int64_t capacity, uint32_t* const keys[2],
                               uint32_t* const vals[2], uint32_t* hist, uint32_t* totals, int bits, int first_bit)
{
    if (capacity <= 0) return;
    if (ngs_sort_items((size_t)capacity) == 4)
        launch_sort_passes<4>(s, count, capacity, keys, vals, hist, totals, bits, first_bit);
    else
        launch_sort_passes<NGS_SORT_ITEMS>(s, count, capacity, keys, vals, hist, totals, bits, first_bit);
}

// The depth order obtained AFTER the tile-level sort instead of before it: one workgroup per tile fetches the depth
// of each of its Gaussians (a gather from the per-Gaussian depth array) and sorts its list by the 32 depth bits,
// stably, in LDS (segment_sort_bucket_kernel).  What that kernel cannot take (more than 4096 entries, or depths
// crowding into one bucket) goes to the general kernel, a small grid that finds its tiles in a work list and costs a
// few microseconds when the list is empty.  The cost of this route grows with the number of INSTANCES (and steeply
// once lists outgrow a workgroup), the Gaussian-level sort's with the number of Gaussians: the forward pass picks
// whichever is cheaper for the call.
void ngs_launch_sort_segments(hipStream_t s, int T, const uint2* ranges, const float* depth, uint32_t* const keys[2],
                              uint32_t* const vals[2], uint32_t* hard, const DepthRank& dr, const uint32_t* misc,
                              uint2* subs, uint32_t subs_cap)
{
    // work list 0: what the first kernel hands on; work list 1: what the split hands on
    if (dr.key_bits > 0)
        hipLaunchKernelGGL((segment_sort_bucket_kernel<ST, true>), dim3(T), dim3(ST), 0, s, T, ranges, depth, keys[0],
                           dr, misc, vals[0], hard, -1, 0);
    else
        hipLaunchKernelGGL((segment_sort_bucket_kernel<ST, false>), dim3(T), dim3(ST), 0, s, T, ranges, depth, keys[0],
                           dr, misc, vals[0], hard, -1, 0);
    const int split_grid = T < 2048 ? T : 2048;
    if (dr.key_bits > 0)
        hipLaunchKernelGGL(segment_split_kernel<true>, dim3(split_grid), dim3(ST), 0, s, T, ranges, depth, keys[0],
                           vals[0], keys[1], vals[1], dr, misc, hard, 0, 1, subs, subs_cap);
    else
        hipLaunchKernelGGL(segment_split_kernel<false>, dim3(split_grid), dim3(ST), 0, s, T, ranges, depth, keys[0],
                           vals[0], keys[1], vals[1], dr, misc, hard, 0, 1, subs, subs_cap);
    hipLaunchKernelGGL(segment_sort_sub_kernel, dim3(2048), dim3(ST), 0, s, subs, subs_cap, hard, keys[1], vals[1],
                       vals[0]);
    hipLaunchKernelGGL(segment_sort_long_kernel, dim3(T < 512 ? T : 512), dim3(ST), 0, s, T, ranges, depth, keys[0],
                       vals[0], keys[1], vals[1], hard, 1, dr.key_bits > 0 ? dr.idx_mask() : 0xffffffffu);
}
