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
__device__ __forceinline__ uint32_t wave_rank(uint32_t d, bool valid, uint32_t* cnt, uint64_t lanes_below)
{
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
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

__global__ __launch_bounds__(ST) void sort_hist_kernel(const uint32_t* __restrict__ count, int64_t capacity,
                                                       const uint32_t* __restrict__ keys, int shift,
                                                       uint32_t mask, int nblocks, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t s_cnt[SWAVES][256];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int64_t n = pair_count(count, capacity);
    if ((int64_t)blockIdx.x * STILE >= n) {  // a block past the end (launches are sized by the capacity)
        hist[(size_t)tid * nblocks + blockIdx.x] = 0;
        return;
    }
    for (int w = 0; w < SWAVES; w++) s_cnt[w][tid] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * STILE;
#pragma unroll
    for (int i = 0; i < SI; i++) {
        const int64_t p = base + i * ST + tid;
        if (p < n) atomicAdd(&s_cnt[wave][digit_of(keys[p], shift, mask)], 1u);
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
    __shared__ uint32_t s_key[STILE];
    __shared__ uint32_t s_val[STILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n = pair_count(count, capacity);
    const int64_t base = (int64_t)blockIdx.x * STILE;
    if (base >= n) return;  // block-uniform
    const int n_valid = (int)min((int64_t)STILE, n - base);
    for (int w = 0; w < SWAVES; w++) s_cnt[w][tid] = 0;
    __syncthreads();

    // Wave-striped ownership keeps input order: wave w owns slots [w*64*SI, (w+1)*64*SI), item i of lane l is
    // slot w*64*SI + i*64 + l.
    uint32_t key[SI], val[SI], rank[SI];
    const uint64_t lanes_below = (1ull << lane) - 1ull;
#pragma unroll
    for (int i = 0; i < SI; i++) {
        const int slot = wave * (64 * SI) + i * 64 + lane;
        const bool valid = slot < n_valid;
        key[i] = valid ? kin[base + slot] : ~0u;
        val[i] = valid ? vin[base + slot] : 0u;
    }
#pragma unroll
    for (int i = 0; i < SI; i++) {
        const int slot = wave * (64 * SI) + i * 64 + lane;
        rank[i] = wave_rank(digit_of(key[i], shift, mask), slot < n_valid, s_cnt[wave], lanes_below);
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
    for (int i = 0; i < SI; i++) {
        const int slot = wave * (64 * SI) + i * 64 + lane;
        if (slot < n_valid) {
            const uint32_t d = digit_of(key[i], shift, mask);
            const uint32_t pos = s_start[d] + s_cnt[wave][d] + rank[i];
            s_key[pos] = key[i];
            s_val[pos] = val[i];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SI; i++) {
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

// Segments with min_n < n <= THREADS*ITEMS: the whole segment lives in registers, no global traffic between passes.
// Longer segments get more threads rather than more items per thread: the passes are a chain of barriers, and a
// workgroup's latency is what a launch with few such tiles waits for.
template <int ITEMS, int THREADS>
__global__ __launch_bounds__(THREADS) void segment_sort_small_kernel(const uint2* __restrict__ ranges, int min_n,
                                                                     const float* __restrict__ depth,
                                                                     uint32_t* __restrict__ vals0)
{
    constexpr int CAP = THREADS * ITEMS, WAVES = THREADS / 64;
    __shared__ uint32_t s_cnt[WAVES][256];
    __shared__ uint32_t s_start[256];
    __shared__ uint32_t s_wtot[4];
    __shared__ uint32_t s_red[2 * WAVES];
    __shared__ uint32_t s_key[CAP];
    __shared__ uint32_t s_val[CAP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint2 range = ranges[blockIdx.x];
    const int n = (int)(range.y - range.x);
    // block-uniform: another launch (or nobody: an empty tile, or a single entry, which is sorted) owns this tile
    if (n <= min_n || n > CAP || n < 2) return;
    const size_t lo = range.x;
    const int quota = ((((n + WAVES - 1) / WAVES) + 63) / 64) * 64;  // slots per wave
    uint32_t key[ITEMS], val[ITEMS], pos[ITEMS];
    uint32_t k_or = 0u, k_and = ~0u;
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const int slot = wave * quota + i * 64 + lane;
        const bool valid = i * 64 < quota && slot < n;
        val[i] = valid ? vals0[lo + slot] : 0u;
    }
    // all list entries are on their way before the first dependent depth gather is issued
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const int slot = wave * quota + i * 64 + lane;
        const bool valid = i * 64 < quota && slot < n;
        key[i] = valid ? __float_as_uint(depth[val[i]]) : ~0u;
        if (valid) { k_or |= key[i]; k_and &= key[i]; }
    }
    // bits that differ somewhere in the segment
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        k_or |= (uint32_t)__shfl_xor((int)k_or, o, 64);
        k_and &= (uint32_t)__shfl_xor((int)k_and, o, 64);
    }
    if (lane == 0) { s_red[wave] = k_or; s_red[WAVES + wave] = k_and; }
    __syncthreads();
    uint32_t varying;
    {
        uint32_t o = 0u, a = ~0u;
        for (int w = 0; w < WAVES; w++) { o |= s_red[w]; a &= s_red[WAVES + w]; }
        varying = o ^ a;
    }
    for (int shift = 0; shift < 32; shift += 8) {
        if (((varying >> shift) & 0xffu) == 0u) continue;  // digit constant over the segment (block-uniform)
        uint32_t tot;
        block_rank_pass<ITEMS, WAVES>(key, shift, n, quota, s_cnt, s_start, s_wtot, pos, tot);
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const int slot = wave * quota + i * 64 + lane;
            if (i * 64 < quota && slot < n) { s_key[pos[i]] = key[i]; s_val[pos[i]] = val[i]; }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const int slot = wave * quota + i * 64 + lane;
            if (i * 64 < quota && slot < n) { key[i] = s_key[slot]; val[i] = s_val[slot]; }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const int slot = wave * quota + i * 64 + lane;
        if (i * 64 < quota && slot < n) vals0[lo + slot] = val[i];
    }
}

// Segments longer than STILE pairs: the same four passes, chunk by chunk, ping-ponging through the spare buffers
// (depth keys in keys0 / keys1, the list in vals0 / vals1; four passes end where they started, in vals0).
__global__ __launch_bounds__(ST) void segment_sort_long_kernel(const uint2* __restrict__ ranges,
                                                               const float* __restrict__ depth, uint32_t* keys0,
                                                               uint32_t* vals0, uint32_t* keys1, uint32_t* vals1)
{
    __shared__ uint32_t s_cnt[SWAVES][256];
    __shared__ uint32_t s_start[256];
    __shared__ uint32_t s_base[256];
    __shared__ uint32_t s_wtot[SWAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint2 range = ranges[blockIdx.x];
    const int n = (int)(range.y - range.x);
    if (n <= STILE) return;  // block-uniform
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
            const uint32_t k = pass == 0 ? __float_as_uint(depth[vsrc[lo + i]]) : ksrc[lo + i];
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
                val[i] = valid ? vsrc[lo + c0 + slot] : 0u;
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
}

static int tile_passes(int bits) { return (bits + 7) / 8; }

int ngs_sort_input_buffer(int bits) { return tile_passes(bits) & 1; }

void ngs_launch_sort_tile_bits(hipStream_t s, const uint32_t* count, int64_t capacity, uint32_t* const keys[2],
                               uint32_t* const vals[2], uint32_t* hist, uint32_t* totals, int bits)
{
    if (capacity <= 0) return;
    const int nblocks = (int)((capacity + STILE - 1) / STILE);
    int cur = ngs_sort_input_buffer(bits);
    // the bits are spread evenly over the passes (13 bits: 7 + 6, not 8 + 5): fewer buckets in a pass mean longer
    // runs per bucket in the scatter, i.e. fewer partially written memory lines
    int passes_left = tile_passes(bits);
    for (int shift = 0, width; shift < bits; shift += width, passes_left--) {
        width = (bits - shift + passes_left - 1) / passes_left;
        const uint32_t mask = (1u << width) - 1u;
        hipLaunchKernelGGL(sort_hist_kernel, dim3(nblocks), dim3(ST), 0, s, count, capacity, keys[cur], shift, mask,
                           nblocks, hist);
        hipLaunchKernelGGL(sort_scan_kernel, dim3(256), dim3(SCAN_T), 0, s, nblocks, hist, totals);
        hipLaunchKernelGGL(sort_scatter_kernel, dim3(nblocks), dim3(ST), 0, s, count, capacity, keys[cur], vals[cur],
                           keys[cur ^ 1], vals[cur ^ 1], shift, mask, nblocks, hist, totals);
        cur ^= 1;
    }
}

// The depth order obtained AFTER the tile-level sort instead of before it: one workgroup per tile fetches the depth
// of each of its Gaussians (a gather from the per-Gaussian depth array), sorts its segment by the 32 depth bits,
// stably, entirely in registers + LDS (four 8-bit passes, passes whose digit is constant over the segment are
// skipped) and writes the list back in place; segments longer than 4096 pairs run the same passes ping-ponging
// through global memory.  Its cost grows with the number of INSTANCES (and steeply once lists outgrow a workgroup),
// the Gaussian-level sort's with the number of Gaussians: the forward pass picks whichever is cheaper for the call.
void ngs_launch_sort_segments(hipStream_t s, int T, const uint2* ranges, const float* depth, uint32_t* const keys[2],
                              uint32_t* const vals[2])
{
    // every tile with two or more entries is claimed by exactly one of the three launches (by segment length); the
    // others exit at once
    hipLaunchKernelGGL((segment_sort_small_kernel<8, ST>), dim3(T), dim3(ST), 0, s, ranges, 0, depth, vals[0]);
    hipLaunchKernelGGL((segment_sort_small_kernel<8, 2 * ST>), dim3(T), dim3(2 * ST), 0, s, ranges, ST * 8, depth,
                       vals[0]);
    hipLaunchKernelGGL(segment_sort_long_kernel, dim3(T), dim3(ST), 0, s, ranges, depth, keys[0], vals[0], keys[1],
                       vals[1]);
}
