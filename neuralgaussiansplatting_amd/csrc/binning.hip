// Instance binning: offsets of each Gaussian's tile instances, key emission
// and per-tile ranges of the sorted list.  Integer work, HBM bound.  Results
// are those of the reference's InclusiveSum + duplicateWithKeys + SortPairs +
// identifyTileRanges (cuda_rasterizer/rasterizer_impl.cu:70-138,277-318); the
// schedule is not.  The reference emits instances in index order with 64-bit
// (tile | depth) keys and sorts all of them on all key bits.  Here the
// GAUSSIANS are sorted by depth first (a sort of P items instead of the depth
// half of a sort of R >= P items), instances are emitted in that order with the
// tile id as their only key, and the stable tile-level sort then leaves every
// tile's list in depth order, ties in index order -- the order a stable sort on
// (tile, depth) of index-ordered instances gives.  The prefix sums are
// two-level (block sums, one workgroup scanning them, the emission kernel
// rescanning inside its block), and keys are emitted by whole waves into
// contiguous output runs instead of by one serial loop per Gaussian.
#include "ngs_common.h"

#define SCAN_THREADS 1024
#define SCAN_REGS 8

// Exclusive scan of the per-block tile counts by a single workgroup.  Thread t
// owns a contiguous run of entries (a multiple of four).
__global__ __launch_bounds__(SCAN_THREADS) void scan_block_sums_kernel(int n, const uint32_t* __restrict__ in,
                                                                        uint32_t* __restrict__ out,
                                                                        const uint32_t* __restrict__ bext,
                                                                        uint32_t* __restrict__ misc,
                                                                        unsigned long long* mailbox)
{
    __shared__ uint32_t wave_tot[SCAN_THREADS / 64];
    if (blockIdx.x == 1) {
        // A second workgroup, beside the scan rather than inside it (the scan's registers are spoken for): the frame's
        // depth bounds from the per-block ones, which hold the high halves of the depth bits, the smallest rounded
        // down in the low 16 bits of a word, the largest rounded up in the high 16.
        __shared__ uint32_t wave_lo[SCAN_THREADS / 64], wave_hi[SCAN_THREADS / 64];
        uint32_t dlo = 0xffffu, dhi = 0u;
        for (int base = 0; base < n; base += 4 * SCAN_THREADS) {
            uint32_t e[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {  // four loads in flight per trip; 16 K blocks (4 M Gaussians) take one trip
                const int i = base + c * SCAN_THREADS + (int)threadIdx.x;
                e[c] = i < n ? bext[i] : 0xffffu;
            }
#pragma unroll
            for (int c = 0; c < 4; c++) {
                dlo = min(dlo, e[c] & 0xffffu);
                dhi = max(dhi, e[c] >> 16);
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            dlo = min(dlo, (uint32_t)__shfl_xor((int)dlo, o, 64));
            dhi = max(dhi, (uint32_t)__shfl_xor((int)dhi, o, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            wave_lo[threadIdx.x >> 6] = dlo;
            wave_hi[threadIdx.x >> 6] = dhi;
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            dlo = threadIdx.x < SCAN_THREADS / 64 ? wave_lo[threadIdx.x] : 0xffffu;
            dhi = threadIdx.x < SCAN_THREADS / 64 ? wave_hi[threadIdx.x] : 0u;
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) {
                dlo = min(dlo, (uint32_t)__shfl_xor((int)dlo, o, 64));
                dhi = max(dhi, (uint32_t)__shfl_xor((int)dhi, o, 64));
            }
            if (threadIdx.x == 0) {  // bounds of the visible depths' bits (not necessarily attained)
                misc[NGS_MISC_DMIN] = dlo << 16;
                misc[NGS_MISC_DMAX] = dhi << 16;
            }
        }
        return;
    }
    // A thread's run is a whole number of 16-byte groups; up to SCAN_REGS groups (8.4 M Gaussians) are fetched with
    // independent vector loads and stay in registers for the second sweep -- one memory round trip instead of one per
    // entry.  Longer runs fall back to two scalar sweeps.
    const int groups = (n + 4 * SCAN_THREADS - 1) / (4 * SCAN_THREADS);
    const int per = 4 * groups;
    const int lo = threadIdx.x * per;
    const int hi = min(n, lo + per);
    const bool in_regs = groups <= SCAN_REGS;
    __shared__ unsigned long long wave_wide[SCAN_THREADS / 64];
    uint32_t s = 0, marks = 0;
    unsigned long long wide = 0;  // the same sum without wrap-around, to notice a count that does not fit
    uint4 held[SCAN_REGS];
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < SCAN_REGS; j++) {
            held[j] = make_uint4(0, 0, 0, 0);
            const int at = lo + 4 * j;
            if (j < groups && at < n) {
                if (at + 4 <= n) {
                    held[j] = *(const uint4*)(in + at);
                } else {
                    held[j].x = in[at];
                    if (at + 1 < n) held[j].y = in[at + 1];
                    if (at + 2 < n) held[j].z = in[at + 2];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < SCAN_REGS; j++) {
            // the top bit of an entry is not part of the count: preprocess_fwd's "culled although prefiltered" mark
            marks |= held[j].x | held[j].y | held[j].z | held[j].w;
            held[j].x &= 0x7fffffffu; held[j].y &= 0x7fffffffu; held[j].z &= 0x7fffffffu; held[j].w &= 0x7fffffffu;
            s += held[j].x + held[j].y + held[j].z + held[j].w;
            wide += (unsigned long long)held[j].x + held[j].y + held[j].z + held[j].w;
        }
    } else {
        for (int i = lo; i < hi; i++) {
            marks |= in[i];
            s += in[i] & 0x7fffffffu;
            wide += in[i] & 0x7fffffffu;
        }
    }
    const bool marked = __ballot((marks & 0x80000000u) != 0u) != 0ull;  // wave-uniform
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan_u32(s, lane);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) wide += __shfl_xor(wide, o);
    __shared__ uint32_t wave_marked[SCAN_THREADS / 64];
    if (lane == 63) {
        wave_tot[wave] = incl;
        wave_wide[wave] = wide;
        wave_marked[wave] = marked ? 1u : 0u;
    }
    __syncthreads();
    uint32_t base = 0, total = 0, flags = 0;
    unsigned long long total_wide = 0;
    for (int w = 0; w < SCAN_THREADS / 64; w++) {
        const uint32_t t = wave_tot[w];
        if (w < wave) base += t;
        total += t;
        total_wide += wave_wide[w];
        flags |= wave_marked[w];  // error flag 1: a point was culled although the caller said they were prefiltered
    }
    uint32_t run = base + incl - s;
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < SCAN_REGS; j++) {
            const int at = lo + 4 * j;
            uint4 o;
            o.x = run; run += held[j].x;
            o.y = run; run += held[j].y;
            o.z = run; run += held[j].z;
            o.w = run; run += held[j].w;
            if (j < groups && at < n) {
                if (at + 4 <= n) {
                    *(uint4*)(out + at) = o;
                } else {
                    out[at] = o.x;
                    if (at + 1 < n) out[at + 1] = o.y;
                    if (at + 2 < n) out[at + 2] = o.z;
                }
            }
        }
    } else {
        for (int i = lo; i < hi; i++) {
            const uint32_t v = in[i] & 0x7fffffffu;
            out[i] = run;
            run += v;
        }
    }
    if (threadIdx.x == 0 && misc) {  // misc == NULL: a scan of offsets only (the count has been posted before)
        if (total_wide > 0x7fffffffull) {  // the count is handed on as a signed 32-bit number
            flags |= 2u;
            total = 0xffffffffu;
        }
        misc[0] = total;
        misc[1] = flags;  // written, not merged: nobody has to clear the word before a frame
        misc[NGS_MISC_T_SCAN] = (uint32_t)wall_clock64();  // the binning stages start here (see render_fwd_kernel)
        // The host is spinning on this word of pinned host memory (it holds ~0 until now): the instance count and
        // the error flag reach it without a copy engine or an interrupt in between.
        if (mailbox)
            __hip_atomic_store(mailbox, ((unsigned long long)flags << 32) | total, __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Gather of the per-Gaussian rectangles into depth order, the tile counts in that order (a rectangle's area) and their
// per-block sums (first level of the offset scan).  One lane per position of the order.  Whether a position holds
// a Gaussian that emits instances is read off its sorted key (culled Gaussians sort to the end under a key of all
// ones), so the one dependent access per position is the rectangle's.
__global__ __launch_bounds__(NGS_GAUSS_BLOCK) void depth_order_kernel(int P, const uint32_t* __restrict__ order,
                                                                      const uint32_t* __restrict__ sorted_keys,
                                                                      const uint2* __restrict__ rect,
                                                                      uint32_t* __restrict__ s_tiles,
                                                                      uint2* __restrict__ s_rect,
                                                                      uint32_t* __restrict__ s_block_sums)
{
    __shared__ uint32_t wave_tot[NGS_GAUSS_BLOCK / 64];
    const int k = blockIdx.x * NGS_GAUSS_BLOCK + threadIdx.x;
    uint32_t t = 0;
    if (k < P) {
        if (sorted_keys[k] != 0xffffffffu) {
            const uint2 r = rect[order[k]];
            t = ((r.x >> 16) - (r.x & 0xffffu)) * ((r.y >> 16) - (r.y & 0xffffu));
            s_rect[k] = r;
        }
        s_tiles[k] = t;
    }
    const uint32_t ws = wave_sum_u32(t);
    if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = ws;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t sum = 0;
#pragma unroll
        for (int w = 0; w < NGS_GAUSS_BLOCK / 64; w++) sum += wave_tot[w];
        s_block_sums[blockIdx.x] = sum;
    }
}

// One workgroup per 256 Gaussians (the same blocks as the preprocess kernel).
// Each wave owns 64 consecutive Gaussians, whose instances occupy one
// contiguous run of the output; the wave walks that run 64 entries at a time,
// each lane locating its Gaussian by binary search over the wave's prefix sums,
// so key/value stores are fully coalesced whatever the footprint sizes are.
// Instances beyond `capacity` (a binning buffer sized from an estimate that turned out too small) are dropped and
// flagged in misc[1]; the host then runs the stages again with a buffer that fits.
// `order` == NULL: lane k emits Gaussian k; else Gaussian order[k] (rect / tiles / block_offs are then the arrays
// in that order).
__global__ __launch_bounds__(NGS_GAUSS_BLOCK) void duplicate_kernel(int P, const uint2* __restrict__ rect,
                                                                    const uint32_t* __restrict__ tiles,
                                                                    const uint32_t* __restrict__ block_offs,
                                                                    const uint32_t* __restrict__ order,
                                                                    const float* __restrict__ depth,
                                                                    const DepthRank dr, int grid_x,
                                                                    uint32_t* __restrict__ keys,
                                                                    uint32_t* __restrict__ vals, uint32_t capacity,
                                                                    uint32_t* __restrict__ misc,
                                                                    uint4* __restrict__ clear, int clear_n)
{
    // The tile ranges (and the counters behind them) have to be zero before the range kernel runs, two launches on:
    // the first workgroups of this one see to it, which saves a fill launch.
    for (int k = blockIdx.x * NGS_GAUSS_BLOCK + threadIdx.x; k < clear_n; k += gridDim.x * NGS_GAUSS_BLOCK)
        clear[k] = make_uint4(0u, 0u, 0u, 0u);
    __shared__ uint32_t s_incl[NGS_GAUSS_BLOCK];  // per wave: inclusive prefix of tiles within the wave
    __shared__ uint2 s_rect[NGS_GAUSS_BLOCK];
    __shared__ uint2 s_id[NGS_GAUSS_BLOCK];  // x: list word (Gaussian index | low part of the depth rank), y: the
                                             // high part of the rank, which goes into every key of the Gaussian
    __shared__ uint32_t s_excl[NGS_GAUSS_BLOCK];  // and the exclusive one
    __shared__ uint32_t s_wave_tot[NGS_GAUSS_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int idx = blockIdx.x * NGS_GAUSS_BLOCK + tid;
    // Everything a Gaussian brings is requested at once (a workgroup emits about a thousand instances: a chain of
    // loads that wait for one another at its start would be most of its life); what a culled Gaussian's rectangle and
    // depth hold is never used.
    uint32_t cnt = 0, id = (uint32_t)idx, zbits = 0;
    uint2 rc = make_uint2(0u, 0u);
    const bool ranked = depth && dr.key_bits;
    const uint32_t block_base = block_offs[blockIdx.x];
    const uint32_t dmin = ranked ? misc[NGS_MISC_DMIN] : 0u, dmax = ranked ? misc[NGS_MISC_DMAX] : 0u;
    if (idx < P) {
        cnt = tiles[idx];
        rc = rect[idx];
        if (order) id = order[idx];
        if (ranked) zbits = __float_as_uint(depth[idx]);
    }
    if (cnt) {
        uint32_t rank = 0;
        if (ranked) {
            const uint32_t r = dr.rank(zbits, dmin, dr.shift(dmin, dmax));
            rank = dr.key_part(r);
            id |= dr.word_part(r);
        }
        s_rect[tid] = rc;
        s_id[tid] = make_uint2(id, rank);
    }
    const uint32_t incl = wave_incl_scan_u32(cnt, lane);
    s_incl[tid] = incl;
    s_excl[tid] = incl - cnt;
    if (lane == 63) s_wave_tot[wave] = incl;
    __syncthreads();
    uint32_t wave_base = block_base;
    for (int w = 0; w < wave; w++) wave_base += s_wave_tot[w];
    uint32_t wave_total = s_wave_tot[wave];
    if (wave_base + wave_total > capacity || wave_base + wave_total < wave_base) {  // wave-uniform
        if (lane == 0 && wave_total != 0) atomicOr(&misc[1], 4u);
        wave_total = wave_base < capacity ? capacity - wave_base : 0u;
    }
    const uint32_t* incl_w = s_incl + wave * 64;
    // The loop below is the whole kernel (one trip per 64 instances), and it is bound by the number of vector
    // instructions it issues: nothing in it branches on a lane's values, and the store addresses advance as pointers.
    uint32_t* kp = keys + (size_t)wave_base + lane;
    uint32_t* vp = vals + (size_t)wave_base + lane;
    for (uint32_t k = lane; k < wave_total; k += 64, kp += 64, vp += 64) {
        // smallest j with incl_w[j] > k: six halving steps, always
        int j = 0;
#pragma unroll
        for (int step = 32; step > 0; step >>= 1) j += incl_w[j + step - 1] <= k ? step : 0;
        const uint32_t local = k - s_excl[wave * 64 + j];
        const uint2 r = s_rect[wave * 64 + j];
        const uint32_t x0 = r.x & 0xffffu, x1 = r.x >> 16, y0 = r.y & 0xffffu;
        const uint32_t w = x1 - x0;
        // row and column of the local-th tile of the rectangle: local / w and local % w.  An integer division by a
        // variable costs some 25 vector instructions; below 2^22 a float reciprocal is off by at most one either way.
        uint32_t qy = (uint32_t)((float)local * __builtin_amdgcn_rcpf((float)w));
        uint32_t qx = local - qy * w;
        const bool under = (int32_t)qx < 0;
        qy -= under ? 1u : 0u;
        qx += under ? w : 0u;
        const bool over = qx >= w;
        qy += over ? 1u : 0u;
        qx -= over ? w : 0u;
        if (__builtin_amdgcn_ballot_w64(local >= (1u << 22)) != 0ull) {  // wave-uniform; a rectangle of four million tiles
            qy = local / w;
            qx = local % w;
        }
        const uint32_t ty = y0 + qy, tx = x0 + qx;
        // the tile id above the high part of the depth rank (the low part rides in the list word)
        const uint2 ir = s_id[wave * 64 + j];
        *kp = ((ty * (uint32_t)grid_x + tx) << dr.key_bits) | ir.y;
        *vp = ir.x;
    }
}

// rasterizer_impl.cu:116-138; `ranges` was zeroed by the caller.  L = min(*count, capacity).
__global__ __launch_bounds__(256) void tile_ranges_kernel(const uint32_t* __restrict__ count, int64_t capacity,
                                                          const uint32_t* __restrict__ keys, int qbits,
                                                          uint2* __restrict__ ranges)
{
    const int64_t L = min((int64_t)*count, capacity);
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= L) return;
    const uint32_t cur = keys[idx] >> qbits;
    if (idx == 0) {
        ranges[cur].x = 0;
    } else {
        const uint32_t prev = keys[idx - 1] >> qbits;
        if (cur != prev) {
            ranges[prev].y = (uint32_t)idx;
            ranges[cur].x = (uint32_t)idx;
        }
    }
    if (idx == L - 1) ranges[cur].y = (uint32_t)L;
}

// The same with four keys per thread (one 16-byte load plus the key before them).
__global__ __launch_bounds__(256) void tile_ranges_vec_kernel(const uint32_t* __restrict__ count, int64_t capacity,
                                                              const uint32_t* __restrict__ keys, int qbits,
                                                              uint2* __restrict__ ranges)
{
    const int64_t L = min((int64_t)*count, capacity);
    const int64_t first = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (first >= L) return;
    uint32_t k[4];
    uint32_t prev = first > 0 ? keys[first - 1] : 0u;  // both loads in flight before either is used
    if (first + 4 <= L) {
        const uint4 v = *(const uint4*)(keys + first);
        k[0] = v.x; k[1] = v.y; k[2] = v.z; k[3] = v.w;
    } else {
        for (int j = 0; j < 4; j++) k[j] = first + j < L ? keys[first + j] : 0u;
    }
    prev >>= qbits;
#pragma unroll
    for (int j = 0; j < 4; j++) k[j] >>= qbits;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int64_t idx = first + j;
        if (idx < L) {
            const uint32_t cur = k[j];
            if (idx == 0) {
                ranges[cur].x = 0;
            } else if (cur != prev) {
                ranges[prev].y = (uint32_t)idx;
                ranges[cur].x = (uint32_t)idx;
            }
            if (idx == L - 1) ranges[cur].y = (uint32_t)L;
            prev = cur;
        }
    }
}

void ngs_launch_scan_block_sums(hipStream_t s, int nblocks, const uint32_t* block_sums, uint32_t* block_offs,
                                const uint32_t* block_depths, uint32_t* misc, unsigned long long* mailbox)
{
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(block_depths ? 2 : 1), dim3(SCAN_THREADS), 0, s, nblocks,
                       block_sums, block_offs, block_depths, misc, mailbox);
}

void ngs_launch_depth_order(hipStream_t s, int P, const uint32_t* order, const uint32_t* sorted_keys, const uint2* rect,
                            uint32_t* s_tiles, uint2* s_rect, uint32_t* s_block_sums, uint32_t* s_block_offs)
{
    const int nb = (P + NGS_GAUSS_BLOCK - 1) / NGS_GAUSS_BLOCK;
    hipLaunchKernelGGL(depth_order_kernel, dim3(nb), dim3(NGS_GAUSS_BLOCK), 0, s, P, order, sorted_keys, rect, s_tiles,
                       s_rect, s_block_sums);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, nb, s_block_sums, s_block_offs,
                       (const uint32_t*)nullptr, (uint32_t*)nullptr, (unsigned long long*)nullptr);
}

void ngs_launch_duplicate(hipStream_t s, int P, const uint2* rect, const uint32_t* tiles, const uint32_t* block_offs,
                          const uint32_t* order, const float* depth, const DepthRank& dr, int grid_x, uint32_t* keys,
                          uint32_t* vals, int64_t capacity, uint32_t* misc, void* clear, size_t clear_bytes)
{
    const int nb = (P + NGS_GAUSS_BLOCK - 1) / NGS_GAUSS_BLOCK;
    hipLaunchKernelGGL(duplicate_kernel, dim3(nb), dim3(NGS_GAUSS_BLOCK), 0, s, P, rect, tiles, block_offs, order,
                       depth, dr, grid_x, keys, vals, (uint32_t)capacity, misc, (uint4*)clear, (int)(clear_bytes / 16));
}

void ngs_launch_tile_ranges(hipStream_t s, const uint32_t* count, int64_t capacity, const uint32_t* keys, int qbits,
                            int T, uint2* ranges, size_t clear_bytes)
{
    if (clear_bytes) (void)hipMemsetAsync(ranges, 0, clear_bytes, s);
    if (capacity > 0 && (uintptr_t)keys % 16 == 0)
        hipLaunchKernelGGL(tile_ranges_vec_kernel, dim3((unsigned)((capacity + 1023) / 1024)), dim3(256), 0, s, count,
                           capacity, keys, qbits, ranges);
    else if (capacity > 0)
        hipLaunchKernelGGL(tile_ranges_kernel, dim3((unsigned)((capacity + 255) / 256)), dim3(256), 0, s, count,
                           capacity, keys, qbits, ranges);
}
