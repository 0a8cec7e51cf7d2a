// Stable least-significant-digit radix sort of (uint64 key, uint32 value)
// pairs on key bits [0,end_bit), written for wave64: 8-bit digits, 2048 pairs
// per workgroup, ballot-based ranking.  It produces the result the reference
// obtains from cub::DeviceRadixSort::SortPairs
// (cuda_rasterizer/rasterizer_impl.cu:303-308): keys ascending on the masked
// bits, ties in input order.
//
// Per pass: (1) per-workgroup digit histogram, stored digit-major; (2) one
// workgroup per digit turns its row of the histogram into exclusive prefixes
// and records the digit total; (3) scatter: ranks pairs inside the workgroup in
// input order, reorders them through LDS so that each digit's run leaves as
// contiguous stores, and writes them at digit base + workgroup prefix + rank.
#include "ngs_common.h"

#define ST NGS_SORT_THREADS
#define SI NGS_SORT_ITEMS
#define STILE NGS_SORT_TILE
#define SWAVES (ST / 64)

__device__ __forceinline__ uint32_t digit_of(uint64_t key, int shift, uint32_t mask)
{
    return (uint32_t)(key >> shift) & mask;
}

__global__ __launch_bounds__(ST) void sort_hist_kernel(int64_t n, const uint64_t* __restrict__ keys, int shift,
                                                       uint32_t mask, int nblocks, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t s_cnt[SWAVES][256];
    const int tid = threadIdx.x, wave = tid >> 6;
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
__global__ __launch_bounds__(256) void sort_scan_kernel(int nblocks, uint32_t* __restrict__ hist,
                                                        uint32_t* __restrict__ totals)
{
    __shared__ uint32_t wave_tot[4];
    __shared__ uint32_t s_carry;
    uint32_t* row = hist + (size_t)blockIdx.x * nblocks;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int start = 0; start < nblocks; start += 256) {
        const int i = start + tid;
        const uint32_t v = i < nblocks ? row[i] : 0u;
        const uint32_t incl = wave_incl_scan_u32(v, lane);
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t base = s_carry;
        for (int w = 0; w < wave; w++) base += wave_tot[w];
        if (i < nblocks) row[i] = base + incl - v;
        __syncthreads();
        if (tid == 255) s_carry = base + incl;
        __syncthreads();
    }
    if (tid == 0) totals[blockIdx.x] = s_carry;
}

__global__ __launch_bounds__(ST) void sort_scatter_kernel(int64_t n, const uint64_t* __restrict__ kin,
                                                          const uint32_t* __restrict__ vin,
                                                          uint64_t* __restrict__ kout, uint32_t* __restrict__ vout,
                                                          int shift, uint32_t mask, int nblocks,
                                                          const uint32_t* __restrict__ hist,
                                                          const uint32_t* __restrict__ totals)
{
    __shared__ uint32_t s_cnt[SWAVES][256];  // per-wave digit counts, then per-wave exclusive bases
    __shared__ uint32_t s_start[256];        // first local slot of each digit
    __shared__ uint32_t s_gofs[256];         // global position of local slot 0 of each digit (minus s_start)
    __shared__ uint32_t s_wtot[SWAVES];
    __shared__ uint64_t s_key[STILE];
    __shared__ uint32_t s_val[STILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * STILE;
    const int n_valid = (int)min((int64_t)STILE, n - base);
    for (int w = 0; w < SWAVES; w++) s_cnt[w][tid] = 0;
    __syncthreads();

    // Wave-striped ownership keeps input order: wave w owns slots [w*512, w*512+512), item i of lane l is slot
    // w*512 + i*64 + l.
    uint64_t key[SI];
    uint32_t val[SI], rank[SI];
    const uint64_t lanes_below = (1ull << lane) - 1ull;
#pragma unroll
    for (int i = 0; i < SI; i++) {
        const int slot = wave * (64 * SI) + i * 64 + lane;
        const bool valid = slot < n_valid;
        key[i] = valid ? kin[base + slot] : ~0ull;
        val[i] = valid ? vin[base + slot] : 0u;
    }
#pragma unroll
    for (int i = 0; i < SI; i++) {
        const int slot = wave * (64 * SI) + i * 64 + lane;
        const bool valid = slot < n_valid;
        const uint32_t d = digit_of(key[i], shift, mask);
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const uint64_t m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t before = s_cnt[wave][d];
        rank[i] = before + (uint32_t)__popcll(peers & lanes_below);
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers & lanes_below) == 0) s_cnt[wave][d] = before + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
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
    __shared__ uint32_t s_gwtot[SWAVES];
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
            const uint64_t k = s_key[pos];
            const uint32_t g = s_gofs[digit_of(k, shift, mask)] + (uint32_t)pos;
            kout[g] = k;
            vout[g] = s_val[pos];
        }
    }
}

static int sort_passes(int end_bit) { return (end_bit + 7) / 8; }

int ngs_sort_input_buffer(int end_bit) { return sort_passes(end_bit) & 1; }

void ngs_launch_sort_pairs(hipStream_t s, int64_t R, uint64_t* const keys[2], uint32_t* const vals[2], uint32_t* hist,
                           uint32_t* totals, int end_bit)
{
    if (R <= 0) return;
    const int nblocks = (int)((R + STILE - 1) / STILE);
    int cur = ngs_sort_input_buffer(end_bit);
    for (int shift = 0; shift < end_bit; shift += 8) {
        const int bits = end_bit - shift < 8 ? end_bit - shift : 8;
        const uint32_t mask = (1u << bits) - 1u;
        hipLaunchKernelGGL(sort_hist_kernel, dim3(nblocks), dim3(ST), 0, s, R, keys[cur], shift, mask, nblocks, hist);
        hipLaunchKernelGGL(sort_scan_kernel, dim3(256), dim3(256), 0, s, nblocks, hist, totals);
        hipLaunchKernelGGL(sort_scatter_kernel, dim3(nblocks), dim3(ST), 0, s, R, keys[cur], vals[cur], keys[cur ^ 1],
                           vals[cur ^ 1], shift, mask, nblocks, hist, totals);
        cur ^= 1;
    }
}
