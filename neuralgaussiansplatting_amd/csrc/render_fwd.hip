// Per-tile front-to-back alpha compositing (forward).  One 256-thread
// workgroup per 16x16 tile, four wave64s each covering a 8x8 pixel block.
//
// The four waves of a tile run DECOUPLED: there is no workgroup barrier in the
// kernel.  Each wave reads the tile's depth-sorted list on its own, 64 splats
// at a time: lane l fetches splat l of the batch (the next batch's loads are
// issued before the current one is processed), tests whether the splat can
// reach the wave's block at all (minimum of its quadratic form over the block
// against the opacity threshold, render_common.h) and appends the survivors,
// in list order, to the wave's private LDS slice.  When the slice cannot take
// the next batch's survivors (or the list ends) the wave WALKS it: the
// survivors -- up to 64 of them, from several batches -- are tested against
// each of the block's four 4x4 quadrants, every 16-lane row of the wave (one
// quadrant) gets its own list of the record slots that concern it, and the
// four rows step through their lists side by side, four entries per trip.
// A splat of a few pixels' reach touches one or two quadrants of a block, so
// the rows take about two thirds of the steps the block as a whole would;
// classifying only survivors, 64 at a time, keeps the cost of the four lists
// small against the walk they shorten.  A wave whose 64 pixels are all
// saturated simply leaves; nobody waits for the slowest block of the tile.
//
// Per-pixel arithmetic and its order follow cuda_rasterizer/forward.cu:261-374
// of the reference; the per-pixel body is branch-free (predicated).  Skipping a
// splat culled this way cannot change any pixel because every pixel of the
// block (quadrant) would have failed the alpha test.
#include "render_common.h"

#define FWD_SLOTS 64
typedef float fwd_f4 __attribute__((ext_vector_type(4)));
typedef float fwd_f2 __attribute__((ext_vector_type(2)));
typedef uint32_t fwd_u4 __attribute__((ext_vector_type(4)));

// LDS addresses as numbers: a list entry names a record slot directly, no index arithmetic on the way.
__device__ __forceinline__ uint32_t fwd_lds_address(const void* p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
template <typename V>
__device__ __forceinline__ V fwd_lds_read(uint32_t at)
{
#if __HIP_DEVICE_COMPILE__
    return *(__attribute__((address_space(3))) const V*)at;
#else
    return V{};  // the host pass only parses this
#endif
}


#ifdef NGS_FWD_STATS
// Diagnostic build: [0] wave-batches read, [1] survivors of the first test, [2] walks, [3] survivors with a quadrant
// to visit, [4] list entries (row-visits with an entry), [5] wave-steps (visits of a wave, whatever its rows hold),
// [6] lane-visits that blended, [7] row-visits in which a lane blended.
__device__ unsigned long long g_fwd_stats[8];
extern "C" int ngs_debug_fwd_stats(unsigned long long* out)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fwd_stats), sizeof(g_fwd_stats)) != hipSuccess) return 1;
    unsigned long long zero[8] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_fwd_stats), zero, sizeof(zero)) != hipSuccess;
}
#define FSTAT(k, v) fst[k] += (v)
#else
#define FSTAT(k, v)
#endif

#ifndef NGS_FWD_WG_PER_CU
#define NGS_FWD_WG_PER_CU 7
#endif
__global__ __launch_bounds__(NGS_TILE_PIX, NGS_FWD_WG_PER_CU) void render_fwd_kernel(
    int W, int H, int grid_x, const uint2* __restrict__ ranges, const uint32_t* __restrict__ plist,
    const float4* __restrict__ rec_g, const float* __restrict__ bg, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
    float* __restrict__ out_color, float* __restrict__ acc, int acc_f4_per_block, int acc_f4_total,
    uint32_t* __restrict__ misc, unsigned long long* __restrict__ stage_clock, uint32_t seq,
    uint8_t* __restrict__ quad, uint32_t quad_offset)
{
    // per wave, slot k (48 bytes): [0] = {x, y, conic a, c}, [1] = {conic b, opacity, r, g}, [2] = {b, list position + 1,
    // ln(255 opacity), -}; slot FWD_SLOTS is all zeros: a splat of opacity 0, which no pixel blends
    __shared__ float4 s_rec[NGS_TILE_PIX / 64][FWD_SLOTS + 1][3];
    // per wave and quadrant: the LDS addresses of the slots the quadrant has to visit, in list order
    __shared__ uint32_t s_list[NGS_TILE_PIX / 64][4][FWD_SLOTS];
    const int tid = threadIdx.x, lane = tid & 63, row = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // the same in all lanes: addresses formed from it stay scalar
    const int tile = blockIdx.x;
    // This workgroup's share of the backward pass's accumulator rows is cleared here: the stores leave at once and
    // complete behind the compositing, which keeps the vector ALU busy and the memory system idle.
    {
        const int lo = tile * acc_f4_per_block, hi = min(acc_f4_total, lo + acc_f4_per_block);
        for (int i = lo + tid; i < hi; i += NGS_TILE_PIX) stream_store4(acc + 4 * (size_t)i, make_float4(0.f, 0.f, 0.f, 0.f));
        if (tile == 0 && tid == 0) {
            misc[NGS_MISC_ACC_CLEAN] = 1u;
            misc[NGS_MISC_QUAD] = quad_offset;
            // How long the binning stages between the scan and this kernel took, posted to the host, which picks
            // the source of the depth order for later frames by what each one has cost (capi.hip).
            if (stage_clock)
                __hip_atomic_store(stage_clock, ((unsigned long long)seq << 32) |
                                                    (uint32_t)((uint32_t)wall_clock64() - misc[NGS_MISC_T_SCAN]),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    const int tx = tile % grid_x, ty = tile / grid_x;
    const int px = tx * NGS_TILE_W + tile_local_x(tid), py = ty * NGS_TILE_H + tile_local_y(tid);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    // this wave's pixel block; quadrant k spans [sx0 + 4 (k & 1), +3] x [sy0 + 4 (k >> 1), +3]
    const float sx0 = (float)(tx * NGS_TILE_W + (wave & 1) * NGS_WAVE_W), sx1 = sx0 + (NGS_WAVE_W - 1);
    const float sy0 = (float)(ty * NGS_TILE_H + (wave >> 1) * NGS_WAVE_H), sy1 = sy0 + (NGS_WAVE_H - 1);
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    float4(*rec)[3] = s_rec[wave];
    uint8_t* const wave_quad = quad + 4 * (size_t)range.x + wave;  // this block's byte of the tile's first instance
    if (lane < 3) rec[FWD_SLOTS][lane] = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t my_slot_lds = fwd_lds_address(&rec[lane][0]);
    const uint32_t null_lds = fwd_lds_address(&rec[FWD_SLOTS][0]);
    const uint32_t my_list_lds = fwd_lds_address(&s_list[wave][row][0]);
    static_assert(sizeof(s_list[0]) == 64 * sizeof(uint4), "a wave clears its four lists with one 16-byte store per lane");

#ifdef NGS_FWD_STATS
    unsigned long long fst[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    bool done = !inside;
    float T = 1.0f, Cb = 0.0f;
    fwd_f2 Crg = {0.0f, 0.0f};
    uint32_t last_contributor = 0;

    // One entry of a row's list: the splat in the slot at LDS address `at` against the row's 16 pixels.  The two
    // squares of the exponent are formed as pairs (packed FP32 instructions: same roundings, half the issue slots).
    const fwd_f2 pix = {pxf, pyf};
    auto visit = [&](const uint32_t at) {
        const fwd_f4 p = fwd_lds_read<fwd_f4>(at), q = fwd_lds_read<fwd_f4>(at + 16);
        const fwd_f2 c = fwd_lds_read<fwd_f2>(at + 32);
        const fwd_f2 d = p.xy - pix;
        const fwd_f2 sq = (p.zw * d) * d;  // {a dx dx, c dy dy}
        const float dx = d.x, dy = d.y;
        float sum = sq.x + sq.y;
        asm("" : "+v"(sum));  // the rest one by one (left alone, the compiler pairs it up
                                                            // with register moves that cost more than they save)
        float half = -0.5f * sum;
        asm("" : "+v"(half));
        const float power = half - q.x * dx * dy;
        const float alpha = fminf(0.99f, q.y * exp_nonpos(power));
        const float test_T = T * (1 - alpha);
        const bool hit = !done && power <= 0.0f && !(alpha < 1.0f / 255.0f);
        const bool blend = hit && !(test_T < 0.0001f);
        const bool stop = hit && !blend;  // from the masks: one comparison instead of two
        done = done || stop;
        // (colour * alpha) * T added with one rounding, as the reference's build contracts the same expression
        // (forward.cu:355-356 compiled with nvcc's default -fmad=true).  A pixel that does not blend the splat
        // adds (colour * 0) * T, an exact zero, instead of selecting its old value three times.
        FSTAT(5, 1);
        FSTAT(6, __popcll(wave_mask(blend)));
#ifdef NGS_FWD_STATS
        {
            const unsigned long long bm = wave_mask(blend);
            fst[7] += ((bm & 0xffffull) != 0) + ((bm & 0xffff0000ull) != 0) + ((bm & 0xffff00000000ull) != 0) +
                      ((bm & 0xffff000000000000ull) != 0);
        }
#endif
        const float alpha_b = blend ? alpha : 0.0f;
        const fwd_f2 Tp = {T, T}, ab = {alpha_b, alpha_b};
        Crg = __builtin_elementwise_fma(q.zw * ab, Tp, Crg);
        Cb = fmaf(c.x * alpha_b, T, Cb);
        T = blend ? test_T : T;
        last_contributor = blend ? __float_as_uint(c.y) : last_contributor;
    };

    // The walk over the `count` survivors parked in the wave's slots.
    auto walk = [&](const int count) {
        unsigned bits = 0;
        if (lane < count) {
            const float4 p = rec[lane][0], q = rec[lane][1], c = rec[lane][2];
            bits = splat_reaches_quadrants(p.x, p.y, p.z, q.x, p.w, c.z, sx0, sy0);
            // The backward pass walks the same quadrants' lists: it is handed the bits instead of repeating the
            // tests.  Every instance the wave looks at gets its byte exactly once, here or where the first test
            // drops it.
            wave_quad[4u * (__float_as_uint(c.y) - 1u)] = (uint8_t)bits;
        }
        // a quadrant whose pixels are all saturated (or outside the image) takes no list
        const unsigned long long open = wave_mask(!done);
        const bool r0 = (bits & 1u) && (open & 0xffffull), r1 = (bits & 2u) && (open & 0xffff0000ull);
        const bool r2 = (bits & 4u) && (open & 0xffff00000000ull), r3 = (bits & 8u) && (open & 0xffff000000000000ull);
        // every list entry names the empty slot until a survivor takes it, so a row that runs out of entries before
        // the others (or before the trip of four ends) blends nothing
        ((uint4*)&s_list[wave][0][0])[lane] = make_uint4(null_lds, null_lds, null_lds, null_lds);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const unsigned long long m0 = wave_mask(r0), m1 = wave_mask(r1), m2 = wave_mask(r2), m3 = wave_mask(r3);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (r0) s_list[wave][0][__popcll(m0 & below)] = my_slot_lds;
        if (r1) s_list[wave][1][__popcll(m1 & below)] = my_slot_lds;
        if (r2) s_list[wave][2][__popcll(m2 & below)] = my_slot_lds;
        if (r3) s_list[wave][3][__popcll(m3 & below)] = my_slot_lds;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int steps = max(max(__popcll(m0), __popcll(m1)), max(__popcll(m2), __popcll(m3)));
        FSTAT(2, 1);
        FSTAT(3, __popcll(wave_mask(bits != 0)));
        FSTAT(4, __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3));
        // Four entries per trip: their addresses come as one 16-byte read, and all records are requested before the
        // first is used, so their LDS latency passes behind the arithmetic.
        for (int it = 0; it < steps; it += 4) {
            const fwd_u4 w = fwd_lds_read<fwd_u4>(my_list_lds + 4 * it);
            visit(w.x);
            if (it + 1 < steps) visit(w.y);
            if (it + 2 < steps) visit(w.z);
            if (it + 3 < steps) visit(w.w);
            // has the whole block saturated?  (a saturated pixel ignores further splats anyway)
            if (wave_mask(!done) == 0ull) break;
        }
    };

    // registers of the batch in flight
    float4 na = make_float4(0, 0, 0, 0), nb = na, nc = na;
    if (lane < n) {
        const float4* r = rec_g + 4 * (size_t)plist[range.x + lane];  // one 64-byte line per splat
        na = r[0];
        nb = r[1];
        nc = r[2];
    }
    int pending = 0;  // survivors parked and not yet walked (wave-uniform)
    for (int base = 0; base < n || pending > 0; base += 64) {
        if (wave_mask(!done) == 0ull) break;  // every pixel of the block is saturated (or outside the image)
        const float4 a = na, b = nb, c = nc;
        const bool have = base + lane < n;
        // First test, per batch: a lower bound of the splat's quadratic form over the block from the block's distance
        // to the centre along either axis alone (for a given dx the form is at least dx^2 (a - b^2 / c), whatever dy:
        // the record carries both such factors, zero for a conic that is not positive definite), against the same
        // threshold and margin as splat_reaches().  It keeps everything the exact test keeps and costs a quarter;
        // what it keeps for nothing is dropped by the exact test per quadrant when the parked survivors are walked.
        bool reach = false;
        if (have) {
            const float xl = sx0 - a.x, xh = sx1 - a.x, yl = sy0 - a.y, yh = sy1 - a.y;
            const float gx = fmaxf(fmaxf(xl, -xh), 0.0f), gy = fmaxf(fmaxf(yl, -yh), 0.0f);
            const float lb = fmaxf(gx * gx * c.z, gy * gy * c.w);
            const float mx = fmaxf(fabsf(xl), fabsf(xh)), my = fmaxf(fabsf(yl), fabsf(yh));
            const float mag = a.z * mx * mx + fabsf(2.0f * a.w) * mx * my + b.x * my * my;
            reach = !(0.5f * lb > c.y + 0.01f + 1e-5f * mag);
        }
        if (have && !reach) wave_quad[4u * (uint32_t)(base + lane)] = 0;
        const unsigned long long mask = wave_mask(reach);
        const int cnt = __popcll(mask);
        FSTAT(0, base < n ? 1 : 0);
        FSTAT(1, cnt);
        if (pending + cnt > FWD_SLOTS || base >= n) {
            walk(pending);
            pending = 0;
            if (wave_mask(!done) == 0ull) break;
        }
        const int nxt = base + 64 + lane;
        if (nxt < n) {
            const float4* r = rec_g + 4 * (size_t)plist[range.x + nxt];
            na = r[0];
            nb = r[1];
            nc = r[2];
        }
        if (reach) {
            const int slot = pending + __popcll(mask & ((1ull << lane) - 1ull));
            rec[slot][0] = make_float4(a.x, a.y, a.z, b.x);
            rec[slot][1] = make_float4(a.w, b.y, b.z, b.w);
            rec[slot][2] = make_float4(c.x, __uint_as_float((uint32_t)(base + lane + 1)), c.y, 0.0f);
        }
        pending += cnt;
    }
#ifdef NGS_FWD_STATS
    if (lane == 0)
        for (int k = 0; k < 8; k++) atomicAdd(&g_fwd_stats[k], fst[k]);
#endif
    if (inside) {
        const size_t tp = (size_t)tile * NGS_TILE_PIX + tid;
        final_T[tp] = T;
        n_contrib[tp] = last_contributor;
        const size_t pid = (size_t)W * py + px, plane = (size_t)W * H;
        out_color[pid] = Crg.x + T * bg[0];
        out_color[plane + pid] = Crg.y + T * bg[1];
        out_color[2 * plane + pid] = Cb + T * bg[2];
    }
}

void ngs_launch_render_fwd(hipStream_t s, const NgsCamera& cam, const uint2* ranges, const uint32_t* plist,
                           const float4* rec, const float* bg, float* final_T,
                           uint32_t* n_contrib, float* out_color, float* acc, int P, uint32_t* misc,
                           unsigned long long* stage_clock, uint32_t seq, uint8_t* quad, uint32_t quad_offset)
{
    const int T = cam.grid_x * cam.grid_y;
    const long long f4 = (long long)P * (NGS_ACC_STRIDE / 4);  // P <= NGS_MAX_GAUSSIANS keeps this inside an int
    hipLaunchKernelGGL(render_fwd_kernel, dim3(T), dim3(NGS_TILE_PIX), 0, s, cam.W, cam.H, cam.grid_x, ranges, plist,
                       rec, bg, final_T, n_contrib, out_color, acc, (int)((f4 + T - 1) / T), (int)f4, misc, stage_clock, seq, quad,
                       quad_offset);
}
