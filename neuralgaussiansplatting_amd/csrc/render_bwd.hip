// Per-tile back-to-front compositing gradient.  Same tiling and LDS staging
// as the forward kernel, traversed in reverse; per-pixel arithmetic follows
// cuda_rasterizer/backward.cu:399-557 of the reference.  What differs:
//
//  * The traversal starts at the tile's deepest contributor (the maximum of
//    the per-pixel last-contributor indices the forward pass stored) instead
//    of at the end of the tile's list: entries behind it are skipped by every
//    pixel in the reference too (backward.cu:486-488), so they are never staged.
//  * Every 16-lane row of a wave owns one 4x4 pixel quadrant and walks its OWN
//    list of splats: per batch of staged splats each wave ballots, quadrant by
//    quadrant, which splats can reach the quadrant and lie in front of its
//    deepest contributor, compacts the survivors into four short index lists,
//    and then all four rows step through their lists side by side.  A small
//    quadrant is touched by far fewer splats than a tile, and a touching splat
//    covers most of its pixels, so few lanes run the arithmetic for nothing.
//  * What a pixel hands on per splat: the reference adds nine gradient terms
//    per (pixel, splat) pair.  Five of them are products of a per-splat
//    constant (conic, opacity) with a moment of dL/dG over the pixel's offset
//    from the splat centre; here the pixels sum the moments, and the conic is
//    applied once per Gaussian by the per-Gaussian backward kernel.
//  * Reduction: the reference issues nine global float atomics per (pixel,
//    splat) pair.  Here the nine partials are summed over the 16 pixels of the
//    quadrant in registers (a reduce-scatter of DPP row exchanges: after four
//    steps nine lanes of the row hold one total each), then across the wave's
//    steps in a table of sums the wave owns in LDS (plain read, add, write:
//    no LDS float atomic unless two rows meet on one splat), the workgroup's
//    four tables are added up when a round of 128 splats is flushed, and each
//    splat's row leaves the workgroup once per round as ONE 64-byte-row group
//    of global atomics into the [P][16] accumulator.
#include "render_common.h"

typedef float v2f __attribute__((ext_vector_type(2)));  // pairs of channels go through the packed-FP32 instructions

// One step of the reduce-scatter inside a 16-lane row.  `ctrl` is a DPP pattern that pairs every lane with a lane
// whose `upper` flag is the opposite; afterwards the lanes with upper == false hold a summed over each pair, the
// others b.
template <int ctrl>
__device__ __forceinline__ float row_fold(float a, float b, bool upper)
{
    const float keep = upper ? b : a, give = upper ? a : b;
    return keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give), ctrl, 0xf,
                                                                        0xf, false));
}
template <int ctrl>
__device__ __forceinline__ float row_pair_sum(float a)
{
    return a + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), ctrl, 0xf, 0xf,
                                                                     false));
}
#define DPP_ROW_MIRROR 0x140       // lane i <-> 15 - i
#define DPP_ROW_HALF_MIRROR 0x141  // lane i <-> 7 - i inside each half row
#define DPP_QUAD_XOR2 0x4E         // quad_perm [2,3,0,1]
#define DPP_QUAD_XOR1 0xB1         // quad_perm [1,0,3,2]

#define BWD_BATCH 128  // splats staged per round

#ifdef NGS_BWD_STATS
// Diagnostic build: utilisation counters of the walk.  [0] wave-steps, [1] row-steps with an entry, [2] live
// lane-steps, [3] wave-steps left at the early exit, [4] wave-batches, [5] sum over batches of the longest wave's
// steps, [6] sum over batches and waves of steps, [7] staged splats.
__device__ unsigned long long g_bwd_stats[8];
extern "C" int ngs_debug_bwd_stats(unsigned long long* out)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bwd_stats), sizeof(g_bwd_stats)) != hipSuccess) return 1;
    unsigned long long zero[8] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_stats), zero, sizeof(zero)) != hipSuccess;
}
#define STAT(k, v) st[k] += (v)
#else
#define STAT(k, v)
#endif

__global__ __launch_bounds__(NGS_TILE_PIX, 6) void render_bwd_kernel(
    int W, int H, int grid_x, const uint2* __restrict__ ranges, const uint32_t* __restrict__ plist,
    const float4* __restrict__ splat_rec, const float* __restrict__ bg, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dpix, float* __restrict__ acc, uint32_t* __restrict__ misc)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) misc[NGS_MISC_ACC_CLEAN] = 0u;  // the rows are about to be used
    // record k: [0] = {x, y, conic a, c}, [1] = {conic b, opacity, -, the forward pass's reach bits of the instance},
    // [2] = {r, g, b, Gaussian id bits}
    __shared__ float4 s_rec[BWD_BATCH][3];
    // One table of sums per wave: a wave's LDS operations complete in order, so it adds to its own table with a plain
    // read, add and write.  An LDS float atomic costs about three cycles of the CU's LDS per active lane, ten times
    // an integer one or a store (tools/probes/lds_atomic_probe.hip); it is kept for the steps in which several rows
    // of the wave work on the same splat and so meet in one entry.
    __shared__ float s_acc[NGS_TILE_PIX / 64][BWD_BATCH][ACC_USED];
    __shared__ uint8_t s_list[NGS_TILE_PIX / 64][4][BWD_BATCH];  // per wave and quadrant: staged slots to visit
    __shared__ uint32_t s_wmax[NGS_TILE_PIX / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = lane >> 4, li = lane & 15;
    const int tile = blockIdx.x;
    const int tx = tile % grid_x, ty = tile / grid_x;
    const int px = tx * NGS_TILE_W + tile_local_x(tid), py = ty * NGS_TILE_H + tile_local_y(tid);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];

    const size_t tp = (size_t)tile * NGS_TILE_PIX + tid;
    const float T_final = inside ? final_T[tp] : 0.0f;
    float T = T_final;
    const uint32_t last_contributor = inside ? n_contrib[tp] : 0u;
    // deepest contributor of each quadrant (16-lane row), of this block (wave) and of the tile
    uint32_t qmax = last_contributor;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) qmax = max(qmax, (uint32_t)__shfl_xor((int)qmax, o, 64));
    const uint32_t qmax0 = (uint32_t)__builtin_amdgcn_readlane((int)qmax, 0);
    const uint32_t qmax1 = (uint32_t)__builtin_amdgcn_readlane((int)qmax, 16);
    const uint32_t qmax2 = (uint32_t)__builtin_amdgcn_readlane((int)qmax, 32);
    const uint32_t qmax3 = (uint32_t)__builtin_amdgcn_readlane((int)qmax, 48);
    const uint32_t wmax = max(max(qmax0, qmax1), max(qmax2, qmax3));
    if (lane == 0) s_wmax[wave] = wmax;
    // The walk reads list entries, and the records they name, one and two steps ahead of knowing whether they exist:
    // whatever it can come across must be a valid slot number and a finite record (a row without an entry runs the
    // arithmetic with alpha = 0, which leaves its state alone only if the colour it multiplies by 0 is finite).
    for (int k = tid; k < (int)(sizeof(s_list) / sizeof(uint32_t)); k += NGS_TILE_PIX) ((uint32_t*)&s_list[0][0][0])[k] = 0u;
    for (int k = tid; k < BWD_BATCH * 3; k += NGS_TILE_PIX) (&s_rec[0][0])[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const uint32_t tmax = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));
    if (tmax == 0) return;  // nothing in this tile was blended (block-uniform)

    float dpr = 0.0f, dpg = 0.0f, dpb = 0.0f;
    if (inside) {
        const size_t pid = (size_t)W * py + px, plane = (size_t)W * H;
        dpr = dL_dpix[pid];
        dpg = dL_dpix[plane + pid];
        dpb = dL_dpix[2 * plane + pid];
    }
    float bgdot = 0.0f;
    bgdot += bg[0] * dpr;
    bgdot += bg[1] * dpg;
    bgdot += bg[2] * dpb;
    const float bg_term = -T_final * bgdot;  // d(background's share of the pixel)/d(alpha) is this over (1 - alpha)
    v2f rec_rg = {0.0f, 0.0f};
    float rec_b = 0.0f;
    const v2f dp_rg = {dpr, dpg};
    const v2f pix = {pxf, pyf};
    // After the reduce-scatter the even lanes of a row hold partials 0..7 (index bits taken from lane bits 3, 2, 1
    // in that order of significance 1, 2, 4) and the odd lanes partial 8; lane 1 speaks for the odd ones.
    const bool up8 = (li & 8) != 0, up4 = (li & 4) != 0, up2 = (li & 2) != 0, up1 = (li & 1) != 0;
    const int my_slot = !up1 ? ((up2 ? 4 : 0) + (up4 ? 2 : 0) + (up8 ? 1 : 0)) : (li == 1 ? 8 : -1);
    uint8_t* const my_list = &s_list[wave][row][0];

    // Zero-based list positions tmax-1 ... 0 are walked in that order, BWD_BATCH per round;
    // staged slot s of round r holds list position tmax-1 - r*BWD_BATCH - s.
    const int rounds = (int)((tmax + BWD_BATCH - 1) / BWD_BATCH);
#ifdef NGS_BWD_STATS
    __shared__ int s_steps[4];
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // the list entry a thread stages next: fetched a round ahead, so that a round's staging is one dependent memory
    // round trip (the records) instead of two
    // which quadrants of which block each instance can reach, as the forward pass found it: one byte per 8x8 block
    const uint32_t* const quad = (const uint32_t*)((const char*)plist + (size_t)misc[NGS_MISC_QUAD] * NGS_ALIGN);
    uint32_t next_id = 0, next_quad = 0;
    if (tid < BWD_BATCH && (int)tmax - 1 - tid >= 0) {
        next_id = plist[range.x + (uint32_t)((int)tmax - 1 - tid)];
        next_quad = quad[range.x + (uint32_t)((int)tmax - 1 - tid)];
    }
    for (int r = 0; r < rounds; r++) {
        __syncthreads();  // previous batch fully flushed before its LDS is reused
        const int first = (int)tmax - 1 - r * BWD_BATCH;  // list position of slot 0
        const int nb = min(BWD_BATCH, first + 1);
        if (tid < nb) {
            const uint32_t id = next_id, reach_bits = next_quad;
            if (first - BWD_BATCH - tid >= 0) {
                next_id = plist[range.x + (uint32_t)(first - BWD_BATCH - tid)];
                next_quad = quad[range.x + (uint32_t)(first - BWD_BATCH - tid)];
            }
            const float4* gr = splat_rec + 4 * (size_t)id;  // one 64-byte line per splat
            const float4 a = gr[0];
            const float4 b = gr[1];
            const float c = gr[2].x;
            s_rec[tid][0] = make_float4(a.x, a.y, a.z, b.x);
            s_rec[tid][1] = make_float4(a.w, b.y, 0.0f, __uint_as_float(reach_bits));
            s_rec[tid][2] = make_float4(b.z, b.w, c, __uint_as_float(id));
        }
        static_assert(sizeof(s_acc) % sizeof(float4) == 0, "the tables are cleared with 16-byte stores");
#pragma unroll
        for (int c = 0; c * NGS_TILE_PIX < (int)(sizeof(s_acc) / sizeof(float4)); c++)
            if (c * NGS_TILE_PIX + tid < (int)(sizeof(s_acc) / sizeof(float4)))
                ((float4*)&s_acc[0][0][0])[c * NGS_TILE_PIX + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();

        // ---- which staged splats does each quadrant of this block have to visit? ----
        int n0 = 0, n1 = 0, n2 = 0, n3 = 0;  // wave-uniform list lengths
        for (int chunk = 0; chunk < nb; chunk += 64) {
            const int mine = chunk + lane;
            bool r0 = false, r1 = false, r2 = false, r3 = false;
            if (mine < nb) {
                const uint32_t pos = (uint32_t)(first - mine);
                const unsigned bits = __float_as_uint(s_rec[mine][1].w) >> (8 * wave);
                r0 = pos < qmax0 && (bits & 1u);
                r1 = pos < qmax1 && (bits & 2u);
                r2 = pos < qmax2 && (bits & 4u);
                r3 = pos < qmax3 && (bits & 8u);
            }
            const unsigned long long m0 = __ballot(r0), m1 = __ballot(r1), m2 = __ballot(r2), m3 = __ballot(r3);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (r0) s_list[wave][0][n0 + __popcll(m0 & below)] = (uint8_t)mine;
            if (r1) s_list[wave][1][n1 + __popcll(m1 & below)] = (uint8_t)mine;
            if (r2) s_list[wave][2][n2 + __popcll(m2 & below)] = (uint8_t)mine;
            if (r3) s_list[wave][3][n3 + __popcll(m3 & below)] = (uint8_t)mine;
            n0 += __popcll(m0);
            n1 += __popcll(m1);
            n2 += __popcll(m2);
            n3 += __popcll(m3);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int my_n = row == 0 ? n0 : row == 1 ? n1 : row == 2 ? n2 : n3;
        const int steps = max(max(n0, n1), max(n2, n3));
        // Steps in which two rows of this wave work on the same splat (their sums then meet in one table entry): one
        // bit per step, found 64 steps at a time.
        unsigned long long dup[BWD_BATCH / 64];
#pragma unroll
        for (int ch = 0; ch < BWD_BATCH / 64; ch++) {
            dup[ch] = 0ull;
            if (ch * 64 < steps) {  // wave-uniform
                const int t = ch * 64 + lane;
                const int e0 = t < n0 ? (int)s_list[wave][0][t] : -1, e1 = t < n1 ? (int)s_list[wave][1][t] : -2;
                const int e2 = t < n2 ? (int)s_list[wave][2][t] : -3, e3 = t < n3 ? (int)s_list[wave][3][t] : -4;
                dup[ch] = wave_mask(e0 == e1 || e0 == e2 || e0 == e3 || e1 == e2 || e1 == e3 || e2 == e3);
            }
        }

        // ---- the four rows walk their lists side by side ----
        // Slot addresses as 24-bit multiply-adds: the 48- and 36-byte strides would otherwise cost a quarter-rate
        // 32/64-bit integer multiply each, twice per step.
        const char* const rec_bytes = (const char*)&s_rec[0][0];
        const int behind = first - (int)last_contributor;  // slots up to this one lie behind everything the pixel blended
        float* const acc_lane = &s_acc[wave][0][0] + max(my_slot, 0);
        // One step of a row: splat `j` of the batch (record p, q, col) against the row's 16 pixels.
        auto step = [&](const int it, const int j, const float4 p, const float4 q, const float4 col, const bool shared) {
            const bool have = it < my_n;
            // same evaluation order as the forward pass (and the reference): alpha must come out bit-identical;
            // the two squares as a pair (packed FP32: same roundings, half the issue slots)
            const v2f d = (v2f){p.x, p.y} - pix;
            const v2f sq = ((v2f){p.z, p.w} * d) * d;  // {a dx dx, c dy dy}
            const float dx = d.x, dy = d.y;
            float sum = sq.x + sq.y;
            asm("" : "+v"(sum));  // the rest one by one (left alone, the compiler pairs it up with register moves)
            float half = -0.5f * sum;
            asm("" : "+v"(half));
            const float power = half - q.x * dx * dy;
            const float G = exp_nonpos(power);
            const float alpha = fminf(0.99f, q.y * G);
            // first - j is the reference's `contributor` after its decrement; the pixel blended the splat if that
            // lies below its last contributor, i.e. if j lies above `behind`.  The alpha test comes last and on its
            // own, so that the wave's mask of live lanes is that comparison's result as it stands.
            const float alpha_t = (have && j > behind && !(power > 0.0f)) ? alpha : 0.0f;
            const bool live = !(alpha_t < 1.0f / 255.0f);
            STAT(0, 1);
            STAT(1, __popcll(wave_mask(have)) / 16);
            STAT(2, __popcll(wave_mask(live)));
            if (wave_mask(live) == 0ull) {
                STAT(3, 1);
                return;  // wave-uniform
            }

            // Every lane evaluates the body.  A dead lane (pixel that does not blend this splat) runs it with
            // alpha = 0, which leaves its state untouched without a single select: T / (1 - 0) = T, the
            // colour behind it becomes 0 * colour + 1 * rec = rec, and every partial below is an exact zero
            // (G and dL/dalpha are forced to 0 as well).
            // the table entry this lane will add to, on its way while the sums are formed
            // (its LDS address is formed once and kept: left to itself the compiler forms it again for the write)
            typedef __attribute__((address_space(3))) float lds_float;
            lds_float* entry = (lds_float*)(acc_lane + __umul24((unsigned)j, (unsigned)ACC_USED));
            asm volatile("" : "+v"(entry));
            const float before = *entry;
            asm volatile("" :: "v"(col.w));  // the colour comes as one 16-byte LDS read (a 12-byte one costs twice the LDS cycles)
            const float alpha_l = live ? alpha_t : 0.0f;
            const float om = 1.f - alpha_l;
            const float inv = __builtin_amdgcn_rcpf(om);  // v_rcp_f32, 1 ulp; exactly 1 for a dead lane
            const float Tn = T * inv;
            const float w = alpha_l * Tn;
            const v2f col_rg = {col.x, col.y};
            // `rec` is the colour accumulated behind this splat (the reference's accum_rec after its update,
            // backward.cu:503-513)
            // From here on only gradients are formed (compared within a tolerance, and summed by atomics in no
            // fixed order anyway): products and sums are fused where that saves an instruction.
            const v2f d_rg = col_rg - rec_rg;
            float dL_dalpha = fmaf(col.z - rec_b, dpb, fmaf(d_rg.y, dp_rg.y, d_rg.x * dp_rg.x));
            dL_dalpha = fmaf(dL_dalpha, Tn, bg_term * inv);
            // What leaves the pixel are the moments of g = G dL/dalpha over the pixel offsets: the splat's conic and
            // opacity, which turn them into the gradients of its position, conic and opacity, are the same for every
            // pixel and are applied once per splat when the batch is flushed.
            const float g = live ? G * dL_dalpha : 0.0f;
            const float go = q.y * g;  // dL/dG
            const float m1x = go * dx, m1y = go * dy;
            const float m2xx = m1x * dx, m2xy = m1x * dy, m2yy = m1y * dy;
            const v2f v67 = w * dp_rg;  // ACC_R, ACC_G
            const float v8 = w * dpb;   // ACC_B
            // state for the next (nearer) splat: transmittance in front of this one, and the colour behind
            // the next one = this splat blended over what was behind it (same expression the reference
            // evaluates one step later from its last_alpha / last_color)
            T = Tn;
            rec_rg = __builtin_elementwise_fma((v2f){alpha_l, alpha_l}, col_rg, om * rec_rg);
            rec_b = fmaf(alpha_l, col.z, om * rec_b);

            // Sum over the quadrant's 16 pixels: four exchange steps, each halving the number of registers.
            const float a01 = row_fold<DPP_ROW_MIRROR>(m1x, m1y, up8);
            const float a23 = row_fold<DPP_ROW_MIRROR>(m2xx, m2xy, up8);
            const float a45 = row_fold<DPP_ROW_MIRROR>(m2yy, g, up8);
            const float a67 = row_fold<DPP_ROW_MIRROR>(v67.x, v67.y, up8);
            const float a8 = row_pair_sum<DPP_ROW_MIRROR>(v8);
            const float b03 = row_fold<DPP_ROW_HALF_MIRROR>(a01, a23, up4);
            const float b47 = row_fold<DPP_ROW_HALF_MIRROR>(a45, a67, up4);
            const float b8 = row_pair_sum<DPP_ROW_HALF_MIRROR>(a8);
            const float c07 = row_fold<DPP_QUAD_XOR2>(b03, b47, up2);
            const float c8 = row_pair_sum<DPP_QUAD_XOR2>(b8);
            const float total = row_fold<DPP_QUAD_XOR1>(c07, c8, up1);
            asm volatile("" ::"v"(total));  // formed here, as one DPP add: not taken apart and half of it moved into the branch below
            if (have && my_slot >= 0) {
                if (shared) {  // wave-uniform: rows that work on the same splat in this step meet in one entry
                    if (total != 0.0f) atomicAdd((float*)entry, total);
                } else {
                    *entry = before + total;
                }
            }
        };
        auto record_of = [&](const int j) {
            return (const float4*)(rec_bytes + __umul24((unsigned)j, (unsigned)sizeof(float4[3])));
        };
        // read ahead without asking whether an entry exists: a stale byte is a valid slot number, and `have` keeps
        // a row from using what lies behind the end of its list
        auto list_at = [&](const int k) { return (int)my_list[min(k, BWD_BATCH - 1)]; };
        // Two steps per trip, the records of the two alternating between two sets of registers: step k requests
        // the record of step k+1 and the list entry of step k+2 before its own arithmetic.
        int ja = list_at(0), jb = list_at(1);
        const float4* ra = record_of(ja);
        float4 pa = ra[0], qa = ra[1], ca = ra[2];
        for (int it = 0; it < steps; it += 2) {
            const float4* rb = record_of(jb);
            const float4 pb = rb[0], qb = rb[1], cb = rb[2];
            const int jc = list_at(it + 2);
            step(it, ja, pa, qa, ca, (dup[(it >> 6) & (BWD_BATCH / 64 - 1)] >> (it & 63)) & 1ull);
            if (it + 1 >= steps) break;
            ra = record_of(jc);
            pa = ra[0]; qa = ra[1]; ca = ra[2];
            const int jd = list_at(it + 3);
            step(it + 1, jb, pb, qb, cb, (dup[((it + 1) >> 6) & (BWD_BATCH / 64 - 1)] >> ((it + 1) & 63)) & 1ull);
            ja = jc;
            jb = jd;
        }
#ifdef NGS_BWD_STATS
        if (lane == 0) s_steps[wave] = steps;
        st[4] += 1;
#endif
        __syncthreads();
#ifdef NGS_BWD_STATS
        if (tid == 0) {
            st[5] += max(max(s_steps[0], s_steps[1]), max(s_steps[2], s_steps[3]));
            st[6] += s_steps[0] + s_steps[1] + s_steps[2] + s_steps[3];
            st[7] += nb;
        }
#endif
        // Flush: 16 lanes per splat, lane c adds slot c of the splat's 64-byte accumulator row.
        // Flush: 16 lanes per splat, lane c adds slot c of the splat's 64-byte accumulator row (the four waves'
        // tables summed).
        const int c = tid & 15;
        for (int j = tid >> 4; j < nb; j += NGS_TILE_PIX / 16) {
            if (c < ACC_USED) {
                const float v = (s_acc[0][j][c] + s_acc[1][j][c]) + (s_acc[2][j][c] + s_acc[3][j][c]);
                if (v != 0.0f)
                    atomicAdd(acc + (size_t)__float_as_uint(s_rec[j][2].w) * NGS_ACC_STRIDE + c, v);
            }
        }
    }
#ifdef NGS_BWD_STATS
    if (lane == 0)
        for (int k = 0; k < 8; k++)
            if (st[k]) atomicAdd(&g_bwd_stats[k], st[k]);
#endif
}

void ngs_launch_render_bwd(hipStream_t s, const NgsCamera& cam, const uint2* ranges, const uint32_t* plist,
                           const float4* rec, const float* bg, const float* final_T,
                           const uint32_t* n_contrib, const float* dL_dpix, float* acc, uint32_t* misc)
{
    hipLaunchKernelGGL(render_bwd_kernel, dim3(cam.grid_x * cam.grid_y), dim3(NGS_TILE_PIX), 0, s, cam.W, cam.H,
                       cam.grid_x, ranges, plist, rec, bg, final_T, n_contrib, dL_dpix, acc, misc);
}

// The forward pass leaves the accumulator rows cleared and says so in misc[NGS_MISC_ACC_CLEAN]; a second backward
// pass over the same state finds the word at 0 and clears them itself.
__global__ __launch_bounds__(256) void zero_acc_unless_clean_kernel(float* __restrict__ acc, long long n_f4,
                                                                    const uint32_t* __restrict__ misc)
{
    if (misc[NGS_MISC_ACC_CLEAN] == 1u) return;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_f4; i += (long long)gridDim.x * 256)
        stream_store4(acc + 4 * i, make_float4(0.f, 0.f, 0.f, 0.f));
}

void ngs_launch_zero_acc_unless_clean(hipStream_t s, float* acc, int P, const uint32_t* misc)
{
    const long long n_f4 = (long long)P * (NGS_ACC_STRIDE / 4);
    // a small grid: the launch normally finds nothing to do, and the rare launch that does can take its time
    const long long blocks = (n_f4 + 256 * 8 - 1) / (256 * 8);
    hipLaunchKernelGGL(zero_acc_unless_clean_kernel, dim3((unsigned)(blocks < 512 ? blocks : 512)), dim3(256), 0, s,
                       acc, n_f4, misc);
}
