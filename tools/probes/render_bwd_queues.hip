// EXPERIMENT, not part of the library (kept for the record; drop it into csrc/render_bwd.hip to rebuild it).
// A backward compositing kernel with decoupled waves and per-quadrant ring queues, written to remove the two
// imbalances counters showed in the shipped kernel at 3 M Gaussians / 1080p: rows of a wave stepping to the longest
// of their four lists (row balance 0.85) and the four waves of a tile meeting at barriers (0.84).
// Outcome on MI355X: bit-compatible results, 3 % fewer steps walked, but 0.91 ms against 0.87 ms.  The row imbalance
// turned out to be inherent (quadrants saturate at different depths, so their total work differs: balance 0.87 even
// with queues), and the barrier waits of the shipped kernel are already hidden by the six workgroups resident per
// CU: the vector ALU issue port is the limiter, so only fewer or cheaper instructions help, and this form adds
// queue upkeep and runs at five waves per SIMD instead of six.
// Per-tile back-to-front compositing gradient.  Same tiling as the forward
// kernel (one 256-thread workgroup per 16x16 tile, four wave64s each covering
// an 8x8 pixel block), traversed in reverse; per-pixel arithmetic follows
// cuda_rasterizer/backward.cu:399-557 of the reference.  What differs:
//
//  * The four waves of a tile run DECOUPLED: there is no workgroup barrier in
//    the kernel.  Each wave starts at the deepest contributor of its own block
//    (the maximum of the per-pixel last-contributor indices the forward pass
//    stored): entries behind it are skipped by every pixel in the reference too
//    (backward.cu:486-488), so they are never fetched.
//  * Every 16-lane row of a wave owns one 4x4 pixel quadrant and walks its OWN
//    queue of splats.  The wave reads the tile's list 64 candidates at a time
//    (lane = candidate), tests each against its four quadrants (can the splat
//    reach the quadrant at all, does it lie in front of the quadrant's deepest
//    contributor), parks the survivors' records in a ring of slots in LDS and
//    appends each survivor to the queue of every quadrant it reaches.  The four
//    rows then step through their queues side by side, each at its own
//    position: a row with a short list in one stretch of the depth order simply
//    runs ahead into the next one instead of idling until the other rows have
//    caught up, as long as the ring has room.  A small quadrant is touched by
//    far fewer splats than a tile and a touching splat covers most of its
//    pixels, so few lanes run the arithmetic for nothing.
//  * Reduction: the reference issues nine global float atomics per (pixel,
//    splat) pair.  Here the nine partials are summed over the 16 pixels of the
//    quadrant in registers (a reduce-scatter of DPP row exchanges: after four
//    steps nine lanes of the row hold one total each), then across the wave's
//    quadrants in the slot's accumulator row in LDS, and a slot's row leaves
//    the wave once, when every queue has passed the slot, as one group of
//    global atomics into the [P][16] accumulator (seven slots per instruction).
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

#ifndef BWD_SLOTS
#define BWD_SLOTS 64  // survivor slots per wave (a ring; at most BWD_SLOTS - 1 are in use)
#endif
#define BWD_QRING 128  // entries per quadrant queue (a ring of slot numbers; power of two, >= BWD_SLOTS)
#ifndef BWD_RUN
#define BWD_RUN 16  // steps the rows walk between two looks at the queues
#endif
#ifndef BWD_FREE_MIN
#define BWD_FREE_MIN 28  // free slots wanted before another 64 candidates are tested (about what they yield)
#endif
#define BWD_WAVES (NGS_TILE_PIX / 64)

#ifdef NGS_BWD_STATS
// instrumented builds only: [0] wave steps walked, [1] row steps with a splat, [2] candidates tested,
// [3] survivors parked, [4] walks started, [5] candidates tested again after a chunk did not fit
__device__ unsigned long long g_bwd_stats[8];
extern "C" int ngs_debug_bwd_stats(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bwd_stats), sizeof g_bwd_stats) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_stats), z, sizeof z) != hipSuccess) return 1;
    }
    return 0;
}
#define STAT_ADD(i, v)                                                                  \
    do {                                                                                \
        const unsigned long long v_ = (unsigned long long)(v);                          \
        if (lane == 0) atomicAdd(&g_bwd_stats[i], v_);                                  \
    } while (0)
#else
#define STAT_ADD(i, v) do { } while (0)
#endif

__global__ __launch_bounds__(NGS_TILE_PIX) void render_bwd_kernel(
    int W, int H, int grid_x, const uint2* __restrict__ ranges, const uint32_t* __restrict__ plist,
    const float4* __restrict__ splat_rec, const float* __restrict__ bg, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dpix, float* __restrict__ acc)
{
    // per wave, slot s: [0] = {x, y, list position bits, -}, [1] = {conic a, b, c, opacity}, [2] = {r, g, b, Gaussian id bits}
    __shared__ float4 s_rec[BWD_WAVES][BWD_SLOTS][3];
    __shared__ float s_acc[BWD_WAVES][BWD_SLOTS][ACC_USED];
    __shared__ uint8_t s_q[BWD_WAVES][4][BWD_QRING];  // per wave and quadrant: slots to visit, in depth order
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = lane >> 4, li = lane & 15;
    const int tile = blockIdx.x;
    const int tx = tile % grid_x, ty = tile / grid_x;
    const int px = tx * NGS_TILE_W + tile_local_x(tid), py = ty * NGS_TILE_H + tile_local_y(tid);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    // corner pixel of this wave's 8x8 block; quadrant k spans [bx0 + 4 (k & 1), +3] x [by0 + 4 (k >> 1), +3]
    const float bx0 = (float)(tx * NGS_TILE_W + (wave & 1) * NGS_WAVE_W);
    const float by0 = (float)(ty * NGS_TILE_H + (wave >> 1) * NGS_WAVE_H);
    const uint2 range = ranges[tile];

    const size_t tp = (size_t)tile * NGS_TILE_PIX + tid;
    const float T_final = inside ? final_T[tp] : 0.0f;
    float T = T_final;
    const uint32_t last_contributor = inside ? n_contrib[tp] : 0u;
    // deepest contributor of each quadrant (16-lane row) and of this block (wave)
    uint32_t qmax = last_contributor;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) qmax = max(qmax, (uint32_t)__shfl_xor((int)qmax, o, 64));
    const uint32_t qmax0 = (uint32_t)__builtin_amdgcn_readlane((int)qmax, 0);
    const uint32_t qmax1 = (uint32_t)__builtin_amdgcn_readlane((int)qmax, 16);
    const uint32_t qmax2 = (uint32_t)__builtin_amdgcn_readlane((int)qmax, 32);
    const uint32_t qmax3 = (uint32_t)__builtin_amdgcn_readlane((int)qmax, 48);
    const uint32_t wmax = max(max(qmax0, qmax1), max(qmax2, qmax3));
    if (wmax == 0) return;  // nothing in this block was blended (wave-uniform; the kernel has no barrier)

    float4(*const rec)[3] = s_rec[wave];
    float(*const accw)[ACC_USED] = s_acc[wave];
    uint8_t* const my_q = &s_q[wave][row][0];
    // free slots hold zero accumulators: zero all of them once, the flush re-zeroes what it has sent
    for (int i = lane; i < BWD_SLOTS * ACC_USED; i += 64) (&accw[0][0])[i] = 0.0f;

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
    v2f rec_rg = {0.0f, 0.0f};
    float rec_b = 0.0f;
    const v2f dp_rg = {dpr, dpg};
    const v2f ddel = {(float)(0.5 * W), (float)(0.5 * H)};  // d(pixel offset)/d(ndc)
    const v2f pix = {pxf, pyf};
    // After the reduce-scatter the even lanes of a row hold partials 0..7 (index bits taken from lane bits 3, 2, 1
    // in that order of significance 1, 2, 4) and the odd lanes partial 8; lane 1 speaks for the odd ones.
    const bool up8 = (li & 8) != 0, up4 = (li & 4) != 0, up2 = (li & 2) != 0, up1 = (li & 1) != 0;
    const int my_slot = !up1 ? ((up2 ? 4 : 0) + (up4 ? 2 : 0) + (up8 ? 1 : 0)) : (li == 1 ? 8 : -1);
    const unsigned long long below = (1ull << lane) - 1ull;
    // flush mapping: lane l sends component l % 9 of the (l / 9)-th slot of a group of seven
    const int fl_slot = lane / ACC_USED, fl_comp = lane - fl_slot * ACC_USED;

    // ---- wave-uniform bookkeeping ----
    // Zero-based list positions wmax-1 ... 0 are read in that order, 64 candidates at a time.
    int next_pos = (int)wmax - 1;          // list position of lane 0's candidate in the next chunk
    uint32_t head = 0;                     // slots handed out so far; slot number = count % BWD_SLOTS
    uint32_t flushed = 0;                  // every slot handed out before this count has been sent and is free again
    uint32_t prod0 = 0, prod1 = 0, prod2 = 0, prod3 = 0;  // entries appended to each quadrant's queue
    uint32_t cons0 = 0, cons1 = 0, cons2 = 0, cons3 = 0;  // entries each quadrant has walked

    // the chunk in flight: its list entries and records are requested one walk ahead of their use
    uint32_t nid = 0;
    float4 na = make_float4(0, 0, 0, 0), nb = na;
    float nc = 0.0f;
    int loaded_pos = next_pos;  // position the registers above were fetched for
    if (next_pos - lane >= 0) {
        nid = plist[range.x + (uint32_t)(next_pos - lane)];
        const float4* gr = splat_rec + 4 * (size_t)nid;  // one 64-byte line per splat
        na = gr[0];
        nb = gr[1];
        nc = gr[2].x;
    }

    for (;;) {
        // ---- send the accumulator rows of slots every queue has passed, and learn how many slots are free ----
        uint32_t a0 = prod0 - cons0, a1 = prod1 - cons1, a2 = prod2 - cons2, a3 = prod3 - cons3;
        {
            // oldest slot still waiting in a queue: the one farthest behind `head`
            const uint32_t my_a = row == 0 ? a0 : row == 1 ? a1 : row == 2 ? a2 : a3;
            const uint32_t my_c = row == 0 ? cons0 : row == 1 ? cons1 : row == 2 ? cons2 : cons3;
            uint32_t dist = 0;
            if (my_a != 0) {
                const uint32_t e = my_q[my_c & (BWD_QRING - 1)];
                dist = (head % BWD_SLOTS + BWD_SLOTS - e) % BWD_SLOTS;  // 1 .. BWD_SLOTS-1 for a slot in use
            }
            const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)dist, 0);
            const uint32_t d1 = (uint32_t)__builtin_amdgcn_readlane((int)dist, 16);
            const uint32_t d2 = (uint32_t)__builtin_amdgcn_readlane((int)dist, 32);
            const uint32_t d3 = (uint32_t)__builtin_amdgcn_readlane((int)dist, 48);
            const uint32_t tail = head - max(max(d0, d1), max(d2, d3));
            while (flushed != tail) {
                const uint32_t n = min(7u, tail - flushed);
                if ((uint32_t)fl_slot < n) {
                    const uint32_t s = (flushed + (uint32_t)fl_slot) % BWD_SLOTS;
                    const float v = accw[s][fl_comp];
                    if (v != 0.0f) {
                        atomicAdd(acc + (size_t)__float_as_uint(rec[s][2].w) * NGS_ACC_STRIDE + fl_comp, v);
                        accw[s][fl_comp] = 0.0f;
                    }
                }
                flushed += n;
            }
        }
        const bool input_left = next_pos >= 0;
        const uint32_t most = max(max(a0, a1), max(a2, a3));
        if (!input_left && most == 0) break;  // every queue is empty, all rows sent (flushed == head)

        // ---- test more candidates while some quadrant is short of work and the ring has room ----
        uint32_t free_slots = BWD_SLOTS - 1 - (head - flushed);
        while (next_pos >= 0 && free_slots >= (max(max(a0, a1), max(a2, a3)) == 0 ? 1u : (uint32_t)BWD_FREE_MIN) &&
               min(min(a0, a1), min(a2, a3)) < (uint32_t)BWD_RUN) {
            const int pos = next_pos - lane;
            if (loaded_pos != next_pos) {  // the previous chunk did not fit as a whole: fetch from where it stopped
                STAT_ADD(5, 64);
                nid = 0;
                na = make_float4(0, 0, 0, 0);
                nb = na;
                nc = 0.0f;
                if (pos >= 0) {
                    nid = plist[range.x + (uint32_t)pos];
                    const float4* gr = splat_rec + 4 * (size_t)nid;
                    na = gr[0];
                    nb = gr[1];
                    nc = gr[2].x;
                }
            }
            const uint32_t cid = nid;
            const float4 ca = na, cb = nb;
            const float cc = nc;
            unsigned bits = 0;
            if (pos >= 0) {
                const unsigned reach = splat_reaches_quadrants(ca.x, ca.y, ca.z, ca.w, cb.x, splat_tau(cb.y), bx0, by0);
                const uint32_t p = (uint32_t)pos;
                bits = (p < qmax0 ? reach & 1u : 0u) | (p < qmax1 ? reach & 2u : 0u) | (p < qmax2 ? reach & 4u : 0u) |
                       (p < qmax3 ? reach & 8u : 0u);
            }
            const unsigned long long surv = __ballot(bits != 0);
            const uint32_t rank = (uint32_t)__popcll(surv & below);
            const bool place = bits != 0 && rank < free_slots;
            const unsigned long long placed = __ballot(place);
            if (place) {
                const uint32_t s = (head + rank) % BWD_SLOTS;
                rec[s][0] = make_float4(ca.x, ca.y, __uint_as_float((uint32_t)pos), 0.0f);
                rec[s][1] = make_float4(ca.z, ca.w, cb.x, cb.y);
                rec[s][2] = make_float4(cb.z, cb.w, cc, __uint_as_float(cid));
            }
            const unsigned long long m0 = __ballot(place && (bits & 1u)), m1 = __ballot(place && (bits & 2u));
            const unsigned long long m2 = __ballot(place && (bits & 4u)), m3 = __ballot(place && (bits & 8u));
            if (place) {
                const uint8_t s = (uint8_t)((head + rank) % BWD_SLOTS);
                if (bits & 1u) s_q[wave][0][(prod0 + (uint32_t)__popcll(m0 & below)) & (BWD_QRING - 1)] = s;
                if (bits & 2u) s_q[wave][1][(prod1 + (uint32_t)__popcll(m1 & below)) & (BWD_QRING - 1)] = s;
                if (bits & 4u) s_q[wave][2][(prod2 + (uint32_t)__popcll(m2 & below)) & (BWD_QRING - 1)] = s;
                if (bits & 8u) s_q[wave][3][(prod3 + (uint32_t)__popcll(m3 & below)) & (BWD_QRING - 1)] = s;
            }
            prod0 += (uint32_t)__popcll(m0);
            prod1 += (uint32_t)__popcll(m1);
            prod2 += (uint32_t)__popcll(m2);
            prod3 += (uint32_t)__popcll(m3);
            const uint32_t n_placed = (uint32_t)__popcll(placed);
            head += n_placed;
            free_slots -= n_placed;
            STAT_ADD(3, n_placed);
            // the next chunk starts behind the last candidate dealt with: all 64 if every survivor found a slot,
            // else at the first survivor that did not
            const unsigned long long unplaced = surv & ~placed;
            const int done = unplaced ? (int)__builtin_ctzll(unplaced) : 64;
            STAT_ADD(2, done);
            next_pos -= done;
            a0 = prod0 - cons0; a1 = prod1 - cons1; a2 = prod2 - cons2; a3 = prod3 - cons3;
            if (done == 64) {  // the usual case: request the chunk after this one now, it is used a walk later
                loaded_pos = next_pos;
                const int np = next_pos - lane;
                if (np >= 0) {
                    nid = plist[range.x + (uint32_t)np];
                    const float4* gr = splat_rec + 4 * (size_t)nid;
                    na = gr[0];
                    nb = gr[1];
                    nc = gr[2].x;
                }
            } else {
                loaded_pos = -2;  // forces a fetch at the rewound position
            }
            if (unplaced) break;  // the ring is full
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- the four rows walk their queues side by side for a while ----
        const uint32_t my_a = row == 0 ? a0 : row == 1 ? a1 : row == 2 ? a2 : a3;
        const uint32_t my_c = row == 0 ? cons0 : row == 1 ? cons1 : row == 2 ? cons2 : cons3;
        const uint32_t longest = max(max(a0, a1), max(a2, a3));
        // while there is input left the walk stops after BWD_RUN steps so that short queues can be refilled; once
        // the list is exhausted the queues are simply drained
        const int steps = (int)(next_pos >= 0 ? min(longest, (uint32_t)BWD_RUN) : longest);
        const int my_n = (int)min(my_a, (uint32_t)steps);
        STAT_ADD(0, steps);
        STAT_ADD(1, min(a0, (uint32_t)steps) + min(a1, (uint32_t)steps) + min(a2, (uint32_t)steps) + min(a3, (uint32_t)steps));
        STAT_ADD(4, 1);
        // The records of the next step are requested before the arithmetic of the current one, so the LDS round
        // trips (queue entry, then record) overlap with it.
        int j_next = 0 < my_n ? (int)my_q[my_c & (BWD_QRING - 1)] : 0;
        float4 p_next = rec[j_next][0], q_next = rec[j_next][1], col_next = rec[j_next][2];
        int j_after = 1 < my_n ? (int)my_q[(my_c + 1) & (BWD_QRING - 1)] : 0;
        for (int it = 0; it < steps; it++) {
            const bool have = it < my_n;
            const int j = j_next;
            const float4 p = p_next, q = q_next, col = col_next;
            j_next = j_after;
            p_next = rec[j_next][0];
            q_next = rec[j_next][1];
            col_next = rec[j_next][2];
            j_after = it + 2 < my_n ? (int)my_q[(my_c + (uint32_t)it + 2) & (BWD_QRING - 1)] : 0;
            const uint32_t pos = __float_as_uint(p.z);  // the reference's `contributor` after its decrement
            const v2f d = (v2f){p.x, p.y} - pix;  // {dx, dy}
            const float dx = d.x, dy = d.y;
            // same evaluation order as the forward pass (and the reference): alpha must come out bit-identical
            const float power = -0.5f * (q.x * dx * dx + q.z * dy * dy) - q.y * dx * dy;
            const float G = exp_nonpos(power);
            const float alpha = fminf(0.99f, q.w * G);
            const bool live = have && pos < last_contributor && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
#ifdef NGS_BWD_STATS
            {
                const unsigned long long m = __ballot(live);
                const unsigned long long a2 = m | (m >> 1), b2 = (a2 | (a2 >> 4)) & 0x0505050505050505ull;
                STAT_ADD(6, __popcll(m));   // live lanes
                STAT_ADD(7, __popcll(b2));  // 2x2 pixel blocks with a live pixel
            }
#endif
            if (__ballot(live) == 0ull) continue;  // wave-uniform

            // Every lane evaluates the body.  A dead lane (pixel that does not blend this splat) runs it with
            // alpha = 0, which leaves its state untouched without a single select: T / (1 - 0) = T, the
            // colour behind it becomes 0 * colour + 1 * rec = rec, and every partial below is an exact zero
            // (G and dL/dalpha are forced to 0 as well).
            const float alpha_l = live ? alpha : 0.0f;
            const float om = 1.f - alpha_l;
            const float inv = __builtin_amdgcn_rcpf(om);  // v_rcp_f32, 1 ulp; exactly 1 for a dead lane
            const float Tn = T * inv;
            const float w = alpha_l * Tn;
            const v2f col_rg = {col.x, col.y};
            // `rec` is the colour accumulated behind this splat (the reference's accum_rec after its update,
            // backward.cu:503-513)
            const v2f d_rg = (col_rg - rec_rg) * dp_rg;
            float dL_dalpha = d_rg.x + d_rg.y;
            dL_dalpha += (col.z - rec_b) * dpb;
            dL_dalpha *= Tn;
            dL_dalpha += (-T_final * inv) * bgdot;
            dL_dalpha = live ? dL_dalpha : 0.0f;
            const float Gl = live ? G : 0.0f;
            const float dL_dG = q.w * dL_dalpha;
            const v2f gd = Gl * d;  // {G dx, G dy}
            const v2f gx = {gd.x, gd.x}, gy = {gd.y, gd.y};
            // {dG/ddelx, dG/ddely} = -gdx * {a, b} - gdy * {b, c}
            const v2f dG = -gx * (v2f){q.x, q.y} - gy * (v2f){q.y, q.z};
            const v2f v01 = (dL_dG * dG) * ddel;            // ACC_DX, ACC_DY
            const v2f h = -0.5f * gd;
            const v2f v23 = ((v2f){h.x, h.x} * d) * dL_dG;  // ACC_CON_A, ACC_CON_B
            const float v4 = h.y * dy * dL_dG;              // ACC_CON_C
            const float v5 = Gl * dL_dalpha;                // ACC_OPAC
            const v2f v67 = w * dp_rg;                      // ACC_R, ACC_G
            const float v8 = w * dpb;                       // ACC_B
            // state for the next (nearer) splat: transmittance in front of this one, and the colour behind
            // the next one = this splat blended over what was behind it (same expression the reference
            // evaluates one step later from its last_alpha / last_color)
            T = Tn;
            rec_rg = alpha_l * col_rg + om * rec_rg;
            rec_b = alpha_l * col.z + om * rec_b;

            // Sum over the quadrant's 16 pixels: four exchange steps, each halving the number of registers.
            const float a01 = row_fold<DPP_ROW_MIRROR>(v01.x, v01.y, up8);
            const float a23 = row_fold<DPP_ROW_MIRROR>(v23.x, v23.y, up8);
            const float a45 = row_fold<DPP_ROW_MIRROR>(v4, v5, up8);
            const float a67 = row_fold<DPP_ROW_MIRROR>(v67.x, v67.y, up8);
            const float a8 = row_pair_sum<DPP_ROW_MIRROR>(v8);
            const float b03 = row_fold<DPP_ROW_HALF_MIRROR>(a01, a23, up4);
            const float b47 = row_fold<DPP_ROW_HALF_MIRROR>(a45, a67, up4);
            const float b8 = row_pair_sum<DPP_ROW_HALF_MIRROR>(a8);
            const float c07 = row_fold<DPP_QUAD_XOR2>(b03, b47, up2);
            const float c8 = row_pair_sum<DPP_QUAD_XOR2>(b8);
            const float total = row_fold<DPP_QUAD_XOR1>(c07, c8, up1);
            if (have && my_slot >= 0 && total != 0.0f) atomicAdd(&accw[j][my_slot], total);
        }
        cons0 += min(a0, (uint32_t)steps);
        cons1 += min(a1, (uint32_t)steps);
        cons2 += min(a2, (uint32_t)steps);
        cons3 += min(a3, (uint32_t)steps);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

void ngs_launch_render_bwd(hipStream_t s, const NgsCamera& cam, const uint2* ranges, const uint32_t* plist,
                           const float4* rec, const float* bg, const float* final_T,
                           const uint32_t* n_contrib, const float* dL_dpix, float* acc)
{
    hipLaunchKernelGGL(render_bwd_kernel, dim3(cam.grid_x * cam.grid_y), dim3(NGS_TILE_PIX), 0, s, cam.W, cam.H,
                       cam.grid_x, ranges, plist, rec, bg, final_T, n_contrib, dL_dpix, acc);
}
