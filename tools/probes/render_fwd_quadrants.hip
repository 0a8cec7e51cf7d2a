// EXPERIMENT, not part of the library: the forward compositing kernel with the backward kernel's per-quadrant walk
// (every 16-lane row has its own list of the 64-splat batch's survivors).  Bit-identical images and contributor
// counts, 0.366 ms at 3 M Gaussians / 1080p against 0.364 for the block-wide walk with two splats per trip: with 64
// splats per batch a row walks about a dozen entries per batch, and building four lists per batch costs what the
// shorter walk saves.  (A record of that experiment: it is written against the library's headers and launch signature
// as they were then.  The library's kernel now walks per quadrant too, after parking survivors across batches and
// classifying 64 of them at a time, which is what makes the four lists pay -- see csrc/render_fwd.hip.)
// Per-tile front-to-back alpha compositing (forward).  One 256-thread
// workgroup per 16x16 tile, four wave64s each covering a 8x8 pixel block.
//
// The four waves of a tile run DECOUPLED: there is no workgroup barrier in the
// kernel.  Each wave walks the tile's depth-sorted list on its own, 64 splats
// at a time: lane l fetches splat l of the batch (the next batch's loads are
// issued before the current one is processed), tests whether the splat can
// reach the wave's block at all (minimum of its quadratic form over the block against the opacity threshold, render_common.h),
// parks the 48-byte record in the wave's private LDS slice, and the wave then
// walks only the set bits of the wave-uniform ballot mask, reading each record
// back as three broadcast LDS loads.  A wave whose 64 pixels are all saturated
// simply leaves; nobody waits for the slowest block of the tile.
//
// Per-pixel arithmetic and its order follow cuda_rasterizer/forward.cu:261-374
// of the reference; the per-pixel body is branch-free (predicated).  Skipping a
// splat culled this way cannot change any pixel because every pixel
// of the block would have failed the alpha test.
#include "render_common.h"

__global__ __launch_bounds__(NGS_TILE_PIX) void render_fwd_kernel(
    int W, int H, int grid_x, const uint2* __restrict__ ranges, const uint32_t* __restrict__ plist,
    const float4* __restrict__ rec_g, const float* __restrict__ bg, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
    float* __restrict__ out_color, float* __restrict__ acc, int acc_f4_per_block, int acc_f4_total,
    uint32_t* __restrict__ misc)
{
    // per wave, record k: [0] = {x, y, -, -}, [1] = {conic a, b, c, opacity}, [2] = {r, g, b, -}; 64-byte slots, so
    // the address of slot k is a shift-and-add
    __shared__ float4 s_rec[NGS_TILE_PIX / 64][64][4];
    // per wave and quadrant: the slots of the batch the quadrant has to visit, in list order (a few spare bytes: the
    // walk reads entries up to three steps ahead of knowing whether they exist)
    __shared__ uint8_t s_list[NGS_TILE_PIX / 64][4][64 + 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = lane >> 4;
    const int tile = blockIdx.x;
    // This workgroup's share of the backward pass's accumulator rows is cleared here: the stores leave at once and
    // complete behind the compositing, which keeps the vector ALU busy and the memory system idle.
    {
        const int lo = tile * acc_f4_per_block, hi = min(acc_f4_total, lo + acc_f4_per_block);
        for (int i = lo + tid; i < hi; i += NGS_TILE_PIX) stream_store4(acc + 4 * (size_t)i, make_float4(0.f, 0.f, 0.f, 0.f));
        if (tile == 0 && tid == 0) misc[NGS_MISC_ACC_CLEAN] = 1u;
    }
    const int tx = tile % grid_x, ty = tile / grid_x;
    const int px = tx * NGS_TILE_W + tile_local_x(tid), py = ty * NGS_TILE_H + tile_local_y(tid);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    // corner pixel of this wave's 8x8 block; quadrant k spans [bx0 + 4 (k & 1), +3] x [by0 + 4 (k >> 1), +3]
    const float bx0 = (float)(tx * NGS_TILE_W + (wave & 1) * NGS_WAVE_W);
    const float by0 = (float)(ty * NGS_TILE_H + (wave >> 1) * NGS_WAVE_H);
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    float4(*rec)[4] = s_rec[wave];
    // The walk reads list entries, and the records they name, ahead of knowing whether they exist: whatever it can
    // come across must be a valid slot and a finite record (a row without an entry blends with alpha = 0, which
    // leaves its colour alone only if the record's colour is finite).  Wave-private LDS: no workgroup barrier.
    for (int k = lane; k < 64 * 4; k += 64) (&rec[0][0])[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = lane; k < (int)sizeof(s_list[0]) / 4; k += 64) ((uint32_t*)&s_list[wave][0][0])[k] = 0u;
    const uint8_t* const my_list = &s_list[wave][row][0];

    bool done = !inside;
    float T = 1.0f, Cr = 0.0f, Cg = 0.0f, Cb = 0.0f;
    uint32_t last_contributor = 0;

    // registers of the batch in flight
    float4 na = make_float4(0, 0, 0, 0), nb = na;
    float nc = 0.0f;
    if (lane < n) {
        const float4* r = rec_g + 4 * (size_t)plist[range.x + lane];  // one 64-byte line per splat
        na = r[0];
        nb = r[1];
        nc = r[2].x;
    }
    for (int base = 0; base < n; base += 64) {
        // which quadrants (16-lane rows) still have an unsaturated pixel?
        const unsigned long long open = wave_mask(!done);
        if (open == 0ull) break;  // every pixel of the block is saturated (or outside the image)
        const float4 a = na, b = nb;
        const float c = nc;
        const bool have = base + lane < n;
        const int nxt = base + 64 + lane;
        if (nxt < n) {
            const float4* r = rec_g + 4 * (size_t)plist[range.x + nxt];
            na = r[0];
            nb = r[1];
            nc = r[2].x;
        }
        // Every 16-lane row (one 4x4 quadrant) gets its own list of the batch's splats that can reach it: a splat of
        // this workload reaches a quadrant or two of the block, so rows that walked the block's list together would
        // spend most of their steps on splats that do not concern them.
        unsigned bits = 0;
        if (have) {
            bits = splat_reaches_quadrants(a.x, a.y, a.z, a.w, b.x, splat_tau(b.y), bx0, by0);
            if (bits) {
                rec[lane][0] = make_float4(a.x, a.y, 0.0f, 0.0f);
                rec[lane][1] = make_float4(a.z, a.w, b.x, b.y);
                rec[lane][2] = make_float4(b.z, b.w, c, 0.0f);
            }
        }
        const bool r0 = (bits & 1u) && (open & 0xffffull), r1 = (bits & 2u) && (open & 0xffff0000ull);
        const bool r2 = (bits & 4u) && (open & 0xffff00000000ull), r3 = (bits & 8u) && (open & 0xffff000000000000ull);
        const unsigned long long m0 = wave_mask(r0), m1 = wave_mask(r1), m2 = wave_mask(r2), m3 = wave_mask(r3);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (r0) s_list[wave][0][__popcll(m0 & below)] = (uint8_t)lane;
        if (r1) s_list[wave][1][__popcll(m1 & below)] = (uint8_t)lane;
        if (r2) s_list[wave][2][__popcll(m2 & below)] = (uint8_t)lane;
        if (r3) s_list[wave][3][__popcll(m3 & below)] = (uint8_t)lane;
        const int n0 = __popcll(m0), n1 = __popcll(m1), n2 = __popcll(m2), n3 = __popcll(m3);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int my_n = row == 0 ? n0 : row == 1 ? n1 : row == 2 ? n2 : n3;
        const int steps = max(max(n0, n1), max(n2, n3));

        // One step of a row: splat j of the batch (record p, q, col) against the row's 16 pixels; arithmetic and
        // its order as forward.cu:327-359.
        auto step = [&](const int it, const int j, const float4 p, const float4 q, const float4 col) {
            const float dx = p.x - pxf, dy = p.y - pyf;
            const float power = -0.5f * (q.x * dx * dx + q.z * dy * dy) - q.y * dx * dy;
            const float alpha = fminf(0.99f, q.w * exp_nonpos(power));
            const float test_T = T * (1 - alpha);
            const bool hit = it < my_n && !done && power <= 0.0f && !(alpha < 1.0f / 255.0f);
            const bool stop = hit && test_T < 0.0001f;
            const bool blend = hit && !stop;
            done = done || stop;
            // (colour * alpha) * T added with one rounding, as the reference's build contracts the same expression
            // (forward.cu:355-356 compiled with nvcc's default -fmad=true).  A pixel that does not blend the splat
            // adds (colour * 0) * T, an exact zero, instead of selecting its old value three times.
            const float alpha_b = blend ? alpha : 0.0f;
            Cr = fmaf(col.x * alpha_b, T, Cr);
            Cg = fmaf(col.y * alpha_b, T, Cg);
            Cb = fmaf(col.z * alpha_b, T, Cb);
            T = blend ? test_T : T;
            last_contributor = blend ? (uint32_t)(base + j + 1) : last_contributor;
        };
        // Two steps per trip: both records are requested before the first is used.
        for (int it = 0; it < steps; it += 2) {
            const int j0 = my_list[it], j1 = my_list[it + 1];
            const float4 p0 = rec[j0][0], q0 = rec[j0][1], c0 = rec[j0][2];
            const float4 p1 = rec[j1][0], q1 = rec[j1][1], c1 = rec[j1][2];
            asm volatile("" ::"v"(c0.w), "v"(c1.w));  // 16-byte LDS reads (a 12-byte one costs twice the LDS cycles)
            step(it, j0, p0, q0, c0);
            step(it + 1, j1, p1, q1, c1);
            // has the whole block saturated?  (a saturated pixel ignores further splats anyway)
            if (wave_mask(!done) == 0ull) break;
        }
    }
    if (inside) {
        const size_t tp = (size_t)tile * NGS_TILE_PIX + tid;
        final_T[tp] = T;
        n_contrib[tp] = last_contributor;
        const size_t pid = (size_t)W * py + px, plane = (size_t)W * H;
        out_color[pid] = Cr + T * bg[0];
        out_color[plane + pid] = Cg + T * bg[1];
        out_color[2 * plane + pid] = Cb + T * bg[2];
    }
}

void ngs_launch_render_fwd(hipStream_t s, const NgsCamera& cam, const uint2* ranges, const uint32_t* plist,
                           const float4* rec, const float* bg, float* final_T,
                           uint32_t* n_contrib, float* out_color, float* acc, int P, uint32_t* misc)
{
    const int T = cam.grid_x * cam.grid_y;
    const long long f4 = (long long)P * (NGS_ACC_STRIDE / 4);  // P <= NGS_MAX_GAUSSIANS keeps this inside an int
    hipLaunchKernelGGL(render_fwd_kernel, dim3(T), dim3(NGS_TILE_PIX), 0, s, cam.W, cam.H, cam.grid_x, ranges, plist,
                       rec, bg, final_T, n_contrib, out_color, acc, (int)((f4 + T - 1) / T), (int)f4, misc);
}
