// Per-tile front-to-back alpha compositing (forward).  One 256-thread
// workgroup per 16x16 tile, four wave64s each covering a 8x8 pixel block.
//
// The four waves of a tile run DECOUPLED: there is no workgroup barrier in the
// kernel.  Each wave walks the tile's depth-sorted list on its own, 64 splats
// at a time: lane l fetches splat l of the batch (the next batch's loads are
// issued before the current one is processed), tests whether the splat can
// reach the wave's block at all (minimum of its quadratic form over the block against the opacity threshold, render_common.h),
// parks the 48-byte record in the wave's private LDS slice, and the wave then
// walks only the set bits of the wave-uniform ballot mask, four at a time, reading each record
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
    uint32_t* __restrict__ misc, unsigned long long* __restrict__ stage_clock, uint32_t seq)
{
    // per wave, record k: [0] = {x, y, -, -}, [1] = {conic a, b, c, opacity}, [2] = {r, g, b, -}; 64-byte slots, so
    // the address of slot k is a shift-and-add (a 48-byte stride costs a quarter-rate 64-bit multiply-add per visit)
    __shared__ float4 s_rec[NGS_TILE_PIX / 64][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    // This workgroup's share of the backward pass's accumulator rows is cleared here: the stores leave at once and
    // complete behind the compositing, which keeps the vector ALU busy and the memory system idle.
    {
        const int lo = tile * acc_f4_per_block, hi = min(acc_f4_total, lo + acc_f4_per_block);
        for (int i = lo + tid; i < hi; i += NGS_TILE_PIX) stream_store4(acc + 4 * (size_t)i, make_float4(0.f, 0.f, 0.f, 0.f));
        if (tile == 0 && tid == 0) {
            misc[NGS_MISC_ACC_CLEAN] = 1u;
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
    // this wave's pixel block
    const float sx0 = (float)(tx * NGS_TILE_W + (wave & 1) * NGS_WAVE_W), sx1 = sx0 + (NGS_WAVE_W - 1);
    const float sy0 = (float)(ty * NGS_TILE_H + (wave >> 1) * NGS_WAVE_H), sy1 = sy0 + (NGS_WAVE_H - 1);
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    float4(*rec)[4] = s_rec[wave];

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
        if (wave_mask(!done) == 0ull) break;  // every pixel of the block is saturated (or outside the image)
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
        bool reach = false;
        if (have) {
            reach = splat_reaches(a.x, a.y, a.z, a.w, b.x, splat_tau(b.y), sx0, sx1, sy0, sy1);
            if (reach) {
                rec[lane][0] = make_float4(a.x, a.y, 0.0f, 0.0f);
                rec[lane][1] = make_float4(a.z, a.w, b.x, b.y);
                rec[lane][2] = make_float4(b.z, b.w, c, 0.0f);
            }
        }
        unsigned long long mask = wave_mask(reach);
        // One splat against the wave's 64 pixels.
        auto visit = [&](const int j, const float4 p, const float4 q, const float4 col) {
            const float dx = p.x - pxf, dy = p.y - pyf;
            const float power = -0.5f * (q.x * dx * dx + q.z * dy * dy) - q.y * dx * dy;
            const float alpha = fminf(0.99f, q.w * exp_nonpos(power));
            const float test_T = T * (1 - alpha);
            const bool hit = !done && power <= 0.0f && !(alpha < 1.0f / 255.0f);
            const bool blend = hit && !(test_T < 0.0001f);
            const bool stop = hit && !blend;  // from the masks: one comparison instead of two
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
        // Four surviving splats per trip: all records are requested before the first is used, so their LDS latency
        // passes behind the arithmetic, and the loop's scalar bookkeeping is paid once per four.
        while (mask) {
            int j[4];
            bool on[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                on[k] = mask != 0ull;
                j[k] = on[k] ? lowest_lane(mask) : j[k ? k - 1 : 0];
                mask &= mask - 1ull;  // 0 stays 0
            }
            float4 p[4], q[4], c[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                p[k] = rec[j[k]][0]; q[k] = rec[j[k]][1]; c[k] = rec[j[k]][2];
                asm volatile("" ::"v"(c[k].w));  // 16-byte LDS reads (a 12-byte one costs twice the LDS cycles)
            }
            visit(j[0], p[0], q[0], c[0]);
            if (on[1]) visit(j[1], p[1], q[1], c[1]);
            if (on[2]) visit(j[2], p[2], q[2], c[2]);
            if (on[3]) visit(j[3], p[3], q[3], c[3]);
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
                           uint32_t* n_contrib, float* out_color, float* acc, int P, uint32_t* misc,
                           unsigned long long* stage_clock, uint32_t seq)
{
    const int T = cam.grid_x * cam.grid_y;
    const long long f4 = (long long)P * (NGS_ACC_STRIDE / 4);  // P <= NGS_MAX_GAUSSIANS keeps this inside an int
    hipLaunchKernelGGL(render_fwd_kernel, dim3(T), dim3(NGS_TILE_PIX), 0, s, cam.W, cam.H, cam.grid_x, ranges, plist,
                       rec, bg, final_T, n_contrib, out_color, acc, (int)((f4 + T - 1) / T), (int)f4, misc, stage_clock, seq);
}
