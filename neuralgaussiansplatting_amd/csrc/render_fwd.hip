// Per-tile front-to-back alpha compositing (forward).  One 256-thread
// workgroup per 16x16 tile, four wave64s each covering a 16x4 pixel strip.
// Batches of 256 depth-sorted splats are staged into LDS as 48-byte records
// (position + reach, conic + opacity, colour), so the inner loop never touches
// global memory.  Each wave then ballots, 64 splats at a time, which staged
// splats can reach its strip at all (opacity-aware bounding box,
// render_common.h) and walks only the set bits of that wave-uniform mask; the
// other splats cost it nothing.  The per-pixel body is branch-free: one group
// of three broadcast LDS reads, then predicated arithmetic.
// Per-pixel arithmetic and its order follow cuda_rasterizer/forward.cu:261-374
// of the reference; skipping a splat whose box misses the strip cannot change
// any pixel because every pixel of the strip would have failed the alpha test.
#include "render_common.h"

__global__ __launch_bounds__(NGS_TILE_PIX) void render_fwd_kernel(
    int W, int H, int grid_x, const uint2* __restrict__ ranges, const uint32_t* __restrict__ plist,
    const float4* __restrict__ g0, const float4* __restrict__ g1, const float* __restrict__ g2,
    const float* __restrict__ bg, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
    float* __restrict__ out_color)
{
    // record k: [0] = {x, y, reach x, reach y}, [1] = {conic a, b, c, opacity}, [2] = {r, g, b, -}
    __shared__ float4 s_rec[NGS_TILE_PIX][3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const int tx = tile % grid_x, ty = tile / grid_x;
    const int px = tx * NGS_TILE_W + (tid & 15), py = ty * NGS_TILE_H + (tid >> 4);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    // this wave's pixel strip
    const float sx0 = (float)(tx * NGS_TILE_W), sx1 = sx0 + (NGS_TILE_W - 1);
    const float sy0 = (float)(ty * NGS_TILE_H + wave * 4), sy1 = sy0 + 3.0f;
    const uint2 range = ranges[tile];
    int todo = (int)(range.y - range.x);
    const int rounds = (todo + NGS_TILE_PIX - 1) / NGS_TILE_PIX;

    bool done = !inside;
    float T = 1.0f, Cr = 0.0f, Cg = 0.0f, Cb = 0.0f;
    uint32_t last_contributor = 0;

    for (int r = 0; r < rounds; r++, todo -= NGS_TILE_PIX) {
        if (__syncthreads_count(done) == NGS_TILE_PIX) break;
        const uint32_t src = range.x + r * NGS_TILE_PIX + tid;
        if (src < range.y) {
            const uint32_t id = plist[src];
            const float4 a = g0[id];
            const float4 b = g1[id];
            const float c = g2[id];
            const float2 h = splat_half_extent(a.z, a.w, b.x, b.y);
            s_rec[tid][0] = make_float4(a.x, a.y, h.x, h.y);
            s_rec[tid][1] = make_float4(a.z, a.w, b.x, b.y);
            s_rec[tid][2] = make_float4(b.z, b.w, c, 0.0f);
        }
        __syncthreads();
        const int nb = min(NGS_TILE_PIX, todo);
        if (__ballot(!done) == 0ull) continue;  // this strip is finished; keep helping with the staging
        const uint32_t base = (uint32_t)(r * NGS_TILE_PIX + 1);
        for (int chunk = 0; chunk < nb; chunk += 64) {
            const int mine = chunk + lane;
            bool reach = false;
            if (mine < nb) {
                const float4 p = s_rec[mine][0];
                reach = splat_touches(p.x, p.y, make_float2(p.z, p.w), sx0, sx1, sy0, sy1);
            }
            unsigned long long mask = __ballot(reach);
            while (mask) {
                const int j = chunk + lowest_lane(mask);
                mask &= mask - 1ull;
                const float4 p = s_rec[j][0];
                const float4 q = s_rec[j][1];
                const float4 c = s_rec[j][2];
                const float dx = p.x - pxf, dy = p.y - pyf;
                const float power = -0.5f * (q.x * dx * dx + q.z * dy * dy) - q.y * dx * dy;
                const float alpha = fminf(0.99f, q.w * expf(power));
                const float test_T = T * (1 - alpha);
                const bool hit = !done && power <= 0.0f && !(alpha < 1.0f / 255.0f);
                const bool stop = hit && test_T < 0.0001f;
                const bool blend = hit && !stop;
                done = done || stop;
                Cr = blend ? Cr + c.x * alpha * T : Cr;
                Cg = blend ? Cg + c.y * alpha * T : Cg;
                Cb = blend ? Cb + c.z * alpha * T : Cb;
                T = blend ? test_T : T;
                last_contributor = blend ? base + (uint32_t)j : last_contributor;
                if (__ballot(!done) == 0ull) break;
            }
            if (__ballot(!done) == 0ull) break;
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
                           const float4* g0, const float4* g1, const float* g2, const float* bg, float* final_T,
                           uint32_t* n_contrib, float* out_color)
{
    hipLaunchKernelGGL(render_fwd_kernel, dim3(cam.grid_x * cam.grid_y), dim3(NGS_TILE_PIX), 0, s, cam.W, cam.H,
                       cam.grid_x, ranges, plist, g0, g1, g2, bg, final_T, n_contrib, out_color);
}
