// Per-tile front-to-back alpha compositing (forward).  One 256-thread
// workgroup per 16x16 tile, four wave64s each covering a 16x4 pixel strip.
// Batches of 256 depth-sorted splats are staged into LDS (position, conic,
// opacity AND colour, so the inner loop never touches global memory) and every
// lane walks the batch for its own pixel.  Per-pixel arithmetic and its order
// follow cuda_rasterizer/forward.cu:261-374 of the reference.
#include "ngs_common.h"

__global__ __launch_bounds__(NGS_TILE_PIX) void render_fwd_kernel(
    int W, int H, int grid_x, const uint2* __restrict__ ranges, const uint32_t* __restrict__ plist,
    const float4* __restrict__ g0, const float4* __restrict__ g1, const float* __restrict__ g2,
    const float* __restrict__ bg, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
    float* __restrict__ out_color)
{
    __shared__ float4 s_a[NGS_TILE_PIX];  // x, y, conic a, conic b
    __shared__ float4 s_b[NGS_TILE_PIX];  // conic c, opacity, r, g
    __shared__ float s_c[NGS_TILE_PIX];   // b
    const int tid = threadIdx.x;
    const int tile = blockIdx.x;
    const int tx = tile % grid_x, ty = tile / grid_x;
    const int px = tx * NGS_TILE_W + (tid & 15), py = ty * NGS_TILE_H + (tid >> 4);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];
    int todo = (int)(range.y - range.x);
    const int rounds = (todo + NGS_TILE_PIX - 1) / NGS_TILE_PIX;

    bool done = !inside;
    float T = 1.0f, Cr = 0.0f, Cg = 0.0f, Cb = 0.0f;
    uint32_t contributor = 0, last_contributor = 0;

    for (int r = 0; r < rounds; r++, todo -= NGS_TILE_PIX) {
        if (__syncthreads_count(done) == NGS_TILE_PIX) break;
        const uint32_t src = range.x + r * NGS_TILE_PIX + tid;
        if (src < range.y) {
            const uint32_t id = plist[src];
            s_a[tid] = g0[id];
            s_b[tid] = g1[id];
            s_c[tid] = g2[id];
        }
        __syncthreads();
        const int nb = min(NGS_TILE_PIX, todo);
        for (int j = 0; !done && j < nb; j++) {
            contributor++;
            const float4 a = s_a[j];
            const float4 b = s_b[j];
            const float dx = a.x - pxf, dy = a.y - pyf;
            const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
            if (power > 0.0f) continue;
            const float alpha = fminf(0.99f, b.y * expf(power));
            if (alpha < 1.0f / 255.0f) continue;
            const float test_T = T * (1 - alpha);
            if (test_T < 0.0001f) {
                done = true;
                continue;
            }
            const float w = alpha * T;
            Cr += b.z * alpha * T;
            Cg += b.w * alpha * T;
            Cb += s_c[j] * alpha * T;
            (void)w;
            T = test_T;
            last_contributor = contributor;
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
