// Index-map path of the reference's second rasterizer (r2 = submodules/rasterizer2): every Gaussian centre is
// splatted as a screen-space square of half-width 3/depth pixels and each pixel keeps the NEAREST centre; the
// outputs are the winner's index, its view direction (raw and as a 24-value positional encoding), its depth and
// features 25..63 of its 64-value feature vector (r2/cuda_rasterizer/rasterizer_impl.cu:851-984).
//
// The reference gets there with a prefix scan, one (pixel | depth) key per covered pixel, a 63-bit radix sort of
// all of them, a range search and a gather (rasterizer_impl.cu:914-981).  The first entry of a pixel's sorted range
// is the entry with the smallest depth bits, ties going to the lower Gaussian index (stable sort, entries emitted
// in index order) -- which is the minimum of the 64-bit word (depth bits << 32 | index).  So here each covered
// pixel takes one 64-bit atomic minimum into a W x H buffer and a second kernel resolves the winners: no scan,
// no keys, no sort, no ranges, and the only scratch is 8 bytes per pixel.
#include "ngs_common.h"

#define IDX_FEATURES 64      // r2/cuda_rasterizer/config.h:3
#define IDX_FREQS 4          // r2/cuda_rasterizer/auxiliary.h:33
#define IDX_FIRST_COPIED 25  // features 25..63 come from the Gaussian (rasterizer_impl.cu:797-799)
#define IDX_PI 3.14159265358979323846f

// r2/cuda_rasterizer/auxiliary.h:38-41 (double arithmetic, rounded once to float)
__device__ __forceinline__ float idx_ndc_to_pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

// Per-Gaussian stage (r2/cuda_rasterizer/raster.cu:10-66): is the centre in front of the near plane and inside the
// image, and which pixel square [x0,x1) x [y0,y1) does it cover.
struct IdxSquare {
    int visible, x0, y0, x1, y1;
    float depth;
};

__device__ __forceinline__ IdxSquare idx_square(float px, float py, float pz, const float* __restrict__ view,
                                                const float* __restrict__ proj, int W, int H)
{
    IdxSquare q = {0, 0, 0, 0, 0, 0.0f};
    const float4 ph = xform44(px, py, pz, proj);
    const float pw = 1.0f / (ph.w + 0.0000001f);
    const float projx = ph.x * pw, projy = ph.y * pw;
    const float3 pv = xform43(px, py, pz, view);
    if (pv.z > 0.2f) {
        const float fx = idx_ndc_to_pix(projx, W), fy = idx_ndc_to_pix(projy, H);
        const int cx = (int)fx, cy = (int)fy;  // v_cvt_i32_f32: truncates like the reference's conversion
        const float rad = 3.0f / pv.z;         // raster.cu:46,85
        q.x0 = (int)fmaxf(0.0f, fx - rad);
        q.y0 = (int)fmaxf(0.0f, fy - rad);
        q.x1 = (int)fminf((float)W, fx + rad + 1.0f);
        q.y1 = (int)fminf((float)H, fy + rad + 1.0f);
        q.depth = pv.z;
        q.visible = !(cx < 0 || cx >= W || cy < 0 || cy >= H);
    }
    return q;
}

// The per-Gaussian stage fused with the "key emission": one lane per Gaussian.
__global__ __launch_bounds__(NGS_GAUSS_BLOCK) void idxmap_splat_kernel(int P, int W, int H,
                                                                      const float* __restrict__ means,
                                                                      const float* __restrict__ view,
                                                                      const float* __restrict__ proj,
                                                                      unsigned long long* __restrict__ zbuf,
                                                                      int* __restrict__ radii,
                                                                      uint32_t* __restrict__ counter)
{
    const int idx = blockIdx.x * NGS_GAUSS_BLOCK + threadIdx.x;
    uint32_t touched = 0;
    if (idx < P) {
        const IdxSquare q = idx_square(means[3 * idx], means[3 * idx + 1], means[3 * idx + 2], view, proj, W, H);
        if (q.visible) {
            touched = (uint32_t)((q.x1 - q.x0) * (q.y1 - q.y0));
            const unsigned long long key = ((unsigned long long)__float_as_uint(q.depth) << 32) | (uint32_t)idx;
            for (int y = q.y0; y < q.y1; y++)
                for (int x = q.x0; x < q.x1; x++) {
                    unsigned long long* cell = zbuf + (size_t)y * W + x;
                    // values only ever decrease, so a cell already below the key can be skipped unlocked
                    if (key < *(volatile unsigned long long*)cell) atomicMin(cell, key);
                }
        }
        if (radii) radii[idx] = q.visible;
    }
    // Entry count (the reference's return value), only for a caller that asks: 64 counters share the additions,
    // since tens of thousands of waves adding to one address serialise at the memory side.
    if (counter) {
        const uint32_t ws = wave_sum_u32(touched);
        if ((threadIdx.x & 63) == 0 && ws) atomicAdd(counter + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & 63), ws);
    }
}

// 16 lanes per pixel share the pixel's 64-float feature row so that it leaves as one contiguous 256-byte run, and
// the work is spread evenly: lanes 0..11 each evaluate ONE sine/cosine pair of the positional encoding (the
// expensive part), lanes 7..15 each copy one 16-byte group of the winner's features 28..63, lane 6 copies features
// 25..27 and lane 15 writes the per-pixel scalars.  Every element of every output is written (zeros where no centre
// landed).
__global__ __launch_bounds__(256) void idxmap_resolve_kernel(int W, int H, const unsigned long long* __restrict__ zbuf,
                                                             const float* __restrict__ means,
                                                             const float* __restrict__ campos,
                                                             const float* __restrict__ feature_vector,
                                                             int* __restrict__ out_idx, float* __restrict__ out_color,
                                                             float* __restrict__ out_depth,
                                                             float* __restrict__ out_feature)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t pix = t >> 4;
    const int c = (int)(t & 15);
    if (pix >= (size_t)W * H) return;
    const unsigned long long cell = zbuf[pix];
    float* row = out_feature + pix * IDX_FEATURES;
    if (cell == ~0ull) {
        *(float4*)(row + 4 * c) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (c == 15) {
            out_idx[pix] = -1;
            out_color[3 * pix] = 0.0f; out_color[3 * pix + 1] = 0.0f; out_color[3 * pix + 2] = 0.0f;
            out_depth[pix] = 0.0f;
        }
        return;
    }
    const uint32_t g = (uint32_t)cell;
    const float depth = __uint_as_float((uint32_t)(cell >> 32));
    // view direction of the winner (rasterizer_impl.cu:768-771)
    float3 d = make_float3(means[3 * (size_t)g] - campos[0], means[3 * (size_t)g + 1] - campos[1],
                           means[3 * (size_t)g + 2] - campos[2]);
    const float len = sqrtf(f3_dot(d, d));
    d = make_float3(d.x / len, d.y / len, d.z / len);
    const float* fv = feature_vector + (size_t)g * IDX_FEATURES;
    if (c < 3 * IDX_FREQS) {
        // positional encoding (rasterizer_impl.cu:26-42): pair (dim, freq) -> elements 1 + dim*8 + freq*2, +1
        const int dim = c >> 2, fr = c & 3;
        const float x = dim == 0 ? d.x : dim == 1 ? d.y : d.z;
        const float arg = (float)(1 << fr) * IDX_PI * x;
        float sn, cs;
        sincosf(arg, &sn, &cs);
        row[1 + 2 * c] = sn;
        row[2 + 2 * c] = cs;
    }
    if (c >= 7) {
        *(float4*)(row + 4 * c) = *(const float4*)(fv + 4 * c);
    } else if (c == 6) {
        row[25] = fv[25]; row[26] = fv[26]; row[27] = fv[27];
    }
    if (c == 15) {
        row[0] = depth;
        out_idx[pix] = (int)g;
        out_color[3 * pix] = d.x; out_color[3 * pix + 1] = d.y; out_color[3 * pix + 2] = d.z;
        out_depth[pix] = depth;
    }
}

// Backward (rasterizer_impl.cu:820-847): the gradient of feature-map elements 25..63 of every pixel flows to the
// Gaussian the pixel shows.  Same 16-lanes-per-pixel mapping; float atomics make the sum well defined (the
// reference adds from concurrent threads without them).  dL_dfeature_vector is zeroed by the entry point first.
__global__ __launch_bounds__(256) void idxmap_backward_kernel(int P, int W, int H, const int* __restrict__ out_idx,
                                                              const float* __restrict__ dL_dfeature,
                                                              float* __restrict__ dL_dfeature_vector)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t pix = t >> 4;
    const int c = (int)(t & 15);
    if (pix >= (size_t)W * H || c < IDX_FIRST_COPIED / 4) return;
    const int g = out_idx[pix];
    if (g < 0 || g >= P) return;
    const float4 v = *(const float4*)(dL_dfeature + pix * IDX_FEATURES + 4 * c);
    float* dst = dL_dfeature_vector + (size_t)g * IDX_FEATURES + 4 * c;
    if (4 * c >= IDX_FIRST_COPIED) atomicAdd(dst, v.x);
    atomicAdd(dst + 1, v.y);
    atomicAdd(dst + 2, v.z);
    atomicAdd(dst + 3, v.w);
}

// The same sum as a gather, for callers that can supply the camera matrices: 16 lanes per Gaussian rebuild its
// square, visit its pixels in row-major order and add the rows of the pixels that show it.  No atomics, the
// additions happen in pixel order (the order of a sequential loop over the image), and every element of the
// Gaussian's output row is written, zeros included.
__global__ __launch_bounds__(256) void idxmap_backward_gather_kernel(int P, int W, int H,
                                                                     const float* __restrict__ means,
                                                                     const float* __restrict__ view,
                                                                     const float* __restrict__ proj,
                                                                     const int* __restrict__ out_idx,
                                                                     const float* __restrict__ dL_dfeature,
                                                                     float* __restrict__ dL_dfeature_vector)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t g = t >> 4;
    const int c = (int)(t & 15);
    if (g >= (size_t)P) return;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const IdxSquare q = idx_square(means[3 * g], means[3 * g + 1], means[3 * g + 2], view, proj, W, H);
    if (q.visible) {
        // The 16 lanes of a Gaussian look at 16 pixels of its square at a time (one round trip to the index map
        // instead of one per pixel), then add the rows of the pixels that matched, in pixel order.
        const int w = q.x1 - q.x0, npx = w * (q.y1 - q.y0);
        const int shift = (int)(threadIdx.x & 48);  // position of this Gaussian's 16 lanes in the wave's ballot
        for (int base = 0; base < npx; base += 16) {
            const int k = base + c;
            bool match = false;
            if (k < npx) {
                const int dy = k / w;
                match = out_idx[(size_t)(q.y0 + dy) * W + (q.x0 + (k - dy * w))] == (int)g;
            }
            unsigned hits = (unsigned)(__ballot(match) >> shift) & 0xffffu;
            while (hits) {
                const int kk = base + __builtin_ctz(hits);
                hits &= hits - 1u;
                const int dy = kk / w;
                const size_t pix = (size_t)(q.y0 + dy) * W + (q.x0 + (kk - dy * w));
                if (c >= IDX_FIRST_COPIED / 4) {
                    const float4 v = *(const float4*)(dL_dfeature + pix * IDX_FEATURES + 4 * c);
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
            }
        }
        if (4 * c < IDX_FIRST_COPIED) acc.x = 0.0f;  // element 24 belongs to the positional encoding
    }
    stream_store4(dL_dfeature_vector + g * IDX_FEATURES + 4 * c, acc);
}

size_t ngs_idxmap_workspace_size(int W, int H) { return (size_t)W * H * 8 + 256; }

// Sums the 64 partial entry counts into one device word (for callers that must not synchronise to read it).
__global__ __launch_bounds__(64) void idxmap_count_kernel(const uint32_t* __restrict__ partial, int* __restrict__ out)
{
    const uint32_t total = wave_sum_u32(partial[threadIdx.x]);
    if (threadIdx.x == 0) *out = (int)total;
}

void ngs_launch_idxmap_forward(hipStream_t s, int P, int W, int H, const float* means, const float* view,
                               const float* proj, const float* campos, const float* feature_vector, char* workspace,
                               int* out_idx, float* out_color, float* out_depth, float* out_feature, int* radii,
                               bool count_entries)
{
    unsigned long long* zbuf = (unsigned long long*)workspace;
    uint32_t* counter = count_entries ? (uint32_t*)(workspace + (size_t)W * H * 8) : nullptr;  // 64 partial counts
    (void)hipMemsetAsync(zbuf, 0xFF, (size_t)W * H * 8, s);
    if (counter) (void)hipMemsetAsync(counter, 0, 256, s);
    if (P > 0)
        hipLaunchKernelGGL(idxmap_splat_kernel, dim3((P + NGS_GAUSS_BLOCK - 1) / NGS_GAUSS_BLOCK),
                           dim3(NGS_GAUSS_BLOCK), 0, s, P, W, H, means, view, proj, zbuf, radii, counter);
    const size_t threads = (size_t)W * H * 16;
    hipLaunchKernelGGL(idxmap_resolve_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, W, H, zbuf,
                       means, campos, feature_vector, out_idx, out_color, out_depth, out_feature);
}

void ngs_launch_idxmap_backward(hipStream_t s, int P, int W, int H, const float* means, const float* view,
                                const float* proj, const int* out_idx, const float* dL_dfeature,
                                float* dL_dfeature_vector)
{
    if (means && view && proj) {
        const size_t lanes = (size_t)P * 16;
        hipLaunchKernelGGL(idxmap_backward_gather_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s, P, W,
                           H, means, view, proj, out_idx, dL_dfeature, dL_dfeature_vector);
        return;
    }
    (void)hipMemsetAsync(dL_dfeature_vector, 0, (size_t)P * IDX_FEATURES * sizeof(float), s);
    const size_t threads = (size_t)W * H * 16;
    hipLaunchKernelGGL(idxmap_backward_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, P, W, H,
                       out_idx, dL_dfeature, dL_dfeature_vector);
}

// ---------------------------------------------------------------------------
// ngs_idxmap_forward_static: the forward pass for callers that cannot allocate device memory and cannot hand over a
// workspace either (code under HIP graph capture, serving loops).  Its z-buffer lives in a file-scope device array
// of fixed size; every call on a GPU shares that GPU's copy.
// ---------------------------------------------------------------------------
#include <stdio.h>

#include <mutex>

#include "../../include/ngs_rasterizer.h"

#define IDX_STATIC_PIXELS NGS_IDXMAP_STATIC_MAX_PIXELS
#define IDX_STATIC_BYTES ((size_t)IDX_STATIC_PIXELS * 8 + 256)
#define IDX_STATIC_MAX_DEVICES 64

// One copy per GPU the library runs on, held until the process ends.  In host code this name is only a handle
// for hipGetSymbolAddress; the address kernels and memsets use is the one that call returns.
static __device__ __align__(256) unsigned char scratch[IDX_STATIC_BYTES];

static std::mutex g_static_lock;
static char* g_static_ptr[IDX_STATIC_MAX_DEVICES];  // per device: address of that device's copy of `scratch`

// Address of the current device's copy, resolved once per device with that device current.
static hipError_t idx_static_workspace(char** out)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= IDX_STATIC_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> hold(g_static_lock);
    if (!g_static_ptr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        g_static_ptr[dev] = (char*)p;
    }
    *out = g_static_ptr[dev];
    return hipSuccess;
}

static int static_fail(const char* what, const char* detail)
{
    char msg[256];
    snprintf(msg, sizeof msg, "ngs_idxmap_forward_static: %s", what);
    return ngs_set_error(msg, detail);
}

extern "C" int ngs_idxmap_forward_static(int P, int width, int height, const float* means3D, const float* viewmatrix,
                                         const float* projmatrix, const float* feature_vector, const float* campos,
                                         int* out_idx, float* out_color, float* out_depth, float* out_feature,
                                         int* radii, int* num_rendered_device, void* stream)
{
    ngs_clear_error();
    hipStream_t st = (hipStream_t)stream;
    if (P < 0 || P > NGS_MAX_GAUSSIANS || width <= 0 || height <= 0) return static_fail("bad sizes", nullptr);
    if (width > 32767 || height > 32767 || (long long)width * height > (long long)IDX_STATIC_PIXELS)
        return static_fail("image larger than the built-in z-buffer (NGS_IDXMAP_STATIC_MAX_PIXELS pixels); "
                           "use ngs_idxmap_forward with a workspace", nullptr);
    if (!out_idx || !out_color || !out_depth || !out_feature) return static_fail("missing required pointer", nullptr);
    if (P > 0 && (!means3D || !viewmatrix || !projmatrix || !feature_vector || !campos))
        return static_fail("missing required pointer", nullptr);
    if (((uintptr_t)out_feature | (uintptr_t)feature_vector) % 16 != 0)
        return static_fail("feature arrays must be 16-byte aligned", nullptr);
    char* ws = nullptr;
    const hipError_t e = idx_static_workspace(&ws);
    if (e != hipSuccess) return static_fail("hipGetSymbolAddress(scratch)", hipGetErrorString(e));
    // ngs_launch_idxmap_forward resets the part of the workspace this image size uses (all-ones z-buffer, zeroed
    // counters), so nothing of an earlier call's contents is read.
    ngs_launch_idxmap_forward(st, P, width, height, means3D, viewmatrix, projmatrix, campos, feature_vector, ws,
                              out_idx, out_color, out_depth, out_feature, radii, num_rendered_device != nullptr);
    if (num_rendered_device)
        hipLaunchKernelGGL(idxmap_count_kernel, dim3(1), dim3(64), 0, st,
                           (const uint32_t*)(ws + (size_t)width * height * 8), num_rendered_device);
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) return static_fail("launch", hipGetErrorString(le));
    return 0;
}
