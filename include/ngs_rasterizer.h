/*
 * ngs_rasterizer.h -- C ABI of libngs_rasterizer.so, the MI355X (gfx950)
 * differentiable Gaussian rasterizer.
 *
 * The three compute entry points take exactly the pointer lists of the
 * reference's inner C++ API, CudaRasterizer::Rasterizer::{forward, backward,
 * markVisible} (submodules/diff-gaussian-rasterization/cuda_rasterizer/
 * rasterizer.h:24-84), i.e. what its torch binding (rasterize_points.cu:35-217)
 * calls, with two mechanical changes: the std::function<char*(size_t)> buffer
 * callbacks become a C function pointer plus a context pointer, and a stream
 * handle is added (the reference always uses the legacy default stream).
 *
 * All array arguments are DEVICE pointers to contiguous float32 / int32 data
 * unless noted.  Optional inputs are signalled by NULL exactly as in the
 * reference (an empty tensor's data pointer): colors_precomp == NULL selects
 * the SH path, cov3D_precomp == NULL the scale/rotation path.
 *
 * Return value: 0 on success, non-zero on failure; ngs_last_error() then
 * returns a message (thread-local storage, valid until the next call on that
 * thread).  Nothing is retained by the library between calls except one 64-byte
 * word of pinned host memory per calling thread, through which the instance
 * count reaches the host (NGS_NO_MAILBOX=1 in the environment disables it).
 */
#ifndef NGS_RASTERIZER_H
#define NGS_RASTERIZER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Grows a caller-owned device buffer to at least `bytes` and returns its base
 * address.  Replaces std::function<char*(size_t)> (rasterizer.h:32-34; the
 * reference binds it to a torch tensor resize, rasterize_points.cu:27-33). */
typedef char* (*ngs_alloc_fn)(void* ctx, size_t bytes);

/*
 * Rasterizer::forward (rasterizer.h:31-53, rasterizer_impl.cu:198-336).
 * P Gaussians, D active SH degree, M SH coefficients per Gaussian in `shs`
 * ([P][M][3]).  viewmatrix / projmatrix are 4x4 in the reference's row-vector
 * memory order; out_color is [3][height][width]; radii is [P] or NULL.
 * Blocks the host once (it waits for the instance count, as the reference does
 * at rasterizer_impl.cu:281).  *num_rendered receives the
 * number of (Gaussian, tile) instances, the reference's return value.
 * With prefiltered != 0, a Gaussian failing the near-plane test makes the call
 * fail (the reference aborts the kernel, auxiliary.h:156-160).  With debug != 0
 * every stage is followed by a stream synchronisation and an error check
 * (auxiliary.h:166-173).
 */
int ngs_rasterize_forward(ngs_alloc_fn geometry_alloc, void* geometry_ctx, ngs_alloc_fn binning_alloc,
                          void* binning_ctx, ngs_alloc_fn image_alloc, void* image_ctx, int P, int D, int M,
                          const float* background, int width, int height, const float* means3D, const float* shs,
                          const float* colors_precomp, const float* opacities, const float* scales,
                          float scale_modifier, const float* rotations, const float* cov3D_precomp,
                          const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                          float tan_fovy, int prefiltered, float* out_color, int* radii, int debug, void* stream,
                          int* num_rendered);

/*
 * Rasterizer::backward (rasterizer.h:55-84, rasterizer_impl.cu:340-434).
 * geom_buffer / binning_buffer / image_buffer are the buffers the forward call
 * filled (or copies of them at addresses with the same residue modulo 256;
 * they are re-carved from P, R and the image size).  Every output element is WRITTEN (zeros for culled Gaussians), so
 * the caller need not zero-fill; dL_dconic ([P][2][2]) may be NULL, and so may dL_dsh (the SH coefficient gradient
 * is then not produced; its share of dL_dmean3D still is -- see ngs_sh_gradient_from_views).
 * Shapes: dL_dpix [3][H][W]; dL_dmean2D [P][3]; dL_dopacity [P]; dL_dcolor
 * [P][3]; dL_dmean3D [P][3]; dL_dcov3D [P][6]; dL_dsh [P][M][3]; dL_dscale
 * [P][3]; dL_drot [P][4].
 */
int ngs_rasterize_backward(int P, int D, int M, int R, const float* background, int width, int height,
                           const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                           float scale_modifier, const float* rotations, const float* cov3D_precomp,
                           const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                           float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer,
                           char* image_buffer, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic,
                           float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                           float* dL_dscale, float* dL_drot, int debug, void* stream);

/* Rasterizer::markVisible (rasterizer.h:24-29, rasterizer_impl.cu:141-153):
 * present[i] = 1 if Gaussian i passes the near-plane test, else 0 (bytes). */
int ngs_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     unsigned char* present, void* stream);

/* Bytes the allocators will be asked for: the counterpart of
 * required<GeometryState / BinningState / ImageState>() (rasterizer_impl.h:67-73). */
size_t ngs_required_geometry_bytes(int P);
size_t ngs_required_binning_bytes(long long R);
size_t ngs_required_image_bytes(int width, int height);

/*
 * View-factored SH gradient, for callers that sum gradients over several views (view-sharded data parallelism).
 * No counterpart in the reference, which renders one view per step on one GPU (train.py:76-78).
 * Per view, dL_dsh[k] = basis_k(view direction) * masked dL_dcolor (backward.cu:42-137), so instead of exchanging
 * 3*M floats per Gaussian and view, ranks exchange the masked colour gradient (3 floats) and every rank rebuilds
 *     dL_dsh[i][k][c] = sum over v of basis_k(means3D[i] - campos[v]) * dL_dcolor_views[v][i][c]
 * adding the views in index order (identical bits on every rank; equal to the per-view dL_dsh for V == 1).
 *
 * ngs_masked_color_gradient: out[i][c] = dL_dcolor[i][c], or 0 where channel c of Gaussian i was clamped in the
 * forward pass whose geometry buffer is given (out may alias dL_dcolor).
 * ngs_sh_gradient_from_views: campos holds V positions campos_stride floats apart, dL_dcolor_views holds V
 * arrays [P][3] view_stride floats apart (so both may point into one gathered buffer); dL_dsh is [P][M][3] and
 * every element is written.
 */
int ngs_masked_color_gradient(int P, const char* geom_buffer, const float* dL_dcolor, float* out, void* stream);
int ngs_sh_gradient_from_views(int P, int D, int M, int V, const float* means3D, const float* campos,
                               long long campos_stride, const float* dL_dcolor_views, long long view_stride,
                               float* dL_dsh, void* stream);

/*
 * Introspection for parity tests: byte offsets (relative to the 256-byte
 * aligned base of each buffer) of the intermediate arrays, in this order:
 *  geometry[10]: splat_record(16 floats per Gaussian: x, y, conic a, b, c, opacity, r, g, b, 7 unused),
 *                depth(float), tile_rect(uint2 packed lo|hi<<16), tiles_touched(u32),
 *                internal_radii(i32), clamped_bits(u8), block_sums(u32),
 *                block_offsets(u32), misc(u32[8]), grad_accumulator(float[16])
 *  binning[2]:   sorted_keys(u64), point_list(u32)   (the unsorted pairs do not survive the sort's ping-pong)
 *  image[3]:     final_T(float, per tile in thread order: [tile][2x2 blocks][8][8]), n_contrib(u32, tile-major), ranges(uint2[tiles])
 * No counterpart in the reference (its state structs are private, rasterizer_impl.h:29-65).
 */
int ngs_state_layout(int P, long long R, int width, int height, size_t* geometry, size_t* binning, size_t* image);

/*
 * Per-stage timing with HIP events recorded on the caller's stream around each
 * stage of ngs_rasterize_forward / _backward.  No counterpart in the reference
 * (its only timer is a torch.cuda.Event pair around a whole iteration,
 * train.py:44-45,67,95).  One device per process; not thread-safe.
 * ngs_profile_read synchronises with the recorded events and returns, per
 * stage, the summed duration in ms and the number of timed launches since the
 * last reset; both arrays hold ngs_profile_stage_count() entries.
 */
void ngs_profile_enable(int on);
int ngs_profile_stage_count(void);
const char* ngs_profile_stage_name(int stage);
int ngs_profile_read(double* total_ms, long long* launches, int reset);

/* Message of the last failure on the calling thread ("" if none). */
const char* ngs_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
