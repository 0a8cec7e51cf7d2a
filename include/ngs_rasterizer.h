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
 * word of pinned host memory per calling thread and device, through which the
 * instance count reaches the host (never freed; NGS_NO_MAILBOX=1 in the
 * environment disables it).
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
 * ngs_rasterize_forward with the binning buffer sized BEFORE the instance count is known, so that the host never
 * has to wait for the count in the middle of the call (the reference does, rasterizer_impl.cu:280-285, and so does
 * ngs_rasterize_forward: the GPU idles there while the count travels to the host and the next kernels travel back).
 * No counterpart in the reference.  Arguments as ngs_rasterize_forward, plus binning_capacity = the number of
 * (Gaussian, tile) instances the binning buffer shall hold: the binning allocator is asked for
 * ngs_required_binning_bytes(binning_capacity) right away and every stage is queued at once; the kernels take the
 * count from device memory.  The host reads the count only after the last stage has been queued (for the return
 * value, and to compare it with the capacity).  If the scene produces MORE instances than the capacity, the stages
 * after the count are queued a second time with a buffer of the right size (one more call of the binning
 * allocator, made after the stream has been synchronised, so the allocator may free or move the first buffer):
 * the results are the same, the call is slower.  A good capacity is the previous call's count plus a
 * few percent.  binning_capacity == 0 is ngs_rasterize_forward.
 */
int ngs_rasterize_forward_presized(ngs_alloc_fn geometry_alloc, void* geometry_ctx, ngs_alloc_fn binning_alloc,
                                   void* binning_ctx, ngs_alloc_fn image_alloc, void* image_ctx, int P, int D, int M,
                                   const float* background, int width, int height, const float* means3D,
                                   const float* shs, const float* colors_precomp, const float* opacities,
                                   const float* scales, float scale_modifier, const float* rotations,
                                   const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                                   const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered,
                                   float* out_color, int* radii, int debug, void* stream, int* num_rendered,
                                   long long binning_capacity);

/*
 * Rasterizer::backward (rasterizer.h:55-84, rasterizer_impl.cu:340-434).
 * geom_buffer / binning_buffer / image_buffer are the buffers the forward call
 * filled (or whole copies of them at addresses with the same residue modulo 256;
 * they are re-carved from P and the image size; of the binning buffer the
 * first 4 * R bytes past the 256-byte boundary, the sorted point list, are read, and the forward compositing
 * kernel's per-quadrant reach bytes, which lie where a word of geom_buffer says: the three buffers must be those
 * of one forward call).  Every output element is WRITTEN (zeros for culled Gaussians), so
 * the caller need not zero-fill; dL_dconic ([P][2][2]) may be NULL, and so may dL_dsh (the SH coefficient gradient
 * is then not produced; its share of dL_dmean3D still is -- see ngs_sh_gradient_from_views).
 * rotations, dL_drot and dL_dconic must be 16-byte aligned (an error otherwise); D and M as in the forward call.
 * Shapes: dL_dpix [3][H][W]; dL_dmean2D [P][3]; dL_dopacity [P]; dL_dcolor
 * [P][3]; dL_dmean3D [P][3]; dL_dcov3D [P][6]; dL_dsh [P][M][3]; dL_dscale
 * [P][3]; dL_drot [P][4].
  * The backward pass writes into geom_buffer (the gradient accumulator rows the forward pass left cleared, and the
 * word that says whether they are clear): backward calls over one saved state must not overlap, on whatever
 * streams; a second backward call over the same state clears the rows itself first.  A geom_buffer must come from
 * this library's forward pass.
 */
int ngs_rasterize_backward(int P, int D, int M, int R, const float* background, int width, int height,
                           const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                           float scale_modifier, const float* rotations, const float* cov3D_precomp,
                           const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                           float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer,
                           char* image_buffer, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic,
                           float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                           float* dL_dscale, float* dL_drot, int debug, void* stream);

/*
 * ngs_rasterize_backward with the optimiser step of the SH coefficients fused in.  No counterpart in the
 * reference, where the coefficient gradient is written to memory (rasterize_points.cu:156) and torch.optim.Adam
 * reads it back (scene/gaussian_model.py:262-263,275; train.py:124): here the gradient rows never leave the chip.
 * Arguments as ngs_rasterize_backward, except: `shs` ([P][16][3]) is updated IN PLACE, there is no dL_dsh (and no
 * colors_precomp), and sh_exp_avg / sh_exp_avg_sq are Adam's moments of the same shape.  lr_dc applies to
 * coefficient 0 of every row, lr_rest to the others; constants and `step` as for ngs_adam_step.  Requires M == 16
 * and 16-byte aligned arrays.  Single-view steps only: a caller that sums gradients over views must exchange them
 * before the optimiser runs.
 */
int ngs_rasterize_backward_sh_adam(int P, int D, int M, int R, const float* background, int width, int height,
                                   const float* means3D, float* shs, const float* scales, float scale_modifier,
                                   const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                                   const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy,
                                   const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
                                   const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                                   float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dscale,
                                   float* dL_drot, float* sh_exp_avg, float* sh_exp_avg_sq, double lr_dc, double lr_rest,
                                   double beta1, double beta2, double eps, int step, float grad_scale, int debug,
                                   void* stream);

/*
 * Raw-parameter entry points.  The reference's extension only ever sees ACTIVATED values: its Python side applies
 * sigmoid to the opacity, exp to the scales and normalize to the rotations on every iteration
 * (scene/gaussian_model.py:36-43, properties at :115-127, called from gaussian_renderer/__init__.py:55-69) and
 * autograd differentiates those operators on the way back.  A trainer that keeps the optimiser's raw parameters on
 * the device can hand those to the library instead; the activations and their backward then run inside the
 * per-Gaussian kernels and no activated copy or activation graph exists.
 *
 * ngs_rasterize_forward_raw: ngs_rasterize_forward_presized with opacities_raw [P], scales_raw [P][3] and
 * rotations_raw [P][4] in place of the activated arrays (and no cov3D_precomp).  The image, radii and state
 * buffers are exactly those of ngs_rasterize_forward_presized called on sigmoid(opacities_raw), exp(scales_raw)
 * and rotations_raw / max(|rotations_raw|, 1e-12) as ngs_activate_parameters computes them.
 *
 * ngs_rasterize_backward_raw: ngs_rasterize_backward for a forward pass made by ngs_rasterize_forward_raw; takes
 * the raw arrays again (plus opacities_raw, which the activated variant does not need) and returns dL_dopacity_raw,
 * dL_dscale_raw and dL_drot_raw, the gradients wrt the RAW values: dL_dopacity * s (1 - s), dL_dscale * exp(raw),
 * (dL_drot - q (q . dL_drot)) / |raw| with s, q the activated values.  All other outputs as ngs_rasterize_backward.
 *
 * ngs_rasterize_backward_train: the backward pass with torch.optim.Adam's step (scene/gaussian_model.py:255-275,
 * train.py:124-125) of all five parameter groups fused into the per-Gaussian kernel: positions, SH coefficients,
 * raw opacities, raw scales and raw rotations are updated IN PLACE, no gradient is written to memory.  `state`
 * holds, per group, the parameter array (what the forward pass read), Adam's two moments, the learning rate and
 * the 1-based count of steps this group has taken including this one (torch counts per parameter; a group whose
 * lr is 0 keeps its values but still advances its moments).  features.lr applies to coefficient 0 of every SH row,
 * features_rest_lr to the others (the reference's "f_dc" and "f_rest" groups).  The gradient is multiplied by
 * grad_scale first.  When xyz_gradient_accum is not NULL the densification statistics of
 * scene/gaussian_model.py:550-552 and train.py:115 are updated for Gaussians with radii > 0 (all three arrays are
 * then required).  Requires SH rows of 16 coefficients and 16-byte aligned arrays.  Single-view steps only: a
 * caller that sums gradients over views must exchange them before the optimiser runs.  Every Gaussian takes the
 * step, rendered or not (zero gradient), as with torch.optim.Adam on dense gradients.
 *
 * ngs_activate_parameters: the three activations as the kernels above compute them, for callers that keep
 * activated copies next to the raw values (any output may be NULL).
 */
typedef struct ngs_adam_group {
    float* param;      /* updated in place */
    float* exp_avg;    /* Adam's first moment, same shape */
    float* exp_avg_sq; /* Adam's second moment, same shape */
    double lr;
    int step;          /* >= 1 */
} ngs_adam_group;

typedef struct ngs_train_state {
    ngs_adam_group xyz;      /* [P][3] */
    ngs_adam_group features; /* [P][16][3]; lr of coefficient 0 */
    double features_rest_lr; /* lr of coefficients 1..15 */
    ngs_adam_group opacity;  /* [P] raw */
    ngs_adam_group scaling;  /* [P][3] raw */
    ngs_adam_group rotation; /* [P][4] raw */
    double beta1, beta2, eps;
    float grad_scale;
    float* xyz_gradient_accum; /* [P] or NULL */
    float* denom;              /* [P] or NULL */
    int* max_radii2D;          /* [P] or NULL */
} ngs_train_state;

int ngs_rasterize_forward_raw(ngs_alloc_fn geometry_alloc, void* geometry_ctx, ngs_alloc_fn binning_alloc,
                              void* binning_ctx, ngs_alloc_fn image_alloc, void* image_ctx, int P, int D, int M,
                              const float* background, int width, int height, const float* means3D, const float* shs,
                              const float* colors_precomp, const float* opacities_raw, const float* scales_raw,
                              float scale_modifier, const float* rotations_raw, const float* viewmatrix,
                              const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                              int prefiltered, float* out_color, int* radii, int debug, void* stream,
                              int* num_rendered, long long binning_capacity);
int ngs_rasterize_backward_raw(int P, int D, int M, int R, const float* background, int width, int height,
                               const float* means3D, const float* shs, const float* colors_precomp,
                               const float* opacities_raw, const float* scales_raw, float scale_modifier,
                               const float* rotations_raw, const float* viewmatrix, const float* projmatrix,
                               const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                               char* geom_buffer, char* binning_buffer, char* image_buffer, const float* dL_dpix,
                               float* dL_dmean2D, float* dL_dconic, float* dL_dopacity_raw, float* dL_dcolor,
                               float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale_raw,
                               float* dL_drot_raw, int debug, void* stream);
int ngs_rasterize_backward_train(int P, int D, int R, const float* background, int width, int height,
                                 float scale_modifier, const float* viewmatrix, const float* projmatrix,
                                 const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                                 char* geom_buffer, char* binning_buffer, char* image_buffer, const float* dL_dpix,
                                 const ngs_train_state* state, int debug, void* stream);
int ngs_activate_parameters(int P, const float* opacities_raw, const float* scales_raw, const float* rotations_raw,
                            float* opacities, float* scales, float* rotations, void* stream);

/*
 * ngs_rasterize_backward for callers that exchange gradients between GPUs.  No counterpart in the reference
 * (single GPU).  Arguments as ngs_rasterize_backward, plus: masked_color [P][3] receives the colour gradient with
 * the channels clamped in the forward pass zeroed (what ngs_masked_color_gradient computes from dL_dcolor) as soon
 * as the compositing stage has produced it, i.e. BEFORE the per-Gaussian stage is queued; color_ready (may be NULL)
 * is then called on the host, once, with color_ready_ctx.  An event the callback records on `stream` therefore
 * completes when masked_color is final, and a collective that waits for that event (on another stream) runs
 * concurrently with the per-Gaussian stage.  With P == 0 nothing is written and the callback is not called.
 */
int ngs_rasterize_backward_early_color(int P, int D, int M, int R, const float* background, int width, int height,
                                       const float* means3D, const float* shs, const float* colors_precomp,
                                       const float* scales, float scale_modifier, const float* rotations,
                                       const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                                       const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                                       char* geom_buffer, char* binning_buffer, char* image_buffer,
                                       const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                                       float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                                       float* dL_dscale, float* dL_drot, int debug, void* stream, float* masked_color,
                                       void (*color_ready)(void*), void* color_ready_ctx);

/*
 * ngs_rasterize_backward_early_color with the per-Gaussian stage queued in n_chunks pieces, so that a collective on
 * the gradients of the Gaussians of chunk k can start while the later chunks are still being computed.  No
 * counterpart in the reference (single GPU).  chunks[k] covers Gaussians [first, first + count); the chunks tile
 * [0, P) in order and every `first` is a multiple of 256.  The five gradients a multi-GPU caller exchanges may be
 * redirected per chunk: a non-NULL pointer receives the chunk's rows of that array (row 0 = Gaussian `first`; dL_drot
 * and dL_dsh 16-byte aligned) instead of rows [first, first + count) of the whole array, which may then be NULL if
 * every chunk redirects it.  All other outputs go to the whole arrays as in ngs_rasterize_backward.  chunk_ready (may
 * be NULL) is called on the host with (chunk_ready_ctx, k) right after chunk k's kernel has been queued on `stream`:
 * an event recorded there completes when the chunk's gradients are final.  masked_color may be NULL (then there is
 * no hand-over and color_ready is not called).  Results are those of the unchunked call, bit for bit.
 */
typedef struct ngs_gradient_chunk {
    int first, count;
    float* dL_dmean3D; /* [count][3] or NULL */
    float* dL_dopacity; /* [count] */
    float* dL_dscale; /* [count][3] */
    float* dL_drot; /* [count][4] */
    float* dL_dsh; /* [count][M][3] */
} ngs_gradient_chunk;
int ngs_rasterize_backward_chunked(int P, int D, int M, int R, const float* background, int width, int height,
                                   const float* means3D, const float* shs, const float* colors_precomp,
                                   const float* scales, float scale_modifier, const float* rotations,
                                   const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                                   const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                                   char* geom_buffer, char* binning_buffer, char* image_buffer, const float* dL_dpix,
                                   float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                                   float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                                   int debug, void* stream, float* masked_color, void (*color_ready)(void*),
                                   void* color_ready_ctx, int n_chunks, const ngs_gradient_chunk* chunks,
                                   void (*chunk_ready)(void*, int), void* chunk_ready_ctx);

/* Rasterizer::markVisible (rasterizer.h:24-29, rasterizer_impl.cu:141-153):
 * present[i] = 1 if Gaussian i passes the near-plane test, else 0 (bytes). */
int ngs_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     unsigned char* present, void* stream);

/* Bytes the allocators will be asked for: the counterpart of
 * required<GeometryState / BinningState / ImageState>() (rasterizer_impl.h:67-73).  R is the instance count, or
 * the capacity given to ngs_rasterize_forward_presized. */
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
 *
 * ngs_sh_adam_from_views: as ngs_sh_gradient_from_views, but the summed gradient is not written: the Adam step of
 * the SH coefficients is applied to `shs` ([P][16][3], in place) and its moments directly (arguments as in
 * ngs_rasterize_backward_sh_adam; grad_scale = 1/V turns the sum into the mean over the views).
 *
 * ngs_view_statistics: one view's densification statistics in summable form, so that they can ride in the same
 * all-reduce as the gradients: norm_out[i] = |dL_dmean2D[i].xy| and vis_out[i] = 1 where radii[i] > 0, zeros
 * elsewhere (the increments of xyz_gradient_accum and denom, scene/gaussian_model.py:550-552); max_radii2D (may be
 * NULL) takes the running maximum of radii in place (train.py:115), which ranks merge with an integer max.
 */
int ngs_view_statistics(int P, const float* dL_dmean2D, const int* radii, float* norm_out, float* vis_out,
                        int* max_radii2D, void* stream);
int ngs_masked_color_gradient(int P, const char* geom_buffer, const float* dL_dcolor, float* out, void* stream);
int ngs_sh_adam_from_views(int P, int D, int M, int V, const float* means3D, const float* campos,
                           long long campos_stride, const float* dL_dcolor_views, long long view_stride, float* shs,
                           float* sh_exp_avg, float* sh_exp_avg_sq, double lr_dc, double lr_rest, double beta1,
                           double beta2, double eps, int step, float grad_scale, void* stream);
int ngs_sh_gradient_from_views(int P, int D, int M, int V, const float* means3D, const float* campos,
                               long long campos_stride, const float* dL_dcolor_views, long long view_stride,
                               float* dL_dsh, void* stream);

/*
 * Training-step companions of the rasterizer.  The reference has no native counterpart: it runs these parts of an
 * iteration as chains of PyTorch operators (file:line below).
 *
 * ngs_l1_ssim_loss: loss = (1 - lambda) * mean|image - gt| + lambda * (1 - SSIM(image, gt)) with the 11x11 Gaussian
 * window (sigma 1.5, zero padding) of utils/loss_utils.py:17-63, combined as at train.py:90-92, and its gradient wrt
 * `image`.  image, gt_image, dL_dimage: [3][height][width]; loss_terms: 3 floats on the device = {L1 mean, SSIM
 * mean, loss}; workspace: ngs_loss_workspace_bytes(width, height) bytes of device memory.
 *
 * ngs_adam_step: torch.optim.Adam's update (no weight decay, no amsgrad; gaussian_model.py:275) of a flat tensor of
 * n floats whose activation is the identity (positions, SH coefficients).  Elements whose index modulo `period` is
 * below `head` use lr_head, the rest lr_tail: with period = 3*M, head = 3 one call covers the reference's "f_dc"
 * and "f_rest" groups (gaussian_model.py:262-263) on a single [P][M][3] tensor.  `step` counts from 1.  The
 * gradient is multiplied by grad_scale first (1/views for a mean over a view batch).  Arrays 16-byte aligned.
 *
 * ngs_adam_step_activated: the same update for opacity, scaling and rotation (index 0, 1, 2 of each array of
 * pointers; [P], [P][3], [P][4]), whose raw values pass through sigmoid, exp and normalise
 * (gaussian_model.py:36-43, 113-141): grad[] is the gradient wrt the ACTIVATED value, as ngs_rasterize_backward
 * returns it; the activation's backward is applied here, raw[] is updated in place and activated[] receives the new
 * activated values for the next forward pass.  When xyz_gradient_accum is not NULL the densification statistics
 * are updated as well for Gaussians with radii > 0: xyz_gradient_accum += |dL_dmean2D.xy|, denom += 1
 * (gaussian_model.py:550-552), max_radii2D = max(max_radii2D, radii) (train.py:115).  step[] holds one count per
 * group: torch keeps it per parameter, and the reference's opacity reset (gaussian_model.py:331-334 with train.py:122-127)
 * makes the opacity group miss a step the other two take.
 */
size_t ngs_loss_workspace_bytes(int width, int height);
int ngs_l1_ssim_loss(int width, int height, float lambda_dssim, const float* image, const float* gt_image,
                     float* workspace, float* loss_terms, float* dL_dimage, void* stream);
int ngs_adam_step(long long n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, double lr_head,
                  double lr_tail, int period, int head, double beta1, double beta2, double eps, int step,
                  float grad_scale, void* stream);
int ngs_adam_step_activated(int P, float* const raw[3], const float* const grad[3], float* const exp_avg[3],
                            float* const exp_avg_sq[3], float* const activated[3], const double lr[3], double beta1,
                            double beta2, double eps, const int step[3], float grad_scale, const float* dL_dmean2D,
                            const int* radii, float* xyz_gradient_accum, float* denom, int* max_radii2D,
                            void* stream);

/*
 * Index-map path of the reference's second rasterizer (r2 = submodules/rasterizer2), the operator behind
 * rasterizer2._GetIdxMap2 (r2/rasterizer2/__init__.py:31-138; binding r2/rasterize_points.cu:183-295).
 *
 * ngs_idxmap_forward takes the pointer list of CudaRasterizer::Rasterizer::getidxmap
 * (r2/cuda_rasterizer/rasterizer.h:61-85, rasterizer_impl.cu:851-984) with the three buffer callbacks replaced by
 * one caller-owned workspace of ngs_idxmap_workspace_bytes(width, height) bytes (the reference's three buffers
 * never leave its binding either), plus a stream.  Every Gaussian centre in front of the near plane and inside the
 * image is splatted as a square of half-width 3/depth pixels; each pixel shows the nearest centre:
 *   out_idx [H][W]: its index or -1;  out_color [H][W][3]: its unit view direction;  out_depth [H][W];
 *   out_feature [H][W][64]: {depth, 24-value positional encoding of the direction, features 25..63 of the
 *   Gaussian's feature_vector [P][64]}.
 * Every output element is written (zeros where no centre landed), so the caller need not zero-fill.  normal,
 * scales, rotations, sh, M and the tan_fov values are accepted and ignored, as in the reference; radii ([P], 1 for
 * a Gaussian that was splatted) and num_rendered (the number of (Gaussian, pixel) entries the reference would have
 * sorted, its return value) may be NULL -- asking for num_rendered costs a host synchronisation.
 *
 * ngs_idxmap_backward = getidxmap_backward (rasterizer.h:87-96, rasterizer_impl.cu:986-1007):
 * dL_dfeature_vector[g][25..63] = sum over the pixels showing g of dL_dout_feature[pixel][25..63]; the whole
 * [P][64] output is written.  The reference adds without atomics from concurrent threads; here the sum is well
 * defined.  viewmatrix / projmatrix are an addition to the reference's list and may be NULL: with them (and
 * means3D) each Gaussian gathers the pixels of its own square in pixel order, without atomics; without them the
 * pixels scatter with float atomics (sum exact up to addition order).
 */
size_t ngs_idxmap_workspace_bytes(int width, int height);
int ngs_idxmap_forward(int P, int M, int width, int height, const float* means3D, const float* normal,
                       const float* scales, const float* rotations, const float* viewmatrix, const float* projmatrix,
                       float tan_fovx, float tan_fovy, const float* feature_vector, const float* sh,
                       const float* campos, int* out_idx, float* out_color, float* out_depth, float* out_feature,
                       int* radii, char* workspace, int debug, void* stream, int* num_rendered);
int ngs_idxmap_backward(int P, int width, int height, const float* means3D, const float* campos, const int* out_idx,
                        const float* dL_dout_feature, float* dL_dfeature_vector, const float* viewmatrix,
                        const float* projmatrix, int debug, void* stream);

/*
 * ngs_idxmap_forward for callers that cannot allocate device memory: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  It has NO counterpart in the reference (whose binding allocates three
 * scratch tensors per call, r2/rasterize_points.cu:205-213).  Same kernels and same results as ngs_idxmap_forward,
 * but the z-buffer (8 bytes per pixel + 256) lives in a device array built into the library instead of a
 * caller-supplied workspace, and the call itself allocates nothing, frees nothing and never synchronises.
 *  - Limit: width * height <= NGS_IDXMAP_STATIC_MAX_PIXELS (2^21, enough for 1920x1080); larger images are refused
 *    with an error.  Every GPU the library runs on holds its own 16 MiB + 256 B copy of the array until the process
 *    ends.
 *  - All calls on one GPU share that GPU's array: calls that overlap in time on the same GPU (different streams or
 *    threads) race; calls on one stream, or otherwise ordered, are safe.  Each call resets what it uses.
 *  - Arguments as ngs_idxmap_forward without the ones the reference ignores, the workspace and the debug flag;
 *    num_rendered_device (may be NULL) is a DEVICE pointer that receives the entry count on the stream, so no host
 *    synchronisation is needed; radii may be NULL.
  * The first call on each device resolves the array's address (under a lock; the runtime may load the code object
 * and allocate the array then): make one call per device outside any stream capture or no-allocation region before
 * relying on the guarantees above.  `stream` must belong to the device that is current at the call.
 */
#define NGS_IDXMAP_STATIC_MAX_PIXELS (1 << 21)
int ngs_idxmap_forward_static(int P, int width, int height, const float* means3D, const float* viewmatrix,
                              const float* projmatrix, const float* feature_vector, const float* campos, int* out_idx,
                              float* out_color, float* out_depth, float* out_feature, int* radii,
                              int* num_rendered_device, void* stream);

/*
 * Scene initialisation: mean_dist2[i] = mean of the squared distances from point i to its three nearest other
 * points = SimpleKNN::knn (submodules/simple-knn/simple_knn.h:16-19, simple_knn.cu:185-221), the routine behind
 * simple_knn._C.distCUDA2 (spatial.cu:15-26, ext.cpp:15-17), whose result scene/gaussian_model.py:160-161 turns into
 * the initial scales.  points [P][3], mean_dist2 [P] (every element written).  The search is exact, as the
 * reference's is; with fewer than three other points the FLT_MAX sentinels of the missing neighbours enter the
 * mean as they do in the reference (+inf for P < 3, FLT_MAX / 3 for P == 3).  The reference allocates its temporaries inside the call; here they live in a caller-owned
 * workspace of ngs_knn_workspace_bytes(P) bytes (any alignment; about 52 bytes per point).
 */
size_t ngs_knn_workspace_bytes(int P);
int ngs_knn_mean_dist2(int P, const float* points, float* mean_dist2, char* workspace, int debug, void* stream);

/*
 * Introspection for parity tests: byte offsets (relative to the 256-byte
 * aligned base of each buffer) of the intermediate arrays, in this order:
 *  geometry[10]: splat_record(16 floats per Gaussian: x, y, conic a, b, c, opacity, r, g, b, 7 unused),
 *                depth(float), tile_rect(uint2 packed lo|hi<<16), tiles_touched(u32),
 *                internal_radii(i32), clamped_bits(u8), block_sums(u32),
 *                block_offsets(u32), misc(u32[8]), grad_accumulator(float[16])
 *  binning[1]:   point_list(u32), always 0.  The sorted 64-bit keys (tile << 32 | depth bits) of the reference's
 *                BinningState are not stored: nothing reads them after the sort; entry i of the point list lies in
 *                the tile whose range holds i, and its depth bits are depth[point_list[i]]
 *  image[3]:     final_T(float, per tile in thread order: [tile][2x2 blocks][2x2 quadrants][4][4]), n_contrib(u32, tile-major), ranges(uint2[tiles])
 * No counterpart in the reference (its state structs are private, rasterizer_impl.h:29-65).
 */
int ngs_state_layout(int P, long long R, int width, int height, size_t* geometry, size_t* binning, size_t* image);

/*
 * Per-stage timing with HIP events recorded on the caller's stream around each
 * stage of ngs_rasterize_forward / _backward and around the other entry points' kernels.  No counterpart in the reference
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

/*
 * For tests and measurements: where ngs_rasterize_forward takes the depth order of the tile lists from.  -1 (the
 * default): it chooses per call from P and the instance count; 0: always the per-tile depth sort after the
 * tile-level sort; 1: always the depth sort of the Gaussians before their instances are emitted; 2 and 3: the
 * per-tile depth sort with the depth rank that instances carry through the tile-level sort made too coarse to stand
 * in for the depth (so that every depth is fetched after all), and with no rank carried at all.  The lists are the
 * same in every case.  The setting belongs to the calling thread (forward calls of other threads are not affected).
 * ngs_debug_depth_order_taken returns what the calling thread's last forward call took (0 or 1; -1 before the
 * first call).  No counterpart in the reference (which sorts all instances on the full 64-bit key,
 * rasterizer_impl.cu:303-308).
 */
void ngs_debug_depth_order(int choice);
int ngs_debug_depth_order_taken(void);
/*
 * What the default choice goes by: the device clocks the stages between the scan and the compositing kernel on every
 * frame, and the library keeps, per calling thread and device and for the current width, height and P (to within a factor
 * of two), the last three
 * such durations with either source of the depth order; once both have three, the one with the cheaper best frame
 * (per instance) is taken and the other tried again every 128 frames (every 1024 when it was more than a quarter behind).  Returns those two durations in
 * microseconds as of the last forward call of the calling thread on the current device (-1: not measured yet);
 * a frame's figure is known to the host only once a later forward call has looked at it.  Return value 1: the platform
 * has no coherent host memory for the device to post into, the choice is by a fixed cost model.
 */
int ngs_debug_stage_clock(double* us_per_tile_sort, double* us_gaussians_first);

/* Message of the last failure on the calling thread ("" if none). */
const char* ngs_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
