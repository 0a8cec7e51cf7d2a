// Shared declarations for the gfx950 rasterizer library: state layouts carved
// out of the three caller-owned scratch buffers, launch prototypes of every
// stage, and the device math both per-Gaussian kernels (forward and backward)
// must evaluate identically.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <math.h>
#include <stdint.h>

#define NGS_TILE_W 16
#define NGS_TILE_H 16
#define NGS_TILE_PIX 256
#ifndef NGS_GAUSS_BLOCK
#define NGS_GAUSS_BLOCK 256  // Gaussians per workgroup in the per-Gaussian kernels; also the scan granule
#endif
#define NGS_MAX_GAUSSIANS (1 << 27)  // entry points refuse more: per-Gaussian offsets are formed in 32-bit lanes
#define NGS_ACC_STRIDE 16    // floats per Gaussian in the 64-byte gradient accumulator rows
#define NGS_ALIGN 256

// Words of the geometry buffer's `misc` array beyond [0] (instance count) and [1] (flags: 1 = a Gaussian was culled
// although the caller said `prefiltered`, 2 = more than 2^31-1 instances, both written by the scan kernel; 4 = an
// emission pass met more instances than the binning buffer's capacity and was cut short -- informational: the host
// decides from the count, queues the stages again with a buffer that fits, and the bit stays set).
#define NGS_MISC_DMIN 2       // lower bound of the visible Gaussians' depth bits
#define NGS_MISC_DMAX 3       // upper bound
#define NGS_MISC_ACC_CLEAN 4  // 1: the gradient accumulator rows are all zero (the forward pass left them so)
#define NGS_MISC_T_SCAN 5     // low 32 bits of the device's constant-rate clock when the scan kernel finished
#define NGS_MISC_QUAD 6       // where BinLayout::quad lies behind the point list, in units of NGS_ALIGN bytes (the
                              // backward pass is told the instance count, not the capacity the buffer was laid out for)
// Slots of one accumulator row filled by the backward compositing kernel, summed over the pixels a Gaussian was
// blended at (d = Gaussian centre - pixel, G = exp(power), o = opacity): DX, DY = sum of o dL/dalpha G d;
// CON_A, CON_B, CON_C = sum of o dL/dalpha G (dx^2, dx dy, dy^2); OPAC = sum of dL/dalpha G = dL/dopacity;
// R, G, B = dL/dcolour.  The per-Gaussian backward kernel turns the first five into dL/dmean2D and dL/dconic with
// the Gaussian's conic (backward.cu:545-551).
enum { ACC_DX = 0, ACC_DY, ACC_CON_A, ACC_CON_B, ACC_CON_C, ACC_OPAC, ACC_R, ACC_G, ACC_B, ACC_USED };

// ---------------------------------------------------------------------------
// Scratch-buffer layouts.  Offsets are relative to the 256-byte aligned base of
// the buffer the caller's allocator returned; forward and backward compute the
// same offsets from (P), (R) and (W,H), the way the reference re-carves its
// chunks (cuda_rasterizer/rasterizer_impl.cu:155-194,371-373).
// ---------------------------------------------------------------------------
struct GeomLayout {
    size_t rec;        // float4[P][4] one 64-byte splat record per Gaussian, so that the compositing kernels'
                       // gather touches ONE cache line per splat: [0] = {pixel x, pixel y, conic a, conic b},
                       // [1] = {conic c, opacity, red, green}, [2] = {blue, -, -, -}, [3] unused
    size_t depth;      // float [P]  view-space z
    size_t rect;       // uint2 [P]  packed tile rectangle: x = min.x | max.x<<16, y = min.y | max.y<<16
    size_t tiles;      // u32   [P]  tiles touched
    size_t radii;      // i32   [P]  internal radii when the caller passes none
    size_t clamped;    // u8    [P]  bit ch set when colour channel ch was clamped at 0
    size_t block_sums; // u32   [ceil(P/256)] tiles touched per 256-Gaussian block (top bit: a point of the block was culled
                       //            although the caller said the points were prefiltered)
    size_t block_offs; // u32   [ceil(P/256)] exclusive prefix of block_sums
    size_t misc;       // u32   [8]  [0] total instances R, [1] error flags (1 prefiltered point culled, 2 count
                       //            beyond 2^31-1, 4 count beyond the binning buffer's capacity), [2] / [3] a lower /
                       //            upper bound of the depth bits of the visible Gaussians, within 2^16 of the extremes
                       //            (0xffff0000 / 0 when there is none), [4] 1 while the accumulator rows are all zero
    size_t acc;        // float [P][16] gradient accumulator rows (backward)
    // Depth order of the Gaussians (forward only): instances are emitted in this order, so that the stable
    // tile-level sort leaves every tile's list sorted by depth.
    size_t dkeys[2];   // u32   [P]  depth bits (all ones for a culled Gaussian), ping/pong of the radix passes
    size_t order[2];   // u32   [P]  Gaussian indices, ping/pong; order[0] ends up sorted by (depth bits, index)
    size_t s_tiles;    // u32   [P]  tiles touched, in depth order
    size_t s_rect;     // uint2 [P]  packed tile rectangle, in depth order
    size_t s_block_sums; // u32 [ceil(P/256)] tiles touched per 256 Gaussians in depth order
    size_t s_block_offs; // u32 [ceil(P/256)] exclusive prefix of s_block_sums
    size_t dhist;      // u32   [256 * ngs_sort_blocks(P)] histogram of the depth passes
    size_t dtotals;    // u32   [256]
    size_t block_depths; // u32 [ceil(P/256)] depth bounds of the block's visible Gaussians, high halves of the depth
                       //            bits: low 16 = smallest, rounded down; high 16 = largest, rounded up (0xffff and 0
                       //            for a block without a visible Gaussian)
    size_t total;
};

struct ImageLayout {
    size_t final_T;   // float[T*256] tile-major: pixel (tile, ly*16+lx)
    size_t n_contrib; // u32  [T*256] tile-major
    size_t ranges;    // uint2[T]
    size_t hard;      // u32[4 + 2T]: work lists of the per-tile sort's later launches; [0], [1] = lengths, list k at 4 + kT;
                      // [2] = number of sub-lists the split of long lists made
    size_t total;
};

// Sized by a CAPACITY in instances (the count itself when the buffer is allocated after the count is known, an
// estimate when it is allocated before).  The sorted point list comes first, so its address does not depend on
// the capacity: that is all the backward pass needs from this buffer.
struct BinLayout {
    size_t vals[2];  // u32[C] ping/pong; vals[0] (offset 0) is the sorted point list
    size_t keys[2];  // u32[C] each: tile ids, ping/pong of the tile-level sort (then depth-bit scratch of the general
                     // per-tile sort)
    size_t hist;     // u32[256 * ngs_sort_blocks(C)] digit-major per-block histogram
    size_t totals;   // u32[256] per-digit totals
    size_t quad;     // u8[C][4] per instance and 8x8 block of its tile (byte k: block k of the tile, in wave order): bit q
                     // set when the splat can reach 4x4 quadrant q of the block.  Written by the forward compositing
                     // kernel for every instance it looks at, read by the backward one instead of repeating the tests.
    size_t total;
};

static inline size_t ngs_align_up(size_t v) { return (v + NGS_ALIGN - 1) & ~(size_t)(NGS_ALIGN - 1); }

#define NGS_SORT_THREADS 256
#define NGS_SORT_ITEMS 16
#define NGS_SORT_TILE (NGS_SORT_THREADS * NGS_SORT_ITEMS)
// Pairs per workgroup of a radix pass over `count` pairs: 4096 (the more pairs a workgroup holds, the longer the runs
// it writes per digit: at 3 M pairs and 8-bit digits 1024 cost 40 us more over four passes), but 1024 below a million
// pairs (the instances of a small scene), where 4096 would give fewer workgroups than there are CUs.
#define NGS_SORT_SMALL ((size_t)1 << 20)
static inline int ngs_sort_items(size_t count) { return count < NGS_SORT_SMALL ? 4 : NGS_SORT_ITEMS; }
static inline size_t ngs_sort_blocks(size_t count)
{
    const size_t tile = (size_t)NGS_SORT_THREADS * ngs_sort_items(count);
    return (count + tile - 1) / tile;
}

static inline GeomLayout ngs_geom_layout(size_t P)
{
    GeomLayout l;
    size_t o = 0, nb = (P + NGS_GAUSS_BLOCK - 1) / NGS_GAUSS_BLOCK;
    l.rec = o; o = ngs_align_up(o + P * 64);
    l.depth = o; o = ngs_align_up(o + P * 4);
    l.rect = o; o = ngs_align_up(o + P * 8);
    l.tiles = o; o = ngs_align_up(o + P * 4);
    l.radii = o; o = ngs_align_up(o + P * 4);
    l.clamped = o; o = ngs_align_up(o + P);
    l.block_sums = o; o = ngs_align_up(o + nb * 4);
    l.block_offs = o; o = ngs_align_up(o + nb * 4);
    l.misc = o; o = ngs_align_up(o + 32);
    l.acc = o; o = ngs_align_up(o + P * NGS_ACC_STRIDE * 4);
    for (int k = 0; k < 2; k++) { l.dkeys[k] = o; o = ngs_align_up(o + P * 4); }
    for (int k = 0; k < 2; k++) { l.order[k] = o; o = ngs_align_up(o + P * 4); }
    l.s_tiles = o; o = ngs_align_up(o + P * 4);
    l.s_rect = o; o = ngs_align_up(o + P * 8);
    l.s_block_sums = o; o = ngs_align_up(o + nb * 4);
    l.s_block_offs = o; o = ngs_align_up(o + nb * 4);
    l.dhist = o; o = ngs_align_up(o + ngs_sort_blocks(P) * 256 * 4);
    l.dtotals = o; o = ngs_align_up(o + 256 * 4);
    l.block_depths = o; o = ngs_align_up(o + nb * 4);
    l.total = o + NGS_ALIGN;
    return l;
}

// Depth rank that rides with every instance through the tile-level sort, so that the per-tile depth sort reads it
// with the list instead of gathering a depth per entry: r = (depth bits - a lower bound of the frame's) >> shift,
// `key_bits` + `word_bits` bits long.  Its high key_bits sit in the low bits of the instance's 32-bit sort key, below
// the tile id (the tile-level passes skip them); its low word_bits sit in the instance's list word above the
// idx_bits the Gaussian index needs.  The bit pattern of a positive float grows with its value, so r never decreases
// with depth, and with shift == 0 (the frame's depth span fits the rank) it orders exactly like the depth bits.
// key_bits == 0: no rank is carried (too few bits beside the tile id to be worth it).
struct DepthRank {
    int key_bits, word_bits, idx_bits;
    int min_shift;  // tests: a rank at least this coarse, whatever the frame's depth span
    __host__ __device__ uint32_t idx_mask() const { return idx_bits >= 32 ? 0xffffffffu : (1u << idx_bits) - 1u; }
    __host__ __device__ int shift(uint32_t dmin, uint32_t dmax) const
    {
        const uint32_t span = dmax > dmin ? dmax - dmin : 0u;
        const int len = span ? 32 - __builtin_clz(span) : 0;  // bit length of the span
        const int sh = len > key_bits + word_bits ? len - (key_bits + word_bits) : 0;
        return sh > min_shift ? sh : min_shift;
    }
    // the rank of a depth, and its two parts as they are stored
    __host__ __device__ uint32_t rank(uint32_t bits, uint32_t dmin, int sh) const
    {
        return ((bits - dmin) >> sh) & ((1u << (key_bits + word_bits)) - 1u);  // the mask changes nothing for a depth
    }                                                                           // between the frame's extremes
    __host__ __device__ uint32_t key_part(uint32_t r) const { return r >> word_bits; }
    __host__ __device__ uint32_t word_part(uint32_t r) const
    {
        return word_bits ? (r & ((1u << word_bits) - 1u)) << idx_bits : 0u;
    }
    // and back: the rank from a sort key and a list word
    __host__ __device__ uint32_t join(uint32_t key, uint32_t word) const
    {
        const uint32_t hi = key & ((1u << key_bits) - 1u);
        return word_bits ? (hi << word_bits) | (word >> idx_bits) : hi;
    }
};
// P Gaussians, tile ids of tile_bits bits.  The rank is at most 31 bits long (the span of positive floats).
static inline DepthRank ngs_depth_rank(int P, int tile_bits, bool wanted)
{
    DepthRank d = {0, 0, 32, 0};
    if (!wanted || 32 - tile_bits < 12) return d;
    d.key_bits = 32 - tile_bits;
    if (d.key_bits > 31) d.key_bits = 31;
    int idx_bits = 1;
    while (idx_bits < 32 && ((uint64_t)1 << idx_bits) < (uint64_t)P) idx_bits++;
    d.word_bits = 32 - idx_bits;
    if (d.word_bits > 31 - d.key_bits) d.word_bits = 31 - d.key_bits;
    d.idx_bits = d.word_bits ? idx_bits : 32;
    return d;
}

static inline ImageLayout ngs_image_layout(int W, int H)
{
    ImageLayout l;
    size_t T = (size_t)((W + NGS_TILE_W - 1) / NGS_TILE_W) * ((H + NGS_TILE_H - 1) / NGS_TILE_H);
    size_t o = 0;
    l.final_T = o; o = ngs_align_up(o + T * NGS_TILE_PIX * 4);
    l.n_contrib = o; o = ngs_align_up(o + T * NGS_TILE_PIX * 4);
    l.ranges = o; o = ngs_align_up(o + T * 8);
    l.hard = o; o = ngs_align_up(o + (2 * T + 4) * 4);
    l.total = o + NGS_ALIGN;
    return l;
}

static inline BinLayout ngs_bin_layout(size_t C)
{
    BinLayout l;
    size_t o = 0, nb = ngs_sort_blocks(C);
    l.vals[0] = o; o = ngs_align_up(o + C * 4);
    l.vals[1] = o; o = ngs_align_up(o + C * 4);
    l.keys[0] = o; o = ngs_align_up(o + C * 4);
    l.keys[1] = o; o = ngs_align_up(o + C * 4);
    l.hist = o; o = ngs_align_up(o + nb * 256 * 4);
    l.totals = o; o = ngs_align_up(o + 256 * 4);
    l.quad = o; o = ngs_align_up(o + C * 4);
    l.total = o + NGS_ALIGN;
    return l;
}

static inline char* ngs_align_ptr(char* p)
{
    return (char*)(((uintptr_t)p + NGS_ALIGN - 1) & ~(uintptr_t)(NGS_ALIGN - 1));
}

// Per-call camera constants.  The scalars travel to kernels by value; the
// matrices stay in device memory (the caller's tensors) and are read through
// wave-uniform scalar loads, so no host copy or synchronisation is needed.
struct NgsCamera {
    const float* view;    // 16 floats, device
    const float* proj;    // 16 floats, device
    const float* campos;  // 3 floats, device
    float tan_fovx, tan_fovy, focal_x, focal_y;
    int W, H, grid_x, grid_y;
};

// Adam (torch.optim.Adam's single-tensor update, no weight decay / amsgrad).  The scalars are formed on the host in
// double and rounded to float once, as torch does with its Python floats: one_minus_beta1 = float(1 - beta1),
// one_minus_beta2 = float(1 - beta2), bc2_sqrt = float(sqrt(1 - beta2^t)); a group's step is float(lr / (1 - beta1^t)).
struct AdamConst {
    float one_minus_beta1, beta2, one_minus_beta2, eps, bc2_sqrt, grad_scale;
};

static inline AdamConst ngs_adam_const(double beta1, double beta2, double eps, int step, float grad_scale)
{
    AdamConst k;
    k.one_minus_beta1 = (float)(1.0 - beta1);
    k.beta2 = (float)beta2;
    k.one_minus_beta2 = (float)(1.0 - beta2);
    k.eps = (float)eps;
    k.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
    k.grad_scale = grad_scale;
    return k;
}

static inline float ngs_adam_step_size(double lr, double beta1, int step)
{
    return (float)(lr / (1.0 - pow(beta1, (double)step)));
}

// Optimiser state of the SH coefficients for the per-Gaussian backward kernel's fused update: with `param` set the
// kernel applies Adam to the [P][16][3] coefficient rows in place instead of writing their gradient (the DC
// coefficient uses step_head, the others step_tail: two parameter groups in the reference,
// scene/gaussian_model.py:262-263).
struct ShAdamArgs {
    float* param;
    float* exp_avg;
    float* exp_avg_sq;
    float step_head, step_tail;
    AdamConst k;
};

// Optimiser state of one further parameter group for the same kernel: step = float(lr / (1 - beta1^t)) and
// bc2_sqrt = float(sqrt(1 - beta2^t)) with t the number of steps THIS group has taken.
struct AdamGroupArgs {
    float* param;
    float* exp_avg;
    float* exp_avg_sq;
    float step, bc2_sqrt;
};

// Positions, raw opacities, raw scales and raw rotations (scene/gaussian_model.py:255-261) for the fused update of
// the per-Gaussian backward kernel; xyz.param == NULL switches it off.  grad_accum / denom / max_radii are the
// densification statistics of scene/gaussian_model.py:550-552 and train.py:115, or NULL.
struct GeoAdamArgs {
    AdamGroupArgs xyz, opacity, scaling, rotation;
    AdamConst k;  // bc2_sqrt unused: every group brings its own
    float* grad_accum;
    float* denom;
    int* max_radii;
};

// ---------------------------------------------------------------------------
// Stage launchers (each defined in the .hip file named in the comment).
// ---------------------------------------------------------------------------
// capi.hip: the calling thread's error message (what ngs_last_error returns); ngs_set_error returns 1
int ngs_set_error(const char* what, const char* detail);
void ngs_clear_error(void);
// preprocess_fwd.hip
// dkeys / order: depth bits (all ones when culled) and the Gaussian's own index, the input of the depth sort (both NULL:
// not written, for a frame that is known to take the per-tile sort)
// raw != 0: scales, rots and opac are the optimiser's raw parameters; the kernel applies exp / normalise / sigmoid
void ngs_launch_preprocess_fwd(hipStream_t s, int P, int D, int M, const float* means, const float* scales, float mod,
                               const float* rots, const float* opac, int raw, const float* shs, const float* cov_pre,
                               const float* col_pre, const NgsCamera& cam, int prefiltered, int* radii, float4* rec,
                               float* depth, uint2* rect, uint32_t* tiles, uint8_t* clamped,
                               uint32_t* block_sums, uint32_t* block_depths, uint32_t* misc, uint32_t* dkeys,
                               uint32_t* order);
void ngs_launch_mark_visible(hipStream_t s, int P, const float* means, const float* view, uint8_t* present);
// binning.hip
// mailbox: device-visible address of a 64-bit word in pinned host memory that receives (error flag << 32 | total),
// or NULL.  block_depths: per-block depth bounds (GeomLayout::block_depths), reduced into misc[NGS_MISC_DMIN / _DMAX].
void ngs_launch_scan_block_sums(hipStream_t s, int nblocks, const uint32_t* block_sums, uint32_t* block_offs,
                                const uint32_t* block_depths, uint32_t* misc, unsigned long long* mailbox);
// The stages between the scan and the compositing take the instance count from DEVICE memory (`count`, the word
// the scan kernel wrote) and a capacity: launches are sized by the capacity, kernels work on min(count, capacity)
// instances, so they can be queued before the host knows the count and never step outside the buffer.
// ngs_launch_depth_order: from the Gaussian indices sorted by depth and their sorted keys (all ones = emits no
// instance), the rectangles and tile counts in that order and the exclusive prefix of the counts per block of 256.
// ngs_launch_duplicate then emits the instances of the k-th Gaussian IN DEPTH ORDER at its offset; the value of each
// instance is order[k].
void ngs_launch_depth_order(hipStream_t s, int P, const uint32_t* order, const uint32_t* sorted_keys, const uint2* rect,
                            uint32_t* s_tiles, uint2* s_rect, uint32_t* s_block_sums, uint32_t* s_block_offs);
// Keys are (tile id << dr.key_bits) | high part of the depth rank, list words Gaussian index | low part of the rank
// << dr.idx_bits (see DepthRank); dr.key_bits == 0: the tile id and the index alone.
// `clear`: clear_bytes bytes (16-byte aligned, a multiple of 16) the kernel zeroes on the side -- the tile ranges and the
// counters behind them, which the range kernel two launches on expects zero.
void ngs_launch_duplicate(hipStream_t s, int P, const uint2* rect, const uint32_t* tiles, const uint32_t* block_offs,
                          const uint32_t* order, const float* depth, const DepthRank& dr, int grid_x,
                          uint32_t* tile_keys, uint32_t* vals, int64_t capacity, uint32_t* misc, void* clear,
                          size_t clear_bytes);
// clear_bytes: bytes to zero starting at `ranges` before the kernel runs (8 T and the counter that follows), 0 when
// ngs_launch_duplicate has seen to it; the tile id of a key is key >> qbits
void ngs_launch_tile_ranges(hipStream_t s, const uint32_t* count, int64_t capacity, const uint32_t* tile_keys, int qbits,
                            int T, uint2* ranges, size_t clear_bytes);
// radix_sort.hip: stable LSD radix sort of (32-bit key, 32-bit value) pairs on bits [first_bit, first_bit + bits) of
// the key, result in keys[0]/vals[0].  It runs twice per frame: over the Gaussians' depth bits (all 32), and over the tile
// ids of the instances emitted in that depth order -- stable, so every tile's list comes out sorted by depth.
// ngs_sort_input_buffer returns the buffer index (0 or 1) the unsorted input must be written to.
// `count` (device) may be NULL: the number of pairs is then `capacity` itself (the caller knows it).
int ngs_sort_input_buffer(int bits);
void ngs_launch_sort_tile_bits(hipStream_t s, const uint32_t* count, int64_t capacity, uint32_t* const keys[2],
                               uint32_t* const vals[2], uint32_t* hist, uint32_t* totals, int bits, int first_bit = 0);
// Per-tile depth sort of vals[0] (lists in index order inside each tile) by the depth of the listed Gaussians;
// keys[] and vals[1] are scratch for lists that do not fit a workgroup.
// hard: u32[4 + 2T] with hard[0] == hard[1] == hard[2] == 0 on entry (see ImageLayout); subs: room for subs_cap
// (first, end) pairs (the tile-level sort's histogram, idle by then).  dr.key_bits != 0: keys[0] and vals[0]
// carry every entry's depth rank (the list leaves without it); misc: the geometry buffer's word array (depth extremes
// in [2], [3]).
void ngs_launch_sort_segments(hipStream_t s, int T, const uint2* ranges, const float* depth, uint32_t* const keys[2],
                              uint32_t* const vals[2], uint32_t* hard, const DepthRank& dr, const uint32_t* misc,
                              uint2* subs, uint32_t subs_cap);
// render_fwd.hip
// acc / P / misc: the compositing kernel also clears the P gradient accumulator rows for the backward pass (it is
// bound by arithmetic and leaves the memory system idle) and says so in misc[NGS_MISC_ACC_CLEAN].
// quad / quad_offset: BinLayout::quad of the buffer plist lies in (plist at its start), and its distance from plist in
// units of NGS_ALIGN bytes, which the kernel leaves in misc[NGS_MISC_QUAD] for the backward pass.
// stage_clock (or NULL): a word of host memory that receives (seq << 32) | ticks of the device's constant-rate clock
// that passed between the end of the scan kernel and the start of this kernel, i.e. the duration of the binning stages.
void ngs_launch_render_fwd(hipStream_t s, const NgsCamera& cam, const uint2* ranges, const uint32_t* plist,
                           const float4* rec, const float* bg, float* final_T,
                           uint32_t* n_contrib, float* out_color, float* acc, int P, uint32_t* misc,
                           unsigned long long* stage_clock, uint32_t seq, uint8_t* quad, uint32_t quad_offset);
// render_bwd.hip
// Clears the accumulator rows unless misc[NGS_MISC_ACC_CLEAN] says they are clear already.
void ngs_launch_zero_acc_unless_clean(hipStream_t s, float* acc, int P, const uint32_t* misc);
// misc: the kernel marks the accumulator rows as used (misc[NGS_MISC_ACC_CLEAN] = 0) and finds the forward pass's
// per-quadrant reach bits through misc[NGS_MISC_QUAD], behind plist.
void ngs_launch_render_bwd(hipStream_t s, const NgsCamera& cam, const uint2* ranges, const uint32_t* plist,
                           const float4* rec, const float* bg, const float* final_T,
                           const uint32_t* n_contrib, const float* dL_dpix, float* acc, uint32_t* misc);
// preprocess_bwd.hip
// opac_raw != NULL: scales, rots and opac_raw are the optimiser's raw parameters (the kernel applies exp /
// normalise / sigmoid itself) and the three gradients come out wrt those raw values.  geo (needs raw parameters):
// the kernel steps all four groups in place and writes no gradient at all; the dL_* pointers may then be NULL.
void ngs_launch_preprocess_bwd(hipStream_t s, int P, int D, int M, const float* means, const int* radii,
                               const float* shs, const uint8_t* clamped, const float* scales, const float* rots,
                               const float* opac_raw, float mod, const float* cov_pre, const NgsCamera& cam,
                               const float* acc, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                               float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                               float* dL_drot, const ShAdamArgs* sh_adam, const GeoAdamArgs* geo);

// sh_views.hip: per-Gaussian colour gradient with the clamp mask applied, and the SH coefficient gradient summed
// over V views from those masked colour gradients.
void ngs_launch_masked_color(hipStream_t s, int P, const uint8_t* clamped, const float* dL_dcolor, float* out);
void ngs_launch_masked_color_from_acc(hipStream_t s, int P, const uint8_t* clamped, const float* acc, float* out);
void ngs_launch_sh_from_views(hipStream_t s, int P, int D, int M, int V, const float* means, const float* campos,
                              long long campos_stride, const float* colors, long long colors_stride, float* dL_dsh,
                              const ShAdamArgs* sh_adam);

// train_ops.hip: fused L1 + SSIM image loss (value and gradient) and fused Adam + parameter activations.
size_t ngs_loss_workspace_floats(int W, int H);
void ngs_launch_l1_ssim(hipStream_t s, int W, int H, float lambda_dssim, const float* img, const float* gt,
                        float* workspace, float* loss_terms, float* dL_dimg);
void ngs_launch_adam_flat(hipStream_t s, long long n, float* param, const float* grad, float* exp_avg,
                          float* exp_avg_sq, double lr_head, double lr_tail, int period, int head, double beta1,
                          double beta2, double eps, int step, float grad_scale);
void ngs_launch_adam_activated(hipStream_t s, int P, float* const raw[3], const float* const grad[3],
                               float* const exp_avg[3], float* const exp_avg_sq[3], float* const activated[3],
                               const double lr[3], double beta1, double beta2, double eps, const int step[3],
                               float grad_scale, const float* dL_dmean2D, const int* radii, float* grad_accum,
                               float* denom, int* max_radii);
void ngs_launch_view_statistics(hipStream_t s, int P, const float* dL_dmean2D, const int* radii, float* norm_out,
                                float* vis_out, int* max_radii);
void ngs_launch_activate(hipStream_t s, int P, const float* op_raw, const float* sc_raw, const float* ro_raw,
                         float* op, float* sc, float* ro);

// idxmap.hip: nearest-centre index map of the reference's second rasterizer and its backward.
// knn.hip: mean squared distance to the three nearest neighbours (scene initialisation)
size_t ngs_knn_workspace_size(int P);
void ngs_launch_knn(hipStream_t s, int P, const float* points, char* workspace, float* mean_dist2);
// idxmap.hip
size_t ngs_idxmap_workspace_size(int W, int H);
void ngs_launch_idxmap_forward(hipStream_t s, int P, int W, int H, const float* means, const float* view,
                               const float* proj, const float* campos, const float* feature_vector, char* workspace,
                               int* out_idx, float* out_color, float* out_depth, float* out_feature, int* radii,
                               bool count_entries);
// means/view/proj all non-NULL: gather per Gaussian (no atomics); otherwise scatter with float atomics.
void ngs_launch_idxmap_backward(hipStream_t s, int P, int W, int H, const float* means, const float* view,
                                const float* proj, const int* out_idx, const float* dL_dfeature,
                                float* dL_dfeature_vector);

#ifdef __HIPCC__
// ---------------------------------------------------------------------------
// Device math shared by preprocess forward and backward.  Matrices are stored
// c[col][row] and multiplied in the order of the reference's vendored GLM
// (third_party/glm/glm/detail/type_mat3x3.inl:486-519) so that forward and
// backward, and the CPU checker, see bit-identical covariance values.  These
// translation units are compiled with -ffp-contract=off.
// ---------------------------------------------------------------------------
struct Mat3 { float c[3][3]; };

__device__ __forceinline__ Mat3 m3_mul(const Mat3& a, const Mat3& b)
{
    Mat3 r;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++)
            r.c[j][i] = a.c[0][i] * b.c[j][0] + a.c[1][i] * b.c[j][1] + a.c[2][i] * b.c[j][2];
    return r;
}

__device__ __forceinline__ Mat3 m3_tr(const Mat3& a)
{
    Mat3 r;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++) r.c[j][i] = a.c[i][j];
    return r;
}

__device__ __forceinline__ Mat3 m3_diag(float a, float b, float c)
{
    Mat3 r;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++) r.c[j][i] = 0.0f;
    r.c[0][0] = a; r.c[1][1] = b; r.c[2][2] = c;
    return r;
}

// Rotation matrix of quaternion (r,x,y,z) used as given (cuda_rasterizer/forward.cu:126-138).
__device__ __forceinline__ Mat3 m3_from_quat(float r, float x, float y, float z)
{
    Mat3 R;
    R.c[0][0] = 1.f - 2.f * (y * y + z * z); R.c[0][1] = 2.f * (x * y - r * z); R.c[0][2] = 2.f * (x * z + r * y);
    R.c[1][0] = 2.f * (x * y + r * z); R.c[1][1] = 1.f - 2.f * (x * x + z * z); R.c[1][2] = 2.f * (y * z - r * x);
    R.c[2][0] = 2.f * (x * z - r * y); R.c[2][1] = 2.f * (y * z + r * x); R.c[2][2] = 1.f - 2.f * (x * x + y * y);
    return R;
}

// World covariance (xx,xy,xz,yy,yz,zz) from scale and quaternion (cuda_rasterizer/forward.cu:118-152).
__device__ __forceinline__ void cov3d_from_scale_rot(float sx, float sy, float sz, float mod, float qr, float qx,
                                                     float qy, float qz, float out[6])
{
    Mat3 M = m3_mul(m3_diag(mod * sx, mod * sy, mod * sz), m3_from_quat(qr, qx, qy, qz));
    Mat3 S = m3_mul(m3_tr(M), M);
    out[0] = S.c[0][0]; out[1] = S.c[0][1]; out[2] = S.c[0][2];
    out[3] = S.c[1][1]; out[4] = S.c[1][2]; out[5] = S.c[2][2];
}

__device__ __forceinline__ float3 xform43(float px, float py, float pz, const float* m)
{
    return make_float3(m[0] * px + m[4] * py + m[8] * pz + m[12], m[1] * px + m[5] * py + m[9] * pz + m[13],
                       m[2] * px + m[6] * py + m[10] * pz + m[14]);
}

__device__ __forceinline__ float4 xform44(float px, float py, float pz, const float* m)
{
    return make_float4(m[0] * px + m[4] * py + m[8] * pz + m[12], m[1] * px + m[5] * py + m[9] * pz + m[13],
                       m[2] * px + m[6] * py + m[10] * pz + m[14], m[3] * px + m[7] * py + m[11] * pz + m[15]);
}

// Screen-space covariance and the intermediates its backward needs
// (cuda_rasterizer/forward.cu:74-113, backward.cu:166-199).
struct Cov2D {
    Mat3 T, W, V;
    float3 t;
    float txtz, tytz, limx, limy, a, b, c;
};

__device__ __forceinline__ Cov2D cov2d_eval(float mx, float my, float mz, const NgsCamera& cam,
                                            const float* __restrict__ vm, const float cov6[6])
{
    Cov2D k;
    k.t = xform43(mx, my, mz, vm);
    k.limx = 1.3f * cam.tan_fovx;
    k.limy = 1.3f * cam.tan_fovy;
    k.txtz = k.t.x / k.t.z;
    k.tytz = k.t.y / k.t.z;
    k.t.x = fminf(k.limx, fmaxf(-k.limx, k.txtz)) * k.t.z;
    k.t.y = fminf(k.limy, fmaxf(-k.limy, k.tytz)) * k.t.z;
    Mat3 J;
    J.c[0][0] = cam.focal_x / k.t.z; J.c[0][1] = 0.0f; J.c[0][2] = -(cam.focal_x * k.t.x) / (k.t.z * k.t.z);
    J.c[1][0] = 0.0f; J.c[1][1] = cam.focal_y / k.t.z; J.c[1][2] = -(cam.focal_y * k.t.y) / (k.t.z * k.t.z);
    J.c[2][0] = 0.0f; J.c[2][1] = 0.0f; J.c[2][2] = 0.0f;
    k.W.c[0][0] = vm[0]; k.W.c[0][1] = vm[4]; k.W.c[0][2] = vm[8];
    k.W.c[1][0] = vm[1]; k.W.c[1][1] = vm[5]; k.W.c[1][2] = vm[9];
    k.W.c[2][0] = vm[2]; k.W.c[2][1] = vm[6]; k.W.c[2][2] = vm[10];
    k.T = m3_mul(k.W, J);
    k.V.c[0][0] = cov6[0]; k.V.c[0][1] = cov6[1]; k.V.c[0][2] = cov6[2];
    k.V.c[1][0] = cov6[1]; k.V.c[1][1] = cov6[3]; k.V.c[1][2] = cov6[4];
    k.V.c[2][0] = cov6[2]; k.V.c[2][1] = cov6[4]; k.V.c[2][2] = cov6[5];
    Mat3 cov = m3_mul(m3_mul(m3_tr(k.T), m3_tr(k.V)), k.T);
    k.a = cov.c[0][0] + 0.3f;
    k.b = cov.c[0][1];
    k.c = cov.c[1][1] + 0.3f;
    return k;
}

// 16-byte streaming accesses for data touched once per pass (SH rows, gradient tensors, optimiser state): the
// non-temporal hint keeps such streams from evicting each other and the reused state from the caches.
typedef float ngs_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 stream_load4(const float* p)
{
    const ngs_v4f v = __builtin_nontemporal_load((const ngs_v4f*)p);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void stream_store4(float* p, float4 v)
{
    __builtin_nontemporal_store((ngs_v4f){v.x, v.y, v.z, v.w}, (ngs_v4f*)p);
}

// The activations between the optimiser's raw parameters and what the rasterizer consumes
// (scene/gaussian_model.py:36-43): sigmoid for the opacity, exp for the scales, x / max(|x|, 1e-12) for the
// rotation, and the backward of the last one: d raw = (d act - act (act . d act)) / |raw|.
__device__ __forceinline__ float sigmoid_act(float raw) { return 1.0f / (1.0f + expf(-raw)); }
__device__ __forceinline__ float4 normalize_quat(float4 raw, float& len)
{
    len = fmaxf(sqrtf(raw.x * raw.x + raw.y * raw.y + raw.z * raw.z + raw.w * raw.w), 1e-12f);
    return make_float4(raw.x / len, raw.y / len, raw.z / len, raw.w / len);
}
__device__ __forceinline__ float4 normalize_quat_backward(float4 a, float len, float4 g)
{
    const float dot = a.x * g.x + a.y * g.y + a.z * g.z + a.w * g.w;
    return make_float4((g.x - a.x * dot) / len, (g.y - a.y * dot) / len, (g.z - a.z * dot) / len,
                       (g.w - a.w * dot) / len);
}

// bc2_sqrt is passed apart from k for callers whose parameter groups have taken different numbers of steps
__device__ __forceinline__ float adam_update(float p, float g, float& m, float& v, float step, float bc2_sqrt,
                                             const AdamConst& k)
{
    m = m + k.one_minus_beta1 * (g - m);
    v = v * k.beta2 + k.one_minus_beta2 * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + k.eps;
    return p - step * (m / denom);
}
__device__ __forceinline__ float adam_update(float p, float g, float& m, float& v, float step, const AdamConst& k)
{
    return adam_update(p, g, m, v, step, k.bc2_sqrt, k);
}

// Spherical-harmonics constants (cuda_rasterizer/auxiliary.h:22-39).
#define NGS_SH_C0 0.28209479177387814f
#define NGS_SH_C1 0.4886025119029199f
#define NGS_SH_C2_0 1.0925484305920792f
#define NGS_SH_C2_1 -1.0925484305920792f
#define NGS_SH_C2_2 0.31539156525252005f
#define NGS_SH_C2_3 -1.0925484305920792f
#define NGS_SH_C2_4 0.5462742152960396f
#define NGS_SH_C3_0 -0.5900435899266435f
#define NGS_SH_C3_1 2.890611442640554f
#define NGS_SH_C3_2 -0.4570457994644658f
#define NGS_SH_C3_3 0.3731763325901154f
#define NGS_SH_C3_4 -0.4570457994644658f
#define NGS_SH_C3_5 1.445305721320277f
#define NGS_SH_C3_6 -0.5900435899266435f

__device__ __forceinline__ float3 f3_scale(float s, float3 v) { return make_float3(s * v.x, s * v.y, s * v.z); }
__device__ __forceinline__ float3 f3_add(float3 a, float3 b) { return make_float3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ float3 f3_sub(float3 a, float3 b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float f3_dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// d(colour)/d(sh[k]) for a unit view direction (x, y, z): the SH basis value of coefficient k
// (cuda_rasterizer/backward.cu:49-137).  Sets w[0 .. (deg+1)^2).  Shared by the per-view SH backward and by the
// view-factored SH gradient so both form the same products.
__device__ __forceinline__ void sh_basis_weights(int deg, float x, float y, float z, float* w)
{
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    w[0] = NGS_SH_C0;
    if (deg > 0) {
        w[1] = -NGS_SH_C1 * y;
        w[2] = NGS_SH_C1 * z;
        w[3] = -NGS_SH_C1 * x;
        if (deg > 1) {
            w[4] = NGS_SH_C2_0 * xy;
            w[5] = NGS_SH_C2_1 * yz;
            w[6] = NGS_SH_C2_2 * (2.f * zz - xx - yy);
            w[7] = NGS_SH_C2_3 * xz;
            w[8] = NGS_SH_C2_4 * (xx - yy);
            if (deg > 2) {
                w[9] = NGS_SH_C3_0 * y * (3.f * xx - yy);
                w[10] = NGS_SH_C3_1 * xy * z;
                w[11] = NGS_SH_C3_2 * y * (4.f * zz - xx - yy);
                w[12] = NGS_SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
                w[13] = NGS_SH_C3_4 * x * (4.f * zz - xx - yy);
                w[14] = NGS_SH_C3_5 * z * (xx - yy);
                w[15] = NGS_SH_C3_6 * x * (xx - 3.f * yy);
            }
        }
    }
}

// Sum over the 64 lanes of a wave; the total is valid in lane 63 only.
// Row-wise inclusive scan by DPP row shifts, then two row broadcasts.
__device__ __forceinline__ float wave_sum_lane63(float v)
{
#define NGS_DPP_ADD(ctrl, rmask)                                                                               \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, \
                                                               false))
    NGS_DPP_ADD(0x111, 0xf);  // row_shr:1
    NGS_DPP_ADD(0x112, 0xf);  // row_shr:2
    NGS_DPP_ADD(0x114, 0xf);  // row_shr:4
    NGS_DPP_ADD(0x118, 0xf);  // row_shr:8
    NGS_DPP_ADD(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
    NGS_DPP_ADD(0x143, 0xc);  // row_bcast:31 into rows 2 and 3
#undef NGS_DPP_ADD
    return v;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Inclusive prefix sum across the 64 lanes of a wave.
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t n = __shfl_up(v, o, 64);
        if (lane >= o) v += n;
    }
    return v;
}
#endif  // __HIPCC__
