// extern "C" entry points of libngs_rasterizer.so (declared in
// include/ngs_rasterizer.h): carve the caller's scratch buffers, run the stages
// in order on the caller's stream, report errors by return code.  Pipeline
// order follows cuda_rasterizer/rasterizer_impl.cu:198-434 of the reference.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "../../include/ngs_rasterizer.h"
#include <chrono>
#include <thread>
#include "ngs_common.h"

static thread_local char g_err[512] = "";

static int fail(const char* what, const char* detail)
{
    snprintf(g_err, sizeof g_err, "%s%s%s", what, detail ? ": " : "", detail ? detail : "");
    return 1;
}

// for entry points defined in other translation units
int ngs_set_error(const char* what, const char* detail) { return fail(what, detail); }
void ngs_clear_error(void) { g_err[0] = 0; }

#define NGS_HIP(call)                                                   \
    do {                                                                \
        hipError_t e_ = (call);                                         \
        if (e_ != hipSuccess) return fail(#call, hipGetErrorString(e_)); \
    } while (0)

// After each stage: launch errors always; in debug mode also a stream sync, as
// the reference's CHECK_CUDA does (auxiliary.h:166-173).
#define NGS_STAGE(name)                                                                \
    do {                                                                               \
        hipError_t e_ = hipGetLastError();                                             \
        if (e_ == hipSuccess && debug) e_ = hipStreamSynchronize(st);                  \
        if (e_ != hipSuccess) return fail("stage " name, hipGetErrorString(e_));       \
    } while (0)

// ---------------------------------------------------------------------------
// Optional per-stage timing with HIP events recorded on the caller's stream
// (what a benchmark needs to quote a kernel's duration without a profiler).
// A ring of event sets lets many calls be in flight; totals are harvested when
// a slot is reused and when the caller reads them.
// ---------------------------------------------------------------------------
enum {
    ST_PREPROCESS, ST_SCAN, ST_DEPTHSORT, ST_DUPLICATE, ST_SORT, ST_RANGES, ST_SEGSORT, ST_RENDER_FWD,
    ST_BWD_ZERO, ST_RENDER_BWD, ST_PREPROCESS_BWD, ST_SH_VIEWS, ST_LOSS, ST_ADAM_FLAT, ST_ADAM_ACTIVATED,
    ST_IDXMAP_FWD, ST_IDXMAP_BWD, ST_KNN, ST_COUNT
};
static const char* const g_stage_names[ST_COUNT] = {
    "preprocess_fwd", "scan_block_sums", "depth_sort_gaussians", "duplicate_with_keys", "radix_sort_tile_bits", "tile_ranges",
    "segment_sort_depth", "render_fwd",
    "zero_accumulator", "render_bwd", "preprocess_bwd", "sh_gradient_from_views", "l1_ssim_loss", "adam_identity_groups",
    "adam_activated_groups", "idxmap_forward", "idxmap_backward", "knn_mean_dist2"};
#define PROF_RING 32
struct ProfSlot {
    hipEvent_t ev[ST_COUNT][2];
    bool used[ST_COUNT];
};
static struct {
    bool enabled, created;
    ProfSlot ring[PROF_RING];
    unsigned next;
    double total_ms[ST_COUNT];
    long long count[ST_COUNT];
} g_prof;

static void prof_harvest(ProfSlot& s)
{
    for (int i = 0; i < ST_COUNT; i++) {
        if (!s.used[i]) continue;
        float ms = 0.f;
        if (hipEventSynchronize(s.ev[i][1]) == hipSuccess &&
            hipEventElapsedTime(&ms, s.ev[i][0], s.ev[i][1]) == hipSuccess) {
            g_prof.total_ms[i] += ms;
            g_prof.count[i]++;
        }
        s.used[i] = false;
    }
}

static int prof_acquire()
{
    if (!g_prof.enabled) return -1;
    if (!g_prof.created) {
        for (int r = 0; r < PROF_RING; r++)
            for (int i = 0; i < ST_COUNT; i++) {
                g_prof.ring[r].used[i] = false;
                for (int k = 0; k < 2; k++)
                    if (hipEventCreate(&g_prof.ring[r].ev[i][k]) != hipSuccess) return -1;
            }
        g_prof.created = true;
    }
    const int slot = (int)(g_prof.next++ % PROF_RING);
    prof_harvest(g_prof.ring[slot]);
    return slot;
}

static inline void prof_begin(int slot, int stage, hipStream_t st)
{
    if (slot >= 0) (void)hipEventRecord(g_prof.ring[slot].ev[stage][0], st);
}

static inline void prof_end(int slot, int stage, hipStream_t st)
{
    if (slot >= 0) {
        (void)hipEventRecord(g_prof.ring[slot].ev[stage][1], st);
        g_prof.ring[slot].used[stage] = true;
    }
}

extern "C" void ngs_profile_enable(int on) { g_prof.enabled = on != 0; }

extern "C" int ngs_profile_stage_count(void) { return ST_COUNT; }

extern "C" const char* ngs_profile_stage_name(int stage)
{
    return stage >= 0 && stage < ST_COUNT ? g_stage_names[stage] : "";
}

extern "C" int ngs_profile_read(double* total_ms, long long* launches, int reset)
{
    if (g_prof.created)
        for (int r = 0; r < PROF_RING; r++) prof_harvest(g_prof.ring[r]);
    for (int i = 0; i < ST_COUNT; i++) {
        if (total_ms) total_ms[i] = g_prof.total_ms[i];
        if (launches) launches[i] = g_prof.count[i];
        if (reset) { g_prof.total_ms[i] = 0.0; g_prof.count[i] = 0; }
    }
    return ST_COUNT;
}

// rasterizer_impl.cu:35-50
static uint32_t higher_msb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

static NgsCamera make_camera(int W, int H, float tanx, float tany, const float* view, const float* proj,
                             const float* campos)
{
    NgsCamera c;
    c.view = view; c.proj = proj; c.campos = campos;
    c.tan_fovx = tanx; c.tan_fovy = tany;
    c.focal_y = H / (2.0f * tany);
    c.focal_x = W / (2.0f * tanx);
    c.W = W; c.H = H;
    c.grid_x = (W + NGS_TILE_W - 1) / NGS_TILE_W;
    c.grid_y = (H + NGS_TILE_H - 1) / NGS_TILE_H;
    return c;
}

// One 64-bit word of coherent pinned host memory per calling thread and device, written by the scan kernel and
// polled by the host (see ngs_rasterize_forward).  Allocated on first use and never freed (a thread that ends
// leaves its 64 bytes behind: freeing from a thread-exit hook could run after the HIP runtime has shut down);
// NGS_NO_MAILBOX=1 disables it.
#define NGS_MAX_DEVICES 64
// What the binning stages (everything between the scan and the compositing kernel) cost on recent frames with either
// source of the depth order, as the device itself clocked it (render_fwd_kernel posts the figure into word 1 of the
// mailbox).  Kept per calling thread and device, for one (image size, Gaussian count) at a time.
#define NGS_ROUTE_PENDING 8
struct RouteHistory {
    int W, H, P;
    uint32_t seq;  // frames launched with a stage clock so far
    struct { uint32_t seq; int route; double instances; bool valid; } pending[NGS_ROUTE_PENDING];
    double us[2], instances[2];  // last measurement per route (instances == 0: none yet) and the frame's instance count
    double per_instance[2][3];   // the last three measurements per route, in us per instance
    int measured[2];             // how many there have been (the first frames of a process run slow: one is not enough)
    int age[2];                  // decisions since route r was last taken
    int decisions;               // decisions since the history was started
    double ticks_per_us;
};
struct Mailbox {
    unsigned long long* host;  // word 0: instance count and error flags of the frame in flight; word 1: stage clock
    unsigned long long* dev;
    int state;  // 0 = not tried, 1 = ready, -1 = unavailable
    RouteHistory route;
};

static Mailbox* mailbox_get()
{
    static thread_local Mailbox boxes[NGS_MAX_DEVICES];
    int device = -1;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= NGS_MAX_DEVICES) {
        (void)hipGetLastError();
        return nullptr;
    }
    Mailbox& mb = boxes[device];
    if (mb.state == 0) {
        mb.state = -1;
        const char* off = getenv("NGS_NO_MAILBOX");
        if (!(off && off[0] == '1')) {
            void* h = nullptr;
            void* d = nullptr;
            // the device-side address is resolved with this device current: it is only ever handed to kernels
            // launched on this device
            if (hipHostMalloc(&h, 64, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) ==
                    hipSuccess &&
                hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
                mb.host = (unsigned long long*)h;
                mb.dev = (unsigned long long*)d;
                mb.state = 1;
                memset(&mb.route, 0, sizeof mb.route);
                __atomic_store_n(mb.host + 1, 0ull, __ATOMIC_RELEASE);
                int khz = 0;  // rate of the constant clock wall_clock64() reads
                if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) != hipSuccess || khz <= 0) {
                    (void)hipGetLastError();
                    khz = 100000;
                }
                mb.route.ticks_per_us = khz * 1e-3;
            } else {
                (void)hipGetLastError();
            }
        }
    }
    return mb.state == 1 ? &mb : nullptr;
}

// Waits until the scan kernel has posted its word: spinning on it for the first millisecond (a frame's scan is
// normally done well inside that), then looking every 20 us between short sleeps, so that a long wait -- earlier work
// still queued on the stream -- does not keep a core busy.  Returns false -- the caller then falls back to a copy and
// a stream synchronisation, which also surfaces the error of a failed kernel -- when the stream has drained or failed
// without the word arriving.
static bool mailbox_wait(Mailbox* mb, hipStream_t st, uint32_t out[2])
{
    if (!mb) return false;
    auto take = [&](unsigned long long v) {
        out[0] = (uint32_t)v;
        out[1] = (uint32_t)(v >> 32);
        return true;
    };
    auto stream_idle = [&]() {
        if (hipStreamQuery(st) == hipErrorNotReady) return false;
        (void)hipGetLastError();
        return true;
    };
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; spins++) {
        const unsigned long long v = __atomic_load_n(mb->host, __ATOMIC_ACQUIRE);
        if (v != ~0ull) return take(v);
        if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(1)) break;
        __builtin_ia32_pause();
    }
    for (;;) {
        const bool idle = stream_idle();  // looked at BEFORE the word: the word is written before the kernel ends
        const unsigned long long v = __atomic_load_n(mb->host, __ATOMIC_ACQUIRE);
        if (v != ~0ull) return take(v);
        if (idle) return false;  // finished (or failed) without posting
        std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
}

extern "C" const char* ngs_last_error(void) { return g_err; }

extern "C" int ngs_debug_stage_clock(double* us_per_tile_sort, double* us_gaussians_first)
{
    Mailbox* mb = mailbox_get();
    if (us_per_tile_sort) *us_per_tile_sort = mb && mb->route.instances[0] > 0.0 ? mb->route.us[0] : -1.0;
    if (us_gaussians_first) *us_gaussians_first = mb && mb->route.instances[1] > 0.0 ? mb->route.us[1] : -1.0;
    return mb ? 0 : 1;
}

extern "C" size_t ngs_required_geometry_bytes(int P) { return ngs_geom_layout((size_t)(P > 0 ? P : 0)).total; }
extern "C" size_t ngs_required_binning_bytes(long long R) { return ngs_bin_layout((size_t)(R > 0 ? R : 0)).total; }
extern "C" size_t ngs_required_image_bytes(int W, int H) { return ngs_image_layout(W, H).total; }

extern "C" int ngs_state_layout(int P, long long R, int W, int H, size_t* geometry, size_t* binning, size_t* image)
{
    if (geometry) {
        const GeomLayout g = ngs_geom_layout((size_t)P);
        const size_t v[10] = {g.rec, g.depth, g.rect, g.tiles, g.radii, g.clamped, g.block_sums, g.block_offs,
                              g.misc, g.acc};
        memcpy(geometry, v, sizeof v);
    }
    if (binning) binning[0] = ngs_bin_layout((size_t)R).vals[0];
    if (image) {
        const ImageLayout i = ngs_image_layout(W, H);
        image[0] = i.final_T; image[1] = i.n_contrib; image[2] = i.ranges;
    }
    return 0;
}

// -1: the forward pass chooses where the depth order comes from; 0 / 1 force the per-tile sort / the sort of the
// Gaussians; 2 / 3: the per-tile sort with a depth rank too coarse to stand in for the depth / with no rank at all
// (tests and measurements; see ngs_debug_depth_order).
// Both are per calling thread.
static thread_local int g_depth_order_choice = -1;
static thread_local int g_depth_order_taken = -1;
extern "C" void ngs_debug_depth_order(int choice) { g_depth_order_choice = choice; }
extern "C" int ngs_debug_depth_order_taken(void) { return g_depth_order_taken; }

static int forward_impl(const char* fn, ngs_alloc_fn geometry_alloc, void* geometry_ctx, ngs_alloc_fn binning_alloc,
                        void* binning_ctx, ngs_alloc_fn image_alloc, void* image_ctx, int P, int D, int M,
                        const float* background, int width, int height, const float* means3D, const float* shs,
                        const float* colors_precomp, const float* opacities, const float* scales,
                        float scale_modifier, const float* rotations, const float* cov3D_precomp,
                        const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                        float tan_fovy, int prefiltered, float* out_color, int* radii, int debug, void* stream,
                        int* num_rendered, long long binning_capacity, int raw = 0)
{
    g_err[0] = 0;
    hipStream_t st = (hipStream_t)stream;
    if (num_rendered) *num_rendered = 0;
    if (P < 0 || P > NGS_MAX_GAUSSIANS || width <= 0 || height <= 0 || binning_capacity < 0) return fail(fn, "bad sizes");
    if (P == 0) return 0;
    if (!geometry_alloc || !binning_alloc || !image_alloc) return fail(fn, "missing allocator");
    if (!means3D || !opacities || !viewmatrix || !projmatrix || !background || !out_color)
        return fail(fn, "missing required pointer");
    if (!colors_precomp && (!shs || !cam_pos)) return fail(fn, "need shs+cam_pos or colors_precomp");
    if (!colors_precomp && (D < 0 || D > 3 || (D + 1) * (D + 1) > M))
        return fail(fn, "SH degree does not fit the coefficient count");
    if (!cov3D_precomp && (!scales || !rotations)) return fail(fn, "need scales+rotations or cov3D_precomp");
    if (raw && cov3D_precomp) return fail(fn, "raw parameters and cov3D_precomp exclude each other");
    const NgsCamera cam = make_camera(width, height, tan_fovx, tan_fovy, viewmatrix, projmatrix, cam_pos);
    if (cam.grid_x > 65535 || cam.grid_y > 65535) return fail(fn, "image too large");

    const GeomLayout gl = ngs_geom_layout((size_t)P);
    char* gb = geometry_alloc(geometry_ctx, gl.total);
    if (!gb) return fail(fn, "geometry allocator returned NULL");
    gb = ngs_align_ptr(gb);
    const ImageLayout il = ngs_image_layout(width, height);
    char* ib = image_alloc(image_ctx, il.total);
    if (!ib) return fail(fn, "image allocator returned NULL");
    ib = ngs_align_ptr(ib);

    float4* rec = (float4*)(gb + gl.rec);
    float* depth = (float*)(gb + gl.depth);
    uint2* rect = (uint2*)(gb + gl.rect);
    uint32_t* tiles = (uint32_t*)(gb + gl.tiles);
    uint8_t* clamped = (uint8_t*)(gb + gl.clamped);
    uint32_t* block_sums = (uint32_t*)(gb + gl.block_sums);
    uint32_t* block_offs = (uint32_t*)(gb + gl.block_offs);
    uint32_t* misc = (uint32_t*)(gb + gl.misc);
    uint32_t* dkeys[2] = {(uint32_t*)(gb + gl.dkeys[0]), (uint32_t*)(gb + gl.dkeys[1])};
    uint32_t* order[2] = {(uint32_t*)(gb + gl.order[0]), (uint32_t*)(gb + gl.order[1])};
    uint32_t* s_tiles = (uint32_t*)(gb + gl.s_tiles);
    uint2* s_rect = (uint2*)(gb + gl.s_rect);
    uint32_t* s_block_sums = (uint32_t*)(gb + gl.s_block_sums);
    uint32_t* s_block_offs = (uint32_t*)(gb + gl.s_block_offs);
    if (!radii) radii = (int*)(gb + gl.radii);
    const int nblocks = (P + NGS_GAUSS_BLOCK - 1) / NGS_GAUSS_BLOCK;
    static_assert(32 / 8 % 2 == 0, "an even number of depth passes: the unsorted input goes to buffer 0");

    // Where the depth order comes from (both give the same lists): a sort of the P Gaussians before their instances
    // are emitted -- 14 small launches whose cost follows P -- or a sort of every tile's list afterwards -- two
    // launches whose cost follows the instance count and rises steeply once lists outgrow a workgroup (4096
    // entries).  Measured on MI355X the first costs about 50 us + 45 us per million Gaussians, the second about
    // 12 us + 6 us per million instances while no list passes 4096 entries and the instances carry their depth rank
    // (0.085 ms at 1500 per tile on average and 12 M instances), rising steeply where the longest lists do (2900 per
    // tile on average and 24 M instances).
    const int T = cam.grid_x * cam.grid_y;
    // That model only opens the bidding.  How long a tile's list takes to sort depends on how the scene loads the
    // tiles and how its depths crowd, which the host cannot know -- so the device clocks the binning stages of
    // every frame (RouteHistory) and the choice goes by what each source has actually cost on this thread's recent
    // frames of the same size, scaled by the instance count; the one not in use is tried again now and then (rarely
    // when it was far behind).  Both give the same lists, so the choice never shows in the results.
    Mailbox* mb = mailbox_get();
    RouteHistory* rh = mb ? &mb->route : nullptr;
    if (rh) {
        // (a scene that grows or shrinks, as one does under densification, keeps its history until its size has
        // changed by a factor of two: the figures are per instance)
        if (rh->W != width || rh->H != height || P > 2LL * rh->P || 2LL * P < rh->P) {
            const double rate = rh->ticks_per_us;
            const uint32_t seq = rh->seq;
            memset(rh, 0, sizeof *rh);
            rh->ticks_per_us = rate;
            rh->seq = seq;
            rh->W = width; rh->H = height; rh->P = P;
        }
        const unsigned long long word = __atomic_load_n(mb->host + 1, __ATOMIC_ACQUIRE);
        const uint32_t seq = (uint32_t)(word >> 32);
        auto& pend = rh->pending[seq % NGS_ROUTE_PENDING];
        if (pend.valid && pend.seq == seq && pend.instances > 0.0) {
            const int r = pend.route;
            rh->us[r] = (double)(uint32_t)word / rh->ticks_per_us;
            rh->instances[r] = pend.instances;
            rh->per_instance[r][rh->measured[r]++ % 3] = rh->us[r] / pend.instances;
            pend.valid = false;
        }
    }
    auto sort_gaussians_first = [&](int64_t instances) {
        if (g_depth_order_choice >= 0) return g_depth_order_choice == 1;
        const double inst = (double)instances, per_tile = inst / (double)T;
        const double model[2] = {12.0 + 6.1e-6 * inst * (per_tile > 2600.0 ? 1.0 + (per_tile - 2600.0) / 150.0 : 1.0),
                                 50.0 + 45e-6 * (double)P};
        const int by_model = model[1] < model[0] ? 1 : 0;
        if (!rh || inst <= 0.0) return by_model == 1;
        int pick;
        rh->decisions += rh->decisions < (1 << 30);
        if (rh->measured[0] == 0 && rh->measured[1] == 0) {
            pick = by_model;
        } else if ((rh->measured[0] < 3 || rh->measured[1] < 3) && rh->decisions <= 24) {
            // three frames with either before the figures count (a frame's figure reaches the host a call later, so
            // a source may get a frame or two more than that; and should figures keep getting lost, the trying ends
            // after two dozen frames)
            pick = rh->measured[0] == rh->measured[1] ? by_model : rh->measured[1] < rh->measured[0] ? 1 : 0;
        } else if (rh->measured[0] == 0 || rh->measured[1] == 0) {
            pick = by_model;
        } else {
            // the smallest of the last three measurements: a frame that was held up says nothing about its source
            double est[2];
            for (int r = 0; r < 2; r++) {
                double best_r = rh->per_instance[r][0];
                for (int k = 1; k < 3 && k < rh->measured[r]; k++) best_r = fmin(best_r, rh->per_instance[r][k]);
                est[r] = inst * best_r;
            }
            const int best = est[1] < est[0] ? 1 : 0, other = 1 - best;
            pick = ++rh->age[other] >= (est[other] < 1.25 * est[best] ? 128 : 1024) ? other : best;
        }
        rh->age[pick] = 0;
        return pick == 1;
    };
    int64_t capacity = binning_capacity > 0x7fffffffLL ? 0x7fffffffLL : (int64_t)binning_capacity;
    // With a capacity from the caller the choice is made now, and a frame that takes the per-tile sort does not write
    // the Gaussian-level sort's input (8 bytes per Gaussian).
    const bool sort_input = capacity == 0 || sort_gaussians_first(capacity);

    const int ps = prof_acquire();
    prof_begin(ps, ST_PREPROCESS, st);
    ngs_launch_preprocess_fwd(st, P, D, M, means3D, scales, scale_modifier, rotations, opacities, raw, shs,
                              cov3D_precomp, colors_precomp, cam, prefiltered, radii, rec, depth, rect, tiles, clamped,
                              block_sums, (uint32_t*)(gb + gl.block_depths), misc, sort_input ? dkeys[0] : nullptr,
                              sort_input ? order[0] : nullptr);
    prof_end(ps, ST_PREPROCESS, st);
    NGS_STAGE("preprocess");
    prof_begin(ps, ST_SCAN, st);
    if (mb) __atomic_store_n(mb->host, ~0ull, __ATOMIC_RELEASE);
    ngs_launch_scan_block_sums(st, nblocks, block_sums, block_offs, (const uint32_t*)(gb + gl.block_depths), misc,
                               mb ? mb->dev : nullptr);
    prof_end(ps, ST_SCAN, st);
    NGS_STAGE("scan");

    // The host needs the instance count: to size the binning buffer when the caller gave no capacity (the reference
    // blocks on a device-to-host copy at the same point, rasterizer_impl.cu:281), and for the return value either
    // way.  The scan kernel posts the 8 bytes (count, error flags) straight into pinned host memory and the host
    // spins on them; if they do not arrive (a failed kernel, a platform without coherent host memory) the wait
    // falls back to a copy and a stream synchronisation.  With a capacity the wait comes AFTER every stage has been
    // queued, so the GPU never idles while the host learns the count.
    uint32_t host_misc[2] = {0, 0};
    bool count_known = false;
    auto learn_count = [&]() -> int {
        if (!mailbox_wait(mb, st, host_misc)) {
            NGS_HIP(hipMemcpyAsync(host_misc, misc, sizeof host_misc, hipMemcpyDeviceToHost, st));
            NGS_HIP(hipStreamSynchronize(st));
        }
        count_known = true;
        if (host_misc[1] & 1u) return fail(fn, "Point is filtered although prefiltered is set. This shouldn't happen!");
        if ((host_misc[1] & 2u) || host_misc[0] > 0x7fffffffu) return fail(fn, "more than 2^31-1 tile instances");
        return 0;
    };
    const bool early_choice = capacity != 0;
    if (capacity == 0) {
        if (learn_count()) return 1;
        capacity = (int64_t)host_misc[0];
    }

    const int tile_bits = (int)higher_msb((uint32_t)T);
    const int in_buf = ngs_sort_input_buffer(tile_bits);
    uint2* ranges = (uint2*)(ib + il.ranges);
    const size_t ranges_clear = il.hard + 16 - il.ranges;  // the ranges and, behind them, the work lists' counters
    // (made before the first kernel when the caller gave a capacity: the choice must not change once the per-Gaussian
    // kernel has or has not written the Gaussian-level sort's input)
    const bool depth_first = early_choice ? sort_input : sort_gaussians_first(capacity);
    g_depth_order_taken = depth_first ? 1 : 0;
    // Instances emitted in index order carry a depth rank beside the tile id of their key and the Gaussian index of
    // their list word, so that the per-tile sort reads it with the list instead of gathering a depth per entry.
    DepthRank dr = ngs_depth_rank(P, tile_bits, !depth_first && g_depth_order_choice != 3);
    if (g_depth_order_choice == 2) dr.min_shift = 20;
    const int qbits = dr.key_bits;
    if (depth_first) {
        // Gaussians in depth order (ties in index order: the passes are stable), then their tile counts, rectangles
        // and instance offsets in that order.
        prof_begin(ps, ST_DEPTHSORT, st);
        ngs_launch_sort_tile_bits(st, nullptr, P, dkeys, order, (uint32_t*)(gb + gl.dhist), (uint32_t*)(gb + gl.dtotals),
                                  32);
        ngs_launch_depth_order(st, P, order[0], dkeys[0], rect, s_tiles, s_rect, s_block_sums, s_block_offs);
        prof_end(ps, ST_DEPTHSORT, st);
        NGS_STAGE("depth order");
    }
    for (;;) {
        const BinLayout bl = ngs_bin_layout((size_t)capacity);
        char* bb = binning_alloc(binning_ctx, bl.total);
        if (!bb) return fail(fn, "binning allocator returned NULL");
        bb = ngs_align_ptr(bb);
        uint32_t* tkeys[2] = {(uint32_t*)(bb + bl.keys[0]), (uint32_t*)(bb + bl.keys[1])};
        uint32_t* vals[2] = {(uint32_t*)(bb + bl.vals[0]), (uint32_t*)(bb + bl.vals[1])};
        uint32_t* hist = (uint32_t*)(bb + bl.hist);
        uint32_t* totals = (uint32_t*)(bb + bl.totals);
        if (capacity > 0) {
            prof_begin(ps, ST_DUPLICATE, st);
            if (depth_first)
                ngs_launch_duplicate(st, P, s_rect, s_tiles, s_block_offs, order[0], nullptr, dr, cam.grid_x,
                                     tkeys[in_buf], vals[in_buf], capacity, misc, ranges, ranges_clear);
            else
                ngs_launch_duplicate(st, P, rect, tiles, block_offs, nullptr, depth, dr, cam.grid_x, tkeys[in_buf],
                                     vals[in_buf], capacity, misc, ranges, ranges_clear);
            prof_end(ps, ST_DUPLICATE, st);
            NGS_STAGE("duplicateWithKeys");
            prof_begin(ps, ST_SORT, st);
            ngs_launch_sort_tile_bits(st, misc, capacity, tkeys, vals, hist, totals, tile_bits, qbits);
            prof_end(ps, ST_SORT, st);
            NGS_STAGE("sort");
        }
        prof_begin(ps, ST_RANGES, st);
        // (cleared by the emission kernel when that ran: the ranges and the counters of the per-tile sort's work lists)
        ngs_launch_tile_ranges(st, misc, capacity, tkeys[0], qbits, T, ranges, capacity > 0 ? 0 : ranges_clear);
        prof_end(ps, ST_RANGES, st);
        NGS_STAGE("identifyTileRanges");
        if (capacity > 0 && !depth_first) {
            prof_begin(ps, ST_SEGSORT, st);
            ngs_launch_sort_segments(st, T, ranges, depth, tkeys, vals, (uint32_t*)(ib + il.hard), dr, misc, (uint2*)hist,
                                     (uint32_t)(ngs_sort_blocks((size_t)capacity) * 128));
            prof_end(ps, ST_SEGSORT, st);
            NGS_STAGE("sort (depth)");
        }
        prof_begin(ps, ST_RENDER_FWD, st);
        // (a frame whose stages run twice because the estimate was too small is not a measurement)
        auto* pend = rh && capacity > 0 ? &rh->pending[++rh->seq % NGS_ROUTE_PENDING] : nullptr;
        if (pend) {
            pend->seq = rh->seq;
            pend->route = depth_first ? 1 : 0;
            pend->instances = count_known ? (double)host_misc[0] : 0.0;
            pend->valid = true;
        }
        ngs_launch_render_fwd(st, cam, ranges, vals[0], rec, background, (float*)(ib + il.final_T),
                              (uint32_t*)(ib + il.n_contrib), out_color, (float*)(gb + gl.acc), P, misc,
                              pend ? mb->dev + 1 : nullptr, pend ? pend->seq : 0u, (uint8_t*)(bb + bl.quad),
                              (uint32_t)(bl.quad / NGS_ALIGN));
        prof_end(ps, ST_RENDER_FWD, st);
        NGS_STAGE("render");
        if (count_known) break;
        if (learn_count()) return 1;
        if ((int64_t)host_misc[0] <= capacity) {
            if (pend && pend->seq == rh->seq) pend->instances = (double)host_misc[0];
            break;
        }
        if (pend) pend->valid = false;
        // The estimate was too small: the stages above worked on a truncated list (inside the buffer, so harmlessly).
        // Queue them again with a buffer that holds every instance -- after they have drained: the allocator is about
        // to be asked again, and one that frees or moves the first buffer at once (not in stream order) must not pull
        // it from under kernels that are still queued.  This path is the slow one anyway.
        NGS_HIP(hipStreamSynchronize(st));
        capacity = (int64_t)host_misc[0];
    }
    if (num_rendered) *num_rendered = (int)host_misc[0];
    return 0;
}

extern "C" int ngs_rasterize_forward(ngs_alloc_fn geometry_alloc, void* geometry_ctx, ngs_alloc_fn binning_alloc,
                                     void* binning_ctx, ngs_alloc_fn image_alloc, void* image_ctx, int P, int D, int M,
                                     const float* background, int width, int height, const float* means3D,
                                     const float* shs, const float* colors_precomp, const float* opacities,
                                     const float* scales, float scale_modifier, const float* rotations,
                                     const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                                     const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered,
                                     float* out_color, int* radii, int debug, void* stream, int* num_rendered)
{
    return forward_impl("ngs_rasterize_forward", geometry_alloc, geometry_ctx, binning_alloc, binning_ctx, image_alloc,
                        image_ctx, P, D, M, background, width, height, means3D, shs, colors_precomp, opacities, scales,
                        scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy,
                        prefiltered, out_color, radii, debug, stream, num_rendered, 0);
}

extern "C" int ngs_rasterize_forward_presized(ngs_alloc_fn geometry_alloc, void* geometry_ctx,
                                              ngs_alloc_fn binning_alloc, void* binning_ctx, ngs_alloc_fn image_alloc,
                                              void* image_ctx, int P, int D, int M, const float* background, int width,
                                              int height, const float* means3D, const float* shs,
                                              const float* colors_precomp, const float* opacities, const float* scales,
                                              float scale_modifier, const float* rotations,
                                              const float* cov3D_precomp, const float* viewmatrix,
                                              const float* projmatrix, const float* cam_pos, float tan_fovx,
                                              float tan_fovy, int prefiltered, float* out_color, int* radii, int debug,
                                              void* stream, int* num_rendered, long long binning_capacity)
{
    return forward_impl("ngs_rasterize_forward_presized", geometry_alloc, geometry_ctx, binning_alloc, binning_ctx,
                        image_alloc, image_ctx, P, D, M, background, width, height, means3D, shs, colors_precomp,
                        opacities, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos,
                        tan_fovx, tan_fovy, prefiltered, out_color, radii, debug, stream, num_rendered,
                        binning_capacity);
}

extern "C" int ngs_rasterize_forward_raw(ngs_alloc_fn geometry_alloc, void* geometry_ctx, ngs_alloc_fn binning_alloc,
                                         void* binning_ctx, ngs_alloc_fn image_alloc, void* image_ctx, int P, int D,
                                         int M, const float* background, int width, int height, const float* means3D,
                                         const float* shs, const float* colors_precomp, const float* opacities_raw,
                                         const float* scales_raw, float scale_modifier, const float* rotations_raw,
                                         const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                                         float tan_fovx, float tan_fovy, int prefiltered, float* out_color, int* radii,
                                         int debug, void* stream, int* num_rendered, long long binning_capacity)
{
    return forward_impl("ngs_rasterize_forward_raw", geometry_alloc, geometry_ctx, binning_alloc, binning_ctx,
                        image_alloc, image_ctx, P, D, M, background, width, height, means3D, shs, colors_precomp,
                        opacities_raw, scales_raw, scale_modifier, rotations_raw, nullptr, viewmatrix, projmatrix,
                        cam_pos, tan_fovx, tan_fovy, prefiltered, out_color, radii, debug, stream, num_rendered,
                        binning_capacity, 1);
}

// Optional hand-over between the two backward stages (ngs_rasterize_backward_early_color).
struct EarlyColor {
    float* out;
    void (*ready)(void*);
    void* ctx;
};

// The per-Gaussian stage in pieces (ngs_rasterize_backward_chunked)
struct ChunkPlan {
    int n;
    const ngs_gradient_chunk* chunks;
    void (*ready)(void*, int);
    void* ctx;
};

static int backward_impl(const char* fn, int P, int D, int M, int R, const float* background, int width, int height,
                         const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                         float scale_modifier, const float* rotations, const float* cov3D_precomp,
                         const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                         float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
                         const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                         float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                         float* dL_drot, int debug, void* stream, const ShAdamArgs* sh_adam,
                         const EarlyColor* early = nullptr, const float* opacities_raw = nullptr,
                         const GeoAdamArgs* geo = nullptr, const ChunkPlan* plan = nullptr)
{
    g_err[0] = 0;
    hipStream_t st = (hipStream_t)stream;
    (void)colors_precomp;
    if (P < 0 || P > NGS_MAX_GAUSSIANS || R < 0 || width <= 0 || height <= 0) return fail(fn, "bad sizes");
    if (P == 0) return 0;
    if (!geom_buffer || !binning_buffer || !image_buffer) return fail(fn, "missing state buffer");
    if (!means3D || !viewmatrix || !projmatrix || !background || !dL_dpix) return fail(fn, "missing required pointer");
    // the fused optimiser step writes no gradient
    if (!geo && (!dL_dmean2D || !dL_dopacity || !dL_dcolor || !dL_dmean3D || !dL_dcov3D || !dL_dscale || !dL_drot))
        return fail(fn, "missing required pointer");
    if (opacities_raw && cov3D_precomp) return fail(fn, "raw parameters and cov3D_precomp exclude each other");
    if (shs && !campos) return fail(fn, "shs given without campos");
    // the forward pass checks the same (the kernels index sh[0 .. (D+1)^2) of an M-wide row)
    if (shs && (D < 0 || D > 3 || (D + 1) * (D + 1) > M)) return fail(fn, "SH degree does not fit the coefficient count");
    if (!cov3D_precomp && (!scales || !rotations)) return fail(fn, "need scales+rotations or cov3D_precomp");
    // the per-Gaussian kernel moves quaternions and their gradients (and the conic gradient) as 16-byte vectors
    if ((((uintptr_t)dL_drot | (uintptr_t)dL_dconic) % 16 != 0) || (!cov3D_precomp && (uintptr_t)rotations % 16 != 0))
        return fail(fn, "rotations, dL_drot and dL_dconic must be 16-byte aligned");
    const NgsCamera cam = make_camera(width, height, tan_fovx, tan_fovy, viewmatrix, projmatrix, campos);
    const GeomLayout gl = ngs_geom_layout((size_t)P);
    const BinLayout bl = ngs_bin_layout((size_t)R);  // the point list sits at the same place for every capacity
    const ImageLayout il = ngs_image_layout(width, height);
    char* gb = ngs_align_ptr(geom_buffer);
    char* bb = ngs_align_ptr(binning_buffer);
    char* ib = ngs_align_ptr(image_buffer);
    if (!radii) radii = (const int*)(gb + gl.radii);
    float* acc = (float*)(gb + gl.acc);

    const int ps = prof_acquire();
    prof_begin(ps, ST_BWD_ZERO, st);
    uint32_t* misc = (uint32_t*)(gb + gl.misc);
    ngs_launch_zero_acc_unless_clean(st, acc, P, misc);  // normally nothing to do: the forward pass left them clear
    prof_end(ps, ST_BWD_ZERO, st);
    prof_begin(ps, ST_RENDER_BWD, st);
    ngs_launch_render_bwd(st, cam, (const uint2*)(ib + il.ranges), (const uint32_t*)(bb + bl.vals[0]),
                          (const float4*)(gb + gl.rec),
                          background, (const float*)(ib + il.final_T), (const uint32_t*)(ib + il.n_contrib), dL_dpix,
                          acc, misc);
    prof_end(ps, ST_RENDER_BWD, st);
    NGS_STAGE("render backward");
    if (early && early->out) {
        ngs_launch_masked_color_from_acc(st, P, (const uint8_t*)(gb + gl.clamped), acc, early->out);
        NGS_STAGE("early colour gradient");
        if (early->ready) early->ready(early->ctx);
    }
    prof_begin(ps, ST_PREPROCESS_BWD, st);
    // scales == NULL (precomputed covariance) skips the scale/rotation gradients, shs == NULL the SH ones,
    // as in the reference (backward.cu:390-395).
    if (!plan) {
        ngs_launch_preprocess_bwd(st, P, D, M, means3D, radii, shs, (const uint8_t*)(gb + gl.clamped),
                                  cov3D_precomp ? nullptr : scales, rotations, opacities_raw, scale_modifier,
                                  cov3D_precomp, cam, acc, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D,
                                  dL_dcov3D, dL_dsh, dL_dscale, dL_drot, sh_adam, geo);
    } else {
        // The same kernel over one run of Gaussians after the other: every per-Gaussian array is indexed from the
        // pointer it is given, so a chunk is a launch on pointers moved to the chunk's first row.  Chunks start on
        // workgroup boundaries: each workgroup sees the Gaussians it would see in the single launch.
        for (int k = 0; k < plan->n; k++) {
            const ngs_gradient_chunk& c = plan->chunks[k];
            const size_t f = (size_t)c.first;
            auto at = [&](const float* p, size_t per) { return p ? p + per * f : nullptr; };
            auto to = [&](float* redirected, float* whole, size_t per) {
                return redirected ? redirected : whole ? whole + per * f : nullptr;
            };
            ngs_launch_preprocess_bwd(st, c.count, D, M, means3D + 3 * f, radii + f, at(shs, (size_t)3 * M),
                                      (const uint8_t*)(gb + gl.clamped) + f, cov3D_precomp ? nullptr : scales + 3 * f,
                                      at(rotations, 4), at(opacities_raw, 1), scale_modifier, at(cov3D_precomp, 6), cam,
                                      acc + (size_t)NGS_ACC_STRIDE * f, dL_dmean2D + 3 * f, to(nullptr, dL_dconic, 4),
                                      to(c.dL_dopacity, dL_dopacity, 1), dL_dcolor + 3 * f,
                                      to(c.dL_dmean3D, dL_dmean3D, 3), dL_dcov3D + 6 * f,
                                      to(c.dL_dsh, dL_dsh, (size_t)3 * M), to(c.dL_dscale, dL_dscale, 3),
                                      to(c.dL_drot, dL_drot, 4), nullptr, nullptr);
            NGS_STAGE("preprocess backward");
            if (plan->ready) plan->ready(plan->ctx, k);
        }
    }
    prof_end(ps, ST_PREPROCESS_BWD, st);
    NGS_STAGE("preprocess backward");
    return 0;
}

extern "C" int ngs_rasterize_backward(int P, int D, int M, int R, const float* background, int width, int height,
                                      const float* means3D, const float* shs, const float* colors_precomp,
                                      const float* scales, float scale_modifier, const float* rotations,
                                      const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                                      const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                                      char* geom_buffer, char* binning_buffer, char* image_buffer,
                                      const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                                      float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                                      float* dL_dscale, float* dL_drot, int debug, void* stream)
{
    return backward_impl("ngs_rasterize_backward", P, D, M, R, background, width, height, means3D, shs, colors_precomp,
                         scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx,
                         tan_fovy, radii, geom_buffer, binning_buffer, image_buffer, dL_dpix, dL_dmean2D, dL_dconic,
                         dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, debug, stream,
                         nullptr);
}

extern "C" int ngs_rasterize_backward_early_color(
    int P, int D, int M, int R, const float* background, int width, int height, const float* means3D, const float* shs,
    const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
    const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
    float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer, const float* dL_dpix,
    float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D,
    float* dL_dsh, float* dL_dscale, float* dL_drot, int debug, void* stream, float* masked_color,
    void (*color_ready)(void*), void* color_ready_ctx)
{
    if (!masked_color) {
        g_err[0] = 0;
        return fail("ngs_rasterize_backward_early_color", "missing masked_color");
    }
    const EarlyColor early = {masked_color, color_ready, color_ready_ctx};
    return backward_impl("ngs_rasterize_backward_early_color", P, D, M, R, background, width, height, means3D, shs,
                         colors_precomp, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix,
                         campos, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer, image_buffer, dL_dpix,
                         dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale,
                         dL_drot, debug, stream, nullptr, &early);
}

extern "C" int ngs_rasterize_backward_chunked(
    int P, int D, int M, int R, const float* background, int width, int height, const float* means3D, const float* shs,
    const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
    const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
    float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer, const float* dL_dpix,
    float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D,
    float* dL_dsh, float* dL_dscale, float* dL_drot, int debug, void* stream, float* masked_color,
    void (*color_ready)(void*), void* color_ready_ctx, int n_chunks, const ngs_gradient_chunk* chunks,
    void (*chunk_ready)(void*, int), void* chunk_ready_ctx)
{
    const char* fn = "ngs_rasterize_backward_chunked";
    g_err[0] = 0;
    if (n_chunks < 1 || !chunks) return fail(fn, "missing chunks");
    long long next = 0;
    for (int k = 0; k < n_chunks; k++) {
        const ngs_gradient_chunk& c = chunks[k];
        if (c.first != next || c.count < 0 || (c.count == 0 && P != 0)) return fail(fn, "chunks must tile [0, P) in order");
        if (c.first % NGS_GAUSS_BLOCK != 0) return fail(fn, "a chunk must start at a multiple of 256 Gaussians");
        if ((uintptr_t)c.dL_drot % 16 != 0 || (uintptr_t)c.dL_dsh % 16 != 0)
            return fail(fn, "a chunk's dL_drot and dL_dsh must be 16-byte aligned");
        // an array that no chunk redirects must exist as a whole
        if ((!c.dL_dmean3D && !dL_dmean3D) || (!c.dL_dopacity && !dL_dopacity) || (!c.dL_dscale && !dL_dscale) ||
            (!c.dL_drot && !dL_drot))
            return fail(fn, "missing required pointer");
        next += c.count;
    }
    if (next != P) return fail(fn, "chunks must tile [0, P) in order");
    // (backward_impl insists on the whole arrays; a caller that redirects every chunk may leave them out)
    float* const stand_in = (float*)16;  // never dereferenced: every chunk brings its own pointer
    auto whole = [&](float* p, bool all_redirected) { return p ? p : all_redirected ? stand_in : nullptr; };
    bool all[4] = {true, true, true, true};
    for (int k = 0; k < n_chunks; k++) {
        all[0] = all[0] && chunks[k].dL_dmean3D; all[1] = all[1] && chunks[k].dL_dopacity;
        all[2] = all[2] && chunks[k].dL_dscale; all[3] = all[3] && chunks[k].dL_drot;
    }
    const EarlyColor early = {masked_color, color_ready, color_ready_ctx};
    const ChunkPlan plan = {n_chunks, chunks, chunk_ready, chunk_ready_ctx};
    return backward_impl(fn, P, D, M, R, background, width, height, means3D, shs, colors_precomp, scales,
                         scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy,
                         radii, geom_buffer, binning_buffer, image_buffer, dL_dpix, dL_dmean2D, dL_dconic,
                         whole(dL_dopacity, all[1]), dL_dcolor, whole(dL_dmean3D, all[0]), dL_dcov3D, dL_dsh,
                         whole(dL_dscale, all[2]), whole(dL_drot, all[3]), debug, stream, nullptr,
                         masked_color ? &early : nullptr, nullptr, nullptr, &plan);
}

extern "C" int ngs_rasterize_backward_sh_adam(int P, int D, int M, int R, const float* background, int width,
                                              int height, const float* means3D, float* shs, const float* scales,
                                              float scale_modifier, const float* rotations,
                                              const float* cov3D_precomp, const float* viewmatrix,
                                              const float* projmatrix, const float* campos, float tan_fovx,
                                              float tan_fovy, const int* radii, char* geom_buffer,
                                              char* binning_buffer, char* image_buffer, const float* dL_dpix,
                                              float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                                              float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dscale,
                                              float* dL_drot, float* sh_exp_avg, float* sh_exp_avg_sq, double lr_dc,
                                              double lr_rest, double beta1, double beta2, double eps, int step,
                                              float grad_scale, int debug, void* stream)
{
    const char* fn = "ngs_rasterize_backward_sh_adam";
    g_err[0] = 0;
    if (!shs || !sh_exp_avg || !sh_exp_avg_sq) return fail(fn, "missing SH coefficients or optimiser state");
    if (M != 16) return fail(fn, "the fused update needs rows of 16 coefficients");
    if (((uintptr_t)shs | (uintptr_t)sh_exp_avg | (uintptr_t)sh_exp_avg_sq) % 16 != 0)
        return fail(fn, "SH coefficients and optimiser state must be 16-byte aligned");
    if (!(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0 && step >= 1))
        return fail(fn, "bad Adam constants or step < 1");
    ShAdamArgs a;
    a.param = shs;
    a.exp_avg = sh_exp_avg;
    a.exp_avg_sq = sh_exp_avg_sq;
    a.step_head = ngs_adam_step_size(lr_dc, beta1, step);
    a.step_tail = ngs_adam_step_size(lr_rest, beta1, step);
    a.k = ngs_adam_const(beta1, beta2, eps, step, grad_scale);
    return backward_impl(fn, P, D, M, R, background, width, height, means3D, shs, nullptr, scales, scale_modifier,
                         rotations, cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii,
                         geom_buffer, binning_buffer, image_buffer, dL_dpix, dL_dmean2D, dL_dconic, dL_dopacity,
                         dL_dcolor, dL_dmean3D, dL_dcov3D, nullptr, dL_dscale, dL_drot, debug, stream, &a);
}

extern "C" int ngs_rasterize_backward_raw(int P, int D, int M, int R, const float* background, int width, int height,
                                          const float* means3D, const float* shs, const float* colors_precomp,
                                          const float* opacities_raw, const float* scales_raw, float scale_modifier,
                                          const float* rotations_raw, const float* viewmatrix,
                                          const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy,
                                          const int* radii, char* geom_buffer, char* binning_buffer,
                                          char* image_buffer, const float* dL_dpix, float* dL_dmean2D,
                                          float* dL_dconic, float* dL_dopacity_raw, float* dL_dcolor,
                                          float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale_raw,
                                          float* dL_drot_raw, int debug, void* stream)
{
    const char* fn = "ngs_rasterize_backward_raw";
    if (!opacities_raw || !scales_raw || !rotations_raw) {
        g_err[0] = 0;
        return fail(fn, "missing raw opacities, scales or rotations");
    }
    return backward_impl(fn, P, D, M, R, background, width, height, means3D, shs, colors_precomp, scales_raw,
                         scale_modifier, rotations_raw, nullptr, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy,
                         radii, geom_buffer, binning_buffer, image_buffer, dL_dpix, dL_dmean2D, dL_dconic,
                         dL_dopacity_raw, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale_raw, dL_drot_raw, debug,
                         stream, nullptr, nullptr, opacities_raw);
}

static AdamGroupArgs adam_group(const ngs_adam_group& g, double beta1, double beta2)
{
    AdamGroupArgs a;
    a.param = g.param;
    a.exp_avg = g.exp_avg;
    a.exp_avg_sq = g.exp_avg_sq;
    a.step = ngs_adam_step_size(g.lr, beta1, g.step);
    a.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)g.step));
    return a;
}

extern "C" int ngs_rasterize_backward_train(int P, int D, int R, const float* background, int width, int height,
                                            float scale_modifier, const float* viewmatrix, const float* projmatrix,
                                            const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                                            char* geom_buffer, char* binning_buffer, char* image_buffer,
                                            const float* dL_dpix, const ngs_train_state* state, int debug, void* stream)
{
    const char* fn = "ngs_rasterize_backward_train";
    g_err[0] = 0;
    if (!state) return fail(fn, "missing optimiser state");
    const ngs_adam_group* groups[5] = {&state->xyz, &state->features, &state->opacity, &state->scaling,
                                       &state->rotation};
    uintptr_t bits = 0;
    for (const ngs_adam_group* g : groups) {
        if (!g->param || !g->exp_avg || !g->exp_avg_sq) return fail(fn, "missing parameter or optimiser state");
        if (!(g->lr >= 0.0) || g->step < 1) return fail(fn, "bad learning rate or step < 1");
        bits |= (uintptr_t)g->param | (uintptr_t)g->exp_avg | (uintptr_t)g->exp_avg_sq;
    }
    if (!(state->features_rest_lr >= 0.0)) return fail(fn, "bad learning rate or step < 1");
    // SH rows and quaternions move as 16-byte vectors; the other arrays are offered the same way by every allocator
    if (bits % 16 != 0) return fail(fn, "parameters and optimiser state must be 16-byte aligned");
    if (!(state->beta1 >= 0.0 && state->beta1 < 1.0 && state->beta2 >= 0.0 && state->beta2 < 1.0 && state->eps >= 0.0))
        return fail(fn, "bad Adam constants");
    const bool any_stat = state->xyz_gradient_accum || state->denom || state->max_radii2D;
    if (any_stat && !(state->xyz_gradient_accum && state->denom && state->max_radii2D))
        return fail(fn, "densification statistics need xyz_gradient_accum, denom and max_radii2D together");
    ShAdamArgs a;
    a.param = state->features.param;
    a.exp_avg = state->features.exp_avg;
    a.exp_avg_sq = state->features.exp_avg_sq;
    a.step_head = ngs_adam_step_size(state->features.lr, state->beta1, state->features.step);
    a.step_tail = ngs_adam_step_size(state->features_rest_lr, state->beta1, state->features.step);
    a.k = ngs_adam_const(state->beta1, state->beta2, state->eps, state->features.step, state->grad_scale);
    GeoAdamArgs geo;
    geo.xyz = adam_group(state->xyz, state->beta1, state->beta2);
    geo.opacity = adam_group(state->opacity, state->beta1, state->beta2);
    geo.scaling = adam_group(state->scaling, state->beta1, state->beta2);
    geo.rotation = adam_group(state->rotation, state->beta1, state->beta2);
    geo.k = a.k;
    geo.grad_accum = state->xyz_gradient_accum;
    geo.denom = state->denom;
    geo.max_radii = state->max_radii2D;
    return backward_impl(fn, P, D, 16, R, background, width, height, state->xyz.param, state->features.param, nullptr,
                         state->scaling.param, scale_modifier, state->rotation.param, nullptr, viewmatrix, projmatrix,
                         campos, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer, image_buffer, dL_dpix,
                         nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, debug, stream,
                         &a, nullptr, state->opacity.param, &geo);
}

extern "C" int ngs_masked_color_gradient(int P, const char* geom_buffer, const float* dL_dcolor, float* out,
                                         void* stream)
{
    g_err[0] = 0;
    if (P < 0) return fail("ngs_masked_color_gradient", "bad size");
    if (P == 0) return 0;
    if (!geom_buffer || !dL_dcolor || !out) return fail("ngs_masked_color_gradient", "missing required pointer");
    const GeomLayout gl = ngs_geom_layout((size_t)P);
    const char* gb = ngs_align_ptr((char*)geom_buffer);
    const int ps = prof_acquire();
    prof_begin(ps, ST_SH_VIEWS, (hipStream_t)stream);
    ngs_launch_masked_color((hipStream_t)stream, P, (const uint8_t*)(gb + gl.clamped), dL_dcolor, out);
    prof_end(ps, ST_SH_VIEWS, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("stage masked colour gradient", hipGetErrorString(e));
    return 0;
}

extern "C" int ngs_sh_gradient_from_views(int P, int D, int M, int V, const float* means3D, const float* campos,
                                          long long campos_stride, const float* dL_dcolor_views,
                                          long long view_stride, float* dL_dsh, void* stream)
{
    g_err[0] = 0;
    if (P < 0 || P > NGS_MAX_GAUSSIANS || V < 1) return fail("ngs_sh_gradient_from_views", "bad sizes");
    if (D < 0 || D > 3 || (D + 1) * (D + 1) > M)
        return fail("ngs_sh_gradient_from_views", "SH degree does not fit the coefficient count");
    if (P == 0) return 0;
    if (!means3D || !campos || !dL_dcolor_views || !dL_dsh)
        return fail("ngs_sh_gradient_from_views", "missing required pointer");
    if (campos_stride < 3 || view_stride < 3LL * P)
        return fail("ngs_sh_gradient_from_views", "view strides shorter than one view's data");
    const int ps = prof_acquire();
    prof_begin(ps, ST_SH_VIEWS, (hipStream_t)stream);
    ngs_launch_sh_from_views((hipStream_t)stream, P, D, M, V, means3D, campos, campos_stride, dL_dcolor_views,
                             view_stride, dL_dsh, nullptr);
    prof_end(ps, ST_SH_VIEWS, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("stage SH gradient from views", hipGetErrorString(e));
    return 0;
}

extern "C" size_t ngs_loss_workspace_bytes(int width, int height)
{
    return width > 0 && height > 0 ? ngs_loss_workspace_floats(width, height) * sizeof(float) : 0;
}

extern "C" int ngs_l1_ssim_loss(int width, int height, float lambda_dssim, const float* image, const float* gt_image,
                                float* workspace, float* loss_terms, float* dL_dimage, void* stream)
{
    g_err[0] = 0;
    if (width <= 0 || height <= 0) return fail("ngs_l1_ssim_loss", "bad sizes");
    if (!image || !gt_image || !workspace || !loss_terms || !dL_dimage)
        return fail("ngs_l1_ssim_loss", "missing required pointer");
    if ((uintptr_t)workspace % 8 != 0) return fail("ngs_l1_ssim_loss", "workspace must be 8-byte aligned");
    const int ps = prof_acquire();
    prof_begin(ps, ST_LOSS, (hipStream_t)stream);
    ngs_launch_l1_ssim((hipStream_t)stream, width, height, lambda_dssim, image, gt_image, workspace, loss_terms,
                       dL_dimage);
    prof_end(ps, ST_LOSS, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("stage image loss", hipGetErrorString(e));
    return 0;
}

static bool adam_args_ok(double beta1, double beta2, double eps, int step)
{
    return beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0 && step >= 1;
}

extern "C" int ngs_adam_step(long long n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                             double lr_head, double lr_tail, int period, int head, double beta1, double beta2,
                             double eps, int step, float grad_scale, void* stream)
{
    g_err[0] = 0;
    if (n < 0 || period < 1 || head < 0 || head > period) return fail("ngs_adam_step", "bad sizes");
    if (!adam_args_ok(beta1, beta2, eps, step)) return fail("ngs_adam_step", "bad Adam constants or step < 1");
    if (n == 0) return 0;
    if (!param || !grad || !exp_avg || !exp_avg_sq) return fail("ngs_adam_step", "missing required pointer");
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 != 0)
        return fail("ngs_adam_step", "arrays must be 16-byte aligned");
    const int ps = prof_acquire();
    prof_begin(ps, ST_ADAM_FLAT, (hipStream_t)stream);
    ngs_launch_adam_flat((hipStream_t)stream, n, param, grad, exp_avg, exp_avg_sq, lr_head, lr_tail, period, head,
                         beta1, beta2, eps, step, grad_scale);
    prof_end(ps, ST_ADAM_FLAT, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("stage Adam", hipGetErrorString(e));
    return 0;
}

extern "C" int ngs_adam_step_activated(int P, float* const raw[3], const float* const grad[3],
                                       float* const exp_avg[3], float* const exp_avg_sq[3],
                                       float* const activated[3], const double lr[3], double beta1, double beta2,
                                       double eps, const int step[3], float grad_scale, const float* dL_dmean2D,
                                       const int* radii, float* xyz_gradient_accum, float* denom, int* max_radii2D,
                                       void* stream)
{
    g_err[0] = 0;
    if (P < 0) return fail("ngs_adam_step_activated", "bad size");
    if (!step) return fail("ngs_adam_step_activated", "missing required pointer");
    for (int i = 0; i < 3; i++)
        if (!adam_args_ok(beta1, beta2, eps, step[i]))
            return fail("ngs_adam_step_activated", "bad Adam constants or step < 1");
    if (P == 0) return 0;
    if (!raw || !grad || !exp_avg || !exp_avg_sq || !activated || !lr)
        return fail("ngs_adam_step_activated", "missing required pointer");
    for (int i = 0; i < 3; i++)
        if (!raw[i] || !grad[i] || !exp_avg[i] || !exp_avg_sq[i] || !activated[i])
            return fail("ngs_adam_step_activated", "missing required pointer");
    if (((uintptr_t)raw[2] | (uintptr_t)grad[2] | (uintptr_t)exp_avg[2] | (uintptr_t)exp_avg_sq[2] |
         (uintptr_t)activated[2]) % 16 != 0)
        return fail("ngs_adam_step_activated", "rotation arrays must be 16-byte aligned");
    if (xyz_gradient_accum && (!dL_dmean2D || !radii || !denom || !max_radii2D))
        return fail("ngs_adam_step_activated", "densification statistics need dL_dmean2D, radii, denom, max_radii2D");
    const int ps = prof_acquire();
    prof_begin(ps, ST_ADAM_ACTIVATED, (hipStream_t)stream);
    ngs_launch_adam_activated((hipStream_t)stream, P, raw, grad, exp_avg, exp_avg_sq, activated, lr, beta1, beta2, eps,
                              step, grad_scale, dL_dmean2D, radii, xyz_gradient_accum, denom, max_radii2D);
    prof_end(ps, ST_ADAM_ACTIVATED, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("stage Adam (activated groups)", hipGetErrorString(e));
    return 0;
}

extern "C" int ngs_view_statistics(int P, const float* dL_dmean2D, const int* radii, float* norm_out, float* vis_out,
                                   int* max_radii2D, void* stream)
{
    g_err[0] = 0;
    if (P < 0) return fail("ngs_view_statistics", "bad size");
    if (P == 0) return 0;
    if (!dL_dmean2D || !radii || !norm_out || !vis_out) return fail("ngs_view_statistics", "missing required pointer");
    ngs_launch_view_statistics((hipStream_t)stream, P, dL_dmean2D, radii, norm_out, vis_out, max_radii2D);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("stage view statistics", hipGetErrorString(e));
    return 0;
}

extern "C" int ngs_activate_parameters(int P, const float* opacities_raw, const float* scales_raw,
                                       const float* rotations_raw, float* opacities, float* scales, float* rotations,
                                       void* stream)
{
    g_err[0] = 0;
    if (P < 0) return fail("ngs_activate_parameters", "bad size");
    if (P == 0) return 0;
    if ((opacities && !opacities_raw) || (scales && !scales_raw) || (rotations && !rotations_raw))
        return fail("ngs_activate_parameters", "missing required pointer");
    ngs_launch_activate((hipStream_t)stream, P, opacities_raw, scales_raw, rotations_raw, opacities, scales, rotations);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("stage activations", hipGetErrorString(e));
    return 0;
}

extern "C" size_t ngs_idxmap_workspace_bytes(int width, int height)
{
    return width > 0 && height > 0 ? ngs_idxmap_workspace_size(width, height) : 0;
}

extern "C" int ngs_idxmap_forward(int P, int M, int width, int height, const float* means3D, const float* normal,
                                  const float* scales, const float* rotations, const float* viewmatrix,
                                  const float* projmatrix, float tan_fovx, float tan_fovy,
                                  const float* feature_vector, const float* sh, const float* campos, int* out_idx,
                                  float* out_color, float* out_depth, float* out_feature, int* radii, char* workspace,
                                  int debug, void* stream, int* num_rendered)
{
    g_err[0] = 0;
    hipStream_t st = (hipStream_t)stream;
    // accepted for signature parity with the reference, which does not read them either (rasterizer_impl.cu:851-984)
    (void)M; (void)normal; (void)scales; (void)rotations; (void)tan_fovx; (void)tan_fovy; (void)sh;
    if (num_rendered) *num_rendered = 0;
    if (P < 0 || P > NGS_MAX_GAUSSIANS || width <= 0 || height <= 0) return fail("ngs_idxmap_forward", "bad sizes");
    if (width > 32767 || height > 32767)
        return fail("ngs_idxmap_forward", "image too large (the reference walks pixels with 16-bit signed counters)");
    if (!out_idx || !out_color || !out_depth || !out_feature || !workspace)
        return fail("ngs_idxmap_forward", "missing required pointer");
    if (P > 0 && (!means3D || !viewmatrix || !projmatrix || !feature_vector || !campos))
        return fail("ngs_idxmap_forward", "missing required pointer");
    if (((uintptr_t)out_feature | (uintptr_t)feature_vector) % 16 != 0 || (uintptr_t)workspace % 8 != 0)
        return fail("ngs_idxmap_forward", "feature arrays must be 16-byte aligned, the workspace 8-byte aligned");
    const int ps = prof_acquire();
    prof_begin(ps, ST_IDXMAP_FWD, st);
    ngs_launch_idxmap_forward(st, P, width, height, means3D, viewmatrix, projmatrix, campos, feature_vector, workspace,
                              out_idx, out_color, out_depth, out_feature, radii, num_rendered != nullptr);
    prof_end(ps, ST_IDXMAP_FWD, st);
    NGS_STAGE("index map");
    if (num_rendered) {  // only a caller that wants the entry count pays for the synchronisation
        uint32_t part[64];
        NGS_HIP(hipMemcpyAsync(part, workspace + (size_t)width * height * 8, sizeof part, hipMemcpyDeviceToHost, st));
        NGS_HIP(hipStreamSynchronize(st));
        uint32_t n = 0;
        for (int i = 0; i < 64; i++) n += part[i];
        *num_rendered = (int)n;
    }
    return 0;
}

extern "C" int ngs_idxmap_backward(int P, int width, int height, const float* means3D, const float* campos,
                                   const int* out_idx, const float* dL_dout_feature, float* dL_dfeature_vector,
                                   const float* viewmatrix, const float* projmatrix, int debug, void* stream)
{
    g_err[0] = 0;
    hipStream_t st = (hipStream_t)stream;
    (void)campos;  // unused by the reference as well (rasterizer_impl.cu:820-847)
    if (P < 0 || P > NGS_MAX_GAUSSIANS || width <= 0 || height <= 0) return fail("ngs_idxmap_backward", "bad sizes");
    if (P == 0) return 0;
    if (!out_idx || !dL_dout_feature || !dL_dfeature_vector) return fail("ngs_idxmap_backward", "missing required pointer");
    if (((uintptr_t)dL_dout_feature | (uintptr_t)dL_dfeature_vector) % 16 != 0)
        return fail("ngs_idxmap_backward", "feature arrays must be 16-byte aligned");
    const int ps = prof_acquire();
    prof_begin(ps, ST_IDXMAP_BWD, st);
    const bool gather = means3D && viewmatrix && projmatrix;
    ngs_launch_idxmap_backward(st, P, width, height, gather ? means3D : nullptr, viewmatrix, projmatrix, out_idx,
                               dL_dout_feature, dL_dfeature_vector);
    prof_end(ps, ST_IDXMAP_BWD, st);
    NGS_STAGE("index map backward");
    return 0;
}

extern "C" size_t ngs_knn_workspace_bytes(int P) { return P > 0 ? ngs_knn_workspace_size(P) : 0; }

extern "C" int ngs_knn_mean_dist2(int P, const float* points, float* mean_dist2, char* workspace, int debug,
                                  void* stream)
{
    g_err[0] = 0;
    hipStream_t st = (hipStream_t)stream;
    if (P < 0 || P > NGS_MAX_GAUSSIANS) return fail("ngs_knn_mean_dist2", "bad sizes");
    if (P == 0) return 0;
    if (!points || !mean_dist2 || !workspace) return fail("ngs_knn_mean_dist2", "missing required pointer");
    const int ps = prof_acquire();
    prof_begin(ps, ST_KNN, st);
    ngs_launch_knn(st, P, points, workspace, mean_dist2);
    prof_end(ps, ST_KNN, st);
    NGS_STAGE("nearest neighbours");
    return 0;
}

extern "C" int ngs_sh_adam_from_views(int P, int D, int M, int V, const float* means3D, const float* campos,
                                      long long campos_stride, const float* dL_dcolor_views, long long view_stride,
                                      float* shs, float* sh_exp_avg, float* sh_exp_avg_sq, double lr_dc,
                                      double lr_rest, double beta1, double beta2, double eps, int step,
                                      float grad_scale, void* stream)
{
    const char* fn = "ngs_sh_adam_from_views";
    g_err[0] = 0;
    if (P < 0 || P > NGS_MAX_GAUSSIANS || V < 1) return fail(fn, "bad sizes");
    if (M != 16 || D < 0 || D > 3) return fail(fn, "the fused update needs rows of 16 coefficients and a degree of 0..3");
    if (P == 0) return 0;
    if (!means3D || !campos || !dL_dcolor_views || !shs || !sh_exp_avg || !sh_exp_avg_sq)
        return fail(fn, "missing required pointer");
    if (campos_stride < 3 || view_stride < 3LL * P) return fail(fn, "view strides shorter than one view's data");
    if (((uintptr_t)shs | (uintptr_t)sh_exp_avg | (uintptr_t)sh_exp_avg_sq) % 16 != 0)
        return fail(fn, "SH coefficients and optimiser state must be 16-byte aligned");
    if (!(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0 && step >= 1))
        return fail(fn, "bad Adam constants or step < 1");
    ShAdamArgs a;
    a.param = shs;
    a.exp_avg = sh_exp_avg;
    a.exp_avg_sq = sh_exp_avg_sq;
    a.step_head = ngs_adam_step_size(lr_dc, beta1, step);
    a.step_tail = ngs_adam_step_size(lr_rest, beta1, step);
    a.k = ngs_adam_const(beta1, beta2, eps, step, grad_scale);
    const int ps = prof_acquire();
    prof_begin(ps, ST_SH_VIEWS, (hipStream_t)stream);
    ngs_launch_sh_from_views((hipStream_t)stream, P, D, M, V, means3D, campos, campos_stride, dL_dcolor_views,
                             view_stride, shs /* alignment witness for the LDS path; not written as a gradient */, &a);
    prof_end(ps, ST_SH_VIEWS, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("stage SH Adam from views", hipGetErrorString(e));
    return 0;
}

extern "C" int ngs_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                                unsigned char* present, void* stream)
{
    g_err[0] = 0;
    (void)projmatrix;
    if (P < 0) return fail("ngs_mark_visible", "bad size");
    if (P == 0) return 0;
    if (!means3D || !viewmatrix || !present) return fail("ngs_mark_visible", "missing required pointer");
    ngs_launch_mark_visible((hipStream_t)stream, P, means3D, viewmatrix, present);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("stage checkFrustum", hipGetErrorString(e));
    return 0;
}
