// api.hip -- implementation of the C ABI declared in include/polychase_hip.h.
//
// Host-side orchestration only: HBM plane layout, stream ordering, read-backs, and the two pieces of
// GoodFeaturesToTrack that are sequential by definition (greedy min-distance suppression,
// reference cpp/feature_detection/gftt.cc:100-164).  No pixel arithmetic happens on the CPU.
#include <chrono>
#include <ratio>

#include "internal.hpp"

namespace pc {
std::string& last_error() {
    thread_local std::string e;
    return e;
}
}  // namespace pc

namespace {

struct ScopedTimer {
    pc_context* c;
    int cls;
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t s = nullptr;
    ScopedTimer(pc_context* ctx, int k, hipStream_t on = nullptr) : c(ctx), cls(k), s(on ? on : ctx->work) {
        if (!(c->timing_mask & (1u << k))) return;
        auto get = [&]() {
            hipEvent_t e = nullptr;
            if (!c->event_pool.empty()) {
                e = c->event_pool.back();
                c->event_pool.pop_back();
            } else {
                (void)hipEventCreate(&e);
            }
            return e;
        };
        a = get();
        b = get();
        (void)hipEventRecord(a, s);
    }
    ~ScopedTimer() {
        if (!a) return;
        (void)hipEventRecord(b, s);
        c->ranges.push_back({cls, a, b});
    }
};

int collect_timing(pc_context* c) {
    if (c->ranges.empty()) return PC_OK;
    PC_HIP(hipStreamSynchronize(c->prep_stream));
    PC_HIP(hipStreamSynchronize(c->stream));
    for (auto& r : c->ranges) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            c->launches[r.cls] += 1;
            c->total_ms[r.cls] += ms;
        }
        c->event_pool.push_back(r.a);
        c->event_pool.push_back(r.b);
    }
    c->ranges.clear();
    return PC_OK;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Offsets (dx, dy) != (0, 0) with dx^2 + dy^2 < min_distance^2: the neighbourhood inside which the
// reference's greedy loop rejects a candidate (gftt.cc:134-141; its 3x3 cell search with
// cell = cvRound(min_distance) covers exactly these offsets since |dx| < min_distance <= cell + 0.5).
std::vector<int2> suppression_offsets(double min_distance) {
    std::vector<int2> out;
    const double r2 = min_distance * min_distance;
    const int R = (int)std::ceil(min_distance);
    for (int dy = -R; dy <= R; dy++)
        for (int dx = -R; dx <= R; dx++) {
            if (dx == 0 && dy == 0) continue;
            const float fx = (float)dx, fy = (float)dy;
            if ((double)(fx * fx + fy * fy) < r2) out.push_back(make_int2(dx, dy));
        }
    return out;
}

int ensure_kp_capacity(pc_frame* f, int n) {
    if (n <= f->kp_cap) return PC_OK;
    if (f->d_kps) (void)hipFree(f->d_kps);
    f->d_kps = nullptr;
    f->kp_cap = 0;
    const int want = n + n / 2 + 1024;
    PC_HIP(hipMalloc(reinterpret_cast<void**>(&f->d_kps), (size_t)want * sizeof(float2)));
    f->kp_cap = want;
    return PC_OK;
}

// gray already in level-0 interior -> borders, pyrDown chain, Scharr planes
void build_pyramid(pc_context* c, pc_frame* f) {
    ScopedTimer t(c, PC_K_PYRAMID);
    for (int l = 0; l < f->nlevels; l++) {
        if (l > 0) pc::launch_pyrdown(f->levels[l - 1], f->levels[l], c->work);
        pc::launch_border(f->levels[l], f->win, c->work);
        pc::launch_scharr(f->levels[l], c->work);
    }
}

constexpr int kCounterCells = 4;  // counters[0..3] = candidates, accepted, stuck, pad; then cell max keys

int validate_gftt(const pc_gftt_options* opt, int w, int h, pc::GfttGrid* g) {
    // CHECKs of gftt.cc:18-19
    if (!(opt->quality_level > 0 && opt->min_distance >= 0 && opt->max_corners >= 0))
        return fail(PC_E_INVALID, "GFTT options violate quality_level > 0 && min_distance >= 0 && max_corners >= 0");
    if (opt->use_harris) return fail(PC_E_INVALID, "use_harris is not implemented on the HIP path");
    if (opt->block_size != 3 || opt->gradient_size != 3)
        return fail(PC_E_INVALID, "only block_size == 3 and gradient_size == 3 are implemented on the HIP path");
    if (opt->min_distance > 64.0) return fail(PC_E_INVALID, "min_distance > 64 is not supported on the HIP path");
    g->rows = std::max(1, opt->grid_rows);
    g->cols = std::max(1, opt->grid_cols);
    if (g->rows * g->cols > pc::kMaxGridCells) return fail(PC_E_INVALID, "grid_rows*grid_cols must be <= %d", pc::kMaxGridCells);
    g->cell_h = (h + g->rows - 1) / g->rows;
    g->cell_w = (w + g->cols - 1) / g->cols;
    return PC_OK;
}

// Dense part of GoodFeaturesToTrack, fully on the GPU and asynchronous: min-eig map + per-cell max
// (gftt.cc:35,:61-63), threshold + NMS -> candidates (gftt.cc:64-86), exact min-distance suppression
// (gftt.cc:100-164), accepted candidates -> list.  Counts are copied to pinned memory; `ev` fires after.
int detect_phase_a(pc_context* ctx, pc_frame* f, const pc::GfttGrid& grid, const pc_gftt_options& opt,
                   DetectScratch& d) {
    const int w = f->w, h = f->h;
    const size_t npx = (size_t)w * h;
    PC_HIP(ctx->eig.ensure(npx));
    PC_HIP(ctx->cmap.ensure(npx));
    PC_HIP(ctx->state.ensure(npx));
    PC_HIP(d.keys_in.ensure(npx));
    PC_HIP(d.acc_keys.ensure(npx));
    PC_HIP(d.counters.ensure(kCounterCells + pc::kMaxGridCells));
    PC_HIP(d.h_counters.ensure(kCounterCells));
    if (!d.ev) PC_HIP(hipEventCreateWithFlags(&d.ev, hipEventDisableTiming));
    if (ctx->resident_blocks == 0) {
        hipDeviceProp_t prop;
        PC_HIP(hipGetDeviceProperties(&prop, ctx->device));
        ctx->resident_blocks = std::max(1, prop.multiProcessorCount) * 6;  // 256-lane blocks, 34 VGPRs: 8 fit per CU
    }
    const bool suppress = opt.min_distance >= 1;
    if (suppress && ctx->sup_min_distance != opt.min_distance) {
        const std::vector<int2> offs = suppression_offsets(opt.min_distance);
        PC_HIP(ctx->sup_offsets.ensure(offs.size() + 1));
        PC_HIP(hipStreamSynchronize(ctx->work));  // a queued suppression may still read the old table
        PC_HIP(hipMemcpy(ctx->sup_offsets.p, offs.data(), offs.size() * sizeof(int2), hipMemcpyHostToDevice));
        ctx->n_sup_offsets = (int)offs.size();
        ctx->sup_min_distance = opt.min_distance;
    }
    PC_HIP(hipMemsetAsync(d.counters.p, 0, (kCounterCells + pc::kMaxGridCells) * sizeof(uint32_t), ctx->work));
    uint32_t* cell_max = d.counters.p + kCounterCells;
    {
        ScopedTimer t(ctx, PC_K_MINEIG);
        pc::launch_min_eig(f->levels[0], ctx->eig.p, grid, cell_max, ctx->work);
    }
    ctx->eig_owner = f;
    {
        ScopedTimer t(ctx, PC_K_NMS);
        pc::launch_nms_compact(ctx->eig.p, w, h, grid, cell_max, opt.quality_level, d.keys_in.p, (uint32_t)npx,
                               d.counters.p, ctx->cmap.p, ctx->state.p, ctx->work);
    }
    {
        ScopedTimer t(ctx, PC_K_SUPPRESS);
        if (suppress)
            pc::launch_suppress(d.keys_in.p, d.counters.p, (uint32_t)npx, w, h, ctx->cmap.p, ctx->state.p,
                                ctx->sup_offsets.p, ctx->n_sup_offsets, d.counters.p + 2, ctx->resident_blocks,
                                ctx->work);
        pc::launch_collect_accepted(d.keys_in.p, d.counters.p, (uint32_t)npx, ctx->state.p, suppress ? 0 : 1,
                                    d.acc_keys.p, d.counters.p + 1, ctx->work);
    }
    PC_HIP(hipMemcpyAsync(d.h_counters.p, d.counters.p, kCounterCells * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->work));
    PC_HIP(hipEventRecord(d.ev, ctx->work));
    return PC_OK;
}

// Ordering part: sort the accepted corners (value desc, address desc == acceptance order of the
// greedy loop, gftt.cc:98,:157), truncate to max_corners (gftt.cc:160-162), write float2 keypoints.
int detect_phase_b(pc_context* ctx, pc_frame* f, const pc_gftt_options& opt, DetectScratch& d) {
    PC_HIP(hipEventSynchronize(d.ev));
    const uint32_t npx = (uint32_t)((size_t)f->w * f->h);
    const uint32_t n_cand = std::min(d.h_counters.p[0], npx);
    const uint32_t n_acc = std::min(d.h_counters.p[1], n_cand);
    if (d.h_counters.p[2] != 0)
        return fail(PC_E_HIP, "suppression kernel did not converge (%u lanes gave up): grid not resident?", d.h_counters.p[2]);
    f->n_cands = (int)n_cand;
    int n = (int)n_acc;
    if (n > 0) {
        PC_HIP(ctx->keys_out.ensure(n_acc));
        size_t temp_bytes = 0;
        PC_HIP(pc::sort_keys_desc(nullptr, temp_bytes, d.acc_keys.p, ctx->keys_out.p, n_acc, ctx->work));
        PC_HIP(ctx->sort_temp.ensure(temp_bytes));
        {
            ScopedTimer t(ctx, PC_K_SORT);
            PC_HIP(pc::sort_keys_desc(ctx->sort_temp.p, temp_bytes, d.acc_keys.p, ctx->keys_out.p, n_acc, ctx->work));
        }
        if (opt.max_corners > 0) n = std::min(n, opt.max_corners);
        int rc = ensure_kp_capacity(f, n);
        if (rc != PC_OK) return rc;
        pc::launch_keys_to_xy(ctx->keys_out.p, n, f->w, f->d_kps, ctx->work);
    }
    f->n_kps = n;
    f->perm_valid = false;
    return PC_OK;
}

// Stage-level entry points share scratch buffers with work the analyzer queued on prep_stream:
// order `stream` after it.
int join_prep(pc_context* ctx) {
    if (!ctx->prep_dirty) return PC_OK;
    PC_HIP(hipEventRecord(ctx->prep_fence, ctx->prep_stream));
    PC_HIP(hipStreamWaitEvent(ctx->stream, ctx->prep_fence, 0));
    PC_HIP(hipEventRecord(ctx->prep_fence, ctx->copy_stream));
    PC_HIP(hipStreamWaitEvent(ctx->stream, ctx->prep_fence, 0));
    ctx->prep_dirty = false;
    return PC_OK;
}

struct PrepScope {   // image / detection helpers enqueue on prep_stream while one of these is alive
    pc_context* c;
    explicit PrepScope(pc_context* ctx) : c(ctx) {
        c->work = c->prep_stream;
        c->prep_dirty = true;
    }
    ~PrepScope() { c->work = c->stream; }
};

// LK visiting order of the frame's keypoints (counting sort by 64x64 tile) on the current work stream
int order_keypoints_spatially(pc_context* ctx, pc_frame* f, DevBuf<uint32_t>& hist) {
    const int n = f->n_kps;
    f->perm_valid = false;
    if (n <= 0) return PC_OK;
    if (f->perm_cap < n) {
        const int cap = std::max(n + n / 2, 1024);
        if (f->d_perm) PC_HIP(hipFree(f->d_perm));
        f->d_perm = nullptr;
        f->perm_cap = 0;
        PC_HIP(hipMalloc(&f->d_perm, (size_t)cap * sizeof(uint32_t)));
        f->perm_cap = cap;
    }
    PC_HIP(hist.ensure((size_t)pc::bin_num_tiles(f->w, f->h) + 1));
    pc::launch_spatial_bins(f->d_kps, n, f->w, f->h, hist.p, f->d_perm, ctx->work);
    f->perm_valid = true;
    return PC_OK;
}

int check_lk_args(pc_context* ctx, const pc_frame* frame1, const pc_frame* const* targets, int n_targets,
                  const pc_flow_options* opt) {
    if (!ctx || !frame1 || !targets || !opt) return fail(PC_E_INVALID, "null argument");
    if (n_targets < 1 || n_targets > PC_MAX_TARGETS) return fail(PC_E_INVALID, "n_targets must be in [1,%d]", PC_MAX_TARGETS);
    if (frame1->n_kps < 0) return fail(PC_E_STATE, "frame1 has no keypoints (call pc_frame_detect or pc_frame_set_keypoints)");
    if (opt->window_size != frame1->win) return fail(PC_E_INVALID, "window_size %d differs from the frame's pyramid padding %d", opt->window_size, frame1->win);
    for (int t = 0; t < n_targets; t++) {
        if (!targets[t]) return fail(PC_E_INVALID, "null target");
        // cv::calcOpticalFlowPyrLK asserts equal level sizes/types (CV_Assert in lkpyramid.cpp)
        if (targets[t]->w != frame1->w || targets[t]->h != frame1->h || targets[t]->win != frame1->win)
            return fail(PC_E_INVALID, "target %d geometry differs from frame1", t);
    }
    return PC_OK;
}

int run_lk(pc_context* ctx, const pc_frame* frame1, const pc_frame* const* targets, int n_targets,
           const pc_flow_options* opt, int set = 0) {
    const int n = frame1->n_kps;
    const size_t rows = (size_t)n * n_targets;
    PC_HIP(ctx->lk_xy[set].ensure(rows + 1));
    PC_HIP(ctx->lk_status[set].ensure(rows + 1));
    PC_HIP(ctx->lk_err[set].ensure(rows + 1));
    if (n == 0) return PC_OK;
    pc::LKParams p;
    std::memset(&p, 0, sizeof(p));
    int max_level = std::min(opt->max_level, frame1->nlevels - 1);
    for (int t = 0; t < n_targets; t++) max_level = std::min(max_level, targets[t]->nlevels - 1);
    if (max_level < 0) max_level = 0;
    for (int l = 0; l <= max_level; l++) {
        p.src[l] = frame1->levels[l];
        for (int t = 0; t < n_targets; t++) p.tgt[t][l] = targets[t]->levels[l].img;
    }
    p.n_targets = n_targets;
    p.max_level = max_level;
    p.n = n;
    p.pts = frame1->d_kps;
    // visiting order: keypoints binned by image tile (they are stored by corner response); the analyzer
    // prepares it together with the keypoints, the stage-level calls compute it here
    if (frame1->perm_valid) {
        p.perm = frame1->d_perm;
    } else {
        PC_HIP(ctx->lk_perm.ensure((size_t)n));
        PC_HIP(ctx->lk_hist.ensure((size_t)pc::bin_num_tiles(frame1->w, frame1->h) + 1));
        pc::launch_spatial_bins(frame1->d_kps, n, frame1->w, frame1->h, ctx->lk_hist.p, ctx->lk_perm.p, ctx->stream);
        p.perm = ctx->lk_perm.p;
    }
    // TermCriteria clamps of calcOpticalFlowPyrLK
    p.max_iters = std::min(std::max(opt->term_max_iters, 0), 100);
    const double eps = std::min(std::max(opt->term_epsilon, 0.), 10.);
    p.eps_sq = eps * eps;
    p.min_eig_thr = (float)opt->min_eigen_threshold;
    p.out_xy = ctx->lk_xy[set].p;
    p.out_status = ctx->lk_status[set].p;
    p.out_err = ctx->lk_err[set].p;
    ScopedTimer tm(ctx, PC_K_LK, ctx->stream);
    if (!pc::launch_lk(p, frame1->win, ctx->stream)) return fail(PC_E_INVALID, "unsupported window size %d", frame1->win);
    return PC_OK;
}

}  // namespace

extern "C" {

void pc_gftt_default_options(pc_gftt_options* o) {
    o->quality_level = 0.01;
    o->min_distance = 5.0;
    o->block_size = 3;
    o->gradient_size = 3;
    o->max_corners = 0;
    o->use_harris = 0;
    o->harris_k = 0.04;
    o->grid_rows = 4;
    o->grid_cols = 4;
}

void pc_flow_default_options(pc_flow_options* o) {
    o->window_size = 10;
    o->max_level = 3;
    o->term_max_iters = 30;
    o->term_epsilon = 0.01;
    o->min_eigen_threshold = 1e-4;
}

const char* pc_last_error(void) { return pc::last_error().c_str(); }
const char* pc_version(void) { return "polychase_hip 0.1 (gfx950, hand-written HIP)"; }

int pc_context_create(int device_index, pc_context** out) {
    if (!out) return fail(PC_E_INVALID, "null out");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(PC_E_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device_index < 0 || device_index >= count) return fail(PC_E_INVALID, "device index %d out of range [0,%d)", device_index, count);
    PC_HIP(hipSetDevice(device_index));
    pc_context* c = new (std::nothrow) pc_context();
    if (!c) return fail(PC_E_INVALID, "out of host memory");
    c->device = device_index;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->prep_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->prep_fence, hipEventDisableTiming);
    c->work = c->stream;
    if (e != hipSuccess) {
        delete c;
        return fail(PC_E_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    *out = c;
    return PC_OK;
}

void pc_context_destroy(pc_context* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->prep_stream) (void)hipStreamSynchronize(c->prep_stream);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& r : c->ranges) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    c->staging.release();
    c->eig.release();
    c->cmap.release();
    c->state.release();
    c->sup_offsets.release();
    if (c->detect) {
        c->detect->release();
        delete c->detect;
        c->detect = nullptr;
    }
    c->keys_out.release();
    c->sort_temp.release();
    c->lk_xy[0].release();
    c->lk_xy[1].release();
    c->lk_cxy.release();
    c->lk_status[0].release();
    c->lk_status[1].release();
    c->lk_err[0].release();
    c->lk_err[1].release();
    c->lk_cerr.release();
    c->lk_cidx.release();
    c->lk_block_counts.release();
    c->lk_perm.release();
    c->lk_hist.release();
    c->prep_hist.release();
    c->lk_row_offset.release();
    c->h_row_offset.release();
    c->lk_pack.release();
    if (c->copy_stream) {
        (void)hipStreamSynchronize(c->copy_stream);
        (void)hipStreamDestroy(c->copy_stream);
    }
    if (c->prep_stream) (void)hipStreamDestroy(c->prep_stream);
    if (c->prep_fence) (void)hipEventDestroy(c->prep_fence);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int pc_context_synchronize(pc_context* c) {
    if (!c) return fail(PC_E_INVALID, "null context");
    PC_HIP(hipStreamSynchronize(c->prep_stream));
    PC_HIP(hipStreamSynchronize(c->stream));
    PC_HIP(hipStreamSynchronize(c->copy_stream));
    c->prep_dirty = false;
    return PC_OK;
}

void* pc_context_stream(pc_context* c) { return c ? (void*)c->stream : nullptr; }

int pc_context_enable_timing(pc_context* c, int enable) {
    if (!c) return fail(PC_E_INVALID, "null context");
    int rc = collect_timing(c);
    c->timing_mask = (unsigned)enable;
    return rc;
}

int pc_context_get_timing(pc_context* c, int k, int* launches, double* total_ms) {
    if (!c || k < 0 || k >= PC_K_COUNT) return fail(PC_E_INVALID, "bad kernel class");
    int rc = collect_timing(c);
    if (rc != PC_OK) return rc;
    if (launches) *launches = c->launches[k];
    if (total_ms) *total_ms = c->total_ms[k];
    return PC_OK;
}

int pc_context_reset_timing(pc_context* c) {
    if (!c) return fail(PC_E_INVALID, "null context");
    int rc = collect_timing(c);
    for (int k = 0; k < PC_K_COUNT; k++) {
        c->launches[k] = 0;
        c->total_ms[k] = 0;
    }
    return rc;
}

// ---------------------------------------------------------------------------------------------
int pc_frame_create(pc_context* ctx, int width, int height, int window_size, int max_level, pc_frame** out) {
    if (!ctx || !out) return fail(PC_E_INVALID, "null argument");
    *out = nullptr;
    if (width < 1 || height < 1) return fail(PC_E_INVALID, "bad frame size %dx%d", width, height);
    // buildOpticalFlowPyramid: CV_Assert(winSize.width > 2 && winSize.height > 2)
    if (window_size < 3 || window_size > PC_MAX_WINDOW) return fail(PC_E_INVALID, "window_size must be in [3,%d]", PC_MAX_WINDOW);
    if (max_level < 0 || max_level >= PC_MAX_LEVELS) return fail(PC_E_INVALID, "max_level must be in [0,%d]", PC_MAX_LEVELS - 1);
    if ((long long)width * height > (1ll << 30)) return fail(PC_E_INVALID, "frame too large");
    PC_HIP(hipSetDevice(ctx->device));
    pc_frame* f = new (std::nothrow) pc_frame();
    if (!f) return fail(PC_E_INVALID, "out of host memory");
    f->ctx = ctx;
    f->w = width;
    f->h = height;
    f->win = window_size;
    f->max_level = max_level;
    // level geometry: stop when the next level would be <= winSize (lkpyramid.cpp)
    int lw = width, lh = height;
    size_t total = 0;
    size_t img_off[PC_MAX_LEVELS], der_off[PC_MAX_LEVELS];
    for (int l = 0; l <= max_level; l++) {
        pc::Level& L = f->levels[l];
        L.w = lw;
        L.h = lh;
        L.pitch = (int)align_up((size_t)pc::kPadX + lw + window_size, 64);
        const size_t rows = (size_t)lh + 2 * window_size;
        img_off[l] = total;
        total += align_up(rows * L.pitch, 256);
        der_off[l] = total;
        total += align_up(rows * L.pitch * sizeof(int32_t), 256);
        f->nlevels = l + 1;
        lw = (lw + 1) / 2;
        lh = (lh + 1) / 2;
        if (lw <= window_size || lh <= window_size) break;
    }
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&f->slab), total);
    if (e != hipSuccess) {
        delete f;
        return fail(PC_E_HIP, "hipMalloc(%zu) failed: %s", total, hipGetErrorString(e));
    }
    f->slab_bytes = total;
    // zero once: derivative padding must stay zero (derivBorder = BORDER_CONSTANT)
    e = hipMemsetAsync(f->slab, 0, total, ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(f->slab);
        delete f;
        return fail(PC_E_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(e));
    }
    for (int l = 0; l < f->nlevels; l++) {
        pc::Level& L = f->levels[l];
        const size_t interior = (size_t)window_size * L.pitch + pc::kPadX;
        L.img = f->slab + img_off[l] + interior;
        L.der = reinterpret_cast<int32_t*>(f->slab + der_off[l]) + interior;
    }
    int rc = ensure_kp_capacity(f, std::max(4096, (int)((long long)width * height / 16)));
    if (rc != PC_OK) {
        (void)hipFree(f->slab);
        delete f;
        return rc;
    }
    *out = f;
    return PC_OK;
}

void pc_frame_destroy(pc_frame* f) {
    if (!f) return;
    if (f->ctx) {
        (void)hipSetDevice(f->ctx->device);
        (void)hipStreamSynchronize(f->ctx->prep_stream);
        (void)hipStreamSynchronize(f->ctx->stream);
        (void)hipStreamSynchronize(f->ctx->copy_stream);
        if (f->ctx->eig_owner == f) f->ctx->eig_owner = nullptr;
    }
    if (f->slab) (void)hipFree(f->slab);
    if (f->d_kps) (void)hipFree(f->d_kps);
    if (f->d_perm) (void)hipFree(f->d_perm);
    delete f;
}

// channels: 1 / 3 = u8 gray / RGB; elem_size 4 = float32 RGB(A) with `channels` floats per pixel
static int set_image(pc_context* ctx, pc_frame* f, const uint8_t* src, size_t row_pitch, int on_device, int channels,
                     int elem_size = 1) {
    if (!ctx || !f || !src) return fail(PC_E_INVALID, "null argument");
    if (f->ctx != ctx) return fail(PC_E_INVALID, "frame belongs to another context");
    if (elem_size == 4 && channels != 3 && channels != 4) return fail(PC_E_INVALID, "float frames need 3 or 4 channels, got %d", channels);
    if (elem_size == 4 && ((reinterpret_cast<uintptr_t>(src) | row_pitch) & 3)) return fail(PC_E_INVALID, "float frame is not 4-byte aligned");
    const size_t row_bytes = (size_t)f->w * channels * elem_size;
    if (row_pitch < row_bytes) return fail(PC_E_INVALID, "row_pitch %zu < %zu", row_pitch, row_bytes);
    PC_HIP(hipSetDevice(ctx->device));
    if (ctx->work == ctx->stream) {
        int jrc = join_prep(ctx);
        if (jrc != PC_OK) return jrc;
    }
    const uint8_t* d_src = src;
    size_t d_pitch = row_pitch;
    if (!on_device) {
        d_pitch = align_up(row_bytes, 16);
        PC_HIP(ctx->staging.ensure(d_pitch * f->h));
        // the previous frame's kernels may still read the staging buffer: ordered on the same stream
        PC_HIP(hipMemcpy2DAsync(ctx->staging.p, d_pitch, src, row_pitch, row_bytes, f->h, hipMemcpyHostToDevice, ctx->work));
        d_src = ctx->staging.p;
    }
    {
        ScopedTimer t(ctx, PC_K_GRAY);
        if (elem_size == 4) pc::launch_rgbf32_to_gray(reinterpret_cast<const float*>(d_src), d_pitch, channels, f->levels[0], ctx->work);
        else if (channels == 3) pc::launch_rgb2gray(d_src, d_pitch, f->levels[0], ctx->work);
        else pc::launch_copy_gray(d_src, d_pitch, f->levels[0], ctx->work);
    }
    build_pyramid(ctx, f);
    f->n_kps = -1;
    f->n_cands = -1;
    f->perm_valid = false;
    if (ctx->eig_owner == f) ctx->eig_owner = nullptr;
    if (!on_device) PC_HIP(hipStreamSynchronize(ctx->work));  // caller may reuse its host buffer
    return PC_OK;
}

int pc_host_buffer_alloc(size_t bytes, void** out) {
    if (!out) return fail(PC_E_INVALID, "null out");
    *out = nullptr;
    PC_HIP(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return PC_OK;
}
void pc_host_buffer_free(void* buffer) {
    if (buffer) (void)hipHostFree(buffer);
}

int pc_frame_set_rgb(pc_context* ctx, pc_frame* f, const uint8_t* rgb, size_t row_pitch, int on_device) {
    return set_image(ctx, f, rgb, row_pitch, on_device, 3);
}
int pc_frame_set_rgb_f32(pc_context* ctx, pc_frame* f, const float* rgb, size_t row_pitch, int channels, int on_device) {
    return set_image(ctx, f, reinterpret_cast<const uint8_t*>(rgb), row_pitch, on_device, channels, 4);
}
int pc_frame_set_gray(pc_context* ctx, pc_frame* f, const uint8_t* gray, size_t row_pitch, int on_device) {
    return set_image(ctx, f, gray, row_pitch, on_device, 1);
}

int pc_frame_num_levels(const pc_frame* f) { return f ? f->nlevels : 0; }

int pc_frame_level_size(const pc_frame* f, int level, int* width, int* height) {
    if (!f || level < 0 || level >= f->nlevels) return fail(PC_E_INVALID, "bad level");
    if (width) *width = f->levels[level].w;
    if (height) *height = f->levels[level].h;
    return PC_OK;
}

int pc_frame_download_gray(pc_context* ctx, const pc_frame* f, uint8_t* out) {
    if (!ctx || !f || !out) return fail(PC_E_INVALID, "null argument");
    if (int jrc = join_prep(ctx)) return jrc;
    const pc::Level& L = f->levels[0];
    PC_HIP(hipMemcpy2DAsync(out, L.w, L.img, L.pitch, L.w, L.h, hipMemcpyDeviceToHost, ctx->stream));
    PC_HIP(hipStreamSynchronize(ctx->stream));
    return PC_OK;
}

int pc_frame_download_level(pc_context* ctx, const pc_frame* f, int level, uint8_t* out) {
    if (!ctx || !f || !out || level < 0 || level >= f->nlevels) return fail(PC_E_INVALID, "bad argument");
    if (int jrc = join_prep(ctx)) return jrc;
    const pc::Level& L = f->levels[level];
    const int win = f->win;
    const uint8_t* src = L.img - (ptrdiff_t)win * L.pitch - win;
    PC_HIP(hipMemcpy2DAsync(out, L.w + 2 * win, src, L.pitch, L.w + 2 * win, L.h + 2 * win, hipMemcpyDeviceToHost, ctx->stream));
    PC_HIP(hipStreamSynchronize(ctx->stream));
    return PC_OK;
}

int pc_frame_download_deriv(pc_context* ctx, const pc_frame* f, int level, int16_t* out) {
    if (!ctx || !f || !out || level < 0 || level >= f->nlevels) return fail(PC_E_INVALID, "bad argument");
    if (int jrc = join_prep(ctx)) return jrc;
    const pc::Level& L = f->levels[level];
    const int win = f->win;
    const int32_t* src = L.der - (ptrdiff_t)win * L.pitch - win;
    const size_t row = (size_t)(L.w + 2 * win) * 4;
    PC_HIP(hipMemcpy2DAsync(out, row, src, (size_t)L.pitch * 4, row, L.h + 2 * win, hipMemcpyDeviceToHost, ctx->stream));
    PC_HIP(hipStreamSynchronize(ctx->stream));
    return PC_OK;
}

// ---------------------------------------------------------------------------------------------
int pc_frame_detect(pc_context* ctx, pc_frame* f, const pc_gftt_options* opt) {
    if (!ctx || !f || !opt) return fail(PC_E_INVALID, "null argument");
    if (f->ctx != ctx) return fail(PC_E_INVALID, "frame belongs to another context");
    pc::GfttGrid g;
    int rc = validate_gftt(opt, f->w, f->h, &g);
    if (rc != PC_OK) return rc;
    PC_HIP(hipSetDevice(ctx->device));
    if (int jrc = join_prep(ctx)) return jrc;
    if (!ctx->detect) ctx->detect = new DetectScratch();
    if ((rc = detect_phase_a(ctx, f, g, *opt, *ctx->detect)) != PC_OK) return rc;
    if ((rc = detect_phase_b(ctx, f, *opt, *ctx->detect)) != PC_OK) return rc;
    PC_HIP(hipStreamSynchronize(ctx->stream));
    return PC_OK;
}

int pc_frame_download_min_eig(pc_context* ctx, const pc_frame* f, float* out) {
    if (!ctx || !f || !out) return fail(PC_E_INVALID, "null argument");
    if (int jrc = join_prep(ctx)) return jrc;
    if (ctx->eig_owner != f) return fail(PC_E_STATE, "the min-eig scratch map belongs to another frame (call right after pc_frame_detect)");
    PC_HIP(hipMemcpyAsync(out, ctx->eig.p, (size_t)f->w * f->h * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    PC_HIP(hipStreamSynchronize(ctx->stream));
    return PC_OK;
}

int pc_frame_num_candidates(pc_context* ctx, const pc_frame* f, int* out_n) {
    if (!ctx || !f || !out_n) return fail(PC_E_INVALID, "null argument");
    if (f->n_cands < 0) return fail(PC_E_STATE, "pc_frame_detect has not run on this frame");
    *out_n = f->n_cands;
    return PC_OK;
}

int pc_frame_num_keypoints(pc_context* ctx, const pc_frame* f, int* out_n) {
    if (!ctx || !f || !out_n) return fail(PC_E_INVALID, "null argument");
    if (f->n_kps < 0) return fail(PC_E_STATE, "frame has no keypoints");
    *out_n = f->n_kps;
    return PC_OK;
}

int pc_frame_download_keypoints(pc_context* ctx, const pc_frame* f, float* out_xy, int capacity) {
    if (!ctx || !f || (!out_xy && capacity > 0)) return fail(PC_E_INVALID, "null argument");
    if (int jrc = join_prep(ctx)) return jrc;
    if (f->n_kps < 0) return fail(PC_E_STATE, "frame has no keypoints");
    if (capacity < f->n_kps) return fail(PC_E_CAPACITY, "capacity %d < %d keypoints", capacity, f->n_kps);
    if (f->n_kps == 0) return PC_OK;
    PC_HIP(hipMemcpyAsync(out_xy, f->d_kps, (size_t)f->n_kps * sizeof(float2), hipMemcpyDeviceToHost, ctx->stream));
    PC_HIP(hipStreamSynchronize(ctx->stream));
    return PC_OK;
}

int pc_frame_set_keypoints(pc_context* ctx, pc_frame* f, const float* xy, int n) {
    if (!ctx || !f || n < 0 || (!xy && n > 0)) return fail(PC_E_INVALID, "bad argument");
    if (int jrc = join_prep(ctx)) return jrc;
    int rc = ensure_kp_capacity(f, n);
    if (rc != PC_OK) return rc;
    if (n > 0) {
        PC_HIP(hipMemcpyAsync(f->d_kps, xy, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, ctx->stream));
        PC_HIP(hipStreamSynchronize(ctx->stream));
    }
    f->n_kps = n;
    f->perm_valid = false;
    return PC_OK;
}

// ---------------------------------------------------------------------------------------------
int pc_lk_track(pc_context* ctx, const pc_frame* frame1, const pc_frame* const* targets, int n_targets,
                const pc_flow_options* opt, float* next_xy, uint8_t* status, float* err) {
    int rc = check_lk_args(ctx, frame1, targets, n_targets, opt);
    if (rc != PC_OK) return rc;
    if (!next_xy || !status || !err) return fail(PC_E_INVALID, "null output");
    PC_HIP(hipSetDevice(ctx->device));
    if (int jrc = join_prep(ctx)) return jrc;
    rc = run_lk(ctx, frame1, targets, n_targets, opt);
    if (rc != PC_OK) return rc;
    const size_t rows = (size_t)frame1->n_kps * n_targets;
    if (rows > 0) {
        PC_HIP(hipMemcpyAsync(next_xy, ctx->lk_xy[0].p, rows * sizeof(float2), hipMemcpyDeviceToHost, ctx->stream));
        PC_HIP(hipMemcpyAsync(status, ctx->lk_status[0].p, rows, hipMemcpyDeviceToHost, ctx->stream));
        PC_HIP(hipMemcpyAsync(err, ctx->lk_err[0].p, rows * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    }
    PC_HIP(hipStreamSynchronize(ctx->stream));
    return PC_OK;
}

int pc_lk_track_filtered(pc_context* ctx, const pc_frame* frame1, const pc_frame* const* targets, int n_targets,
                         const pc_flow_options* opt, uint32_t* src_indices, float* tgt_xy, float* flow_err,
                         int64_t* row_offset) {
    int rc = check_lk_args(ctx, frame1, targets, n_targets, opt);
    if (rc != PC_OK) return rc;
    if (!row_offset) return fail(PC_E_INVALID, "null row_offset");
    PC_HIP(hipSetDevice(ctx->device));
    if (int jrc = join_prep(ctx)) return jrc;
    rc = run_lk(ctx, frame1, targets, n_targets, opt);
    if (rc != PC_OK) return rc;
    const int n = frame1->n_kps;
    const size_t rows = (size_t)n * n_targets;
    const int nblocks = pc::compact_num_blocks(n);
    PC_HIP(ctx->lk_cxy.ensure(rows + 1));
    PC_HIP(ctx->lk_cerr.ensure(rows + 1));
    PC_HIP(ctx->lk_cidx.ensure(rows + 1));
    PC_HIP(ctx->lk_block_counts.ensure((size_t)nblocks * n_targets + 1));
    PC_HIP(ctx->lk_row_offset.ensure(PC_MAX_TARGETS + 1));
    PC_HIP(ctx->h_row_offset.ensure(PC_MAX_TARGETS + 1));
    {
        ScopedTimer t(ctx, PC_K_COMPACT);
        pc::launch_compact(ctx->lk_xy[0].p, ctx->lk_status[0].p, ctx->lk_err[0].p, n, n_targets, ctx->lk_block_counts.p,
                           ctx->lk_row_offset.p, ctx->lk_cidx.p, ctx->lk_cxy.p, ctx->lk_cerr.p, ctx->stream);
    }
    PC_HIP(hipMemcpyAsync(ctx->h_row_offset.p, ctx->lk_row_offset.p, (size_t)(n_targets + 1) * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
    PC_HIP(hipStreamSynchronize(ctx->stream));
    for (int t = 0; t <= n_targets; t++) row_offset[t] = (int64_t)ctx->h_row_offset.p[t];
    const size_t total = (size_t)row_offset[n_targets];
    if (total > 0) {
        if (!src_indices || !tgt_xy || !flow_err) return fail(PC_E_INVALID, "null output");
        PC_HIP(hipMemcpyAsync(src_indices, ctx->lk_cidx.p, total * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        PC_HIP(hipMemcpyAsync(tgt_xy, ctx->lk_cxy.p, total * sizeof(float2), hipMemcpyDeviceToHost, ctx->stream));
        PC_HIP(hipMemcpyAsync(flow_err, ctx->lk_cerr.p, total * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        PC_HIP(hipStreamSynchronize(ctx->stream));
    }
    return PC_OK;
}


// =============================================================================================
// analyzer: pipelined per-clip engine (see include/polychase_hip.h)
// =============================================================================================
}  // extern "C"  (helpers below are C++)

namespace {

enum DetState { DET_NONE = 0, DET_DENSE = 1, DET_DONE = 3 };

struct Slot {
    pc_frame* frame = nullptr;
    int32_t frame_id = 0;
    bool valid = false;
    DetState det = DET_NONE;
    bool supplied = false;  // keypoints came from the caller (database), not from detection
    DetectScratch scratch;
    hipEvent_t last_read = nullptr;  // `computed` event of the latest job whose LK reads this slot
    hipEvent_t img_ready = nullptr;  // gray + pyramid of the resident frame are complete (prep stream)
    hipEvent_t kps_ready = nullptr;  // keypoints + visiting order are complete (prep stream)
};

struct Job {
    int32_t frame1 = 0;
    int n_kps = 0;
    bool detected = false;
    int n_targets = 0;
    int32_t targets[PC_MAX_TARGETS];
    PinBuf<uint8_t> h_pack;   // the job's records, same layout as pc_context::lk_pack
    size_t o_kps = 0, o_idx = 0, o_xy = 0, o_err = 0, pack_bytes = 0;
    hipEvent_t done = nullptr;      // records of this job are in pinned memory (copy stream)
    hipEvent_t computed = nullptr;  // compaction (+ device-log copies) finished (copy stream)
    hipEvent_t lk_done = nullptr;   // the LK launch finished (main stream)
};

}  // namespace

struct pc_analyzer {
    pc_context* ctx = nullptr;
    int w = 0, h = 0;
    pc_gftt_options gopt;
    pc_flow_options fopt;
    pc::GfttGrid grid;
    std::vector<Slot> slots;
    std::vector<Job> jobs;
    size_t job_head = 0, job_count = 0;  // ring of in-flight jobs
    uint64_t submitted = 0;              // jobs submitted so far: job k writes LK output set k & 1
    hipEvent_t set_free[2] = {nullptr, nullptr};  // `computed` of the last job that used each LK output set
    uint8_t* d_log = nullptr;            // optional device-resident record log
    size_t log_cap = 0, log_used = 0;
    std::vector<PinBuf<long long>> log_hdr;  // one pinned header per job slot
};

namespace {

Slot* find_slot(pc_analyzer* a, int32_t frame_id) {
    const int n = (int)a->slots.size();
    Slot& s = a->slots[(size_t)(((frame_id % n) + n) % n)];
    return (s.valid && s.frame_id == frame_id) ? &s : nullptr;
}

// All three run on the prep stream (the callers hold a PrepScope).
int detect_dense(pc_analyzer* a, Slot& s) {
    int rc = detect_phase_a(a->ctx, s.frame, a->grid, a->gopt, s.scratch);
    if (rc == PC_OK) s.det = DET_DENSE;
    return rc;
}

int detect_finish(pc_analyzer* a, Slot& s) {
    int rc;
    if (s.det == DET_NONE && (rc = detect_dense(a, s)) != PC_OK) return rc;
    if ((rc = detect_phase_b(a->ctx, s.frame, a->gopt, s.scratch)) != PC_OK) return rc;
    if ((rc = order_keypoints_spatially(a->ctx, s.frame, a->ctx->prep_hist)) != PC_OK) return rc;
    PC_HIP(hipEventRecord(s.kps_ready, a->ctx->prep_stream));
    s.det = DET_DONE;
    s.supplied = false;
    return PC_OK;
}

// Ordering phase of the frame that will most likely be the next frame1, if its dense phase has
// already delivered its counters: keeps sort + binning off the LK stream's critical path.
int preorder_if_ready(pc_analyzer* a, int32_t frame_id) {
    Slot* s = find_slot(a, frame_id);
    if (!s || s->det != DET_DENSE || !s->scratch.ev) return PC_OK;
    if (hipEventQuery(s->scratch.ev) != hipSuccess) return PC_OK;
    PrepScope prep(a->ctx);
    return detect_finish(a, *s);
}

}  // namespace

extern "C" {

int pc_analyzer_create(pc_context* ctx, int width, int height, const pc_gftt_options* gftt,
                       const pc_flow_options* flow, int ring_frames, int max_jobs, pc_analyzer** out) {
    if (!ctx || !gftt || !flow || !out) return fail(PC_E_INVALID, "null argument");
    *out = nullptr;
    if (ring_frames < 1 || ring_frames > 4096) return fail(PC_E_INVALID, "ring_frames must be in [1,4096]");
    if (max_jobs < 1 || max_jobs > 64) return fail(PC_E_INVALID, "max_jobs must be in [1,64]");
    pc::GfttGrid grid0;
    {
        int vrc = validate_gftt(gftt, width, height, &grid0);
        if (vrc != PC_OK) return vrc;
    }
    PC_HIP(hipSetDevice(ctx->device));
    pc_analyzer* a = new (std::nothrow) pc_analyzer();
    if (!a) return fail(PC_E_INVALID, "out of host memory");
    a->ctx = ctx;
    a->w = width;
    a->h = height;
    a->gopt = *gftt;
    a->fopt = *flow;
    a->grid = grid0;
    // two extra slots: a frame can be overwritten (prep stream) while the LK launches that read its
    // predecessors in the ring are still running, without the two streams waiting on each other
    a->slots.resize((size_t)ring_frames + 2);
    a->jobs.resize((size_t)max_jobs);
    int rc = PC_OK;
    for (auto& s : a->slots) {
        rc = pc_frame_create(ctx, width, height, flow->window_size, flow->max_level, &s.frame);
        if (rc != PC_OK) break;
        // room for a typical frame's keypoints up front: growing later frees device memory, which synchronises
        const int kp0 = std::max(16384, (width * height) / 32);
        if ((rc = ensure_kp_capacity(s.frame, kp0)) != PC_OK) break;
        if (hipMalloc(reinterpret_cast<void**>(&s.frame->d_perm), (size_t)kp0 * sizeof(uint32_t)) != hipSuccess) {
            rc = fail(PC_E_HIP, "hipMalloc failed");
            break;
        }
        s.frame->perm_cap = kp0;
        if (hipEventCreateWithFlags(&s.img_ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.kps_ready, hipEventDisableTiming) != hipSuccess) {
            rc = fail(PC_E_HIP, "hipEventCreate failed");
            break;
        }
    }
    if (rc == PC_OK)
        for (auto& j : a->jobs)
            if (hipEventCreateWithFlags(&j.done, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&j.computed, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&j.lk_done, hipEventDisableTiming) != hipSuccess) {
                rc = fail(PC_E_HIP, "hipEventCreate failed");
                break;
            }
    if (rc == PC_OK) {
        // Warm the runtime's copy engines: it picks a free SDMA engine per copy and creates an engine's queue the first
        // time it is used (5-8 ms inside some hipMemcpyAsync, observed twice or three times in the first few dozen
        // frames).  A burst of overlapping downloads makes it create them now.
        const size_t chunk = (size_t)4 << 20, burst = 12;
        Job& j0 = a->jobs[0];
        if (ctx->lk_pack.ensure(chunk) != hipSuccess || j0.h_pack.ensure(chunk * burst) != hipSuccess) {
            rc = fail(PC_E_HIP, "allocation failed");
        } else {
            hipStream_t streams[3] = {ctx->copy_stream, ctx->prep_stream, ctx->stream};
            for (int round = 0; round < 3 && rc == PC_OK; round++) {
                for (size_t k = 0; k < burst; k++)
                    if (hipMemcpyAsync(j0.h_pack.p + k * chunk, ctx->lk_pack.p, chunk, hipMemcpyDeviceToHost, streams[k % 3]) !=
                        hipSuccess)
                        rc = fail(PC_E_HIP, "copy engine warm-up failed");
                for (hipStream_t st : streams) (void)hipStreamSynchronize(st);
            }
        }
    }
    if (rc != PC_OK) {
        std::string keep = pc::last_error();
        pc_analyzer_destroy(a);
        pc::last_error() = keep;
        return rc;
    }
    *out = a;
    return PC_OK;
}

void pc_analyzer_destroy(pc_analyzer* a) {
    if (!a) return;
    (void)hipSetDevice(a->ctx->device);
    (void)hipStreamSynchronize(a->ctx->prep_stream);
    (void)hipStreamSynchronize(a->ctx->stream);
    (void)hipStreamSynchronize(a->ctx->copy_stream);
    for (auto& s : a->slots) {
        if (s.frame) pc_frame_destroy(s.frame);
        if (s.img_ready) (void)hipEventDestroy(s.img_ready);
        if (s.kps_ready) (void)hipEventDestroy(s.kps_ready);
        s.scratch.release();
    }
    for (auto& hdr : a->log_hdr) hdr.release();
    for (auto& j : a->jobs) {
        j.h_pack.release();
        if (j.done) (void)hipEventDestroy(j.done);
        if (j.computed) (void)hipEventDestroy(j.computed);
        if (j.lk_done) (void)hipEventDestroy(j.lk_done);
    }
    delete a;
}

static int analyzer_put(pc_analyzer* a, int32_t frame_id, const uint8_t* rgb, size_t row_pitch, int on_device,
                        int will_detect, int channels, int elem_size) {
    if (!a || !rgb) return fail(PC_E_INVALID, "null argument");
    const int n = (int)a->slots.size();
    Slot& s = a->slots[(size_t)(((frame_id % n) + n) % n)];
    PC_HIP(hipSetDevice(a->ctx->device));
    PrepScope prep(a->ctx);
    // an LK launch in flight may still read the frame this slot holds
    if (s.last_read) PC_HIP(hipStreamWaitEvent(a->ctx->prep_stream, s.last_read, 0));
    s.last_read = nullptr;
    int rc = set_image(a->ctx, s.frame, rgb, row_pitch, on_device, channels, elem_size);
    if (rc != PC_OK) {
        s.valid = false;
        return rc;
    }
    PC_HIP(hipEventRecord(s.img_ready, a->ctx->prep_stream));
    s.frame_id = frame_id;
    s.valid = true;
    s.det = DET_NONE;
    s.supplied = false;
    if (will_detect) return detect_dense(a, s);
    return PC_OK;
}

int pc_analyzer_put_frame(pc_analyzer* a, int32_t frame_id, const uint8_t* rgb, size_t row_pitch, int on_device,
                          int will_detect) {
    return analyzer_put(a, frame_id, rgb, row_pitch, on_device, will_detect, 3, 1);
}

int pc_analyzer_put_frame_f32(pc_analyzer* a, int32_t frame_id, const float* rgb, size_t row_pitch, int channels,
                              int on_device, int will_detect) {
    return analyzer_put(a, frame_id, reinterpret_cast<const uint8_t*>(rgb), row_pitch, on_device, will_detect, channels, 4);
}

int pc_analyzer_has_frame(const pc_analyzer* a, int32_t frame_id) {
    if (!a) return 0;
    return find_slot(const_cast<pc_analyzer*>(a), frame_id) != nullptr;
}

int pc_analyzer_set_keypoints(pc_analyzer* a, int32_t frame_id, const float* xy, int n) {
    if (!a || n < 0 || (!xy && n > 0)) return fail(PC_E_INVALID, "bad argument");
    Slot* s = find_slot(a, frame_id);
    if (!s) return fail(PC_E_STATE, "frame %d is not resident", frame_id);
    int rc = ensure_kp_capacity(s->frame, n);
    if (rc != PC_OK) return rc;
    if (n > 0) {
        // resume path (keypoints from the database): pageable source, so the copy is synchronous
        PC_HIP(hipMemcpyAsync(s->frame->d_kps, xy, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, a->ctx->prep_stream));
        PC_HIP(hipStreamSynchronize(a->ctx->prep_stream));
    }
    s->frame->n_kps = n;
    {
        PrepScope prep(a->ctx);
        if ((rc = order_keypoints_spatially(a->ctx, s->frame, a->ctx->prep_hist)) != PC_OK) return rc;
        PC_HIP(hipEventRecord(s->kps_ready, a->ctx->prep_stream));
    }
    s->det = DET_DONE;
    s->supplied = true;
    return PC_OK;
}

namespace {
struct SlowSection {   // POLYCHASE_TRACE_ALLOC: report host-side sections of a call that take more than 2 ms
    const char* name;
    std::chrono::steady_clock::time_point t0;
    explicit SlowSection(const char* n) : name(n), t0(std::chrono::steady_clock::now()) {}
    ~SlowSection() {
        if (!pc::trace_allocations()) return;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms > 2.0) fprintf(stderr, "[polychase_hip] slow host section %s: %.2f ms\n", name, ms);
    }
};
}  // namespace

int pc_analyzer_submit(pc_analyzer* a, int32_t frame1, const int32_t* targets, int n_targets) {
    if (!a || (n_targets > 0 && !targets)) return fail(PC_E_INVALID, "null argument");
    if (n_targets < 0 || n_targets > PC_MAX_TARGETS) return fail(PC_E_INVALID, "n_targets must be in [0,%d]", PC_MAX_TARGETS);
    if (a->job_count == a->jobs.size()) return fail(PC_E_STATE, "%zu jobs already in flight: call pc_analyzer_collect", a->job_count);
    pc_context* ctx = a->ctx;
    PC_HIP(hipSetDevice(ctx->device));
    Slot* s1 = find_slot(a, frame1);
    if (!s1) return fail(PC_E_STATE, "frame1 %d is not resident", frame1);
    const pc_frame* tg[PC_MAX_TARGETS];
    for (int t = 0; t < n_targets; t++) {
        Slot* st = find_slot(a, targets[t]);
        if (!st) return fail(PC_E_STATE, "target frame %d is not resident", targets[t]);
        tg[t] = st->frame;
    }
    int rc;
    // (1) keypoints of frame1: the dense phase ran when the frame became resident; order them now
    bool detected = false;
    if (s1->det != DET_DONE) {
        SlowSection ss("submit/detect_finish");
        PrepScope prep(a->ctx);
        if ((rc = detect_finish(a, *s1)) != PC_OK) return rc;
        detected = true;
    } else {
        detected = !s1->supplied;
    }
    Job& j = a->jobs[(a->job_head + a->job_count) % a->jobs.size()];
    // (2) the LK launch needs this frame's keypoints and the pyramids of the frames it reads -- not the
    // detection of frames that were made resident for later
    {
        SlowSection ss("submit/waits");
        PC_HIP(hipStreamWaitEvent(ctx->stream, s1->kps_ready, 0));
        PC_HIP(hipStreamWaitEvent(ctx->stream, s1->img_ready, 0));
        for (int t = 0; t < n_targets; t++) PC_HIP(hipStreamWaitEvent(ctx->stream, find_slot(a, targets[t])->img_ready, 0));
    }
    const int n = s1->frame->n_kps;
    const size_t rows = (size_t)n * (size_t)std::max(n_targets, 0);
    // packed record layout (= a device-log record without its 128-byte header)
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    j.o_kps = 128;
    j.o_idx = up16(j.o_kps + (size_t)n * 8);
    j.o_xy = up16(j.o_idx + rows * 4);
    j.o_err = up16(j.o_xy + rows * 8);
    j.pack_bytes = up16(j.o_err + rows * 4);
    {
        SlowSection ss("submit/ensure pack");
        PC_HIP(j.h_pack.ensure(j.pack_bytes));
        PC_HIP(ctx->lk_pack.ensure(j.pack_bytes));
    }
    j.frame1 = frame1;
    j.n_kps = n;
    j.detected = detected;
    j.n_targets = n_targets;
    for (int t = 0; t < n_targets; t++) j.targets[t] = targets[t];
    // (3) LK on the main stream, into output set `set`; its compaction (status filter), the device-log copies and the
    // downloads on the copy stream, so that the next LK launch starts right behind this one
    const int set = (int)(a->submitted & 1);
    hipStream_t post = ctx->copy_stream;
    ctx->prep_dirty = true;   // stage-level calls must order themselves behind the side streams
    if (n_targets > 0) {
        if (a->fopt.window_size != s1->frame->win) return fail(PC_E_INVALID, "window size mismatch");
        // the compaction of the job two submits ago read this output set
        SlowSection ss("submit/run_lk");
        if (a->set_free[set]) PC_HIP(hipStreamWaitEvent(ctx->stream, a->set_free[set], 0));
        if ((rc = run_lk(ctx, s1->frame, tg, n_targets, &a->fopt, set)) != PC_OK) return rc;
    }
    SlowSection ss_post("submit/post-stream enqueue");
    PC_HIP(hipEventRecord(j.lk_done, ctx->stream));
    s1->last_read = j.lk_done;
    for (int t = 0; t < n_targets; t++) find_slot(a, targets[t])->last_read = j.lk_done;
    PC_HIP(hipStreamWaitEvent(post, j.lk_done, 0));
    uint8_t* const pack = ctx->lk_pack.p;
    long long* const p_ro = reinterpret_cast<long long*>(pack);
    PC_HIP(hipMemsetAsync(pack, 0, 128, post));
    if (n_targets > 0) {
        const int nblocks = pc::compact_num_blocks(n);
        PC_HIP(ctx->lk_block_counts.ensure((size_t)nblocks * n_targets + 1));
        // the previous job's download reads the pack: it precedes this compaction on the same stream
        ScopedTimer t(ctx, PC_K_COMPACT, post);
        pc::launch_compact(ctx->lk_xy[set].p, ctx->lk_status[set].p, ctx->lk_err[set].p, n, n_targets, ctx->lk_block_counts.p, p_ro,
                           reinterpret_cast<uint32_t*>(pack + j.o_idx), reinterpret_cast<float2*>(pack + j.o_xy),
                           reinterpret_cast<float*>(pack + j.o_err), post);
    }
    pc::launch_copy_keypoints(s1->frame->d_kps, reinterpret_cast<float2*>(pack + j.o_kps), n, post);
    if (a->d_log) {
        // device log: header from pinned memory, the record itself is the pack (one device-to-device copy)
        const size_t o_hdr = a->log_used, end = o_hdr + 128 + j.pack_bytes;
        if (end > a->log_cap) return fail(PC_E_CAPACITY, "device log full (%zu of %zu bytes)", end, a->log_cap);
        const size_t slot_i = (a->job_head + a->job_count) % a->jobs.size();
        if (a->log_hdr.size() != a->jobs.size()) a->log_hdr.resize(a->jobs.size());
        PC_HIP(a->log_hdr[slot_i].ensure(16));
        long long* hh = a->log_hdr[slot_i].p;
        for (int k = 0; k < 16; k++) hh[k] = 0;
        hh[0] = PC_LOG_MAGIC;
        hh[1] = frame1;
        hh[2] = n;
        hh[3] = n_targets;
        for (int t = 0; t < n_targets; t++) hh[4 + t] = targets[t];
        hh[12] = (long long)rows;
        PC_HIP(hipMemcpyAsync(a->d_log + o_hdr, hh, 128, hipMemcpyHostToDevice, post));
        PC_HIP(hipMemcpyAsync(a->d_log + o_hdr + 128, pack, j.pack_bytes, hipMemcpyDeviceToDevice, post));
        a->log_used = end;
    }
    PC_HIP(hipEventRecord(j.computed, post));
    a->set_free[set] = j.computed;
    a->submitted++;
    // (4) download, behind the compaction on the same stream: ONE copy with fixed endpoints (the context's pack ->
    // the job's pinned pack).  The runtime stalls the host for 5-8 ms the first time it sees a buffer as a copy
    // source, so per-frame buffers must not appear here.
    PC_HIP(hipMemcpyAsync(j.h_pack.p, pack, j.pack_bytes, hipMemcpyDeviceToHost, ctx->copy_stream));
    PC_HIP(hipEventRecord(j.done, ctx->copy_stream));
    a->job_count++;
    // (5) while this LK launch runs: order the keypoints of the next frame1
    SlowSection ss("submit/preorder");
    return preorder_if_ready(a, frame1 + 1);
}

int pc_analyzer_pending(const pc_analyzer* a) { return a ? (int)a->job_count : 0; }

int pc_analyzer_set_device_log(pc_analyzer* a, void* d_log, size_t capacity_bytes) {
    if (!a) return fail(PC_E_INVALID, "null analyzer");
    if (d_log && (reinterpret_cast<uintptr_t>(d_log) & 15)) return fail(PC_E_INVALID, "device log must be 16-byte aligned");
    PC_HIP(hipStreamSynchronize(a->ctx->prep_stream));
    PC_HIP(hipStreamSynchronize(a->ctx->stream));
    a->d_log = static_cast<uint8_t*>(d_log);
    a->log_cap = d_log ? capacity_bytes : 0;
    a->log_used = 0;
    return PC_OK;
}

int pc_analyzer_device_log_used(const pc_analyzer* a, size_t* bytes) {
    if (!a || !bytes) return fail(PC_E_INVALID, "null argument");
    *bytes = a->log_used;
    return PC_OK;
}

int pc_analyzer_collect(pc_analyzer* a, pc_frame_result* out) {
    if (!a || !out) return fail(PC_E_INVALID, "null argument");
    if (a->job_count == 0) return fail(PC_E_STATE, "no job in flight");
    Job& j = a->jobs[a->job_head];
    PC_HIP(hipEventSynchronize(j.done));
    // let the runtime retire the finished commands of the other streams now, a few at a time: left alone it does
    // so in one batch of several milliseconds every couple of hundred frames, inside some later launch
    (void)hipStreamQuery(a->ctx->stream);
    (void)hipStreamQuery(a->ctx->prep_stream);
    out->frame1 = j.frame1;
    out->n_keypoints = j.n_kps;
    out->keypoints_detected = j.detected ? 1 : 0;
    out->keypoints_xy = reinterpret_cast<const float*>(j.h_pack.p + j.o_kps);
    out->n_targets = j.n_targets;
    for (int t = 0; t < PC_MAX_TARGETS; t++) out->targets[t] = t < j.n_targets ? j.targets[t] : 0;
    const long long* h_ro = reinterpret_cast<const long long*>(j.h_pack.p);
    for (int t = 0; t <= PC_MAX_TARGETS; t++) out->row_offset[t] = (int64_t)h_ro[std::min(t, j.n_targets)];
    out->src_indices = reinterpret_cast<const uint32_t*>(j.h_pack.p + j.o_idx);
    out->tgt_xy = reinterpret_cast<const float*>(j.h_pack.p + j.o_xy);
    out->flow_err = reinterpret_cast<const float*>(j.h_pack.p + j.o_err);
    a->job_head = (a->job_head + 1) % a->jobs.size();
    a->job_count--;
    return PC_OK;
}


// =============================================================================================
// tracker path: meshes, batched ray casting, PnP accumulation
// =============================================================================================
}  // extern "C"

struct pc_mesh {
    pc_context* ctx = nullptr;
    int n_vertices = 0, n_triangles = 0;
    DevBuf<float> verts;
    DevBuf<uint32_t> tris, mask;
    // LBVH (bvh.hpp): n_triangles - 1 internal nodes + the sorted leaf order
    DevBuf<pc::BvhNode> bvh_nodes;
    DevBuf<uint32_t> bvh_leaf_tri;
    pc::BvhView bvh() const {
        pc::BvhView v;
        v.nodes = bvh_nodes.p;
        v.leaf_tri = bvh_leaf_tri.p;
        v.verts = verts.p;
        v.tris = tris.p;
        v.n_tris = n_triangles;
        return v;
    }
    // per-call scratch
    DevBuf<float2> d_xy;
    DevBuf<uint8_t> d_hit;
    DevBuf<float> d_pos, d_uvt;
    DevBuf<uint32_t> d_prim;
};

struct pc_pnp_problem {
    pc_context* ctx = nullptr;
    int n = 0;
    bool has_weights = false;
    DevBuf<float> X, x, w, partials, out;
    PinBuf<float> h_out;
};

extern "C" {

int pc_mesh_create(pc_context* ctx, const float* vertices, int n_vertices, const uint32_t* triangles,
                   int n_triangles, pc_mesh** out) {
    if (!ctx || !out || n_vertices < 0 || n_triangles < 0 || (n_vertices > 0 && !vertices) ||
        (n_triangles > 0 && !triangles))
        return fail(PC_E_INVALID, "bad argument");
    *out = nullptr;
    for (int i = 0; i < 3 * n_triangles; i++)
        if (triangles[i] >= (uint32_t)n_vertices) return fail(PC_E_INVALID, "triangle index %u out of range", triangles[i]);
    PC_HIP(hipSetDevice(ctx->device));
    pc_mesh* m = new (std::nothrow) pc_mesh();
    if (!m) return fail(PC_E_INVALID, "out of host memory");
    m->ctx = ctx;
    m->n_vertices = n_vertices;
    m->n_triangles = n_triangles;
    const int words = (n_triangles + 31) / 32 + 4;
    hipError_t e = m->verts.ensure((size_t)std::max(1, n_vertices) * 3);
    if (e == hipSuccess) e = m->tris.ensure((size_t)std::max(1, n_triangles) * 3);
    if (e == hipSuccess) e = m->mask.ensure((size_t)words);
    if (e == hipSuccess && n_vertices) e = hipMemcpyAsync(m->verts.p, vertices, (size_t)n_vertices * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && n_triangles) e = hipMemcpyAsync(m->tris.p, triangles, (size_t)n_triangles * 3 * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(m->mask.p, 0, (size_t)words * sizeof(uint32_t), ctx->stream);
    // acceleration structure (rtcCommitScene in the reference, ray_casting.cc:23-63): LBVH built on the GPU
    if (e == hipSuccess && n_triangles > 0) {
        const size_t n = (size_t)n_triangles;
        float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
        for (int v = 0; v < n_vertices; v++)
            for (int k = 0; k < 3; k++) {
                const float x = vertices[3 * (size_t)v + k];
                if (v == 0 || x < lo[k]) lo[k] = x;
                if (v == 0 || x > hi[k]) hi[k] = x;
            }
        const float extent = std::max(std::max(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
        const float pad = 1e-5f * extent + 1e-30f;
        DevBuf<unsigned long long> keys_in, keys_out;
        DevBuf<float> box_lo, box_hi;
        DevBuf<int> links;      // parent (2n-1) | visits (n) | left (n) | right (n)
        DevBuf<uint32_t> bounds;
        DevBuf<uint8_t> sort_temp;
        const size_t temp_bytes = pc::bvh_sort_temp_bytes(n_triangles);
        e = m->bvh_nodes.ensure(n);
        if (e == hipSuccess) e = m->bvh_leaf_tri.ensure(n);
        if (e == hipSuccess) e = keys_in.ensure(n);
        if (e == hipSuccess) e = keys_out.ensure(n);
        if (e == hipSuccess) e = box_lo.ensure(3 * (2 * n));
        if (e == hipSuccess) e = box_hi.ensure(3 * (2 * n));
        if (e == hipSuccess) e = links.ensure(5 * n + 8);
        if (e == hipSuccess) e = bounds.ensure(8);
        if (e == hipSuccess) e = sort_temp.ensure(temp_bytes + 16);
        if (e == hipSuccess) {
            pc::BvhBuildScratch sc;
            sc.keys_in = keys_in.p;
            sc.keys_out = keys_out.p;
            sc.box_lo = box_lo.p;
            sc.box_hi = box_hi.p;
            sc.parent = links.p;
            sc.visits = links.p + 2 * n;
            sc.left = links.p + 3 * n;
            sc.right = links.p + 4 * n;
            sc.bounds = bounds.p;
            e = pc::bvh_build(m->verts.p, m->tris.p, n_triangles, pad, sc, sort_temp.p, temp_bytes, m->bvh_nodes.p,
                              m->bvh_leaf_tri.p, ctx->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        keys_in.release();
        keys_out.release();
        box_lo.release();
        box_hi.release();
        links.release();
        bounds.release();
        sort_temp.release();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        pc_mesh_destroy(m);
        return fail(PC_E_HIP, "mesh upload failed: %s", hipGetErrorString(e));
    }
    *out = m;
    return PC_OK;
}

int pc_mesh_set_mask(pc_context* ctx, pc_mesh* mesh, const uint32_t* mask_words, int n_words) {
    if (!ctx || !mesh || !mask_words) return fail(PC_E_INVALID, "null argument");
    const int need = (mesh->n_triangles + 31) / 32;
    if (n_words < need) return fail(PC_E_INVALID, "mask has %d words, %d needed", n_words, need);
    if (need > 0) {
        PC_HIP(hipMemcpyAsync(mesh->mask.p, mask_words, (size_t)need * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        PC_HIP(hipStreamSynchronize(ctx->stream));
    }
    return PC_OK;
}

void pc_mesh_destroy(pc_mesh* m) {
    if (!m) return;
    if (m->ctx) {
        (void)hipSetDevice(m->ctx->device);
        (void)hipStreamSynchronize(m->ctx->stream);
    }
    m->verts.release();
    m->tris.release();
    m->mask.release();
    m->bvh_nodes.release();
    m->bvh_leaf_tri.release();
    m->d_xy.release();
    m->d_hit.release();
    m->d_pos.release();
    m->d_uvt.release();
    m->d_prim.release();
    delete m;
}

static int raycast_pixels(pc_context* ctx, const pc_mesh* mesh_c, const pc_ray_camera* cam, const float* xy, int n,
                          int check_mask, uint8_t* hit, float* pos, uint32_t* prim, float* uvt, bool sweep) {
    if (!ctx || !mesh_c || !cam || n < 0) return fail(PC_E_INVALID, "bad argument");
    if (n == 0) return PC_OK;
    if (!xy || !hit || !pos || !prim || !uvt) return fail(PC_E_INVALID, "null buffer");
    pc_mesh* mesh = const_cast<pc_mesh*>(mesh_c);
    PC_HIP(hipSetDevice(ctx->device));
    PC_HIP(mesh->d_xy.ensure((size_t)n));
    PC_HIP(mesh->d_hit.ensure((size_t)n));
    PC_HIP(mesh->d_pos.ensure((size_t)n * 3));
    PC_HIP(mesh->d_uvt.ensure((size_t)n * 3));
    PC_HIP(mesh->d_prim.ensure((size_t)n));
    pc::RayCamera rc;
    std::memcpy(rc.m, cam->dir_matrix, sizeof(rc.m));
    std::memcpy(rc.origin, cam->origin, sizeof(rc.origin));
    rc.fx = cam->fx;
    rc.fy = cam->fy;
    rc.cx = cam->cx;
    rc.cy = cam->cy;
    rc.sign = cam->unproject_sign;
    PC_HIP(hipMemcpyAsync(mesh->d_xy.p, xy, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, ctx->stream));
    if (sweep)
        pc::launch_raycast_sweep(mesh->verts.p, mesh->tris.p, mesh->n_triangles, mesh->mask.p, check_mask, rc, mesh->d_xy.p, n,
                                 mesh->d_hit.p, mesh->d_pos.p, mesh->d_prim.p, mesh->d_uvt.p, ctx->stream);
    else
        pc::launch_raycast(mesh->bvh(), mesh->mask.p, check_mask, rc, mesh->d_xy.p, n, mesh->d_hit.p, mesh->d_pos.p,
                           mesh->d_prim.p, mesh->d_uvt.p, ctx->stream);
    PC_HIP(hipMemcpyAsync(hit, mesh->d_hit.p, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    PC_HIP(hipMemcpyAsync(pos, mesh->d_pos.p, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    PC_HIP(hipMemcpyAsync(prim, mesh->d_prim.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    PC_HIP(hipMemcpyAsync(uvt, mesh->d_uvt.p, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    PC_HIP(hipStreamSynchronize(ctx->stream));
    return PC_OK;
}

int pc_raycast_pixels(pc_context* ctx, const pc_mesh* mesh, const pc_ray_camera* cam, const float* xy, int n, int check_mask,
                      uint8_t* hit, float* pos, uint32_t* prim, float* uvt) {
    return raycast_pixels(ctx, mesh, cam, xy, n, check_mask, hit, pos, prim, uvt, false);
}

int pc_raycast_pixels_sweep(pc_context* ctx, const pc_mesh* mesh, const pc_ray_camera* cam, const float* xy, int n,
                            int check_mask, uint8_t* hit, float* pos, uint32_t* prim, float* uvt) {
    return raycast_pixels(ctx, mesh, cam, xy, n, check_mask, hit, pos, prim, uvt, true);
}

int pc_pnp_problem_create(pc_context* ctx, const float* X, const float* x, const float* weights, int n,
                          pc_pnp_problem** out) {
    if (!ctx || !out || n < 1 || !X || !x) return fail(PC_E_INVALID, "bad argument");
    *out = nullptr;
    PC_HIP(hipSetDevice(ctx->device));
    pc_pnp_problem* p = new (std::nothrow) pc_pnp_problem();
    if (!p) return fail(PC_E_INVALID, "out of host memory");
    p->ctx = ctx;
    p->n = n;
    p->has_weights = weights != nullptr;
    const int nb = pc::pnp_num_blocks(n);
    hipError_t e = p->X.ensure((size_t)n * 3);
    if (e == hipSuccess) e = p->x.ensure((size_t)n * 2);
    if (e == hipSuccess && weights) e = p->w.ensure((size_t)n);
    if (e == hipSuccess) e = p->partials.ensure((size_t)nb * 56);
    if (e == hipSuccess) e = p->out.ensure(64);
    if (e == hipSuccess) e = p->h_out.ensure(64);
    if (e == hipSuccess) e = hipMemcpyAsync(p->X.p, X, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(p->x.p, x, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && weights) e = hipMemcpyAsync(p->w.p, weights, (size_t)n * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        pc_pnp_problem_destroy(p);
        return fail(PC_E_HIP, "PnP upload failed: %s", hipGetErrorString(e));
    }
    *out = p;
    return PC_OK;
}

void pc_pnp_problem_destroy(pc_pnp_problem* p) {
    if (!p) return;
    if (p->ctx) {
        (void)hipSetDevice(p->ctx->device);
        (void)hipStreamSynchronize(p->ctx->stream);
    }
    p->X.release();
    p->x.release();
    p->w.release();
    p->partials.release();
    p->out.release();
    p->h_out.release();
    delete p;
}

static pc::PnPParams to_kernel_params(const pc_pnp_params* q) {
    pc::PnPParams p;
    std::memcpy(p.R, q->R, sizeof(p.R));
    std::memcpy(p.t, q->t, sizeof(p.t));
    p.fx = q->fx;
    p.fy = q->fy;
    p.cx = q->cx;
    p.cy = q->cy;
    p.aspect_ratio = q->aspect_ratio;
    p.convention_opencv = q->convention_opencv;
    p.optimize_focal = q->optimize_focal_length;
    p.optimize_pp = q->optimize_principal_point;
    p.loss_type = q->loss_type;
    p.loss_scale = q->loss_scale;
    return p;
}

int pc_pnp_normal_equations(pc_context* ctx, const pc_pnp_problem* prob, const pc_pnp_params* params,
                            float* jtj_lower45, float* jtr9, int* valid) {
    if (!ctx || !prob || !params || !jtj_lower45 || !jtr9) return fail(PC_E_INVALID, "null argument");
    if (params->loss_type < 0 || params->loss_type > 2) return fail(PC_E_INVALID, "Unknown loss type: %d", params->loss_type);
    PC_HIP(hipSetDevice(ctx->device));
    pc::launch_pnp_normal_eq(prob->X.p, prob->x.p, prob->has_weights ? prob->w.p : nullptr, prob->n,
                             to_kernel_params(params), prob->partials.p, prob->out.p, ctx->stream);
    PC_HIP(hipMemcpyAsync(prob->h_out.p, prob->out.p, 56 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    PC_HIP(hipStreamSynchronize(ctx->stream));
    std::memcpy(jtj_lower45, prob->h_out.p, 45 * sizeof(float));
    std::memcpy(jtr9, prob->h_out.p + 45, 9 * sizeof(float));
    if (valid) *valid = (int)prob->h_out.p[54];
    return PC_OK;
}

int pc_pnp_total_cost(pc_context* ctx, const pc_pnp_problem* prob, const pc_pnp_params* params,
                      float max_inlier_error_sq, float* cost, int* valid, int* inliers) {
    if (!ctx || !prob || !params || !cost) return fail(PC_E_INVALID, "null argument");
    if (params->loss_type < 0 || params->loss_type > 2) return fail(PC_E_INVALID, "Unknown loss type: %d", params->loss_type);
    PC_HIP(hipSetDevice(ctx->device));
    pc::launch_pnp_cost(prob->X.p, prob->x.p, prob->has_weights ? prob->w.p : nullptr, prob->n, to_kernel_params(params),
                        max_inlier_error_sq, prob->partials.p, prob->out.p, ctx->stream);
    PC_HIP(hipMemcpyAsync(prob->h_out.p, prob->out.p, 4 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    PC_HIP(hipStreamSynchronize(ctx->stream));
    *cost = prob->h_out.p[0];
    if (valid) *valid = (int)prob->h_out.p[1];
    if (inliers) *inliers = (int)prob->h_out.p[2];
    return PC_OK;
}


// =============================================================================================
// refiner path
// =============================================================================================
}  // extern "C"

struct pc_refine_problem {
    pc_context* ctx = nullptr;
    const pc_mesh* mesh = nullptr;
    int n_frames = 0, n_edges = 0, block_len = 6, opt_f = 0, opt_pp = 0;
    size_t n_kp = 0, n_res = 0;
    DevBuf<int> kp_offset, edge_src, edge_tgt, edge_offset, edge_valid;
    DevBuf<float2> kp_xy, res_tgt_xy;
    DevBuf<double2> edge_cost;
    DevBuf<uint32_t> res_src_kp, prim_cache;
    DevBuf<float> edge_weight;
    DevBuf<double> edge_blocks;
    DevBuf<uint8_t> frame_fixed;
    DevBuf<pc::RefineCamera> cams;
    PinBuf<double2> h_edge_cost;
    std::vector<float> h_edge_weight;
    float model[16], model_inv[16];
};

namespace {

pc::RefineProblemView refine_view(const pc_refine_problem* p) {
    pc::RefineProblemView v;
    v.n_frames = p->n_frames;
    v.n_edges = p->n_edges;
    v.n_tris = p->mesh->n_triangles;
    v.kp_offset = p->kp_offset.p;
    v.kp_xy = p->kp_xy.p;
    v.edge_src = p->edge_src.p;
    v.edge_tgt = p->edge_tgt.p;
    v.edge_offset = p->edge_offset.p;
    v.res_src_kp = p->res_src_kp.p;
    v.res_tgt_xy = p->res_tgt_xy.p;
    v.edge_weight = p->edge_weight.p;
    v.frame_fixed = p->frame_fixed.p;
    v.prim_cache = p->prim_cache.p;
    v.verts = p->mesh->verts.p;
    v.tris = p->mesh->tris.p;
    v.mask = p->mesh->mask.p;
    v.bvh = p->mesh->bvh();
    std::memcpy(v.model, p->model, sizeof(v.model));
    std::memcpy(v.model_inv, p->model_inv, sizeof(v.model_inv));
    return v;
}

int upload_cameras(pc_context* ctx, pc_refine_problem* p, const pc_refine_camera* cameras) {
    std::vector<pc::RefineCamera> h((size_t)p->n_frames);
    for (int f = 0; f < p->n_frames; f++) {
        std::memcpy(h[f].R, cameras[f].R, sizeof(h[f].R));
        std::memcpy(h[f].t, cameras[f].t, sizeof(h[f].t));
        h[f].fx = cameras[f].fx;
        h[f].fy = cameras[f].fy;
        h[f].cx = cameras[f].cx;
        h[f].cy = cameras[f].cy;
        h[f].aspect = cameras[f].aspect_ratio;
        h[f].sign = cameras[f].unproject_sign;
    }
    PC_HIP(hipMemcpyAsync(p->cams.p, h.data(), h.size() * sizeof(pc::RefineCamera), hipMemcpyHostToDevice, ctx->stream));
    PC_HIP(hipStreamSynchronize(ctx->stream));  // `h` is pageable and local
    return PC_OK;
}

template <typename T, typename U>
hipError_t upload(DevBuf<T>& dst, const U* src, size_t n, hipStream_t s) {
    static_assert(sizeof(T) == sizeof(U) || sizeof(T) == 2 * sizeof(U), "layout");
    hipError_t e = dst.ensure(n ? n : 1);
    if (e == hipSuccess && n) e = hipMemcpyAsync(dst.p, src, n * sizeof(T), hipMemcpyHostToDevice, s);
    return e;
}

}  // namespace

extern "C" {

int pc_refine_problem_create(pc_context* ctx, const pc_mesh* mesh, const pc_refine_desc* d, pc_refine_problem** out) {
    if (!ctx || !mesh || !d || !out) return fail(PC_E_INVALID, "null argument");
    *out = nullptr;
    if (d->n_frames < 3) return fail(PC_E_INVALID, "a segment needs more than 2 frames");  // CHECK(traj.Count() > 2)
    if (d->n_edges < 0 || (d->block_len != 6 && d->block_len != 9)) return fail(PC_E_INVALID, "bad problem description");
    if (!d->kp_offset || !d->edge_offset || (d->n_edges > 0 && (!d->edge_src || !d->edge_tgt || !d->edge_weight)))
        return fail(PC_E_INVALID, "null array");
    const size_t n_kp = (size_t)d->kp_offset[d->n_frames], n_res = (size_t)d->edge_offset[d->n_edges];
    for (int e = 0; e < d->n_edges; e++) {
        if (d->edge_src[e] < 0 || d->edge_src[e] >= d->n_frames || d->edge_tgt[e] < 0 || d->edge_tgt[e] >= d->n_frames ||
            d->edge_src[e] == d->edge_tgt[e])
            return fail(PC_E_INVALID, "edge %d connects invalid frames", e);
        const size_t src_kps = (size_t)(d->kp_offset[d->edge_src[e] + 1] - d->kp_offset[d->edge_src[e]]);
        for (int r = d->edge_offset[e]; r < d->edge_offset[e + 1]; r++)
            if (d->res_src_kp[r] >= src_kps) return fail(PC_E_INVALID, "edge %d references keypoint %u of %zu", e, d->res_src_kp[r], src_kps);
    }
    PC_HIP(hipSetDevice(ctx->device));
    pc_refine_problem* p = new (std::nothrow) pc_refine_problem();
    if (!p) return fail(PC_E_INVALID, "out of host memory");
    p->ctx = ctx;
    p->mesh = mesh;
    p->n_frames = d->n_frames;
    p->n_edges = d->n_edges;
    p->block_len = d->block_len;
    p->opt_f = d->optimize_focal_length ? 1 : 0;
    p->opt_pp = d->optimize_principal_point ? 1 : 0;
    p->n_kp = n_kp;
    p->n_res = n_res;
    std::memcpy(p->model, d->model_matrix, sizeof(p->model));
    std::memcpy(p->model_inv, d->model_matrix_inv, sizeof(p->model_inv));
    std::vector<uint8_t> fixed((size_t)d->n_frames, 0);
    fixed.front() = fixed.back() = 1;  // IsGroundTruth (refiner.cc:268-271)
    const int B2 = 2 * d->block_len, nacc = B2 * (B2 + 1) / 2 + B2;
    hipStream_t s = ctx->stream;
    hipError_t e = upload(p->kp_offset, d->kp_offset, (size_t)d->n_frames + 1, s);
    if (e == hipSuccess) e = upload(p->kp_xy, d->kp_xy, n_kp, s);
    if (e == hipSuccess) e = upload(p->edge_src, d->edge_src, (size_t)d->n_edges, s);
    if (e == hipSuccess) e = upload(p->edge_tgt, d->edge_tgt, (size_t)d->n_edges, s);
    if (e == hipSuccess) e = upload(p->edge_offset, d->edge_offset, (size_t)d->n_edges + 1, s);
    if (e == hipSuccess) e = upload(p->res_src_kp, d->res_src_kp, n_res, s);
    if (e == hipSuccess) e = upload(p->res_tgt_xy, d->res_tgt_xy, n_res, s);
    if (e == hipSuccess) e = upload(p->edge_weight, d->edge_weight, (size_t)d->n_edges, s);
    if (e == hipSuccess) e = upload(p->frame_fixed, fixed.data(), fixed.size(), s);
    if (e == hipSuccess) e = p->prim_cache.ensure(n_kp ? n_kp : 1);
    if (e == hipSuccess) e = hipMemsetAsync(p->prim_cache.p, 0xff, (n_kp ? n_kp : 1) * sizeof(uint32_t), s);
    if (e == hipSuccess) e = p->edge_cost.ensure((size_t)std::max(1, d->n_edges));
    if (e == hipSuccess) e = p->h_edge_cost.ensure((size_t)std::max(1, d->n_edges));
    if (e == hipSuccess) e = p->edge_valid.ensure((size_t)std::max(1, d->n_edges));
    if (e == hipSuccess) e = p->edge_blocks.ensure((size_t)std::max(1, d->n_edges) * nacc);
    if (e == hipSuccess) e = p->cams.ensure((size_t)d->n_frames);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        pc_refine_problem_destroy(p);
        return fail(PC_E_HIP, "refine problem upload failed: %s", hipGetErrorString(e));
    }
    p->h_edge_weight.assign(d->edge_weight, d->edge_weight + d->n_edges);
    *out = p;
    return PC_OK;
}

void pc_refine_problem_destroy(pc_refine_problem* p) {
    if (!p) return;
    if (p->ctx) {
        (void)hipSetDevice(p->ctx->device);
        (void)hipStreamSynchronize(p->ctx->stream);
    }
    p->kp_offset.release();
    p->edge_src.release();
    p->edge_tgt.release();
    p->edge_offset.release();
    p->edge_valid.release();
    p->kp_xy.release();
    p->res_tgt_xy.release();
    p->edge_cost.release();
    p->res_src_kp.release();
    p->prim_cache.release();
    p->edge_weight.release();
    p->edge_blocks.release();
    p->frame_fixed.release();
    p->cams.release();
    p->h_edge_cost.release();
    delete p;
}

int pc_refine_total_cost(pc_context* ctx, pc_refine_problem* p, const pc_refine_camera* cameras, int loss_type,
                         float loss_scale, double* cost) {
    if (!ctx || !p || !cameras || !cost) return fail(PC_E_INVALID, "null argument");
    if (loss_type < 0 || loss_type > 2) return fail(PC_E_INVALID, "Unknown loss type: %d", loss_type);
    PC_HIP(hipSetDevice(ctx->device));
    int rc = upload_cameras(ctx, p, cameras);
    if (rc != PC_OK) return rc;
    *cost = 0.0;
    if (p->n_edges == 0) return PC_OK;
    pc::launch_refine_cost(refine_view(p), p->cams.p, loss_type, loss_scale, p->edge_cost.p, ctx->stream);
    PC_HIP(hipMemcpyAsync(p->h_edge_cost.p, p->edge_cost.p, (size_t)p->n_edges * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
    PC_HIP(hipStreamSynchronize(ctx->stream));
    // cost = sum_e edge_weight * (edge loss sum / valid)   (lev_marq.h:812-820), fixed edge order
    double total = 0.0;
    for (int e = 0; e < p->n_edges; e++) {
        const float w = p->h_edge_weight[(size_t)e];
        if (w == 0.0f) continue;
        double edge_cost = p->h_edge_cost.p[e].x;
        if (p->h_edge_cost.p[e].y > 0.0) edge_cost /= p->h_edge_cost.p[e].y;
        total += (double)w * edge_cost;
    }
    *cost = total;
    return PC_OK;
}

int pc_refine_normal_equations(pc_context* ctx, pc_refine_problem* p, const pc_refine_camera* cameras, int loss_type,
                               float loss_scale, double* edge_blocks, int* edge_valid) {
    if (!ctx || !p || !cameras || !edge_blocks) return fail(PC_E_INVALID, "null argument");
    if (loss_type < 0 || loss_type > 2) return fail(PC_E_INVALID, "Unknown loss type: %d", loss_type);
    PC_HIP(hipSetDevice(ctx->device));
    int rc = upload_cameras(ctx, p, cameras);
    if (rc != PC_OK) return rc;
    if (p->n_edges == 0) return PC_OK;
    const int B2 = 2 * p->block_len, nacc = B2 * (B2 + 1) / 2 + B2;
    pc::launch_refine_normal_eq(refine_view(p), p->cams.p, loss_type, loss_scale, p->block_len, p->opt_f, p->opt_pp,
                                p->edge_blocks.p, p->edge_valid.p, ctx->stream);
    PC_HIP(hipMemcpyAsync(edge_blocks, p->edge_blocks.p, (size_t)p->n_edges * nacc * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (edge_valid)
        PC_HIP(hipMemcpyAsync(edge_valid, p->edge_valid.p, (size_t)p->n_edges * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PC_HIP(hipStreamSynchronize(ctx->stream));
    return PC_OK;
}

}  // extern "C"
