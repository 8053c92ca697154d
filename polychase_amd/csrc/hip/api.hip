// api.hip -- implementation of the C ABI declared in include/polychase_hip.h: context, frames, the stage-level
// detect / LK calls, and the helpers the analyzer shares (api_internal.hpp).  The pipelined analyzer lives in
// api_analyzer.hip, the tracking / refinement entry points in api_tracker.hip.
//
// Host-side orchestration only: HBM plane layout, stream ordering, read-backs, and the two pieces of
// GoodFeaturesToTrack that are sequential by definition (greedy min-distance suppression,
// reference cpp/feature_detection/gftt.cc:100-164).  No pixel arithmetic happens on the CPU.
#include "api_internal.hpp"

#include <dirent.h>
#include <unistd.h>

#include <cstring>
#include <mutex>

namespace pc {
std::string& last_error() {
    thread_local std::string e;
    return e;
}
}  // namespace pc

namespace pc_api {

int collect_timing(pc_context* c) {
    if (c->ranges.empty()) return PC_OK;
    PC_HIP(c->sync_side_streams());
    PC_HIP(hipStreamSynchronize(c->stream));
    // launches of one class may overlap (the analyzer's two job lanes): besides the sum of the durations keep the
    // length of the union of the intervals, placed on one time axis relative to the first range's start event
    std::vector<std::pair<float, float>> spans[PC_K_COUNT];
    hipEvent_t const origin = c->ranges.front().a;
    for (auto& r : c->ranges) {
        float ms = 0.f, t0 = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            c->launches[r.cls] += 1;
            c->total_ms[r.cls] += ms;
            if (hipEventElapsedTime(&t0, origin, r.a) == hipSuccess) spans[r.cls].emplace_back(t0, t0 + ms);
        }
        c->event_pool.push_back(r.a);
        c->event_pool.push_back(r.b);
    }
    c->ranges.clear();
    for (int k = 0; k < PC_K_COUNT; k++) {
        auto& v = spans[k];
        std::sort(v.begin(), v.end());
        float lo = 0.f, hi = -1.f;
        for (auto& sp : v) {
            if (hi < lo || sp.first > hi) {
                if (hi >= lo) c->busy_ms[k] += hi - lo;
                lo = sp.first;
                hi = sp.second;
            } else if (sp.second > hi) {
                hi = sp.second;
            }
        }
        if (hi >= lo) c->busy_ms[k] += hi - lo;
    }
    return PC_OK;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

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

// POLYCHASE_PYRAMID_VARIANT=1: the unfused kernels of kernels_image.hip (gray, pyrDown, border, widen, Scharr: the
// cross-check of the fused level kernel)
static bool unfused_pyramid() {
    static const bool v = [] {
        const char* e = getenv("POLYCHASE_PYRAMID_VARIANT");
        return e && atoi(e) == 1;
    }();
    return v;
}
static bool level_fits_fused(const pc::Level& L, int win) { return L.w > win + 1 && L.h > win + 1; }

// gray already in level-0 interior -> borders, pyrDown chain, Scharr planes (levels [first, nlevels))
void build_pyramid(pc_context* c, pc_frame* f, int first) {
    ScopedTimer t(c, PC_K_PYRAMID);
    for (int l = first; l < f->nlevels; l++) {
        if (l > 0 && !unfused_pyramid() && level_fits_fused(f->levels[l], f->win)) {
            pc::LevelSource in{};
            in.kind = pc::SRC_PYR;
            in.parent = f->levels[l - 1];
            pc::launch_level(in, f->levels[l], f->win, c->work);
            continue;
        }
        if (l > 0) pc::launch_pyrdown(f->levels[l - 1], f->levels[l], c->work);
        pc::launch_border(f->levels[l], f->win, c->work);
        pc::launch_widen(f->levels[l], f->win, c->work);
        pc::launch_scharr(f->levels[l], c->work);
    }
}

// DetectScratch::counters layout
static constexpr int kCounterCells = 16;           // header words; the first kHostCells reach the host
static constexpr int kHostCells = 8;
static constexpr int kCntCand = 0, kCntKps = 1, kCntStuck = 2, kCntOverflow = 3, kCntSortParams = 4;
static constexpr int kCellMaxAt = kCounterCells;
static constexpr int kHistAt = kCellMaxAt + pc::kMaxGridCells;
static constexpr int kCursorAt = kHistAt + pc::kSortBuckets;
static constexpr int kTicketsAt = kCursorAt + pc::kSortBuckets;   // three ticket arrays (pc::last_workgroup): NMS, suppression, compaction
// workgroups of the largest launch that takes tickets: the suppression over the candidate buffer's capacity (which a
// larger frame seen earlier may have left bigger than this frame needs) or, on the slow path, over every pixel
static uint32_t ticket_stride(int w, int h, uint32_t cand_cap) {
    const uint32_t npx = (uint32_t)((size_t)w * h);
    return pc::last_workgroup_words((uint32_t)pc::suppress_num_blocks(std::max(npx, cand_cap)) + 1u) + 3u;
}
static constexpr uint32_t kOverflowSort = 1u, kOverflowKeypoints = 2u;   // bits of counters[kCntOverflow]

int validate_gftt(const pc_gftt_options* opt, int w, int h, pc::GfttGrid* g) {
    // CHECKs of gftt.cc:18-19
    if (!(opt->quality_level > 0 && opt->min_distance >= 0 && opt->max_corners >= 0))
        return fail(PC_E_INVALID, "GFTT options violate quality_level > 0 && min_distance >= 0 && max_corners >= 0");
    // cornerEigenValsVecs: Sobel apertures 3 / 5 / 7 or Scharr (-1) (gftt.cc:31-36)
    if (opt->gradient_size != 3 && opt->gradient_size != 5 && opt->gradient_size != 7 && opt->gradient_size != -1)
        return fail(PC_E_INVALID, "gradient_size must be 3, 5, 7 (Sobel) or -1 (Scharr)");
    // cornerMinEigenVal / cornerHarris take any block size (boxFilter); 2 * block_size reads per pixel from kBoxRowsFromBlock on
    if (opt->block_size < 1) return fail(PC_E_INVALID, "block_size must be >= 1");
    // (any min_distance: above kSuppressMaxTableRadius the greedy loop runs against a grid of accepted corners, kernels_gftt.hip)
    g->rows = std::max(1, opt->grid_rows);
    g->cols = std::max(1, opt->grid_cols);
    if (g->rows * g->cols > pc::kMaxGridCells) return fail(PC_E_INVALID, "grid_rows*grid_cols must be <= %d", pc::kMaxGridCells);
    g->cell_h = (h + g->rows - 1) / g->rows;
    g->cell_w = (w + g->cols - 1) / g->cols;
    return PC_OK;
}

// cornerMinEigenVal / cornerHarris of the frame (gftt.cc:31-36) + per-cell maxima: the tiled kernel for the detector's
// default, the general pair of kernels otherwise (their covariance scratch is allocated on first use: not the addon's path)
static int corner_response(pc_context* ctx, const pc_frame* f, DetectScratch& d, const pc::GfttGrid& grid, const pc_gftt_options& opt,
                           uint32_t* cell_max) {
    const int fma = ((ctx->arith & PC_ARITH_SOBEL_FMA) ? 1 : 0) | ((ctx->arith & PC_ARITH_SOBEL_ROW_FMA) ? 2 : 0);
    // POLYCHASE_GFTT_GENERAL=1: the general kernels for the default options too (cross-check of the tiled kernel)
    static const bool force_general = getenv("POLYCHASE_GFTT_GENERAL") && atoi(getenv("POLYCHASE_GFTT_GENERAL")) == 1;
    if (opt.block_size == 3 && opt.gradient_size == 3 && !opt.use_harris && !force_general) {
        pc::launch_min_eig(f->levels[0], d.eig.p, grid, cell_max, fma, ctx->work);
        return PC_OK;
    }
    PC_HIP(d.cov.ensure((size_t)3 * f->w * f->h));
    const bool two_pass = opt.block_size >= pc::kBoxRowsFromBlock;
    if (two_pass) PC_HIP(d.box_rows.ensure((size_t)3 * f->w * f->h));
    if (!pc::launch_corner_response(f->levels[0], d.eig.p, d.cov.p, two_pass ? d.box_rows.p : nullptr, grid, cell_max, opt.block_size, opt.gradient_size, opt.use_harris != 0, opt.harris_k,
                                    fma, ctx->work))
        return fail(PC_E_INVALID, "gradient_size must be 3, 5, 7 (Sobel) or -1 (Scharr)");
    return PC_OK;
}

// Phase A of GoodFeaturesToTrack (see DetectScratch), asynchronous: min-eig map + per-cell max (gftt.cc:35,:61-63),
// threshold + NMS -> candidates (gftt.cc:64-86).  The candidate count is copied to pinned memory; `ev` fires after.
// every buffer a detection of a w x h frame needs (the analyzer calls this per slot at creation: an allocation inside the
// running pipeline synchronises the device)
int detect_reserve(pc_context* ctx, int w, int h, DetectScratch& d) {
    const size_t npx = (size_t)w * h;
    // 3x3 local maxima: at most one per 2x2 block unless the response has plateaus; more than that takes the slow path
    const size_t cand_cap = std::min(npx, npx / 4 + 65536);
    PC_HIP(d.eig.ensure(npx));
    PC_HIP(d.cstate.ensure(npx + 16));
    PC_HIP(d.keys.ensure(cand_cap));
    PC_HIP(d.keys_bucketed.ensure(cand_cap));
    PC_HIP(d.keys_sorted.ensure(cand_cap));
    d.cand_cap = (uint32_t)std::min<size_t>(d.keys.cap, std::min(d.keys_bucketed.cap, d.keys_sorted.cap));
    d.ticket_stride = ticket_stride(w, h, d.cand_cap);
    d.counter_words = kTicketsAt + 3 * (int)d.ticket_stride;
    PC_HIP(d.counters.ensure((size_t)d.counter_words));
    PC_HIP(d.bucket_offsets.ensure(pc::kSortBuckets + 1));
    PC_HIP(d.per_block.ensure((size_t)pc::suppress_num_blocks(d.cand_cap) + 1));
    PC_HIP(d.h_counters.ensure(kHostCells));
    if (!d.ev_b) PC_HIP(hipEventCreateWithFlags(&d.ev_b, hipEventDisableTiming));
    (void)ctx;
    return PC_OK;
}

static int upload_suppression_offsets(pc_context* ctx, const pc_gftt_options& opt) {
    if (!(opt.min_distance >= 1) || ctx->sup_min_distance == opt.min_distance) return PC_OK;
    if (opt.min_distance > pc::kSuppressMaxTableRadius) return PC_OK;   // no table: the grid kernel
    const std::vector<int2> offs = suppression_offsets(opt.min_distance);
    PC_HIP(ctx->sup_offsets.ensure(offs.size() + 1));
    PC_HIP(hipStreamSynchronize(ctx->work));  // a queued suppression may still read the old table
    PC_HIP(hipMemcpy(ctx->sup_offsets.p, offs.data(), offs.size() * sizeof(int2), hipMemcpyHostToDevice));
    ctx->n_sup_offsets = (int)offs.size();
    // the same neighbourhood as half-widths per row (the set is symmetric and contiguous in dx for every dy)
    int R = 0;
    for (const int2& o : offs) R = std::max(R, std::max(std::abs(o.x), std::abs(o.y)));
    std::vector<int> hw((size_t)2 * R + 1, -1);
    for (const int2& o : offs) hw[(size_t)(o.y + R)] = std::max(hw[(size_t)(o.y + R)], std::abs(o.x));
    hw[(size_t)R] = std::max(hw[(size_t)R], 0);   // the centre row always holds the candidate itself
    size_t covered = 0;
    for (int v : hw) covered += v >= 0 ? (size_t)(2 * v + 1) : 0;
    if (covered != offs.size() + 1) return fail(PC_E_INVALID, "suppression neighbourhood is not row-contiguous");
    PC_HIP(ctx->sup_rows.ensure(hw.size()));
    PC_HIP(hipMemcpy(ctx->sup_rows.p, hw.data(), hw.size() * sizeof(int), hipMemcpyHostToDevice));
    ctx->sup_R = R;
    ctx->sup_min_distance = opt.min_distance;
    return PC_OK;
}

static int ensure_perm_capacity(pc_frame* f, int n) {
    if (f->perm_cap >= n) return PC_OK;
    if (f->d_perm) PC_HIP(hipFree(f->d_perm));
    f->d_perm = nullptr;
    f->perm_cap = 0;
    const int want = std::max(n + n / 2, 1024);
    PC_HIP(hipMalloc(&f->d_perm, (size_t)want * 2 * sizeof(uint32_t)));   // order + inverse
    f->perm_cap = want;
    return PC_OK;
}

// GoodFeaturesToTrack, enqueued in one go on the current work stream (see DetectScratch): min-eig map + per-cell max
// (gftt.cc:35,:61-63), threshold + NMS -> candidates (gftt.cc:64-86), their sort (gftt.cc:98), the suppression in that
// order (gftt.cc:100-164), the accepted corners -> keypoints truncated to max_corners (gftt.cc:157-162), the LK visiting
// order.  Every count stays on the device; the counters are copied to pinned memory and `ev_b` fires after.
int detect_enqueue(pc_context* ctx, pc_frame* f, const pc::GfttGrid& grid, const pc_gftt_options& opt, DetectScratch& d,
                   DevBuf<uint32_t>& hist, bool full_launch) {
    const int w = f->w, h = f->h;
    int rc = detect_reserve(ctx, w, h, d);
    if (rc != PC_OK) return rc;
    if ((rc = upload_suppression_offsets(ctx, opt)) != PC_OK) return rc;
    if (f->kp_cap < 4096 && (rc = ensure_kp_capacity(f, 4096)) != PC_OK) return rc;
    if ((rc = ensure_perm_capacity(f, f->kp_cap)) != PC_OK) return rc;
    PC_HIP(hist.ensure((size_t)pc::bin_num_tiles(w, h) + 1));
    uint32_t* const cnt = d.counters.p;
    // the analyzer has the level-0 kernel of the frame zero the counters (detect_clear_list): one command less in the chain
    if (!d.cleared) PC_HIP(hipMemsetAsync(cnt, 0, (size_t)d.counter_words * sizeof(uint32_t), ctx->work));
    uint32_t* const tickets = cnt + kTicketsAt;
    d.cleared = false;
    {
        ScopedTimer t(ctx, PC_K_MINEIG);
        if ((rc = corner_response(ctx, f, d, grid, opt, cnt + kCellMaxAt)) != PC_OK) return rc;
    }
    {
        ScopedTimer t(ctx, PC_K_NMS);
        pc::launch_nms(d.eig.p, w, h, grid, cnt + kCellMaxAt, opt.quality_level, d.keys.p, d.cand_cap, cnt + kCntCand, d.cstate.p,
                       cnt + kCntSortParams, cnt + kHistAt, tickets, d.bucket_offsets.p, hist.p, ctx->work);
    }
    // launches sized for the expected number of candidates (workgroups that find nothing to do still queue for a slot
    // beside the LK wavefronts), the buffers for the capacity
    const uint32_t n_launch = (ctx->cand_hint > 0 && !full_launch) ? std::min(ctx->cand_hint, d.cand_cap) : d.cand_cap;
    {
        ScopedTimer t(ctx, PC_K_SORT);
        pc::launch_bucket_sort(d.keys.p, d.cand_cap, n_launch, cnt + kCntCand, cnt + kCntSortParams, d.bucket_offsets.p,
                               cnt + kCursorAt, d.keys_bucketed.p, d.keys_sorted.p, cnt + kCntOverflow, ctx->work);
    }
    // keypoints beyond the frame's buffer are not written; the count then exceeds the capacity and the slow path redoes it
    const uint32_t limit = opt.max_corners > 0 ? std::min<uint32_t>((uint32_t)opt.max_corners, (uint32_t)f->kp_cap) : (uint32_t)f->kp_cap;
    const bool large_radius = opt.min_distance > pc::kSuppressMaxTableRadius;
    if (large_radius) PC_HIP(d.sup_grid.ensure((size_t)pc::suppress_large_grid_words(w, h, opt.min_distance)));
    {
        ScopedTimer t(ctx, PC_K_SUPPRESS);
        pc::launch_suppress_and_compact(d.keys_sorted.p, n_launch, cnt + kCntCand, w, h, d.eig.p, d.cstate.p, ctx->sup_offsets.p,
                                        ctx->n_sup_offsets, ctx->sup_rows.p, ctx->sup_R, opt.min_distance >= 1, d.per_block.p,
                                        cnt + kCntStuck, limit, f->d_kps,
                                        cnt + kCntKps, hist.p, cnt + kCntOverflow, tickets + d.ticket_stride, d.ticket_stride,
                                        opt.min_distance, large_radius ? d.sup_grid.p : nullptr, ctx->work);
    }
    // the visiting order; the same launch stores the counters in pinned host memory (no copy command behind it)
    pc::launch_spatial_bins_counted(f->d_kps, (int)std::min<uint32_t>(limit, n_launch), cnt + kCntKps, w, h, hist.p, f->d_perm,
                                    f->d_perm + f->perm_cap, cnt, d.h_counters.p, kHostCells, ctx->work);
    PC_HIP(hipEventRecord(d.ev_b, ctx->work));
    f->n_kps = -1;
    f->n_cands = -1;
    f->perm_valid = false;
    return PC_OK;
}

// The slow path, synchronous, for frames the fast path cannot hold (more candidates than one per 2x2 block -- plateaus of
// the response --, a value bucket beyond its LDS buffer, more keypoints than the frame's buffer): candidate and
// keypoint counts on the host, rocPRIM sort, buffers grown to what the frame needs.
static int detect_slow_path(pc_context* ctx, pc_frame* f, const pc::GfttGrid& grid, const pc_gftt_options& opt, DetectScratch& d,
                            DevBuf<uint32_t>& hist) {
    const int w = f->w, h = f->h;
    const uint32_t npx = (uint32_t)((size_t)w * h);
    PC_HIP(hipStreamSynchronize(ctx->work));
    PC_HIP(d.keys.ensure(npx));
    PC_HIP(d.keys_sorted.ensure(npx));
    PC_HIP(d.per_block.ensure((size_t)pc::suppress_num_blocks(npx) + 1));
    PC_HIP(hist.ensure((size_t)pc::bin_num_tiles(w, h) + 1));
    uint32_t* const cnt = d.counters.p;
    d.cleared = false;
    PC_HIP(hipMemsetAsync(cnt, 0, (size_t)d.counter_words * sizeof(uint32_t), ctx->work));
    uint32_t* const tickets = cnt + kTicketsAt;
    if (int crc = corner_response(ctx, f, d, grid, opt, cnt + kCellMaxAt)) return crc;
    pc::launch_nms(d.eig.p, w, h, grid, cnt + kCellMaxAt, opt.quality_level, d.keys.p, npx, cnt + kCntCand, d.cstate.p,
                   cnt + kCntSortParams, cnt + kHistAt, tickets, d.bucket_offsets.p, hist.p, ctx->work);
    PC_HIP(hipMemcpyAsync(d.h_counters.p, cnt, kHostCells * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->work));
    PC_HIP(hipStreamSynchronize(ctx->work));
    const uint32_t n_cand = std::min(d.h_counters.p[kCntCand], npx);
    f->n_cands = (int)n_cand;
    f->n_kps = 0;
    if (n_cand == 0) return PC_OK;
    size_t temp_bytes = 0;
    PC_HIP(pc::sort_keys_desc(nullptr, temp_bytes, d.keys.p, d.keys_sorted.p, n_cand, ctx->work));
    PC_HIP(ctx->sort_temp.ensure(temp_bytes));
    PC_HIP(pc::sort_keys_desc(ctx->sort_temp.p, temp_bytes, d.keys.p, d.keys_sorted.p, n_cand, ctx->work));
    const int cap = opt.max_corners > 0 ? (int)std::min<uint32_t>(n_cand, (uint32_t)opt.max_corners) : (int)n_cand;
    int rc = ensure_kp_capacity(f, cap);
    if (rc != PC_OK) return rc;
    if ((rc = ensure_perm_capacity(f, f->kp_cap)) != PC_OK) return rc;
    const bool large_radius = opt.min_distance > pc::kSuppressMaxTableRadius;
    if (large_radius) PC_HIP(d.sup_grid.ensure((size_t)pc::suppress_large_grid_words(w, h, opt.min_distance)));
    pc::launch_suppress_and_compact(d.keys_sorted.p, n_cand, nullptr, w, h, d.eig.p, d.cstate.p, ctx->sup_offsets.p, ctx->n_sup_offsets,
                                    ctx->sup_rows.p, ctx->sup_R, opt.min_distance >= 1, d.per_block.p, cnt + kCntStuck, (uint32_t)std::max(opt.max_corners, 0), f->d_kps,
                                    cnt + kCntKps, hist.p, nullptr, tickets + d.ticket_stride, d.ticket_stride, opt.min_distance,
                                    large_radius ? d.sup_grid.p : nullptr, ctx->work);
    pc::launch_spatial_bins_counted(f->d_kps, cap, cnt + kCntKps, w, h, hist.p, f->d_perm, f->d_perm + f->perm_cap, nullptr, nullptr, 0,
                                    ctx->work);
    PC_HIP(hipMemcpyAsync(d.h_counters.p, cnt, kHostCells * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->work));
    PC_HIP(hipStreamSynchronize(ctx->work));
    if (d.h_counters.p[kCntStuck] != 0) return fail(PC_E_HIP, "suppression kernel did not converge (%u lanes gave up)", d.h_counters.p[kCntStuck]);
    f->n_kps = (int)std::min<uint32_t>(d.h_counters.p[kCntKps], (uint32_t)cap);
    f->perm_valid = f->n_kps > 0;
    return PC_OK;
}

// The keypoint count reaches the host (waits for ev_b); frames beyond the fast path's bounds are redone on the slow path
int detect_finish(pc_context* ctx, pc_frame* f, const pc::GfttGrid& grid, const pc_gftt_options& opt, DetectScratch& d,
                  DevBuf<uint32_t>& hist, bool* redone) {
    if (redone) *redone = false;
    PC_HIP(hipEventSynchronize(d.ev_b));
    const uint32_t* hc = d.h_counters.p;
    const bool too_many_candidates = hc[kCntCand] > d.cand_cap;
    const bool too_many_keypoints = hc[kCntKps] >= (uint32_t)f->kp_cap && !(opt.max_corners > 0 && opt.max_corners <= f->kp_cap);
    static const bool force_slow = getenv("POLYCHASE_GFTT_SLOW_PATH") != nullptr;   // tests: the slow path must give the same keypoints
    ctx->cand_hint = std::max<uint32_t>(4096u, hc[kCntCand] + hc[kCntCand] / 4);
    if (too_many_candidates || too_many_keypoints || hc[kCntOverflow] != 0 || force_slow) {
        if (redone) *redone = true;
        return detect_slow_path(ctx, f, grid, opt, d, hist);
    }
    if (hc[kCntStuck] != 0) return fail(PC_E_HIP, "suppression kernel did not converge (%u lanes gave up)", hc[kCntStuck]);
    f->n_cands = (int)hc[kCntCand];
    f->n_kps = (int)hc[kCntKps];
    f->perm_valid = f->n_kps > 0;
    return PC_OK;
}

// Stage-level entry points share scratch buffers with work the analyzer queued on prep_stream:
// order `stream` after it.
int join_prep(pc_context* ctx) {
    if (!ctx->prep_dirty) return PC_OK;
    hipStream_t const side[4] = {ctx->prep_stream, ctx->stream_b, ctx->detect_stream[0], ctx->detect_stream[1]};
    for (hipStream_t s : side) {
        if (!s) continue;
        PC_HIP(hipEventRecord(ctx->prep_fence, s));
        PC_HIP(hipStreamWaitEvent(ctx->stream, ctx->prep_fence, 0));
    }
    ctx->prep_dirty = false;
    return PC_OK;
}


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
        PC_HIP(hipMalloc(&f->d_perm, (size_t)cap * 2 * sizeof(uint32_t)));   // order + inverse
        f->perm_cap = cap;
    }
    PC_HIP(hist.ensure((size_t)pc::bin_num_tiles(f->w, f->h) + 1));
    pc::launch_spatial_bins(f->d_kps, n, nullptr, f->w, f->h, hist.p, f->d_perm, f->d_perm + f->perm_cap, ctx->work);
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
           const pc_flow_options* opt, int set) {
    hipStream_t const lk_stream = ctx->lane_stream(set);
    const int n = frame1->n_kps;
    const size_t rows = (size_t)n * n_targets;
    (void)rows;
    PC_HIP(ctx->lk_rec[set].ensure(((size_t)n + 1) * pc::kRecStride));
    ctx->lk_slot_of[set] = nullptr;
    if (n == 0) return PC_OK;
    pc::LKParams p;
    std::memset(&p, 0, sizeof(p));
    int max_level = std::min(opt->max_level, frame1->nlevels - 1);
    for (int t = 0; t < n_targets; t++) max_level = std::min(max_level, targets[t]->nlevels - 1);
    if (max_level < 0) max_level = 0;
    for (int l = 0; l <= max_level; l++) {
        p.src[l] = frame1->levels[l];
        for (int t = 0; t < n_targets; t++) {
            p.tgt[t][l] = targets[t]->levels[l].img;
            p.tgt16[t][l] = targets[t]->levels[l].img16;
        }
    }
    p.n_targets = n_targets;
    p.max_level = max_level;
    p.n = n;
    p.pts = frame1->d_kps;
    // visiting order: keypoints binned by image tile (they are stored by corner response); the analyzer
    // prepares it together with the keypoints, the stage-level calls compute it here
    if (frame1->perm_valid) {
        p.perm = frame1->d_perm;
        ctx->lk_slot_of[set] = frame1->d_perm + frame1->perm_cap;
    } else {
        PC_HIP(ctx->lk_perm.ensure((size_t)n * 2));
        PC_HIP(ctx->lk_hist.ensure((size_t)pc::bin_num_tiles(frame1->w, frame1->h) + 1));
        pc::launch_spatial_bins(frame1->d_kps, n, nullptr, frame1->w, frame1->h, ctx->lk_hist.p, ctx->lk_perm.p, ctx->lk_perm.p + n, lk_stream);
        p.perm = ctx->lk_perm.p;
        ctx->lk_slot_of[set] = ctx->lk_perm.p + n;
    }
    // TermCriteria clamps of calcOpticalFlowPyrLK
    p.max_iters = std::min(std::max(opt->term_max_iters, 0), 100);
    const double eps = std::min(std::max(opt->term_epsilon, 0.), 10.);
    p.eps_sq = eps * eps;
    p.min_eig_thr = (float)opt->min_eigen_threshold;
    p.out_rec = ctx->lk_rec[set].p;
    p.prof = nullptr;
    p.x86_order = (ctx->arith & PC_ARITH_LK_X86_ORDER) ? 1 : 0;
    p.x86_stats = ctx->lk_x86_stats.p;
    p.gate = ctx->lk_gate_next ? ctx->lk_gate.p : nullptr;
    p.gate_value = ctx->lk_gate_next;
    ctx->lk_gate_next = 0;
    if (pc::lk_profile_enabled()) {   // diagnostics build: 16 words per wavefront, the latest launch only
        ctx->lk_prof_rows = (size_t)n / 2 + 1;
        PC_HIP(ctx->lk_prof.ensure(ctx->lk_prof_rows * PC_LK_PROFILE_SLOTS));
        PC_HIP(hipMemsetAsync(ctx->lk_prof.p, 0, ctx->lk_prof_rows * PC_LK_PROFILE_SLOTS * sizeof(unsigned long long), lk_stream));
        p.prof = ctx->lk_prof.p;
    }
    ScopedTimer tm(ctx, PC_K_LK, lk_stream);
    if (!pc::launch_lk(p, frame1->win, lk_stream)) return fail(PC_E_INVALID, "unsupported window size %d", frame1->win);
    return PC_OK;
}

}  // namespace pc_api

using namespace pc_api;

// The engine keeps five streams busy (two job lanes, preparation, frame upload, and whatever the host uses); the HIP runtime
// multiplexes streams onto GPU_MAX_HW_QUEUES = 4 hardware queues by default, and two streams on one queue wait for each
// other's commands: with host frames the upload stream then shares a queue with a lane or the preparation stream
// (generate_optical_flow_database on 300 host frames at 1080p: 2050 frames/s on 4 queues, 2650-2840 with a queue per stream;
// idle queues cost nothing; sixteen cover two contexts -- analysis engine and tracker -- plus the host's own streams).  The
// runtime reads the variable when the process first touches HIP.  Round 3 raised the default in a library CONSTRUCTOR -- a
// side effect of dlopen on every other HIP user of the process; since round 4 it is the explicit pc_runtime_init() below,
// which pc_context_create (and the polychase_core module / polychase_amd.hip when they load the library) call: still
// before this library's first HIP call, never behind the host's back, and it reports whether it came too late.
static bool hsa_runtime_is_up() {
    // the ROCm runtime holds /dev/kfd open from its initialisation on; nothing in this library has touched HIP when this runs
    DIR* d = opendir("/proc/self/fd");
    if (!d) return false;
    bool up = false;
    char link[64], target[256];
    while (struct dirent* e = readdir(d)) {
        if (e->d_name[0] == '.') continue;
        snprintf(link, sizeof(link), "/proc/self/fd/%s", e->d_name);
        const ssize_t n = readlink(link, target, sizeof(target) - 1);
        if (n > 0) {
            target[n] = 0;
            if (strcmp(target, "/dev/kfd") == 0) {
                up = true;
                break;
            }
        }
    }
    closedir(d);
    return up;
}

static std::once_flag g_runtime_once;
static int g_runtime_was_up = 0, g_runtime_queues = 0;

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

int pc_runtime_init(int* runtime_was_up, int* hw_queues) {
    std::call_once(g_runtime_once, [] {
        g_runtime_was_up = hsa_runtime_is_up() ? 1 : 0;
        setenv("GPU_MAX_HW_QUEUES", "16", /*overwrite=*/0);   // a value set by the user wins
        const char* v = getenv("GPU_MAX_HW_QUEUES");
        g_runtime_queues = v ? atoi(v) : 0;
    });
    if (runtime_was_up) *runtime_was_up = g_runtime_was_up;
    if (hw_queues) *hw_queues = g_runtime_queues;
    return PC_OK;
}

int pc_context_create(int device_index, pc_context** out) {
    if (!out) return fail(PC_E_INVALID, "null out");
    *out = nullptr;
    pc_runtime_init(nullptr, nullptr);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(PC_E_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device_index < 0 || device_index >= count) return fail(PC_E_INVALID, "device index %d out of range [0,%d)", device_index, count);
    PC_HIP(hipSetDevice(device_index));
    pc_context* c = new (std::nothrow) pc_context();
    if (!c) return fail(PC_E_INVALID, "out of host memory");
    c->device = device_index;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    {
        const char* g = getenv("POLYCHASE_LK_GATE");
        c->lk_gate_on = !(g && atoi(g) == 0);
        // Under rocprofv3 the hand-over of dispatches is serialised: the gate's polling wavefront then sits in front of the
        // launch it waits for until its 50-ms bail-out (seen in round 2).  A profiled process runs without the gate unless it
        // is asked for explicitly (POLYCHASE_LK_GATE=1).
        if (!g) {
            extern char** environ;
            for (char** e2 = environ; e2 && *e2; ++e2)
                if (strncmp(*e2, "ROCPROFILER_", 12) == 0 || strncmp(*e2, "ROCPROF_", 8) == 0 || strncmp(*e2, "ROCP_TOOL_LIB", 13) == 0) {
                    c->lk_gate_on = false;
                    break;
                }
        }
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream_b, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->prep_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->prep_fence, hipEventDisableTiming);
    {
        const char* v = getenv("POLYCHASE_DETECT_STREAMS");
        c->n_detect = v ? std::max(0, std::min(2, atoi(v))) : 0;
        for (int k = 0; k < c->n_detect && e == hipSuccess; k++)
            e = hipStreamCreateWithFlags(&c->detect_stream[k], hipStreamNonBlocking);
    }
    c->work = c->stream;
    {
        const char* v = getenv("POLYCHASE_COPY_STREAM");
        if (e == hipSuccess && !(v && atoi(v) == 0)) e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
    }
    if (const char* m = getenv("POLYCHASE_ARITH")) {
        const std::string mode(m);
        if (mode == "opencv_x86") c->arith = PC_ARITH_OPENCV_X86;
        else if (mode == "lk_x86") c->arith = PC_ARITH_LK_X86_ORDER;
        else if (mode == "sobel_fma") c->arith = PC_ARITH_SOBEL_FMA;
        else if (mode == "canonical") c->arith = PC_ARITH_CANONICAL;
        else if (mode == "sobel_fma_rows") c->arith = PC_ARITH_SOBEL_FMA | PC_ARITH_SOBEL_ROW_FMA;
        else if (mode == "opencv_x86_rows") c->arith = PC_ARITH_OPENCV_X86 | PC_ARITH_SOBEL_ROW_FMA;
        else if (!mode.empty()) {
            delete c;
            return fail(PC_E_INVALID, "POLYCHASE_ARITH=%s: expected canonical, opencv_x86, lk_x86, sobel_fma, sobel_fma_rows or opencv_x86_rows", m);
        }
    }
    if (e != hipSuccess) {
        delete c;
        return fail(PC_E_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    *out = c;
    return PC_OK;
}

int pc_context_download(pc_context* c, void* dst_host, const void* src_device, size_t bytes) {
    if (!c || (bytes && (!dst_host || !src_device))) return fail(PC_E_INVALID, "null argument");
    PC_HIP(hipSetDevice(c->device));
    if (bytes) PC_HIP(hipMemcpy(dst_host, src_device, bytes, hipMemcpyDeviceToHost));
    return PC_OK;
}

// ---- peer buffers (include/polychase_hip.h: record exchange between ranks) ----
int pc_peer_buffer_alloc(int device_index, size_t bytes, void** device_ptr) {
    if (!device_ptr || bytes == 0) return fail(PC_E_INVALID, "bad argument");
    *device_ptr = nullptr;
    PC_HIP(hipSetDevice(device_index));
    PC_HIP(hipMalloc(device_ptr, bytes));
    return PC_OK;
}
int pc_peer_buffer_free(int device_index, void* device_ptr) {
    if (!device_ptr) return PC_OK;
    PC_HIP(hipSetDevice(device_index));
    PC_HIP(hipFree(device_ptr));
    return PC_OK;
}
int pc_peer_buffer_export(int device_index, void* device_ptr, unsigned char handle[PC_PEER_HANDLE_BYTES]) {
    static_assert(sizeof(hipIpcMemHandle_t) <= PC_PEER_HANDLE_BYTES, "IPC handle larger than PC_PEER_HANDLE_BYTES");
    if (!device_ptr || !handle) return fail(PC_E_INVALID, "null argument");
    PC_HIP(hipSetDevice(device_index));
    hipIpcMemHandle_t h;
    PC_HIP(hipIpcGetMemHandle(&h, device_ptr));
    memset(handle, 0, PC_PEER_HANDLE_BYTES);
    memcpy(handle, &h, sizeof(h));
    return PC_OK;
}
int pc_peer_buffer_open(int device_index, const unsigned char handle[PC_PEER_HANDLE_BYTES], void** device_ptr) {
    if (!device_ptr || !handle) return fail(PC_E_INVALID, "null argument");
    *device_ptr = nullptr;
    PC_HIP(hipSetDevice(device_index));
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    PC_HIP(hipIpcOpenMemHandle(device_ptr, h, hipIpcMemLazyEnablePeerAccess));
    return PC_OK;
}
int pc_peer_buffer_close(int device_index, void* device_ptr) {
    if (!device_ptr) return PC_OK;
    PC_HIP(hipSetDevice(device_index));
    PC_HIP(hipIpcCloseMemHandle(device_ptr));
    return PC_OK;
}
int pc_peer_copy_async(int device_index, void* dst, const void* src, size_t bytes, void* stream) {
    if (bytes && (!dst || !src)) return fail(PC_E_INVALID, "null argument");
    PC_HIP(hipSetDevice(device_index));
    if (bytes) PC_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    return PC_OK;
}
int pc_peer_buffer_download(int device_index, void* dst_host, const void* src_device, size_t bytes) {
    if (bytes && (!dst_host || !src_device)) return fail(PC_E_INVALID, "null argument");
    PC_HIP(hipSetDevice(device_index));
    if (bytes) PC_HIP(hipMemcpy(dst_host, src_device, bytes, hipMemcpyDeviceToHost));
    return PC_OK;
}

int pc_context_set_arithmetic(pc_context* c, int flags) {
    if (!c) return fail(PC_E_INVALID, "null context");
    if (flags & ~(PC_ARITH_OPENCV_X86 | PC_ARITH_SOBEL_ROW_FMA)) return fail(PC_E_INVALID, "unknown arithmetic flags %d", flags);
    c->arith = flags;
    c->eig_owner = nullptr;   // a min-eig map computed under the other mode is not this mode's
    return PC_OK;
}
int pc_context_get_arithmetic(const pc_context* c) { return c ? c->arith : -1; }

void pc_context_destroy(pc_context* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->prep_stream) (void)c->sync_side_streams();
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& r : c->ranges) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    c->staging.release();
    for (auto& b : c->staging2) b.release();
    for (auto& pair : c->staging_ev)
        for (hipEvent_t ev : pair)
            if (ev) (void)hipEventDestroy(ev);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    c->sup_offsets.release();
    c->sup_rows.release();
    if (c->detect) {
        c->detect->release();
        delete c->detect;
        c->detect = nullptr;
    }
    c->sort_temp.release();
    c->lk_rec[0].release();
    c->lk_rec[1].release();
    c->lk_cxy.release();
    c->lk_ustatus.release();
    c->lk_cerr.release();
    c->lk_cidx.release();
    for (auto& b : c->lk_block_counts) b.release();
    c->lk_perm.release();
    c->lk_prof.release();
    c->lk_x86_stats.release();
    c->lk_gate.release();
    c->lk_gate_timed_out.release();
    c->lk_hist.release();
    c->lk_row_offset.release();
    c->h_row_offset.release();
    for (auto& b : c->lk_pack) b.release();
    if (c->stream_b) {
        (void)hipStreamSynchronize(c->stream_b);
        (void)hipStreamDestroy(c->stream_b);
    }
    if (c->prep_stream) (void)hipStreamDestroy(c->prep_stream);
    for (hipStream_t s : c->detect_stream)
        if (s) (void)hipStreamDestroy(s);
    if (c->prep_fence) (void)hipEventDestroy(c->prep_fence);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int pc_context_synchronize(pc_context* c) {
    if (!c) return fail(PC_E_INVALID, "null context");
    PC_HIP(c->sync_side_streams());
    PC_HIP(hipStreamSynchronize(c->stream));
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

int pc_context_get_busy_time(pc_context* c, int k, double* busy_ms) {
    if (!c || !busy_ms || k < 0 || k >= PC_K_COUNT) return fail(PC_E_INVALID, "bad kernel class");
    int rc = collect_timing(c);
    if (rc != PC_OK) return rc;
    *busy_ms = c->busy_ms[k];
    return PC_OK;
}

int pc_debug_lk_profile(pc_context* c, unsigned long long* out) {
    if (!c || !out) return fail(PC_E_INVALID, "null argument");
    for (int k = 0; k < PC_LK_PROFILE_SLOTS; k++) out[k] = 0;
    if (!c->lk_prof.p || c->lk_prof_rows == 0) return PC_OK;
    PC_HIP(hipStreamSynchronize(c->stream));
    std::vector<unsigned long long> h(c->lk_prof_rows * PC_LK_PROFILE_SLOTS);
    PC_HIP(hipMemcpy(h.data(), c->lk_prof.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (size_t r = 0; r < c->lk_prof_rows; r++)
        for (int k = 0; k < PC_LK_PROFILE_SLOTS; k++) out[k] += h[r * PC_LK_PROFILE_SLOTS + k];
    return PC_OK;
}

int pc_debug_lk_x86_stats(pc_context* c, int enable, unsigned long long* out) {
    if (!c) return fail(PC_E_INVALID, "null context");
    PC_HIP(hipDeviceSynchronize());
    if (out) {
        for (int k = 0; k < 4; k++) out[k] = 0;
        if (c->lk_x86_stats.p) PC_HIP(hipMemcpy(out, c->lk_x86_stats.p, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }
    if (enable) {
        PC_HIP(c->lk_x86_stats.ensure(4));
        PC_HIP(hipMemset(c->lk_x86_stats.p, 0, 4 * sizeof(unsigned long long)));
        PC_HIP(hipDeviceSynchronize());   // the fill is asynchronous and the LK launches run on non-blocking streams
    } else {
        c->lk_x86_stats.release();
    }
    return PC_OK;
}

int pc_context_pci_bus_id(pc_context* c, char* buf, int len) {
    if (!c || !buf || len < 16) return fail(PC_E_INVALID, "bad argument");
    PC_HIP(hipDeviceGetPCIBusId(buf, len, c->device));
    return PC_OK;
}

int pc_context_reset_timing(pc_context* c) {
    if (!c) return fail(PC_E_INVALID, "null context");
    int rc = collect_timing(c);
    for (int k = 0; k < PC_K_COUNT; k++) {
        c->launches[k] = 0;
        c->total_ms[k] = 0;
        c->busy_ms[k] = 0;
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
    if (max_level < 0) return fail(PC_E_INVALID, "max_level must be >= 0");
    // OpenCV takes any maxLevel and stops where the next level would be <= winSize: with windows >= 3 px and frames <= 2^30 pixels
    // no pyramid reaches PC_MAX_LEVELS levels, so clamping the request changes nothing
    max_level = std::min(max_level, PC_MAX_LEVELS - 1);
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
    size_t img_off[PC_MAX_LEVELS], der_off[PC_MAX_LEVELS], i16_off[PC_MAX_LEVELS];
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
        i16_off[l] = total;
        total += align_up((rows + 2 * pc::kImg16SlackRows) * L.pitch * sizeof(uint16_t), 256);
        f->nlevels = l + 1;
        lw = (lw + 1) / 2;
        lh = (lh + 1) / 2;
        if (lw <= window_size || lh <= window_size) break;
    }
    // tail slack: the LK staging reads 16 uint16 columns per region row, a few past the last row of the last plane
    total += 256;
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
        L.img16 = reinterpret_cast<uint16_t*>(f->slab + i16_off[l]) + interior + (size_t)pc::kImg16SlackRows * L.pitch;
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
        (void)f->ctx->sync_side_streams();
        (void)hipStreamSynchronize(f->ctx->stream);
        if (f->ctx->eig_owner == f) f->ctx->eig_owner = nullptr;
    }
    if (f->slab) (void)hipFree(f->slab);
    if (f->d_kps) (void)hipFree(f->d_kps);
    if (f->d_perm) (void)hipFree(f->d_perm);
    delete f;
}

}  // extern "C"

// channels: 1 / 3 = u8 gray / RGB; elem_size 4 = float32 RGB(A) with `channels` floats per pixel
int pc_api::detect_counter_words(const DetectScratch& d) { return d.counter_words; }

int pc_api::set_image(pc_context* ctx, pc_frame* f, const uint8_t* src, size_t row_pitch, int on_device, int channels,
                      int elem_size, uint32_t* clear, int clear_words) {
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
    int staged = -1;   // staging buffer of the copy stream this frame's level-0 kernel reads
    if (on_device == PC_FRAME_PINNED_HOST) {
        // page-locked host memory the caller leaves alone until the frame has been consumed: one DMA transfer into the
        // staging buffer, stream-ordered, no wait.  (The kernels can also read such memory themselves -- on_device = 1 --
        // but then a level-0 kernel holds its CUs for as long as PCIe takes: 33 MB of float pixels at 17 GB/s against
        // 50 GB/s for the copy engine.)
        // rows 0 .. h-2 with their pitch, the last row without: a caller's buffer need not extend to a full last pitch
        const size_t bytes = f->h > 0 ? row_pitch * (size_t)(f->h - 1) + row_bytes : 0;
        if (ctx->copy_stream && ctx->work == ctx->prep_stream) {
            // Inside the analyzer the transfer runs on a stream of its own, into one of two staging buffers in turn: the
            // copy of frame f + 1 then overlaps the detection of frame f instead of sitting in front of the frame's kernels
            // on the preparation stream (whose chain of dependent commands per frame, not the GPU's throughput, is what
            // host frames were limited by: 1080p 2050 frames/s against 3040 with frames already on the device).
            const int b = ctx->staging_turn ^= 1;
            staged = b;
            PC_HIP(ctx->staging2[b].ensure(row_pitch * (size_t)f->h));
            if (!ctx->staging_ev[b][0]) {
                PC_HIP(hipEventCreateWithFlags(&ctx->staging_ev[b][0], hipEventDisableTiming));
                PC_HIP(hipEventCreateWithFlags(&ctx->staging_ev[b][1], hipEventDisableTiming));
            } else {
                PC_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->staging_ev[b][1], 0));   // the kernels that read this buffer two frames ago
            }
            PC_HIP(hipMemcpyAsync(ctx->staging2[b].p, src, bytes, hipMemcpyHostToDevice, ctx->copy_stream));
            PC_HIP(hipEventRecord(ctx->staging_ev[b][0], ctx->copy_stream));
            PC_HIP(hipStreamWaitEvent(ctx->work, ctx->staging_ev[b][0], 0));
            d_src = ctx->staging2[b].p;
        } else {
            PC_HIP(ctx->staging.ensure(row_pitch * (size_t)f->h));
            PC_HIP(hipMemcpyAsync(ctx->staging.p, src, bytes, hipMemcpyHostToDevice, ctx->work));
            d_src = ctx->staging.p;
        }
    } else if (!on_device) {
        d_pitch = align_up(row_bytes, 16);
        PC_HIP(ctx->staging.ensure(d_pitch * f->h));
        // the previous frame's kernels may still read the staging buffer: ordered on the same stream
        PC_HIP(hipMemcpy2DAsync(ctx->staging.p, d_pitch, src, row_pitch, row_bytes, f->h, hipMemcpyHostToDevice, ctx->work));
        d_src = ctx->staging.p;
    }
    if (!unfused_pyramid() && level_fits_fused(f->levels[0], f->win)) {
        // level 0 in one pass: gray conversion fused with the padding, the uint16 plane and the Scharr plane
        pc::LevelSource in{};
        in.kind = elem_size == 4 ? pc::SRC_RGBF32 : (channels == 3 ? pc::SRC_RGB8 : pc::SRC_GRAY8);
        in.src = d_src;
        in.src_pitch = d_pitch;
        in.channels = channels;
        in.aligned = ((reinterpret_cast<uintptr_t>(d_src) | d_pitch) & 3) == 0;
        in.clear = clear;
        in.clear_words = clear ? clear_words : 0;
        {
            ScopedTimer t(ctx, PC_K_PYRAMID);
            pc::launch_level(in, f->levels[0], f->win, ctx->work);
        }
        if (staged >= 0) PC_HIP(hipEventRecord(ctx->staging_ev[staged][1], ctx->work));
        build_pyramid(ctx, f, 1);
    } else {
        if (clear) PC_HIP(hipMemsetAsync(clear, 0, (size_t)clear_words * sizeof(uint32_t), ctx->work));
        {
            ScopedTimer t(ctx, PC_K_GRAY);
            if (elem_size == 4) pc::launch_rgbf32_to_gray(reinterpret_cast<const float*>(d_src), d_pitch, channels, f->levels[0], ctx->work);
            else if (channels == 3) pc::launch_rgb2gray(d_src, d_pitch, f->levels[0], ctx->work);
            else pc::launch_copy_gray(d_src, d_pitch, f->levels[0], ctx->work);
        }
        if (staged >= 0) PC_HIP(hipEventRecord(ctx->staging_ev[staged][1], ctx->work));
        build_pyramid(ctx, f, 0);
    }
    f->n_kps = -1;
    f->n_cands = -1;
    f->perm_valid = false;
    if (ctx->eig_owner == f) ctx->eig_owner = nullptr;
    if (!on_device) PC_HIP(hipStreamSynchronize(ctx->work));  // caller may reuse its host buffer
    return PC_OK;
}

extern "C" {

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
    if ((rc = detect_enqueue(ctx, f, g, *opt, *ctx->detect, ctx->lk_hist, /*full_launch=*/true)) != PC_OK) return rc;
    ctx->eig_owner = f;
    if ((rc = detect_finish(ctx, f, g, *opt, *ctx->detect, ctx->lk_hist)) != PC_OK) return rc;
    PC_HIP(hipStreamSynchronize(ctx->stream));
    return PC_OK;
}

int pc_frame_download_min_eig(pc_context* ctx, const pc_frame* f, float* out) {
    if (!ctx || !f || !out) return fail(PC_E_INVALID, "null argument");
    if (int jrc = join_prep(ctx)) return jrc;
    if (ctx->eig_owner != f || !ctx->detect) return fail(PC_E_STATE, "the min-eig scratch map belongs to another frame (call right after pc_frame_detect)");
    PC_HIP(hipMemcpyAsync(out, ctx->detect->eig.p, (size_t)f->w * f->h * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
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
        // the kernel's records are in visiting order: bring them into the [target][n] arrays of the call
        PC_HIP(ctx->lk_cxy.ensure(rows));
        PC_HIP(ctx->lk_cerr.ensure(rows));
        PC_HIP(ctx->lk_ustatus.ensure(rows));
        pc::launch_unpack_records(ctx->lk_rec[0].p, ctx->lk_slot_of[0], frame1->n_kps, n_targets, ctx->lk_cxy.p,
                                  ctx->lk_ustatus.p, ctx->lk_cerr.p, ctx->stream);
        PC_HIP(hipMemcpyAsync(next_xy, ctx->lk_cxy.p, rows * sizeof(float2), hipMemcpyDeviceToHost, ctx->stream));
        PC_HIP(hipMemcpyAsync(status, ctx->lk_ustatus.p, rows, hipMemcpyDeviceToHost, ctx->stream));
        PC_HIP(hipMemcpyAsync(err, ctx->lk_cerr.p, rows * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
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
    PC_HIP(ctx->lk_cxy.ensure(rows + 1));
    PC_HIP(ctx->lk_cerr.ensure(rows + 1));
    PC_HIP(ctx->lk_cidx.ensure(rows + 1));
    if (n > pc::kCompactMaxKeypoints) return fail(PC_E_CAPACITY, "%d keypoints: more than the compaction handles (%d)", n, pc::kCompactMaxKeypoints);
    const size_t scratch_cap_before = ctx->lk_block_counts[0].cap;   // a reallocation changes the capacity (the address may repeat)
    PC_HIP(ctx->lk_block_counts[0].ensure(pc::compact_scratch_words(n, n_targets)));
    PC_HIP(ctx->lk_row_offset.ensure(PC_MAX_TARGETS + 1));
    PC_HIP(ctx->h_row_offset.ensure(PC_MAX_TARGETS + 1));
    {
        ScopedTimer t(ctx, PC_K_COMPACT);
        pc::launch_compact(ctx->lk_rec[0].p, ctx->lk_slot_of[0], n, n_targets, ctx->lk_block_counts[0].p,
                           ctx->lk_block_counts[0].cap != scratch_cap_before, ctx->lk_row_offset.p, ctx->lk_cidx.p, ctx->lk_cxy.p, ctx->lk_cerr.p, ctx->stream);
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

}  // extern "C"
