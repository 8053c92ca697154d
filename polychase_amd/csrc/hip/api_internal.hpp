// api_internal.hpp -- helpers shared by the C-ABI translation units (api.hip, api_analyzer.hip, api_tracker.hip).
#pragma once

#include <chrono>
#include <ratio>

#include "internal.hpp"

namespace pc_api {

// HIP-event timing of one kernel class on one stream (pc_context_enable_timing)
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

// image / detection helpers enqueue on prep_stream while one of these is alive
struct PrepScope {
    pc_context* c;
    explicit PrepScope(pc_context* ctx, hipStream_t s = nullptr) : c(ctx) {
        c->work = s ? s : c->prep_stream;
        c->prep_dirty = true;
    }
    ~PrepScope() { c->work = c->stream; }
};

// POLYCHASE_TRACE_ALLOC: report host-side sections of a call that take more than 2 ms
struct SlowSection {
    const char* name;
    std::chrono::steady_clock::time_point t0;
    explicit SlowSection(const char* n) : name(n), t0(std::chrono::steady_clock::now()) {}
    ~SlowSection() {
        if (!pc::trace_allocations()) return;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms > 2.0) fprintf(stderr, "[polychase_hip] slow host section %s: %.2f ms\n", name, ms);
    }
};

int collect_timing(pc_context* c);
int ensure_kp_capacity(pc_frame* f, int n);
int validate_gftt(const pc_gftt_options* opt, int w, int h, pc::GfttGrid* g);
// GoodFeaturesToTrack (internal.hpp DetectScratch): everything enqueued on the current work stream / the count on the host
int detect_reserve(pc_context* ctx, int w, int h, DetectScratch& d);
int detect_enqueue(pc_context* ctx, pc_frame* f, const pc::GfttGrid& grid, const pc_gftt_options& opt, DetectScratch& d,
                   DevBuf<uint32_t>& hist, bool full_launch = false);
// *redone (may be null): the frame took the slow path, its keypoints were written again just now
int detect_finish(pc_context* ctx, pc_frame* f, const pc::GfttGrid& grid, const pc_gftt_options& opt, DetectScratch& d,
                  DevBuf<uint32_t>& hist, bool* redone = nullptr);
// orders `stream` behind everything queued on the side streams so far
int join_prep(pc_context* ctx);
int order_keypoints_spatially(pc_context* ctx, pc_frame* f, DevBuf<uint32_t>& hist);
int check_lk_args(pc_context* ctx, const pc_frame* frame1, const pc_frame* const* targets, int n_targets,
                  const pc_flow_options* opt);
// LK launch into output set `set` on job lane `set` (0: the context's main stream)
int run_lk(pc_context* ctx, const pc_frame* frame1, const pc_frame* const* targets, int n_targets, const pc_flow_options* opt,
           int set = 0);
// gray (+ pyramid) of a frame from u8 gray / u8 RGB / float32 RGB(A) pixels, host or device, on the work stream
// `clear` (may be null): clear_words words zeroed in front of the frame's kernels (by the level-0 kernel where there is one)
int set_image(pc_context* ctx, pc_frame* f, const uint8_t* src, size_t row_pitch, int on_device, int channels, int elem_size = 1,
              uint32_t* clear = nullptr, int clear_words = 0);
int detect_counter_words(const DetectScratch& d);   // words of DetectScratch::counters a detection expects zeroed

}  // namespace pc_api
