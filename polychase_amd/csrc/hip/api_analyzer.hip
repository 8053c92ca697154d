// api_analyzer.hip -- pc_analyzer: the pipelined per-clip engine behind GenerateOpticalFlowDatabase
// (reference cpp/opticalflow.cc:209-321).  See include/polychase_hip.h for the contract and DESIGN.md section 3
// for the stream / event structure.
#include "api_internal.hpp"

using namespace pc_api;

// =============================================================================================
// analyzer: pipelined per-clip engine (see include/polychase_hip.h)
// =============================================================================================

namespace {

constexpr int kSpareSlots = PC_ANALYZER_SPARE_SLOTS;

enum DetState { DET_NONE = 0, DET_ENQUEUED = 2, DET_DONE = 3 };   // detection enqueued / keypoint count on the host

struct Slot {
    pc_frame* frame = nullptr;
    int32_t frame_id = 0;
    bool valid = false;
    DetState det = DET_NONE;
    bool supplied = false;  // keypoints came from the caller (database), not from detection
    DetectScratch scratch;
    hipEvent_t last_read[2] = {nullptr, nullptr};  // per job lane: `lk_done` of the latest job whose LK reads this slot
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
    bool host_records = true;       // the whole pack was downloaded (else its 128-byte header only)
    hipEvent_t done = nullptr;      // records of this job are in pinned memory (the job's lane)
};

// Issue priority of the helper kernels this analyzer enqueues (common.hpp: helper_priority), from what it observes.
// The preparation stream is a FIFO: pyramid of frame f+9, then its detection.  When that chain falls behind the LK
// launches -- the pyramid of the SECOND newest frame is not complete when a job is collected, i.e. less than one frame of
// slack is left before a launch has to wait for it -- the helpers get the high priority; when no lag has been seen for
// kCalm collects in a row (and it takes kLagsToRaise lags among 16 collects to raise it) they go back to filling the slots LK leaves (the better mode whenever they keep up).
// POLYCHASE_HELPER_PRIO=0 / 1 pins the priority, "auto" (default) is this controller.
struct HelperPriorityControl {
    static constexpr int kCalm = 48;        // collects without a lag before the priority is dropped again
    static constexpr int kLagsToRaise = 3;  // ... of the last 16 collects (one late pyramid at a clip's start is no trend)
    int mode = -1;          // -1 auto, 0 / 1 pinned
    int hi = 0;
    int calm = 0;
    uint32_t window = 0;    // the last 16 observations, newest in bit 0
    uint64_t lag_seen = 0, switched_on = 0;
    void init() {
        const char* v = getenv("POLYCHASE_HELPER_PRIO");
        mode = (v && (v[0] == '0' || v[0] == '1') && v[1] == 0) ? v[0] - '0' : -1;
        hi = mode == 1 ? 1 : 0;
    }
    void observe(bool lagging) {
        if (mode >= 0) return;
        window = ((window << 1) | (lagging ? 1u : 0u)) & 0xffffu;
        if (lagging) {
            lag_seen++;
            calm = 0;
            if (!hi && __builtin_popcount(window) >= kLagsToRaise) {
                hi = 1;
                switched_on++;
            }
        } else if (hi && ++calm >= kCalm) {
            hi = 0;
            calm = 0;
            window = 0;
        }
    }
};
struct HelperPriorityScope {   // the enqueues of one analyzer call carry its priority; stage-level calls on this thread stay at 0
    explicit HelperPriorityScope(int hi) { pc::set_helper_prio(hi); }
    ~HelperPriorityScope() { pc::set_helper_prio(0); }
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
    uint64_t submitted = 0;              // jobs submitted so far: job k runs on lane k & 1 (stream, LK output set, pack)
    bool one_lane = false;               // POLYCHASE_LK_LANES=1 when the analyzer was created: every job on lane 0
    // "the LK launch of a job finished", per lane, handed out round-robin.  Slots remember the event of the last job
    // that read them; an event that has been re-recorded since marks a LATER launch of the same lane, which the
    // lane's stream order puts behind the remembered one -- waiting for it is still correct.
    static constexpr int kLaneEvents = 16;
    hipEvent_t lk_done[2][kLaneEvents] = {};
    bool gate_armed = false;             // a gated launch is ahead: the next one waits for its "all dispatched" signal
    HelperPriorityControl prio;
    int32_t put_newest = 0, put_previous = 0;   // frame ids of the two most recent put_frame calls
    int puts = 0;
    bool host_records = true;            // download every job's records to pinned host memory
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

// detection of the slot's frame, enqueued on the prep stream in one go; `kps_ready` fires when keypoints and visiting
// order are complete
int detect_dense(pc_analyzer* a, Slot& s) {
    hipStream_t const ds = a->ctx->detect_stream_for(s.frame_id);
    PrepScope prep(a->ctx, ds);
    if (ds != a->ctx->prep_stream) PC_HIP(hipStreamWaitEvent(ds, s.img_ready, 0));
    int rc = detect_enqueue(a->ctx, s.frame, a->grid, a->gopt, s.scratch, s.scratch.bin_hist);
    if (rc != PC_OK) return rc;
    PC_HIP(hipEventRecord(s.kps_ready, ds));
    s.det = DET_ENQUEUED;
    s.supplied = false;
    return PC_OK;
}

// the keypoint count on the host (waits for the detection; a frame beyond the fast path's bounds is redone synchronously)
int detect_finish_slot(pc_analyzer* a, Slot& s) {
    int rc;
    if (s.det == DET_NONE && (rc = detect_dense(a, s)) != PC_OK) return rc;
    {
        hipStream_t const ds = a->ctx->detect_stream_for(s.frame_id);
        PrepScope prep(a->ctx, ds);
        bool redone = false;
        if ((rc = detect_finish(a->ctx, s.frame, a->grid, a->gopt, s.scratch, s.scratch.bin_hist, &redone)) != PC_OK) return rc;
        // Only then: recorded unconditionally the event would sit behind the detection of the frame that was made
        // resident a moment ago, and the LK launch of THIS frame1 would wait for it (rounds 1-2 did exactly that: the
        // "preparation chain" that bounded the step was this false dependency)
        if (redone) PC_HIP(hipEventRecord(s.kps_ready, ds));
    }
    s.det = DET_DONE;
    return PC_OK;
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
    // POLYCHASE_LK_LANES=1 (read per analyzer): every job on lane 0, i.e. no two LK launches in flight -- a launch's
    // start-to-end time is then its own duration (the recipe of tools/collect_profiles.sh under rocprofv3 and of bench.py's
    // roofline time base); the product runs two lanes
    a->one_lane = getenv("POLYCHASE_LK_LANES") && atoi(getenv("POLYCHASE_LK_LANES")) == 1;
    a->w = width;
    a->h = height;
    a->gopt = *gftt;
    a->fopt = *flow;
    a->grid = grid0;
    a->prio.init();
    // spare slots: a frame can be overwritten (prep stream) while the LK launches that read its predecessors in the
    // ring are still running, without the streams waiting on each other, and the drivers make frame1 + 9 resident
    // one step before the first launch that reads it (PC_ANALYZER_LOOKAHEAD) so that its pyramid never sits on
    // the critical path of that launch
    a->slots.resize((size_t)ring_frames + kSpareSlots);
    a->jobs.resize((size_t)max_jobs);
    int rc = PC_OK;
    for (auto& s : a->slots) {
        rc = pc_frame_create(ctx, width, height, flow->window_size, flow->max_level, &s.frame);
        if (rc != PC_OK) break;
        // room for a typical frame's keypoints up front: growing later frees device memory, which synchronises
        const int kp0 = std::max(16384, (width * height) / 16);   // candidates of a textured frame: ~1 per 23 px
        if ((rc = ensure_kp_capacity(s.frame, kp0)) != PC_OK) break;
        if (hipMalloc(reinterpret_cast<void**>(&s.frame->d_perm), (size_t)kp0 * 2 * sizeof(uint32_t)) != hipSuccess) {
            rc = fail(PC_E_HIP, "hipMalloc failed");
            break;
        }
        s.frame->perm_cap = kp0;
        if ((rc = detect_reserve(ctx, width, height, s.scratch)) != PC_OK) break;
        if (s.scratch.bin_hist.ensure((size_t)pc::bin_num_tiles(width, height) + 1) != hipSuccess) {
            rc = fail(PC_E_HIP, "allocation failed");
            break;
        }
        if (hipEventCreateWithFlags(&s.img_ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.kps_ready, hipEventDisableTiming) != hipSuccess) {
            rc = fail(PC_E_HIP, "hipEventCreate failed");
            break;
        }
    }
    if (rc == PC_OK)
        for (auto& j : a->jobs)
            if (hipEventCreateWithFlags(&j.done, hipEventDisableTiming) != hipSuccess) {
                rc = fail(PC_E_HIP, "hipEventCreate failed");
                break;
            }
    if (rc == PC_OK)
        for (auto& lane : a->lk_done)
            for (hipEvent_t& e : lane)
                if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) rc = fail(PC_E_HIP, "hipEventCreate failed");
    if (rc == PC_OK) {
        // Warm the runtime's copy engines: it picks a free SDMA engine per copy and creates an engine's queue the first
        // time it is used (5-8 ms inside some hipMemcpyAsync, observed twice or three times in the first few dozen
        // frames).  A burst of overlapping downloads makes it create them now.
        const size_t chunk = (size_t)4 << 20, burst = 12;
        Job& j0 = a->jobs[0];
        if (ctx->lk_pack[0].ensure(chunk) != hipSuccess || ctx->lk_pack[1].ensure(chunk) != hipSuccess ||
            j0.h_pack.ensure(chunk * burst) != hipSuccess) {
            rc = fail(PC_E_HIP, "allocation failed");
        } else {
            hipStream_t streams[3] = {ctx->stream_b, ctx->prep_stream, ctx->stream};
            for (int round = 0; round < 3 && rc == PC_OK; round++) {
                for (size_t k = 0; k < burst; k++)
                    if (hipMemcpyAsync(j0.h_pack.p + k * chunk, ctx->lk_pack[k & 1].p, chunk, hipMemcpyDeviceToHost, streams[k % 3]) !=
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
    (void)a->ctx->sync_side_streams();
    (void)hipStreamSynchronize(a->ctx->stream);
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
    }
    for (auto& lane : a->lk_done)
        for (hipEvent_t e : lane)
            if (e) (void)hipEventDestroy(e);
    delete a;
}

int pc_analyzer_reset(pc_analyzer* a) {
    if (!a) return fail(PC_E_INVALID, "null analyzer");
    if (a->job_count != 0) return fail(PC_E_STATE, "%zu jobs in flight: collect them first", a->job_count);
    PC_HIP(hipSetDevice(a->ctx->device));
    PC_HIP(a->ctx->sync_side_streams());
    PC_HIP(hipStreamSynchronize(a->ctx->stream));
    for (auto& s : a->slots) {
        s.valid = false;
        s.det = DET_NONE;
        s.supplied = false;
        s.frame_id = 0;
        s.last_read[0] = s.last_read[1] = nullptr;
        s.scratch.cleared = false;
    }
    a->job_head = 0;
    a->d_log = nullptr;
    a->log_cap = a->log_used = 0;
    a->host_records = true;
    a->put_newest = a->put_previous = 0;
    a->puts = 0;
    a->prio = HelperPriorityControl();
    a->prio.init();
    return PC_OK;
}

static int analyzer_put(pc_analyzer* a, int32_t frame_id, const uint8_t* rgb, size_t row_pitch, int on_device,
                        int will_detect, int channels, int elem_size) {
    if (!a || !rgb) return fail(PC_E_INVALID, "null argument");
    const int n = (int)a->slots.size();
    Slot& s = a->slots[(size_t)(((frame_id % n) + n) % n)];
    PC_HIP(hipSetDevice(a->ctx->device));
    PrepScope prep(a->ctx);
    HelperPriorityScope prio_scope(a->prio.hi);
    a->put_previous = a->put_newest;
    a->put_newest = frame_id;
    a->puts++;
    // an LK launch in flight may still read the frame this slot holds
    for (hipEvent_t& e : s.last_read) {
        if (e) PC_HIP(hipStreamWaitEvent(a->ctx->prep_stream, e, 0));
        e = nullptr;
    }
    // ... and so may the detection of a frame that was never submitted as frame1
    if (s.valid && s.det == DET_ENQUEUED && a->ctx->n_detect > 0) PC_HIP(hipStreamWaitEvent(a->ctx->prep_stream, s.kps_ready, 0));
    // the detection that follows expects its counters zeroed: the frame's first kernel does it on the way
    const bool clear_here = will_detect && a->ctx->n_detect == 0;
    int rc = set_image(a->ctx, s.frame, rgb, row_pitch, on_device, channels, elem_size, clear_here ? s.scratch.counters.p : nullptr,
                       detect_counter_words(s.scratch));
    s.scratch.cleared = rc == PC_OK && clear_here;
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

int pc_analyzer_frame_ingested(const pc_analyzer* a, int32_t frame_id) {
    if (!a) return 1;
    Slot* s = find_slot(const_cast<pc_analyzer*>(a), frame_id);
    if (!s || !s->img_ready) return 1;   // evicted: whatever read it was ordered before the eviction
    return hipEventQuery(s->img_ready) == hipSuccess ? 1 : 0;
}

int pc_analyzer_set_keypoints(pc_analyzer* a, int32_t frame_id, const float* xy, int n) {
    if (!a || n < 0 || (!xy && n > 0)) return fail(PC_E_INVALID, "bad argument");
    Slot* s = find_slot(a, frame_id);
    if (!s) return fail(PC_E_STATE, "frame %d is not resident", frame_id);
    HelperPriorityScope prio_scope(a->prio.hi);
    int rc = ensure_kp_capacity(s->frame, n);
    if (rc != PC_OK) return rc;
    if (n > 0) {
        // resume path (keypoints from the database): pageable source, so the copy is synchronous
        hipStream_t const ds = a->ctx->detect_stream_for(frame_id);
        PC_HIP(hipStreamSynchronize(ds));   // a detection of this frame in flight would write the same buffer
        PC_HIP(hipMemcpyAsync(s->frame->d_kps, xy, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, ds));
        PC_HIP(hipStreamSynchronize(ds));
    }
    s->frame->n_kps = n;
    {
        hipStream_t const ds = a->ctx->detect_stream_for(frame_id);
        PrepScope prep(a->ctx, ds);
        if ((rc = order_keypoints_spatially(a->ctx, s->frame, s->scratch.bin_hist)) != PC_OK) return rc;
        PC_HIP(hipEventRecord(s->kps_ready, ds));
    }
    s->det = DET_DONE;
    s->supplied = true;
    return PC_OK;
}


int pc_analyzer_submit(pc_analyzer* a, int32_t frame1, const int32_t* targets, int n_targets) {
    if (!a || (n_targets > 0 && !targets)) return fail(PC_E_INVALID, "null argument");
    if (n_targets < 0 || n_targets > PC_MAX_TARGETS) return fail(PC_E_INVALID, "n_targets must be in [0,%d]", PC_MAX_TARGETS);
    if (a->job_count == a->jobs.size()) return fail(PC_E_STATE, "%zu jobs already in flight: call pc_analyzer_collect", a->job_count);
    pc_context* ctx = a->ctx;
    PC_HIP(hipSetDevice(ctx->device));
    HelperPriorityScope prio_scope(a->prio.hi);
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
        if ((rc = detect_finish_slot(a, *s1)) != PC_OK) return rc;
        detected = true;
    } else {
        detected = !s1->supplied;
    }
    Job& j = a->jobs[(a->job_head + a->job_count) % a->jobs.size()];
    // (2) the job's lane; its LK launch needs this frame's keypoints and the pyramids of the frames it reads -- not the
    // detection of frames that were made resident for later
    // POLYCHASE_LK_LANES=1: every job on lane 0, i.e. no two LK launches in flight -- the recipe for per-dispatch durations under
    // rocprofv3 (tools/collect_profiles.sh); the product runs two lanes
    const int lane = a->one_lane ? 0 : (int)(a->submitted & 1);
    hipStream_t const ls = ctx->lane_stream(lane);
    {
        SlowSection ss("submit/waits");
        PC_HIP(hipStreamWaitEvent(ls, s1->kps_ready, 0));
        PC_HIP(hipStreamWaitEvent(ls, s1->img_ready, 0));
        for (int t = 0; t < n_targets; t++) PC_HIP(hipStreamWaitEvent(ls, find_slot(a, targets[t])->img_ready, 0));
    }
    const int n = s1->frame->n_kps;
    if (n > pc::kCompactMaxKeypoints)   // before anything of the job is enqueued
        return fail(PC_E_CAPACITY, "%d keypoints: more than the compaction handles (%d)", n, pc::kCompactMaxKeypoints);
    const size_t rows = (size_t)n * (size_t)std::max(n_targets, 0);
    // packed record layout (= a device-log record without its 128-byte header)
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    j.o_kps = 128;
    j.o_idx = up16(j.o_kps + (size_t)n * 8);
    j.o_xy = up16(j.o_idx + rows * 4);
    j.o_err = up16(j.o_xy + rows * 8);
    j.pack_bytes = up16(j.o_err + rows * 4);
    // a record that does not fit the device log is refused before anything of the job is enqueued (the caller may
    // redirect the log and submit again)
    if (a->d_log && a->log_used + 128 + j.pack_bytes > a->log_cap)
        return fail(PC_E_CAPACITY, "device log full (%zu of %zu bytes)", a->log_used + 128 + j.pack_bytes, a->log_cap);
    {
        SlowSection ss("submit/ensure pack");
        PC_HIP(j.h_pack.ensure(j.pack_bytes));
        PC_HIP(ctx->lk_pack[lane].ensure(j.pack_bytes));
    }
    j.frame1 = frame1;
    j.n_kps = n;
    j.detected = detected;
    j.n_targets = n_targets;
    for (int t = 0; t < n_targets; t++) j.targets[t] = targets[t];
    // (3) everything of this job goes onto its lane, in order: LK into the lane's output set, compaction (status
    // filter) into the lane's pack, device-log append, download.  The other lane holds the neighbouring jobs, so the
    // next LK launch does not wait for any of it.
    ctx->prep_dirty = true;   // stage-level calls must order themselves behind the side streams
    if (n_targets > 0) {
        if (a->fopt.window_size != s1->frame->win) return fail(PC_E_INVALID, "window size mismatch");
        SlowSection ss("submit/run_lk");
        if (ctx->lk_gate_on && n > 0) {
            // two job lanes: this launch starts when the one before it (other lane) has handed out its last workgroup --
            // it fills that launch's tail and does not compete with its body
            if (!ctx->lk_gate.p) {
                PC_HIP(ctx->lk_gate.ensure(16));
                PC_HIP(hipMemset(ctx->lk_gate.p, 0, 16 * sizeof(uint32_t)));
                PC_HIP(ctx->lk_gate_timed_out.ensure(1));
                ctx->lk_gate_timed_out.p[0] = 0u;
            }
            if (*static_cast<volatile uint32_t*>(ctx->lk_gate_timed_out.p) != 0u) {
                // a gate kernel waited in vain: the lanes' streams share a hardware queue, the launch it waited for was
                // queued BEHIND it.  No more gates in this context.
                ctx->lk_gate_on = false;
                if (pc::trace_allocations()) fprintf(stderr, "[polychase_hip] LK gate timed out (streams share a hardware queue): gate switched off\n");
            }
            if (a->gate_armed && ctx->lk_gate_on) pc::launch_lk_gate(ctx->lk_gate.p, ctx->lk_gate_seq, ctx->lk_gate_timed_out.p, ls);
            ctx->lk_gate_next = ++ctx->lk_gate_seq;
            if (ctx->lk_gate_next == 0) ctx->lk_gate_next = ++ctx->lk_gate_seq;
            a->gate_armed = true;
        }
        if ((rc = run_lk(ctx, s1->frame, tg, n_targets, &a->fopt, lane)) != PC_OK) return rc;
    }
    SlowSection ss_post("submit/post enqueue");
    uint8_t* const pack = ctx->lk_pack[lane].p;
    long long* const p_ro = reinterpret_cast<long long*>(pack);
    PC_HIP(hipMemsetAsync(pack, 0, 128, ls));
    if (n_targets > 0) {
        const size_t scratch_cap_before = ctx->lk_block_counts[lane].cap;   // a reallocation changes the capacity (the address may repeat)
        PC_HIP(ctx->lk_block_counts[lane].ensure(pc::compact_scratch_words(n, n_targets)));
        ScopedTimer t(ctx, PC_K_COMPACT, ls);
        pc::launch_compact(ctx->lk_rec[lane].p, ctx->lk_slot_of[lane], n, n_targets, ctx->lk_block_counts[lane].p,
                           ctx->lk_block_counts[lane].cap != scratch_cap_before, p_ro, reinterpret_cast<uint32_t*>(pack + j.o_idx),
                           reinterpret_cast<float2*>(pack + j.o_xy), reinterpret_cast<float*>(pack + j.o_err), ls);
    }
    pc::launch_copy_keypoints(s1->frame->d_kps, reinterpret_cast<float2*>(pack + j.o_kps), n, ls);
    // Everything that reads the resident frames is enqueued: the pyramids (LK), frame1's keypoints (LK, the copy
    // above) and its inverse visiting order (compaction).  A slot may be overwritten once this event has fired.
    hipEvent_t const lk_done = a->lk_done[lane][(a->one_lane ? a->submitted : (a->submitted >> 1)) % pc_analyzer::kLaneEvents];
    PC_HIP(hipEventRecord(lk_done, ls));
    s1->last_read[lane] = lk_done;
    for (int t = 0; t < n_targets; t++) find_slot(a, targets[t])->last_read[lane] = lk_done;
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
        PC_HIP(hipMemcpyAsync(a->d_log + o_hdr, hh, 128, hipMemcpyHostToDevice, ls));
        PC_HIP(hipMemcpyAsync(a->d_log + o_hdr + 128, pack, j.pack_bytes, hipMemcpyDeviceToDevice, ls));
        a->log_used = end;
    }
    a->submitted++;
    // (4) download: ONE copy with fixed endpoints (the lane's pack -> the job's pinned pack).  The runtime stalls the
    // host for 5-8 ms the first time it sees a buffer as a copy source, so per-frame buffers must not appear here.
    if (a->host_records) PC_HIP(hipMemcpyAsync(j.h_pack.p, pack, j.pack_bytes, hipMemcpyDeviceToHost, ls));
    else PC_HIP(hipMemcpyAsync(j.h_pack.p, pack, 128, hipMemcpyDeviceToHost, ls));   // the row offsets only
    j.host_records = a->host_records;
    PC_HIP(hipEventRecord(j.done, ls));
    a->job_count++;
    return PC_OK;
}

int pc_analyzer_pending(const pc_analyzer* a) { return a ? (int)a->job_count : 0; }

int pc_analyzer_set_device_log(pc_analyzer* a, void* d_log, size_t capacity_bytes) {
    if (!a) return fail(PC_E_INVALID, "null analyzer");
    if (d_log && (reinterpret_cast<uintptr_t>(d_log) & 15)) return fail(PC_E_INVALID, "device log must be 16-byte aligned");
    PC_HIP(a->ctx->sync_side_streams());
    PC_HIP(hipStreamSynchronize(a->ctx->stream));
    a->d_log = static_cast<uint8_t*>(d_log);
    a->log_cap = d_log ? capacity_bytes : 0;
    a->log_used = 0;
    return PC_OK;
}

int pc_analyzer_redirect_device_log(pc_analyzer* a, void* d_log, size_t capacity_bytes) {
    if (!a || !d_log) return fail(PC_E_INVALID, "null argument");
    if (reinterpret_cast<uintptr_t>(d_log) & 15) return fail(PC_E_INVALID, "device log must be 16-byte aligned");
    // no synchronisation: the appends already enqueued keep their addresses, the next submit writes to the new buffer
    a->d_log = static_cast<uint8_t*>(d_log);
    a->log_cap = capacity_bytes;
    a->log_used = 0;
    return PC_OK;
}

int pc_analyzer_set_host_records(pc_analyzer* a, int enabled) {
    if (!a) return fail(PC_E_INVALID, "null analyzer");
    a->host_records = enabled != 0;
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
    // is the preparation stream keeping up?  (HelperPriorityControl)
    if (a->puts >= 2) {
        Slot* const sp = find_slot(a, a->put_previous);
        bool lagging = sp && sp->img_ready && hipEventQuery(sp->img_ready) == hipErrorNotReady;
        if (a->ctx->n_detect > 0 && !lagging) {
            // detection on its own stream(s): the pyramid no longer queues behind it, so its lag shows where it is
            // needed -- the keypoints of the frame1 after the next submit (one frame of slack) are not complete yet
            Slot* const sd = find_slot(a, j.frame1 + (int32_t)a->job_count + 1);
            lagging = sd && sd->det == DET_ENQUEUED && sd->kps_ready && hipEventQuery(sd->kps_ready) == hipErrorNotReady;
        }
        a->prio.observe(lagging);
    }
    // let the runtime retire the finished commands of the other streams now, a few at a time: left alone it does
    // so in one batch of several milliseconds every couple of hundred frames, inside some later launch
    (void)hipStreamQuery(a->ctx->stream);
    if (a->ctx->stream_b) (void)hipStreamQuery(a->ctx->stream_b);
    (void)hipStreamQuery(a->ctx->prep_stream);
    for (int k = 0; k < a->ctx->n_detect; k++) (void)hipStreamQuery(a->ctx->detect_stream[k]);
    out->frame1 = j.frame1;
    out->n_keypoints = j.n_kps;
    out->keypoints_detected = j.detected ? 1 : 0;
    out->keypoints_xy = j.host_records ? reinterpret_cast<const float*>(j.h_pack.p + j.o_kps) : nullptr;
    out->n_targets = j.n_targets;
    for (int t = 0; t < PC_MAX_TARGETS; t++) out->targets[t] = t < j.n_targets ? j.targets[t] : 0;
    const long long* h_ro = reinterpret_cast<const long long*>(j.h_pack.p);
    for (int t = 0; t <= PC_MAX_TARGETS; t++) out->row_offset[t] = (int64_t)h_ro[std::min(t, j.n_targets)];
    out->src_indices = j.host_records ? reinterpret_cast<const uint32_t*>(j.h_pack.p + j.o_idx) : nullptr;
    out->tgt_xy = j.host_records ? reinterpret_cast<const float*>(j.h_pack.p + j.o_xy) : nullptr;
    out->flow_err = j.host_records ? reinterpret_cast<const float*>(j.h_pack.p + j.o_err) : nullptr;
    a->job_head = (a->job_head + 1) % a->jobs.size();
    a->job_count--;
    return PC_OK;
}

}  // extern "C"
