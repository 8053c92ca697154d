// internal.hpp -- host-side structures shared by api.hip and analyzer.hip (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../../include/polychase_hip.h"
#include "kernels.hpp"

static_assert(pc::kMaxLevels == PC_MAX_LEVELS, "LKParams holds PC_MAX_LEVELS levels");

namespace pc {

std::string& last_error();

inline int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}

#define PC_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(PC_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// POLYCHASE_TRACE_ALLOC=1: report every (re)allocation of a scratch buffer (each one stalls the pipeline)
inline bool trace_allocations() {
    static const bool on = getenv("POLYCHASE_TRACE_ALLOC") != nullptr;
    return on;
}

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;  // elements
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 2 + 64;   // generous: a reallocation synchronises the device
        if (trace_allocations()) fprintf(stderr, "[polychase_hip] device buffer %zu -> %zu bytes\n", cap * sizeof(T), want * sizeof(T));
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

template <typename T>
struct PinBuf {
    T* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 2 + 64;   // pinned allocations take tens of milliseconds
        if (trace_allocations()) fprintf(stderr, "[polychase_hip] pinned buffer %zu -> %zu bytes\n", cap * sizeof(T), want * sizeof(T));
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&p), want * sizeof(T), hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct TimedRange {
    int cls;
    hipEvent_t a, b;
};

}  // namespace pc

using pc::DevBuf;
using pc::PinBuf;
using pc::TimedRange;
using pc::fail;

// Per-detection buffers of one frame.  Detection is enqueued in one go, without the host in the loop:
//   min-eig map + per-cell max, threshold + NMS -> candidate keys + state bytes + per-bucket counts, bucket sort of the
//   candidates (counts stay on the device), greedy suppression in priority order, ordered compaction -> keypoints, LK
//   visiting order; the counters go to pinned memory (ev_b).  detect_finish() then reads the keypoint count -- and
//   redoes the frame on the slow path (candidate count on the host, rocPRIM sort) if a fast-path bound was exceeded.
struct DetectScratch {
    DevBuf<unsigned long long> keys, keys_bucketed, keys_sorted;
    DevBuf<float> eig;                     // min-eig map (K2 -> K3, K5)
    DevBuf<float> cov;                     // covariance planes of the general corner response (block_size != 3 / Harris), on demand
    DevBuf<double> box_rows;               // row sums of the box filter for large block sizes (kernels.hpp kBoxRowsFromBlock), on demand
    DevBuf<uint8_t> cstate;                // 0 no candidate / 1 candidate / 2 accepted / 3 rejected
    // [0] candidates, [1] keypoints, [2] stuck lanes, [3] fast-path overflow bits, [4] sort range hi, [5] sort shift, [6..7] pad,
    // [8 ..] cell max [kMaxGridCells], bucket counts [kSortBuckets], bucket cursors [kSortBuckets]
    DevBuf<uint32_t> counters;
    DevBuf<uint32_t> bucket_offsets;       // [kSortBuckets + 1]
    DevBuf<uint32_t> per_block;            // accepted candidates per suppression workgroup -> their exclusive scan
    DevBuf<uint32_t> sup_grid;             // min_distance > 64: the accepted corners by grid cell (kernels_gftt.hip), on demand
    DevBuf<uint32_t> bin_hist;             // keypoints per 64x64 tile (the analyzer's detections; stage-level calls use the context's)
    PinBuf<uint32_t> h_counters;           // counters[0..7] after the detection
    hipEvent_t ev_b = nullptr;
    uint32_t cand_cap = 0;                 // candidates the fast path holds
    int counter_words = 0;                 // words of `counters` (header, cell maxima, bucket counts + cursors, tickets)
    uint32_t ticket_stride = 0;            // words between the three ticket arrays at the end of `counters`
    bool cleared = false;                  // `counters` were zeroed in front of the coming detection (by the frame's level-0 kernel)
    void release() {
        keys.release();
        keys_bucketed.release();
        keys_sorted.release();
        eig.release();
        cov.release();
        box_rows.release();
        cstate.release();
        counters.release();
        bucket_offsets.release();
        per_block.release();
        sup_grid.release();
        bin_hist.release();
        h_counters.release();
        if (ev_b) (void)hipEventDestroy(ev_b);
        ev_b = nullptr;
    }
};

struct pc_context {
    int device = 0;
    int arith = PC_ARITH_OPENCV_X86;     // pc_context_set_arithmetic; the default = what a stock x86-64 OpenCV build executes
    // Stage-level calls run on `stream`.  pc_analyzer alternates its jobs (LK launch + compaction + device-log append +
    // record download of one frame1) over two job lanes, `stream` and `stream_b`: the launches of consecutive frames
    // overlap, so the tail of one launch and the gap before the next are filled by the other lane's wavefronts.
    // (HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues, 4 unless raised, and a stream that shares a queue
    // waits behind the other's commands: pc_runtime_init raises the default before the first HIP call.)
    hipStream_t stream = nullptr;
    hipStream_t stream_b = nullptr;
    hipStream_t lane_stream(int lane) const { return lane ? stream_b : stream; }
    // pc_analyzer runs frame preparation (gray, pyramid, detection, keypoint ordering) on its own
    // stream so that it overlaps the LK launch of the previous frame1 on `stream`.  `work` is the stream
    // the image / detection helpers enqueue on: `stream` by default, `prep_stream` inside the analyzer.
    hipStream_t prep_stream = nullptr;
    // Detection of a resident frame is needed only when the frame becomes frame1, nine steps after it was put, while its
    // pyramid is read by the very next launch: the analyzer enqueues detections on their own streams (frames alternate
    // over n_detect of them) so that the image path never queues behind a suppression kernel and the latency-bound
    // detection kernels of neighbouring frames overlap.  n_detect = 0: detection follows the image path on prep_stream.
    hipStream_t detect_stream[2] = {nullptr, nullptr};
    int n_detect = 0;
    hipStream_t detect_stream_for(int32_t frame_id) const {
        return n_detect > 0 ? detect_stream[(unsigned)frame_id % (unsigned)n_detect] : prep_stream;
    }
    hipError_t sync_side_streams() const {
        hipError_t e = copy_stream ? hipStreamSynchronize(copy_stream) : hipSuccess;
        if (e == hipSuccess) e = hipStreamSynchronize(prep_stream);
        if (stream_b && e == hipSuccess) e = hipStreamSynchronize(stream_b);
        for (int k = 0; k < n_detect && e == hipSuccess; k++) e = hipStreamSynchronize(detect_stream[k]);
        return e;
    }
    hipStream_t work = nullptr;
    hipEvent_t prep_fence = nullptr;     // orders `stream` after everything queued on prep_stream so far
    bool prep_dirty = false;
    // staging of host-provided frames
    DevBuf<uint8_t> staging;
    // the analyzer's host frames (PC_FRAME_PINNED_HOST): transfers on a stream of their own, two buffers in turn;
    // staging_ev[b] = {copy into b complete (copy stream), the kernel that reads b has run (preparation stream)}
    hipStream_t copy_stream = nullptr;   // POLYCHASE_COPY_STREAM=0: none (the transfer is enqueued in front of the frame's kernels)
    DevBuf<uint8_t> staging2[2];
    hipEvent_t staging_ev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    int staging_turn = 0;
    // GFTT scratch
    DevBuf<int2> sup_offsets;              // suppression neighbourhood for sup_min_distance
    DevBuf<int> sup_rows;                  // the same as half-widths per row, [2 * sup_R + 1]
    int sup_R = 0;
    double sup_min_distance = -1.0;
    int n_sup_offsets = 0;
    // candidates of the latest finished detection + 25 %: the detections enqueued next size their launches for that many
    // (the count itself stays on the device; a frame with more is redone with launches for the full capacity)
    uint32_t cand_hint = 0;
    struct DetectScratch* detect = nullptr;  // scratch of the stage-level pc_frame_detect
    DevBuf<uint8_t> sort_temp;
    const pc_frame* eig_owner = nullptr;
    // LK scratch
    // raw LK outputs, compaction scratch and packed records: one set per job lane of the analyzer (set 0: stage-level calls)
    DevBuf<float4> lk_rec[2];              // raw records in visiting order (kernels.hpp LKParams::out_rec)
    const uint32_t* lk_slot_of[2] = {nullptr, nullptr};   // inverse visiting order of the latest launch into the set
    DevBuf<float2> lk_cxy;
    DevBuf<uint8_t> lk_ustatus;            // pc_lk_track: unpacked status
    DevBuf<float> lk_cerr;
    DevBuf<uint32_t> lk_cidx, lk_block_counts[2], lk_perm, lk_hist;
    DevBuf<uint32_t> lk_gate;              // LKParams::gate of the analyzer's launches (one word)
    PinBuf<uint32_t> lk_gate_timed_out;    // raised by a gate kernel that gave up (its stream shares a hardware queue with the other lane)
    uint32_t lk_gate_seq = 0;              // value the latest gated launch stores
    uint32_t lk_gate_next = 0;             // run_lk: value for the coming launch (0: not gated)
    bool lk_gate_on = true;                // POLYCHASE_LK_GATE=0 switches the gate off
    DevBuf<unsigned long long> lk_prof;    // pc_debug_lk_profile: 16 words per wavefront of the latest launch
    size_t lk_prof_rows = 0;
    DevBuf<unsigned long long> lk_x86_stats;   // pc_debug_lk_x86_stats: four counters, or unallocated (not counting)
    DevBuf<long long> lk_row_offset;
    // the analyzer's compacted records of one job, packed like a device-log record without its header:
    // row offsets (128 B) | keypoints | src indices | tgt xy | errors, every part 16-byte aligned
    DevBuf<uint8_t> lk_pack[2];
    PinBuf<long long> h_row_offset;
    // timing
    unsigned timing_mask = 0;   // bit k: time kernel class k with HIP events
    std::vector<TimedRange> ranges;
    std::vector<hipEvent_t> event_pool;
    int launches[PC_K_COUNT] = {0};
    double total_ms[PC_K_COUNT] = {0};
    double busy_ms[PC_K_COUNT] = {0};    // time during which at least one launch of the class was executing
};

struct pc_frame {
    pc_context* ctx = nullptr;
    int w = 0, h = 0, win = 0, max_level = 0, nlevels = 0;
    pc::Level levels[PC_MAX_LEVELS];
    uint8_t* slab = nullptr;
    size_t slab_bytes = 0;
    float2* d_kps = nullptr;
    int kp_cap = 0;
    int n_kps = -1;   // -1: none
    int n_cands = -1;
    uint32_t* d_perm = nullptr;   // LK visiting order of d_kps (spatial bins) [perm_cap], then its inverse [perm_cap]; valid iff perm_valid
    int perm_cap = 0;
    bool perm_valid = false;
};

