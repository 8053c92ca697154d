// track_sequence.cc -- reference cpp/tracker.cc:36-213 on the MI355X path.
//
// For the frame being solved, every flow INTO it whose source frame already has a pose contributes
// 3D-2D correspondences: source keypoints are cast onto the mesh under the source camera
// (tracker.cc:64-78; one batched GPU launch per source here instead of one Embree call per match),
// the hits are moved to world space (:80-82) and paired with the tracked positions (:86); PnP then
// starts from this/previous/next frame's pose (:111-119).  LM residual sweeps run on the GPU
// (pnp.cc).  Sequential over frames by construction: frame k needs the poses solved before it.
#include "track_sequence.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <thread>
#include <vector>

#include "gpu_context.h"
#include "numa_pin.h"
#include "pnp.h"
#include "stage_clock.h"

namespace {

constexpr float kMaxInlierError = 12.0f;  // tracker.cc:123 ("FIXME: Make this customizable")

[[noreturn]] void ThrowHip(const char* what) { throw std::runtime_error(std::string(what) + ": " + pc_last_error()); }

// Page-locked host memory (pc_host_buffer_alloc): what the database reader fills is fetched by the GPU's copy engine as
// it is -- a pageable std::vector costs a staging copy on the calling thread for every transfer (round 4: 31 us per
// source frame, a fifth of a tracked frame).
// What a TrackSequence call keeps for the next one (POLYCHASE_TRACK_CACHE=0: nothing): creating page-locked blocks, the
// correspondence set's streams and device arrays is 6-8 ms of every call -- a third of a 50-frame run, and Blender calls
// once per user action.  Page-locked blocks go back to a small process-wide pool instead of to the driver; one correspondence
// set is parked (ParkedSet below).  Both are leaked at process exit on purpose (no GPU call after the runtime's teardown has
// begun); ReleaseTrackerCaches() -- polychase_core.release_cached_engine() -- gives the memory back earlier.
bool TrackCacheEnabled() {
    const char* e = std::getenv("POLYCHASE_TRACK_CACHE");
    return !(e && e[0] == '0');
}

class PinnedPool {
   public:
    static PinnedPool& Get() {
        static PinnedPool* pool = new PinnedPool;   // never destroyed
        return *pool;
    }
    // a block of at least `bytes` (the smallest that fits), or nullptr
    void* Take(size_t bytes, size_t* cap) {
        std::lock_guard<std::mutex> lk(m_);
        int best = -1;
        for (int i = 0; i < static_cast<int>(blocks_.size()); i++)
            if (blocks_[i].cap >= bytes && (best < 0 || blocks_[i].cap < blocks_[best].cap)) best = i;
        if (best < 0) return nullptr;
        void* p = blocks_[best].p;
        *cap = blocks_[best].cap;
        total_ -= blocks_[best].cap;
        blocks_.erase(blocks_.begin() + best);
        return p;
    }
    // false: the pool is full (or off), the caller frees the block
    bool Give(void* p, size_t cap) {
        if (!p || !TrackCacheEnabled()) return false;
        std::lock_guard<std::mutex> lk(m_);
        if (blocks_.size() >= kMaxBlocks || total_ + cap > kMaxBytes) return false;
        blocks_.push_back({p, cap});
        total_ += cap;
        return true;
    }
    void Clear() {
        std::vector<Block> drop;
        {
            std::lock_guard<std::mutex> lk(m_);
            drop.swap(blocks_);
            total_ = 0;
        }
        for (const Block& b : drop) pc_host_buffer_free(b.p);
    }

   private:
    struct Block {
        void* p;
        size_t cap;
    };
    static constexpr size_t kMaxBlocks = 48, kMaxBytes = 256u << 20;
    std::mutex m_;
    std::vector<Block> blocks_;
    size_t total_ = 0;
};

class PinnedBuffer {
   public:
    PinnedBuffer() = default;
    PinnedBuffer(const PinnedBuffer&) = delete;
    PinnedBuffer& operator=(const PinnedBuffer&) = delete;
    ~PinnedBuffer() { Drop(p_, cap_); }
    // at least `bytes`; `keep` bytes of the old contents survive a reallocation
    void Reserve(size_t bytes, size_t keep = 0) {
        if (bytes <= cap_) return;
        const size_t want = bytes + bytes / 2 + 65536;   // pinned allocations take milliseconds: grow rarely
        size_t got = 0;
        void* q = PinnedPool::Get().Take(bytes, &got);
        if (!q) {
            if (pc_host_buffer_alloc(want, &q) != PC_OK) ThrowHip("pc_host_buffer_alloc");
            got = want;
        }
        if (keep) std::memcpy(q, p_, keep);
        Drop(p_, cap_);
        p_ = q;
        cap_ = got;
    }
    uint8_t* data() const { return static_cast<uint8_t*>(p_); }
    void swap(PinnedBuffer& o) {
        std::swap(p_, o.p_);
        std::swap(cap_, o.cap_);
    }

   private:
    static void Drop(void* p, size_t cap) {
        if (p && !PinnedPool::Get().Give(p, cap)) pc_host_buffer_free(p);
    }
    void* p_ = nullptr;
    size_t cap_ = 0;
};

size_t Align16(size_t v) { return (v + 15u) & ~static_cast<size_t>(15u); }

// The matches of every flow INTO one frame, one after the other in a page-locked block: per source the
// src_keypoints_indices (uint32) followed by the tgt_keypoints (2 floats), each 16-byte aligned.  This is the block
// pc_track_solve_frame sends to the GPU in one transfer.
struct MatchBlock {
    struct Flow {
        int32_t source = 0;
        size_t rows = 0, idx_offset = 0, tgt_offset = 0;
    };
    PinnedBuffer buffer;
    size_t bytes = 0;
    std::vector<Flow> flows;
    void Clear() {
        bytes = 0;
        flows.clear();
    }
    // reads the flow source -> target behind what the block already holds; false if the database has no such row
    bool Append(const Database& db, int32_t source, int32_t target) {
        Flow f;
        f.source = source;
        const bool found = db.ReadImagePairMatchesInto(
            source, target,
            [&](size_t rows, uint32_t** idx, float** tgt) {
                f.rows = rows;
                f.idx_offset = Align16(bytes);
                f.tgt_offset = Align16(f.idx_offset + rows * sizeof(uint32_t));
                const size_t end = f.tgt_offset + rows * sizeof(Keypoint);
                buffer.Reserve(end, bytes);
                *idx = reinterpret_cast<uint32_t*>(buffer.data() + f.idx_offset);
                *tgt = reinterpret_cast<float*>(buffer.data() + f.tgt_offset);
                bytes = end;
            },
            nullptr);
        if (found) flows.push_back(f);
        return found;
    }
    const Flow* Find(int32_t source) const {
        for (const Flow& f : flows)
            if (f.source == source) return &f;
        return nullptr;
    }
};

struct PinnedKeypoints {
    PinnedBuffer buffer;
    size_t rows = 0;
    const float* xy() const { return reinterpret_cast<const float*>(buffer.data()); }
    bool Read(const Database& db, int32_t frame) {
        rows = 0;
        return db.ReadKeypointsInto(
            frame,
            [&](size_t n) {
                buffer.Reserve(std::max<size_t>(n, 1) * sizeof(Keypoint));
                return reinterpret_cast<float*>(buffer.data());
            },
            &rows);
    }
};

// Reads the blobs the NEXT frame will need -- the matches of every flow into it from a source that has, or is about to
// get, a pose, and the keypoints of the frame being solved right now (the one new source) -- on a second read
// connection while the GPU works on the current frame, straight into page-locked memory.
class FlowPrefetcher {
   public:
    struct Batch {
        int32_t frame = 0;
        bool valid = false;
        MatchBlock matches;
        int32_t keypoints_frame = 0;
        bool has_keypoints = false;
        PinnedKeypoints keypoints;
    };

    explicit FlowPrefetcher(const std::string& path) {
        const char* env = std::getenv("POLYCHASE_TRACK_PREFETCH");
        if ((env && env[0] == '0') || path.empty() || path == ":memory:") return;
        try {
            db_ = std::make_unique<Database>(path);
        } catch (...) {
            return;   // no second connection: the tracker reads synchronously
        }
        thread_ = std::thread([this] { Run(); });
    }
    ~FlowPrefetcher() {
        if (!thread_.joinable()) return;
        {
            std::lock_guard<std::mutex> lk(mtx_);
            stop_ = true;
        }
        cv_.notify_all();
        thread_.join();
    }
    bool Enabled() const { return thread_.joinable(); }

    // ask for the blobs of `frame`: flows from `sources` into it, and the keypoints of `keypoints_frame`
    void Request(int32_t frame, std::vector<int32_t> sources, int32_t keypoints_frame) {
        if (!Enabled()) return;
        {
            std::lock_guard<std::mutex> lk(mtx_);
            req_frame_ = frame;
            req_sources_ = std::move(sources);
            req_keypoints_frame_ = keypoints_frame;
            has_request_ = true;
            done_ = false;
        }
        cv_.notify_all();
    }
    // the batch of the last request if it was for `frame` (waits for the reader), else nullptr.  The batch stays untouched
    // until two more requests have been made: the tracker holds one batch for the frame on the GPU and one for the frame it
    // is preparing while the reader fills the third.
    Batch* Take(int32_t frame) {
        if (!Enabled()) return nullptr;
        std::unique_lock<std::mutex> lk(mtx_);
        if (!has_request_ && !done_) return nullptr;
        cv_.wait(lk, [this] { return done_ || stop_; });
        Batch& b = batches_[ready_];
        return (done_ && b.valid && b.frame == frame) ? &b : nullptr;
    }

   private:
    void Run() {
        numa::PinThisThreadNearGpu(nullptr, "tracking: read-ahead thread");
        for (;;) {
            int32_t frame, kp_frame;
            std::vector<int32_t> sources;
            {
                std::unique_lock<std::mutex> lk(mtx_);
                cv_.wait(lk, [this] { return has_request_ || stop_; });
                if (stop_) return;
                frame = req_frame_;
                kp_frame = req_keypoints_frame_;
                sources = std::move(req_sources_);
                has_request_ = false;
            }
            Batch& b = batches_[(ready_ + 1) % kBatches];   // the consumer may still hold batches_[ready_] and the one before it
            b.frame = frame;
            b.valid = false;
            b.matches.Clear();
            b.has_keypoints = false;
            try {
                for (int32_t src : sources) b.matches.Append(*db_, src, frame);
                b.keypoints_frame = kp_frame;
                b.has_keypoints = b.keypoints.Read(*db_, kp_frame);
                b.valid = true;
            } catch (...) {
                b.valid = false;   // the consumer falls back to its own connection and reports the error there
            }
            {
                std::lock_guard<std::mutex> lk(mtx_);
                ready_ = (ready_ + 1) % kBatches;
                done_ = true;
            }
            cv_.notify_all();
        }
    }

    std::unique_ptr<Database> db_;
    std::thread thread_;
    std::mutex mtx_;
    std::condition_variable cv_;
    bool stop_ = false, has_request_ = false, done_ = false;
    int32_t req_frame_ = 0, req_keypoints_frame_ = 0;
    std::vector<int32_t> req_sources_;
    static constexpr int kBatches = 3;
    Batch batches_[kBatches];
    int ready_ = 0;
};

// Host-side state that outlives one frame: the device-resident correspondence set, the keypoints of recently used
// source frames (a frame is a source for up to 8 targets: read from SQLite once; page-locked, the GPU's copy of an array
// is made from it once and cached by frame id inside the set) and the match block of the frame when nothing was prefetched.
struct Scratch {
    pc_context* ctx = nullptr;
    pc_corr_set* set = nullptr;
    std::vector<int32_t> sources;
    struct CachedKeypoints {
        int32_t frame = 0;
        bool valid = false;
        uint64_t stamp = 0;
        PinnedKeypoints keypoints;
    };
    CachedKeypoints cache[16];
    uint64_t clock = 0;
    MatchBlock own_matches[3];   // in turn: the blocks of the (up to two) frames on the GPU stay untouched while the next one is read
    int own_turn = 0;
    unsigned long long last_end_tick = 0;   // the previous frame's hand-over on the GPU's clock (stage report)

    // ONE correspondence set per process is parked between runs (its copy stream, event, page-locked result words and the
    // device arrays a frame needs: 1.5-2 ms to create, more to grow to a clip's size).  A run that ended normally hands its
    // set over through pc_corr_set_recycle (waits for the set's streams, forgets the keypoint arrays cached by frame id --
    // the next run may read another database); a run that ended with an exception destroys it as before.
    struct ParkedSet {
        std::mutex m;
        pc_corr_set* set = nullptr;
    };
    static ParkedSet& Parked() {
        static ParkedSet* parked = new ParkedSet;   // never destroyed
        return *parked;
    }
    bool reusable = false;   // set by the run when it has ended normally

    Scratch() : ctx(SharedGpuContext()) {
        GpuSection section;
        if (TrackCacheEnabled()) {
            std::lock_guard<std::mutex> lk(Parked().m);
            set = Parked().set;
            Parked().set = nullptr;
        }
        if (!set && pc_corr_set_create(ctx, &set) != PC_OK) throw std::runtime_error(std::string("pc_corr_set_create: ") + pc_last_error());
    }
    Scratch(const Scratch&) = delete;
    Scratch& operator=(const Scratch&) = delete;
    ~Scratch() {
        GpuSection section;
        if (reusable && TrackCacheEnabled() && pc_corr_set_recycle(ctx, set) == PC_OK) {
            std::lock_guard<std::mutex> lk(Parked().m);
            if (!Parked().set) {
                Parked().set = set;
                set = nullptr;
            }
        }
        pc_corr_set_destroy(set);
    }
    static void ReleaseParked() {
        pc_corr_set* drop = nullptr;
        {
            std::lock_guard<std::mutex> lk(Parked().m);
            drop = Parked().set;
            Parked().set = nullptr;
        }
        if (drop) {
            GpuSection section;
            pc_corr_set_destroy(drop);
        }
    }

    const PinnedKeypoints& KeypointsOf(const Database& db, int32_t frame, FlowPrefetcher::Batch* batch) {
        CachedKeypoints* slot = &cache[0];
        for (auto& c : cache) {
            if (c.valid && c.frame == frame) {
                c.stamp = ++clock;
                return c.keypoints;
            }
            if (c.stamp < slot->stamp) slot = &c;
        }
        if (batch && batch->has_keypoints && batch->keypoints_frame == frame) {
            slot->keypoints.buffer.swap(batch->keypoints.buffer);   // read ahead by the prefetcher
            slot->keypoints.rows = batch->keypoints.rows;
            batch->has_keypoints = false;
        } else {
            slot->keypoints.Read(db, frame);   // a frame without a keypoints row has none (rows = 0)
        }
        slot->frame = frame;
        slot->valid = true;
        slot->stamp = ++clock;
        return slot->keypoints;
    }
};

void SourceCamera(const CameraState& source_camera, const Mat4f& model_matrix, pc_ray_camera* cam) {
    SceneTransformations scene;
    scene.model_matrix = model_matrix;
    scene.view_matrix = source_camera.pose.Rt4x4();
    scene.intrinsics = source_camera.intrinsics;
    MakeRayCamera(scene, cam);
}

const float kNoKeypoint[2] = {0.f, 0.f};

// "The solution should be very close to the previous/next pose" (tracker.cc:111-119)
CameraState InitialGuess(const CameraTrajectory& traj, int32_t frame) {
    for (int32_t candidate : {frame, frame - 1, frame + 1})
        if (traj.IsFrameFilled(candidate)) return *traj.Get(candidate);
    return CameraState{};
}

// the flows into `frame` whose source already has a pose (:43-50), in the database's order; their matches in `block`
// (the prefetched one, or read now)
// `pending`: frames whose solve is in flight count as filled (their poses are there -- on the host or, for the launch directly in
// front, on the device -- before anything that uses them runs)
struct PendingFrames {
    int n = 0;
    int32_t frame[2] = {0, 0};
    bool Has(int32_t f) const { return (n > 0 && frame[0] == f) || (n > 1 && frame[1] == f); }
};
const MatchBlock& GatherMatches(const Database& db, const CameraTrajectory& traj, int32_t frame, Scratch& s,
                                FlowPrefetcher::Batch* batch, std::vector<int32_t>& used_sources, const PendingFrames& pending = PendingFrames{}) {
    StageClock::Scope sc("track/db read");
    s.sources.clear();
    db.FindOpticalFlowsToImage(frame, s.sources);
    used_sources.clear();
    for (int32_t source : s.sources) {
        CHECK_NE(source, frame);
        if (traj.IsFrameFilled(source) || pending.Has(source)) used_sources.push_back(source);   // only frames that already have a pose (:48)
    }
    if (batch) {
        bool complete = true;
        for (int32_t source : used_sources) complete = complete && batch->matches.Find(source) != nullptr;
        if (complete) return batch->matches;
    }
    MatchBlock& own = s.own_matches[s.own_turn = (s.own_turn + 1) % 3];
    own.Clear();
    for (int32_t source : used_sources) own.Append(db, source, frame);
    return own;
}

// ---- the cross-check path (POLYCHASE_TRACK_FUSED=0) and FrameCorrespondences: one pc_corr_set_append per source frame
// (gather, ray cast, model transform, ordered compaction: tracker.cc:52-92), then pc_pnp_solve ----
int GatherCorrespondences(const Database& db, const CameraTrajectory& traj, const Mat4f& model_matrix, int32_t frame,
                          const AcceleratedMesh& mesh, Scratch& s, FlowPrefetcher::Batch* batch = nullptr) {
    if (pc_corr_set_clear(s.ctx, s.set) != PC_OK) ThrowHip("pc_corr_set_clear");
    std::vector<int32_t> used;
    const MatchBlock& block = GatherMatches(db, traj, frame, s, batch, used);
    if (!used.empty()) mesh.SyncMask();   // the mask can be edited between frames through inner_mut(): send the current bits
    for (int32_t source : used) {
        const MatchBlock::Flow* f = block.Find(source);
        if (!f || f->rows == 0) continue;
        const PinnedKeypoints& kps = s.KeypointsOf(db, source, batch);
        StageClock::Scope sc("track/append (enqueue)");
        pc_ray_camera cam;
        SourceCamera(*traj.Get(source), model_matrix, &cam);
        if (pc_corr_set_append(s.ctx, s.set, mesh.Gpu(), &cam, model_matrix.data(), source, kps.rows ? kps.xy() : kNoKeypoint,
                               static_cast<int>(kps.rows), reinterpret_cast<const uint32_t*>(block.buffer.data() + f->idx_offset),
                               reinterpret_cast<const float*>(block.buffer.data() + f->tgt_offset), static_cast<int>(f->rows),
                               /*check_mask=*/1) != PC_OK)
            ThrowHip("pc_corr_set_append");
    }
    int n = 0;
    {
        StageClock::Scope sc("track/wait for the appends");
        if (pc_corr_set_size(s.ctx, s.set, &n) != PC_OK) {
            // an index past the source's keypoints: the reference's CHECK_LT (tracker.cc:61)
            CHECK(std::string(pc_last_error()).find("out of range") == std::string::npos);
            ThrowHip("pc_corr_set_size");
        }
    }
    return n;
}

std::optional<PnPResult> SolveFrameUnfused(const Database& db, const CameraTrajectory& traj, const Mat4f& model_matrix,
                                           int32_t frame, const AcceleratedMesh& mesh, const PnPOptions& pnp_opts, Scratch& s,
                                           FlowPrefetcher::Batch* batch) {
    GpuSection section;
    const int n = GatherCorrespondences(db, traj, model_matrix, frame, mesh, s, batch);
    if (n < 3) return std::nullopt;  // :95-97
    PnPResult result;
    result.camera = InitialGuess(traj, frame);
    {
        StageClock::Scope sc("track/pnp");
        pc_pnp_problem* prob = nullptr;
        if (pc_pnp_problem_from_set(s.ctx, s.set, &prob) != PC_OK) ThrowHip("pc_pnp_problem_from_set");
        struct Guard {
            pc_pnp_problem* p;
            ~Guard() { pc_pnp_problem_destroy(p); }
        } guard{prob};
        SolvePnPIterativeOnGpu(prob, static_cast<size_t>(n), pnp_opts, result);
    }
    return result;
}

// ---- the product path: SolveFrame (tracker.cc:36-131) = one transfer of the frame's matches, one ray-cast launch over all
// sources, the whole LM loop as one persistent launch, one wait (pc_track_frame_upload / _launch / _finish).  In three steps,
// because only the LAUNCH needs the pose of the frame before: while the GPU solves frame f the host plans frame f + 1 (which
// flows, where their blobs are) and its matches travel to the GPU on the copy stream; when the pose of f arrives the launches
// of f + 1 are enqueued at once, and the caller's callback for f runs beside them. ----
struct CoResidencyLost {};   // pc_track_frame_finish: the persistent launch's workgroups did not all become resident in time

class FrameSolver {
   public:
    FrameSolver(const Database& db, const CameraTrajectory& traj, const Mat4f& model_matrix, const AcceleratedMesh& mesh,
                const PnPOptions& opts, Scratch& scratch)
        : db_(db), traj_(traj), model_(model_matrix), mesh_(mesh), opts_(opts), s_(scratch) {
        const int lt = static_cast<int>(opts.bundle_opts.loss_type);
        if (lt < 0 || lt > 2) throw std::runtime_error("Unknown loss type: " + std::to_string(lt));
    }
    ~FrameSolver() {
        if (!launched_) return;
        GpuSection section;   // an exception on the way: nothing stays in flight on the shared context
        pc_track_solve_result unused;
        (void)pc_track_frame_finish(s_.ctx, s_.set, &unused);
    }

    // host-only part + transfer: which flows end in `frame`, where their matches and the sources' keypoints are
    void Plan(int32_t frame, FlowPrefetcher::Batch* batch, const PendingFrames& pending) {
        frame_ = frame;
        n_sources_ = 0;
        too_many_ = false;
        std::vector<int32_t>& used = used_;
        const MatchBlock& block = GatherMatches(db_, traj_, frame, s_, batch, used, pending);
        for (int32_t source : used) {
            const MatchBlock::Flow* f = block.Find(source);
            if (!f || f->rows == 0) continue;
            if (n_sources_ == 8) {
                // More flows into a frame than the skips of cpp/opticalflow.cc:76-77 make (a database written with another skip
                // set): SolveFrame takes any number (tracker.cc:43-50) -- this frame and the ones after it go to the per-source
                // building blocks, which do too (this used to be a CHECK).
                too_many_ = true;
                n_sources_ = 0;
                return;
            }
            const PinnedKeypoints& kps = s_.KeypointsOf(db_, source, batch);
            source_frames_[n_sources_] = source;
            pc_track_source& q = sources_[n_sources_++];
            std::memset(&q.cam, 0, sizeof(q.cam));
            q.keypoints_key = source;
            q.keypoints_xy = kps.rows ? kps.xy() : kNoKeypoint;
            q.n_keypoints = static_cast<int>(kps.rows);
            q.n_matches = static_cast<int>(f->rows);
            q.idx_offset = f->idx_offset;
            q.tgt_offset = f->tgt_offset;
        }
        if (n_sources_ == 0) return;
        StageClock::Scope sc("track/upload (enqueue)");
        GpuSection section;
        if (pc_track_frame_upload(s_.ctx, s_.set, block.buffer.data(), block.bytes, sources_, n_sources_) != PC_OK)
            ThrowHip("pc_track_frame_upload");
    }

    // The sources' poses are all there now -- on the host, or (`prev`: the solver whose launches were enqueued just before, still
    // in flight) on the device: cameras, initial guess (tracker.cc:111-119), the two launches.  With `prev` the intrinsics must
    // not be under optimisation (the caller checks): the camera `prev` will report then has prev's own initial intrinsics.
    void Launch(const FrameSolver* prev = nullptr) {
        if (n_sources_ == 0) return;
        StageClock::Scope sc("track/launch (enqueue)");
        int chained_source = -1;
        for (int k = 0; k < n_sources_; k++) {
            if (prev && source_frames_[k] == prev->frame_) chained_source = k;
            else SourceCamera(*traj_.Get(source_frames_[k]), model_, &sources_[k].cam);
        }
        // "The solution should be very close to the previous/next pose" (:111-119): this frame's own pose, else frame - 1's, else
        // frame + 1's -- a frame in flight counts as filled
        bool chain_initial = false;
        guess_ = CameraState{};
        for (int32_t candidate : {frame_, frame_ - 1, frame_ + 1}) {
            if (prev && candidate == prev->frame_) {
                chain_initial = true;
                guess_ = prev->guess_;   // its intrinsics (not optimised: the result's too); the pose is on the device
                break;
            }
            if (traj_.IsFrameFilled(candidate)) {
                guess_ = *traj_.Get(candidate);
                break;
            }
        }
        const BundleOptions& bo = opts_.bundle_opts;
        const CameraIntrinsics::Bounds bounds = guess_.intrinsics.GetBounds();
        pc_pnp_camera init;
        init.q_xyzw[0] = guess_.pose.q.x;
        init.q_xyzw[1] = guess_.pose.q.y;
        init.q_xyzw[2] = guess_.pose.q.z;
        init.q_xyzw[3] = guess_.pose.q.w;
        for (int i = 0; i < 3; i++) init.t[i] = guess_.pose.t[i];
        init.fx = guess_.intrinsics.fx;
        init.fy = guess_.intrinsics.fy;
        init.cx = guess_.intrinsics.cx;
        init.cy = guess_.intrinsics.cy;
        init.aspect_ratio = guess_.intrinsics.aspect_ratio;
        init.convention_opencv = guess_.intrinsics.convention == CameraConvention::OpenCV ? 1 : 0;
        pc_pnp_solve_options so;
        so.max_iterations = static_cast<int>(bo.max_iterations);
        so.initial_lambda = bo.initial_lambda;
        so.min_lambda = bo.min_lambda;
        so.max_lambda = bo.max_lambda;
        so.gradient_tol = bo.gradient_tol;
        so.step_tol = bo.step_tol;
        so.loss_type = static_cast<int>(bo.loss_type);
        so.loss_scale = bo.loss_scale;
        so.optimize_focal_length = opts_.optimize_focal_length ? 1 : 0;        // "with more than 3 points" is decided on the device
        so.optimize_principal_point = opts_.optimize_principal_point ? 1 : 0;
        so.f_low = bounds.f_low;
        so.f_high = bounds.f_high;
        so.cx_low = bounds.cx_low;
        so.cx_high = bounds.cx_high;
        so.cy_low = bounds.cy_low;
        so.cy_high = bounds.cy_high;
        so.max_inlier_error = opts_.max_inlier_error;
        so.rounds_hint = 0;
        GpuSection section;
        mesh_.SyncMask();   // the mask can be edited between frames through inner_mut(): the current bits (sent when they changed)
        if (pc_track_frame_launch_chained(s_.ctx, s_.set, mesh_.Gpu(), model_.data(), /*check_mask=*/1, sources_, n_sources_, chained_source,
                                          &init, chain_initial ? 1 : 0, &so) != PC_OK)
            ThrowHip("pc_track_frame_launch");
        launched_ = true;
    }

    // waits for the frame launched last; nullopt: fewer than 3 correspondences (:95-97)
    std::optional<PnPResult> Finish() {
        if (!launched_) return std::nullopt;
        launched_ = false;
        pc_track_solve_result sr;
        {
            StageClock::Scope sc("track/wait for the GPU");
            GpuSection section;
            if (pc_track_frame_finish(s_.ctx, s_.set, &sr) != PC_OK) {
                const std::string why = pc_last_error();
                if (why.find("become resident") != std::string::npos) throw CoResidencyLost{};
                // an index past the source's keypoints: the reference's CHECK_LT (tracker.cc:61)
                CHECK(why.find("out of range") == std::string::npos);
                ThrowHip("pc_track_frame_finish");
            }
        }
        {
            static const char* kPhase[8] = {"track/lm kernel: sweep + publish", "track/lm kernel: wait for workgroups", "track/lm kernel: add partials",
                                            "track/lm kernel: decision", "track/lm kernel: publish decision", "track/lm kernel: fetch parameters",
                                            "track/lm kernel: inlier pass", "track/lm kernel: whole launch"};
            for (int k = 0; k < 8; k++) StageClock::Add(kPhase[k], sr.lm_ticks[k] * 1e-5);   // 100 MHz ticks -> ms
            // from the result of one frame to the first instruction of the next frame's LM kernel, on the GPU's clock: the turn-
            // around through the host (poll, pose, enqueue) + launch latency + the ray-cast kernel
            if (s_.last_end_tick && sr.lm_begin_tick > s_.last_end_tick)
                StageClock::Add("track/gpu: result of a frame -> LM kernel of the next", static_cast<double>(sr.lm_begin_tick - s_.last_end_tick) * 1e-5);
            s_.last_end_tick = sr.lm_end_tick;
            StageClock::Add("track/lm kernel: rounds (count, not ms)", sr.rounds);
            StageClock::Add("track/matches (count, not ms)", sr.n_matches);
            StageClock::Add("track/correspondences (count, not ms)", sr.n_correspondences);
        }
        if (sr.n_correspondences < 3) return std::nullopt;  // :95-97
        PnPResult result;
        result.camera = guess_;
        CameraState& c = result.camera;
        c.pose.q.x = sr.pnp.camera.q_xyzw[0];
        c.pose.q.y = sr.pnp.camera.q_xyzw[1];
        c.pose.q.z = sr.pnp.camera.q_xyzw[2];
        c.pose.q.w = sr.pnp.camera.q_xyzw[3];
        for (int i = 0; i < 3; i++) c.pose.t[i] = sr.pnp.camera.t[i];
        c.intrinsics.fx = sr.pnp.camera.fx;
        c.intrinsics.fy = sr.pnp.camera.fy;
        c.intrinsics.cx = sr.pnp.camera.cx;
        c.intrinsics.cy = sr.pnp.camera.cy;
        BundleStats st;
        st.iterations = static_cast<size_t>(sr.pnp.iterations);
        st.invalid_steps = static_cast<size_t>(sr.pnp.invalid_steps);
        st.initial_cost = sr.pnp.initial_cost;
        st.cost = sr.pnp.cost;
        st.lambda = sr.pnp.lambda;
        st.step_norm = sr.pnp.step_norm;
        st.grad_norm = sr.pnp.grad_norm;
        result.bundle_stats = st;
        result.inlier_ratio = static_cast<Float>(sr.pnp.inliers) / static_cast<Float>(sr.n_correspondences);
        return result;
    }
    int32_t frame() const { return frame_; }
    bool too_many_sources() const { return too_many_; }
    // waits for whatever this solver has in flight and forgets it
    void Drain() {
        if (!launched_) return;
        launched_ = false;
        GpuSection section;
        pc_track_solve_result unused;
        (void)pc_track_frame_finish(s_.ctx, s_.set, &unused);
    }

   private:
    const Database& db_;
    const CameraTrajectory& traj_;
    const Mat4f& model_;
    const AcceleratedMesh& mesh_;
    const PnPOptions& opts_;
    Scratch& s_;
    int32_t frame_ = 0;
    int n_sources_ = 0;
    pc_track_source sources_[8];
    int32_t source_frames_[8];
    std::vector<int32_t> used_;
    CameraState guess_;
    bool launched_ = false;
    bool too_many_ = false;
};

// Process-wide memory of a lost co-residency: the persistent LM launch needs every workgroup on the GPU at once; when a
// run has given up on it twice (the launch, then its one retry -- 100 ms each), the runs of the next
// POLYCHASE_TRACK_FUSED_BACKOFF_S seconds (30) start on the per-source building blocks at once instead of stalling again: Blender
// calls TrackSequence once per user action, and whatever held the GPU a moment ago (a render) is probably still there.
std::atomic<long long> g_fused_lost_until_ms{0};
long long SteadyNowMs() {
    return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
void RememberCoResidencyLost() {
    const char* env = std::getenv("POLYCHASE_TRACK_FUSED_BACKOFF_S");
    const double seconds = env ? std::atof(env) : 30.0;
    g_fused_lost_until_ms.store(SteadyNowMs() + static_cast<long long>(seconds * 1000.0), std::memory_order_relaxed);
}

bool FusedSolve() {
    const char* env = std::getenv("POLYCHASE_TRACK_FUSED");   // read per call: the tests flip it
    if (env && env[0] == '0') return false;
    return SteadyNowMs() >= g_fused_lost_until_ms.load(std::memory_order_relaxed);
}

}  // namespace

void FrameCorrespondences(const Database& database, const CameraTrajectory& camera_traj, const Mat4f& model_matrix,
                          int32_t frame, const AcceleratedMesh& accel_mesh, std::vector<float>& world_points,
                          std::vector<float>& image_points) {
    GpuSection section;
    Scratch scratch;
    const int n = GatherCorrespondences(database, camera_traj, model_matrix, frame, accel_mesh, scratch);
    world_points.assign(3 * static_cast<size_t>(n), 0.f);
    image_points.assign(2 * static_cast<size_t>(n), 0.f);
    if (pc_corr_set_download(scratch.ctx, scratch.set, world_points.data(), image_points.data()) != PC_OK)
        ThrowHip("pc_corr_set_download");
}

void TrackCameraTrajectory(const Database& database, CameraTrajectory& camera_traj, int32_t frame_from,
                           int32_t frame_to_inclusive, const Mat4f& model_matrix, const AcceleratedMesh& accel_mesh,
                           TrackingCallback callback, bool optimize_focal_length, bool optimize_principal_point,
                           const BundleOptions& opts) {
    CHECK(camera_traj.IsValidFrame(std::min(frame_from, frame_to_inclusive)));
    CHECK(camera_traj.IsValidFrame(std::max(frame_from, frame_to_inclusive)));
    CHECK(camera_traj.IsFrameFilled(frame_from));
    PnPOptions pnp_opts;
    pnp_opts.bundle_opts = opts;
    pnp_opts.max_inlier_error = kMaxInlierError;
    pnp_opts.optimize_focal_length = optimize_focal_length;
    pnp_opts.optimize_principal_point = optimize_principal_point;

    StageClock::Begin();
    numa::ScopedPin near_gpu(SharedGpuContext(), "tracking: calling thread (polls the result words)");   // numa_pin.h
    const auto t_entry = std::chrono::steady_clock::now();
    const int32_t step = frame_from < frame_to_inclusive ? 1 : -1;
    // (in this order: the scratch -- whose destructor waits for every stream of the correspondence set, the copy stream included --
    // goes first, the prefetcher with the page-locked blocks those copies read from after it)
    auto since_entry = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_entry).count(); };
    FlowPrefetcher prefetcher(database.Path());
    StageClock::Add("track/start-up: read-ahead thread and connection at (ms)", since_entry());
    Scratch scratch;
    StageClock::Add("track/start-up: correspondence set at (ms)", since_entry());
    std::vector<int32_t> next_sources, wanted;
    // what frame `next` will need, read ahead: flows from the frames that have a pose by then -- filled now, or tracked
    // before it in this run -- and the keypoints of the frame tracked just before it (the one array no cache holds yet)
    auto request = [&](int32_t next) {
        if (!prefetcher.Enabled() || next == frame_to_inclusive + step) return;
        next_sources.clear();
        wanted.clear();
        database.FindOpticalFlowsToImage(next, next_sources);
        for (int32_t src : next_sources) {
            const bool tracked_before = step > 0 ? (src > frame_from && src < next) : (src < frame_from && src > next);
            if (tracked_before || camera_traj.IsFrameFilled(src)) wanted.push_back(src);
        }
        prefetcher.Request(next, wanted, next - step);
    };
    auto report_and_store = [&](int32_t frame, const PnPResult& solved) {
        if (callback) {
            FrameTrackingResult report;
            report.frame = frame;
            report.pose = solved.camera.pose;
            report.intrinsics = solved.camera.intrinsics;
            report.bundle_stats = solved.bundle_stats;
            report.inlier_ratio = solved.inlier_ratio;
            return callback(report);
        }
        return true;
    };
    const int32_t first = frame_from + step, end = frame_to_inclusive + step;
    if (!FusedSolve()) {
        // the cross-check: round 4's path, one frame after the other
        request(first);
        for (int32_t frame = first; frame != end; frame += step) {
            FlowPrefetcher::Batch* batch = prefetcher.Take(frame);
            request(frame + step);
            const std::optional<PnPResult> solved = SolveFrameUnfused(database, camera_traj, model_matrix, frame, accel_mesh, pnp_opts, scratch, batch);
            if (!solved) throw std::runtime_error("Could not track to frame: " + std::to_string(frame) + ". Not enough features.");
            if (!report_and_store(frame, *solved)) {   // the pose of a frame the user stopped at is not stored (:179-186)
                scratch.reusable = true;
                return;
            }
            camera_traj.Set(frame, solved->camera);
        }
        scratch.reusable = true;
        StageClock::Report("TrackCameraTrajectory");
        return;
    }
    if (first != end) {
        // Three solvers take turns.  With `lead` = 2 (the default whenever the intrinsics are not being optimised) the frame
        // AFTER the one on the GPU is always enqueued behind it: its launches take the pose of the frame in front -- one of its
        // source cameras and its initial guess -- from where that frame's LM launch leaves it on the device
        // (pc_track_frame_launch_chained), so the GPU goes from frame to frame without waiting for the host to see a pose
        // (round 5: 56-116 us of turn-around per frame, a third of a frame's time on a slow host).  While the GPU solves `frame`
        // and holds `frame + step`, the host plans `frame + 2 step` (which flows, where their blobs are; the upload) and, when
        // the pose of `frame` arrives, enqueues it.  lead = 1 (POLYCHASE_TRACK_CHAIN=0, or intrinsics under optimisation: the
        // bounds of a frame's solve derive from the intrinsics the frame before it ENDED with): round 5's schedule.
        FrameSolver solvers[3] = {FrameSolver(database, camera_traj, model_matrix, accel_mesh, pnp_opts, scratch),
                                  FrameSolver(database, camera_traj, model_matrix, accel_mesh, pnp_opts, scratch),
                                  FrameSolver(database, camera_traj, model_matrix, accel_mesh, pnp_opts, scratch)};
        const char* chain_env = std::getenv("POLYCHASE_TRACK_CHAIN");   // read per run
        const bool chain = !(chain_env && chain_env[0] == '0') && !optimize_focal_length && !optimize_principal_point;
        const int lead = chain ? 2 : 1;
        bool fused_lost = false, too_many = false;
        const char* lose_env = std::getenv("POLYCHASE_TRACK_TEST_LOSE_AT");   // read per run
        const int lose_at = lose_env ? std::atoi(lose_env) : 0;
        const char* twice_env = std::getenv("POLYCHASE_TRACK_TEST_LOSE_TWICE");
        const bool lose_twice = twice_env && twice_env[0] == '1';
        int finished = 0;
        // solver of frame number k of this run (0 = first): k % 3
        auto solver_of = [&](int32_t f) -> FrameSolver& { return solvers[((f - first) * step) % 3]; };
        auto in_range = [&](int32_t f) { return step > 0 ? (f >= first && f < end) : (f <= first && f > end); };
        // frames [launched_to - ..., launched_to) are in flight; `launched_to` = the next frame to enqueue
        int32_t launched_to = first;
        // plans `f` (the frames in flight count as filled) -- false: more sources than the fused path holds
        auto plan = [&](int32_t f, FlowPrefetcher::Batch* batch, int32_t oldest_in_flight) {
            PendingFrames pending;
            for (int32_t q = oldest_in_flight; q != f; q += step) pending.frame[pending.n++] = q;
            solver_of(f).Plan(f, batch, pending);
            return !solver_of(f).too_many_sources();
        };
        request(first);
        {
            FlowPrefetcher::Batch* batch = prefetcher.Take(first);
            request(first + step);
            StageClock::Add("track/start-up: first batch read at (ms)", since_entry());
            if (!plan(first, batch, first)) too_many = true;
            StageClock::Add("track/start-up: first frame uploaded at (ms)", since_entry());
            if (!too_many) {
                solver_of(first).Launch();
                launched_to = first + step;
            }
            StageClock::Add("track/start-up: first frame launched at (ms)", since_entry());
            // the second frame behind it, chained
            if (!too_many && lead == 2 && in_range(launched_to)) {
                FlowPrefetcher::Batch* b2 = prefetcher.Take(launched_to);
                request(launched_to + step);
                if (plan(launched_to, b2, first)) {
                    solver_of(launched_to).Launch(&solver_of(first));
                    launched_to += step;
                } else {
                    too_many = true;
                }
            }
        }
        // the frames from `f` on with the per-source building blocks (no co-residency requirement, any number of sources)
        auto run_unfused_from = [&](int32_t f) {
            for (; f != end; f += step) {
                const std::optional<PnPResult> r = SolveFrameUnfused(database, camera_traj, model_matrix, f, accel_mesh, pnp_opts, scratch, nullptr);
                if (!r) throw std::runtime_error("Could not track to frame: " + std::to_string(f) + ". Not enough features.");
                if (!report_and_store(f, *r)) return false;
                camera_traj.Set(f, r->camera);
            }
            return true;
        };
        auto drain_all = [&]() {
            for (auto& sv : solvers) sv.Drain();   // (pc_track_frame_finish returns the oldest first whoever asks: results are dropped)
        };
        if (too_many && launched_to == first) {
            if (!run_unfused_from(first)) return;
        } else {
            for (int32_t frame = first; frame != end; frame += step) {
                FrameSolver& now = solver_of(frame);
                // while the GPU works: plan the frame that goes behind what is in flight
                const int32_t target = launched_to;
                bool planned = false;
                if (!too_many && in_range(target)) {
                    FlowPrefetcher::Batch* batch = prefetcher.Take(target);
                    request(target + step);
                    planned = plan(target, batch, frame);
                    if (!planned) too_many = true;
                }
                std::optional<PnPResult> solved;
                auto finish = [&](bool retry) {
                    solved = now.Finish();
                    // POLYCHASE_TRACK_TEST_LOSE_AT=N: the N-th finished frame behaves as if its launch had timed out;
                    // POLYCHASE_TRACK_TEST_LOSE_TWICE=1: and so does its retry (testing aids)
                    if (!retry && lose_at > 0 && ++finished == lose_at) throw CoResidencyLost{};
                    if (retry && lose_twice) throw CoResidencyLost{};
                };
                try {
                    finish(false);
                } catch (const CoResidencyLost&) {
                    // The persistent LM launch needs all its workgroups on the GPU at once; another tenant of the GPU (a render in
                    // the same Blender, an analysis in the same host) can keep that from happening, and the launch then gives the
                    // CUs back after its time limit (100 ms).  Whatever is queued behind it ran on what it left: dropped.  ONE
                    // retry of the frame; if that is lost too, the frame and the rest of this run are solved with the per-source
                    // building blocks (no such requirement), and the next runs start there for a while (RememberCoResidencyLost).
                    StageClock::Add("track/co-residency lost (count, not ms)", 1);
                    drain_all();
                    launched_to = frame;
                    planned = false;
                    try {
                        if (!plan(frame, nullptr, frame)) throw CoResidencyLost{};
                        now.Launch();
                        launched_to = frame + step;
                        finish(true);
                    } catch (const CoResidencyLost&) {
                        drain_all();
                        fused_lost = true;
                        RememberCoResidencyLost();
                    }
                }
                if (fused_lost) {
                    if (!run_unfused_from(frame)) return;
                    break;
                }
                if (!solved) throw std::runtime_error("Could not track to frame: " + std::to_string(frame) + ". Not enough features.");
                if (frame == first)   // connections, page-locked blocks, the device's arrays, the first batch read: paid once per call
                    StageClock::Add("track/from the call to the first pose", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_entry).count());
                // the pose goes in at once -- the next launches need it -- and comes out again if the callback stops the run:
                // the pose of a frame the user stopped at is not stored (:179-186)
                const std::optional<CameraState> before = camera_traj.Get(frame);
                camera_traj.Set(frame, solved->camera);
                struct Restore {
                    CameraTrajectory& traj;
                    int32_t frame;
                    const std::optional<CameraState>& before;
                    bool armed = true;
                    ~Restore() {
                        if (!armed) return;
                        if (before) traj.Set(frame, *before);
                        else traj.Clear(frame);
                    }
                } restore{camera_traj, frame, before};
                // enqueue what was planned: behind frame + step if that one is in flight (then its pose comes from the device)
                if (planned) {
                    solver_of(target).Launch(target != frame + step ? &solver_of(target - step) : nullptr);
                    launched_to += step;
                }
                // after a retry nothing is planned and nothing is in flight: the pipeline is filled again (synchronous reads)
                while (!too_many && in_range(launched_to) && (launched_to - (frame + step)) * step < lead) {
                    const int32_t f = launched_to;
                    if (!plan(f, nullptr, frame + step)) {
                        too_many = true;
                        break;
                    }
                    solver_of(f).Launch(f != frame + step ? &solver_of(f - step) : nullptr);
                    launched_to += step;
                }
                if (!report_and_store(frame, *solved)) {   // (the solvers' destructors wait for the launches that are no longer wanted)
                    scratch.reusable = true;
                    drain_all();
                    return;
                }
                restore.armed = false;
                if (too_many && launched_to == frame + step) {   // nothing in flight any more: the rest on the building blocks
                    if (!run_unfused_from(frame + step)) return;
                    break;
                }
            }
        }
    }
    scratch.reusable = true;   // ended normally: the correspondence set is parked for the next run (Scratch)
    StageClock::Report("TrackCameraTrajectory");
}

namespace {
std::atomic<bool> g_warmed{false};
}

void WarmTrackerCaches() {
    const char* env = std::getenv("POLYCHASE_TRACK_WARM");
    if (!TrackCacheEnabled() || (env && env[0] == '0')) return;
    if (g_warmed.exchange(true)) return;
    try {
        {
            // sizes: a 1080p clip's frame has ~40 k keypoints and ~150 k matches into it (8 flows); everything grows on demand
            Scratch scratch;   // takes the parked set if there is one (then this only tops it up), parks it again on the way out
            GpuSection section;
            if (pc_corr_set_reserve(scratch.ctx, scratch.set, 256 * 1024, 64 * 1024, 12) == PC_OK) scratch.reusable = true;
        }
        // page-locked blocks: the match blocks of the read-ahead batches and of the synchronous reads, the keypoint arrays
        for (int k = 0; k < 6; k++) {
            void* p = nullptr;
            if (pc_host_buffer_alloc(size_t(4) << 20, &p) != PC_OK) break;
            if (!PinnedPool::Get().Give(p, size_t(4) << 20)) pc_host_buffer_free(p);
        }
        for (int k = 0; k < 20; k++) {
            void* p = nullptr;
            if (pc_host_buffer_alloc(size_t(512) << 10, &p) != PC_OK) break;
            if (!PinnedPool::Get().Give(p, size_t(512) << 10)) pc_host_buffer_free(p);
        }
    } catch (...) {
        // no device, no memory: the first TrackSequence call reports it
    }
}

void ReleaseTrackerCaches() {
    g_warmed.store(false, std::memory_order_relaxed);
    g_fused_lost_until_ms.store(0, std::memory_order_relaxed);
    Scratch::ReleaseParked();
    PinnedPool::Get().Clear();
}

void TrackSequence(const std::string& database_path, int32_t frame_from, int32_t frame_to_inclusive,
                   const SceneTransformations& scene_transform, const AcceleratedMesh& accel_mesh,
                   TrackingCallback callback, bool optimize_focal_length, bool optimize_principal_point,
                   BundleOptions opts) {
    const Database database{database_path};
    const int32_t first = std::min(frame_from, frame_to_inclusive);
    const size_t count = static_cast<size_t>(std::abs(frame_to_inclusive - frame_from)) + 1;
    CameraTrajectory trajectory{first, count};
    trajectory.Set(frame_from, CameraState{scene_transform.intrinsics, Pose::FromRt(scene_transform.view_matrix)});
    TrackCameraTrajectory(database, trajectory, frame_from, frame_to_inclusive, scene_transform.model_matrix, accel_mesh,
                          std::move(callback), optimize_focal_length, optimize_principal_point, opts);
}
