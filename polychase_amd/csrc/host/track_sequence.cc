// track_sequence.cc -- reference cpp/tracker.cc:36-213 on the MI355X path.
//
// For the frame being solved, every flow INTO it whose source frame already has a pose contributes
// 3D-2D correspondences: source keypoints are cast onto the mesh under the source camera
// (tracker.cc:64-78; one batched GPU launch per source here instead of one Embree call per match),
// the hits are moved to world space (:80-82) and paired with the tracked positions (:86); PnP then
// starts from this/previous/next frame's pose (:111-119).  LM residual sweeps run on the GPU
// (pnp.cc).  Sequential over frames by construction: frame k needs the poses solved before it.
#include "track_sequence.h"

#include <condition_variable>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <thread>

#include "gpu_context.h"
#include "pnp.h"
#include "stage_clock.h"

namespace {

constexpr float kMaxInlierError = 12.0f;  // tracker.cc:123 ("FIXME: Make this customizable")

// Reads the blobs the NEXT frame will need -- the matches of every flow into it from a source that has, or is about to
// get, a pose, and the keypoints of the frame being solved right now (the one new source) -- on a second read
// connection while the GPU works on the current frame.  SQLite reads were a third of a frame's time.
class FlowPrefetcher {
   public:
    struct Flow {
        int32_t source = 0;
        KeypointsIndices indices;
        Keypoints targets;
    };
    struct Batch {
        int32_t frame = 0;
        bool valid = false;
        std::vector<Flow> flows;     // entries [0, n_flows) are meaningful (the vectors are recycled)
        size_t n_flows = 0;
        int32_t keypoints_frame = 0;
        bool has_keypoints = false;
        Keypoints keypoints;
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
    // the batch of the last request if it was for `frame` (waits for the reader), else nullptr
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
            Batch& b = batches_[1 - ready_];   // the consumer may still be reading batches_[ready_]
            b.frame = frame;
            b.valid = false;
            b.n_flows = 0;
            b.has_keypoints = false;
            try {
                for (int32_t src : sources) {
                    if (b.flows.size() <= b.n_flows) b.flows.emplace_back();
                    Flow& f = b.flows[b.n_flows++];
                    f.source = src;
                    db_->ReadImagePairMatches(src, frame, f.indices, f.targets);
                }
                b.keypoints_frame = kp_frame;
                b.keypoints.clear();
                db_->ReadKeypoints(kp_frame, b.keypoints);
                b.has_keypoints = true;
                b.valid = true;
            } catch (...) {
                b.valid = false;   // the consumer falls back to its own connection and reports the error there
            }
            {
                std::lock_guard<std::mutex> lk(mtx_);
                ready_ = 1 - ready_;
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
    Batch batches_[2];
    int ready_ = 0;
};

// Host-side state that outlives one frame: the device-resident correspondence set, the keypoints of recently used
// source frames (a frame is a source for up to 8 targets: read from SQLite once) and the blobs of the current flow.
struct Scratch {
    pc_context* ctx = nullptr;
    pc_corr_set* set = nullptr;
    std::vector<int32_t> sources;
    struct CachedKeypoints {
        int32_t frame = 0;
        bool valid = false;
        uint64_t stamp = 0;
        Keypoints keypoints;
    };
    CachedKeypoints cache[16];
    uint64_t clock = 0;
    KeypointsIndices indices;
    Keypoints targets;

    Scratch() : ctx(SharedGpuContext()) {
        GpuSection section;
        if (pc_corr_set_create(ctx, &set) != PC_OK) throw std::runtime_error(std::string("pc_corr_set_create: ") + pc_last_error());
    }
    Scratch(const Scratch&) = delete;
    Scratch& operator=(const Scratch&) = delete;
    ~Scratch() {
        GpuSection section;
        pc_corr_set_destroy(set);
    }

    const Keypoints& KeypointsOf(const Database& db, int32_t frame, FlowPrefetcher::Batch* batch) {
        CachedKeypoints* slot = &cache[0];
        for (auto& c : cache) {
            if (c.valid && c.frame == frame) {
                c.stamp = ++clock;
                return c.keypoints;
            }
            if (c.stamp < slot->stamp) slot = &c;
        }
        if (batch && batch->has_keypoints && batch->keypoints_frame == frame) {
            slot->keypoints.swap(batch->keypoints);   // read ahead by the prefetcher
            batch->has_keypoints = false;
        } else {
            slot->keypoints.clear();   // a recycled slot: a frame without a keypoints row must not inherit the old ones
            db.ReadKeypoints(frame, slot->keypoints);
        }
        slot->frame = frame;
        slot->valid = true;
        slot->stamp = ++clock;
        return slot->keypoints;
    }
};

[[noreturn]] void ThrowHip(const char* what) { throw std::runtime_error(std::string(what) + ": " + pc_last_error()); }

// correspondences contributed by one source frame (tracker.cc:52-92): the gather, the ray cast, the model
// transform and the append run on the GPU (pc_corr_set_append); the host only feeds the database blobs
void AppendFromSource(const Database& db, int32_t source_frame, int32_t target_frame, const CameraState& source_camera,
                      const Mat4f& model_matrix, const AcceleratedMesh& mesh, Scratch& s, FlowPrefetcher::Batch* batch) {
    const Keypoints* keypoints;
    const KeypointsIndices* indices = &s.indices;
    const Keypoints* targets = &s.targets;
    {
        StageClock::Scope sc("track/db read");
        keypoints = &s.KeypointsOf(db, source_frame, batch);
        bool prefetched = false;
        if (batch)
            for (size_t k = 0; k < batch->n_flows && !prefetched; k++)
                if (batch->flows[k].source == source_frame) {
                    indices = &batch->flows[k].indices;
                    targets = &batch->flows[k].targets;
                    prefetched = true;
                }
        if (!prefetched) db.ReadImagePairMatches(source_frame, target_frame, s.indices, s.targets);
    }
    CHECK_EQ(indices->size(), targets->size());
    if (indices->empty()) return;
    StageClock::Scope sc("track/append (enqueue)");
    SceneTransformations scene;
    scene.model_matrix = model_matrix;
    scene.view_matrix = source_camera.pose.Rt4x4();
    scene.intrinsics = source_camera.intrinsics;
    pc_ray_camera cam;
    MakeRayCamera(scene, &cam);
    static const float kNoKeypoint[2] = {0.f, 0.f};
    if (pc_corr_set_append(s.ctx, s.set, mesh.Gpu(), &cam, model_matrix.data(), source_frame,
                           keypoints->empty() ? kNoKeypoint : keypoints->front().data(), static_cast<int>(keypoints->size()),
                           indices->data(), targets->front().data(), static_cast<int>(indices->size()),
                           /*check_mask=*/1) != PC_OK)
        ThrowHip("pc_corr_set_append");
}

// "The solution should be very close to the previous/next pose" (tracker.cc:111-119)
CameraState InitialGuess(const CameraTrajectory& traj, int32_t frame) {
    for (int32_t candidate : {frame, frame - 1, frame + 1})
        if (traj.IsFrameFilled(candidate)) return *traj.Get(candidate);
    return CameraState{};
}

// fills s.set with the correspondences of `frame`; returns their number
int GatherCorrespondences(const Database& db, const CameraTrajectory& traj, const Mat4f& model_matrix, int32_t frame,
                          const AcceleratedMesh& mesh, Scratch& s, FlowPrefetcher::Batch* batch = nullptr) {
    if (pc_corr_set_clear(s.ctx, s.set) != PC_OK) ThrowHip("pc_corr_set_clear");
    s.sources.clear();
    db.FindOpticalFlowsToImage(frame, s.sources);
    bool mask_sent = false;
    for (int32_t source : s.sources) {
        CHECK_NE(source, frame);
        if (!traj.IsFrameFilled(source)) continue;  // only frames that already have a pose (:48)
        if (!mask_sent) {   // the mask can be edited between frames through inner_mut(): send the current bits
            mesh.SyncMask();
            mask_sent = true;
        }
        AppendFromSource(db, source, frame, *traj.Get(source), model_matrix, mesh, s, batch);
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

std::optional<PnPResult> SolveFrame(const Database& db, const CameraTrajectory& traj, const Mat4f& model_matrix,
                                    int32_t frame, const AcceleratedMesh& mesh, const PnPOptions& pnp_opts, Scratch& s,
                                    FlowPrefetcher::Batch* batch) {
    // the GPU part of one frame -- correspondences appended, counted, solved, read back -- is one section on the shared
    // context; between frames other threads (ray_cast from Python, a refinement) get their turn
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

    const int32_t step = frame_from < frame_to_inclusive ? 1 : -1;
    Scratch scratch;
    FlowPrefetcher prefetcher(database.Path());
    std::vector<int32_t> next_sources, wanted;
    for (int32_t frame = frame_from + step; frame != frame_to_inclusive + step; frame += step) {
        FlowPrefetcher::Batch* batch = prefetcher.Take(frame);
        // while this frame is solved: read what the next one needs -- flows from the frames that have a pose by
        // then (this one included) and this frame's keypoints, the one array the host cache does not hold yet
        const int32_t next = frame + step;
        if (prefetcher.Enabled() && next != frame_to_inclusive + step) {
            next_sources.clear();
            wanted.clear();
            database.FindOpticalFlowsToImage(next, next_sources);
            for (int32_t src : next_sources)
                if (src == frame || camera_traj.IsFrameFilled(src)) wanted.push_back(src);
            prefetcher.Request(next, wanted, frame);
        }
        const std::optional<PnPResult> solved =
            SolveFrame(database, camera_traj, model_matrix, frame, accel_mesh, pnp_opts, scratch, batch);
        if (!solved)
            throw std::runtime_error("Could not track to frame: " + std::to_string(frame) + ". Not enough features.");
        if (callback) {
            FrameTrackingResult report;
            report.frame = frame;
            report.pose = solved->camera.pose;
            report.intrinsics = solved->camera.intrinsics;
            report.bundle_stats = solved->bundle_stats;
            report.inlier_ratio = solved->inlier_ratio;
            if (!callback(report)) return;  // the pose of a frame the user stopped at is not stored (:179-186)
        }
        camera_traj.Set(frame, solved->camera);
    }
    StageClock::Report("TrackCameraTrajectory");
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
