// track_sequence.cc -- reference cpp/tracker.cc:36-213 on the MI355X path.
//
// For the frame being solved, every flow INTO it whose source frame already has a pose contributes
// 3D-2D correspondences: source keypoints are cast onto the mesh under the source camera
// (tracker.cc:64-78; one batched GPU launch per source here instead of one Embree call per match),
// the hits are moved to world space (:80-82) and paired with the tracked positions (:86); PnP then
// starts from this/previous/next frame's pose (:111-119).  LM residual sweeps run on the GPU
// (pnp.cc).  Sequential over frames by construction: frame k needs the poses solved before it.
#include "track_sequence.h"

#include <cstdlib>
#include <optional>
#include <stdexcept>

#include "gpu_context.h"
#include "pnp.h"
#include "stage_clock.h"

namespace {

constexpr float kMaxInlierError = 12.0f;  // tracker.cc:123 ("FIXME: Make this customizable")

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
        if (pc_corr_set_create(ctx, &set) != PC_OK) throw std::runtime_error(std::string("pc_corr_set_create: ") + pc_last_error());
    }
    Scratch(const Scratch&) = delete;
    Scratch& operator=(const Scratch&) = delete;
    ~Scratch() { pc_corr_set_destroy(set); }

    const Keypoints& KeypointsOf(const Database& db, int32_t frame) {
        CachedKeypoints* slot = &cache[0];
        for (auto& c : cache) {
            if (c.valid && c.frame == frame) {
                c.stamp = ++clock;
                return c.keypoints;
            }
            if (c.stamp < slot->stamp) slot = &c;
        }
        db.ReadKeypoints(frame, slot->keypoints);
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
                      const Mat4f& model_matrix, const AcceleratedMesh& mesh, Scratch& s) {
    const Keypoints* keypoints;
    {
        StageClock::Scope sc("track/db read");
        keypoints = &s.KeypointsOf(db, source_frame);
        db.ReadImagePairMatches(source_frame, target_frame, s.indices, s.targets);
    }
    CHECK_EQ(s.indices.size(), s.targets.size());
    if (s.indices.empty()) return;
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
                           s.indices.data(), s.targets.front().data(), static_cast<int>(s.indices.size()),
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
                          const AcceleratedMesh& mesh, Scratch& s) {
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
        AppendFromSource(db, source, frame, *traj.Get(source), model_matrix, mesh, s);
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
                                    int32_t frame, const AcceleratedMesh& mesh, const PnPOptions& pnp_opts, Scratch& s) {
    const int n = GatherCorrespondences(db, traj, model_matrix, frame, mesh, s);
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
    for (int32_t frame = frame_from + step; frame != frame_to_inclusive + step; frame += step) {
        const std::optional<PnPResult> solved =
            SolveFrame(database, camera_traj, model_matrix, frame, accel_mesh, pnp_opts, scratch);
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
