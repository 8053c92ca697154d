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

#include "pnp.h"

namespace {

constexpr float kMaxInlierError = 12.0f;  // tracker.cc:123 ("FIXME: Make this customizable")

struct Correspondences {
    std::vector<float> world_points;  // n x 3
    std::vector<float> image_points;  // n x 2
    size_t size() const { return image_points.size() / 2; }
    void clear() {
        world_points.clear();
        image_points.clear();
    }
};

struct Scratch {
    std::vector<int32_t> sources;
    Keypoints keypoints;
    ImagePairFlow flow;
    std::vector<float> pixels;
    std::vector<std::optional<RayHit>> hits;
};

Vec3f ToWorld(const Mat4f& model, const Vec3f& p) {
    return {model[0] * p[0] + model[1] * p[1] + model[2] * p[2] + model[3],
            model[4] * p[0] + model[5] * p[1] + model[6] * p[2] + model[7],
            model[8] * p[0] + model[9] * p[1] + model[10] * p[2] + model[11]};
}

// correspondences contributed by one source frame (tracker.cc:52-92)
void AppendFromSource(const Database& db, int32_t source_frame, int32_t target_frame, const CameraState& source_camera,
                      const Mat4f& model_matrix, const AcceleratedMesh& mesh, Scratch& s, Correspondences& out) {
    db.ReadKeypoints(source_frame, s.keypoints);
    db.ReadImagePairFlow(source_frame, target_frame, s.flow);
    CHECK_EQ(s.flow.src_kps_indices.size(), s.flow.tgt_kps.size());
    const size_t matches = s.flow.src_kps_indices.size();
    s.pixels.resize(2 * matches);
    for (size_t i = 0; i < matches; i++) {
        const uint32_t k = s.flow.src_kps_indices[i];
        CHECK_LT(k, s.keypoints.size());
        s.pixels[2 * i] = s.keypoints[k][0];
        s.pixels[2 * i + 1] = s.keypoints[k][1];
    }
    SceneTransformations scene;
    scene.model_matrix = model_matrix;
    scene.view_matrix = source_camera.pose.Rt4x4();
    scene.intrinsics = source_camera.intrinsics;
    mesh.RayCastPixels(scene, s.pixels.data(), matches, /*check_mask=*/true, s.hits);
    for (size_t i = 0; i < matches; i++) {
        if (!s.hits[i]) continue;
        const Vec3f w = ToWorld(model_matrix, s.hits[i]->pos);
        out.world_points.insert(out.world_points.end(), w.begin(), w.end());
        out.image_points.push_back(s.flow.tgt_kps[i][0]);
        out.image_points.push_back(s.flow.tgt_kps[i][1]);
    }
}

// "The solution should be very close to the previous/next pose" (tracker.cc:111-119)
CameraState InitialGuess(const CameraTrajectory& traj, int32_t frame) {
    for (int32_t candidate : {frame, frame - 1, frame + 1})
        if (traj.IsFrameFilled(candidate)) return *traj.Get(candidate);
    return CameraState{};
}

std::optional<PnPResult> SolveFrame(const Database& db, const CameraTrajectory& traj, const Mat4f& model_matrix,
                                    int32_t frame, const AcceleratedMesh& mesh, const PnPOptions& pnp_opts, Scratch& s,
                                    Correspondences& corr) {
    corr.clear();
    s.sources.clear();
    db.FindOpticalFlowsToImage(frame, s.sources);
    for (int32_t source : s.sources) {
        CHECK_NE(source, frame);
        if (!traj.IsFrameFilled(source)) continue;  // only frames that already have a pose (:48)
        AppendFromSource(db, source, frame, *traj.Get(source), model_matrix, mesh, s, corr);
    }
    if (corr.size() < 3) return std::nullopt;  // :95-97
    PnPResult result;
    result.camera = InitialGuess(traj, frame);
    SolvePnPIterative(corr.world_points.data(), corr.image_points.data(), nullptr, corr.size(), pnp_opts, result);
    return result;
}

}  // namespace

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
    Correspondences corr;
    for (int32_t frame = frame_from + step; frame != frame_to_inclusive + step; frame += step) {
        const std::optional<PnPResult> solved =
            SolveFrame(database, camera_traj, model_matrix, frame, accel_mesh, pnp_opts, scratch, corr);
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
