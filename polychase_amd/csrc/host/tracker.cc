// tracker.cc -- SolveFrame / TrackCameraTrajectory / TrackSequence (reference cpp/tracker.cc:36-213).
// Same control flow; the per-match Embree call (:64-78, "TODO: benchmark / vectorize /
// parallelize") becomes one batched GPU ray cast per source frame and the LM residual sweeps run
// on the GPU (pnp.cc).
#include "tracker.h"

#include <cstdlib>
#include <optional>
#include <stdexcept>

#include "pnp.h"

namespace {

struct SolveFrameCache {
    std::vector<float> object_points_worldspace;  // n x 3
    std::vector<float> image_points;              // n x 2
    std::vector<int32_t> flow_frames_ids;
    Keypoints keypoints;
    ImagePairFlow flow;
    std::vector<float> src_xy;
    std::vector<std::optional<RayHit>> hits;
    void Clear() {
        object_points_worldspace.clear();
        image_points.clear();
        flow_frames_ids.clear();
        keypoints.clear();
        flow.Clear();
    }
};

std::optional<PnPResult> SolveFrame(const Database& database, const CameraTrajectory& camera_traj,
                                    const Mat4f& model_matrix, int32_t frame_id, const AcceleratedMesh& accel_mesh,
                                    bool optimize_focal_length, bool optimize_principal_point,
                                    const BundleOptions& bundle_opts, SolveFrameCache& cache) {
    cache.Clear();
    database.FindOpticalFlowsToImage(frame_id, cache.flow_frames_ids);  // :43
    for (int32_t flow_frame_id : cache.flow_frames_ids) {
        CHECK_NE(flow_frame_id, frame_id);
        if (!camera_traj.IsFrameFilled(flow_frame_id)) continue;  // :48
        database.ReadKeypoints(flow_frame_id, cache.keypoints);
        database.ReadImagePairFlow(flow_frame_id, frame_id, cache.flow);
        CHECK_EQ(cache.flow.src_kps_indices.size(), cache.flow.tgt_kps.size());
        const size_t num_matches = cache.flow.src_kps_indices.size();
        const CameraState& camera_state = *camera_traj.Get(flow_frame_id);
        SceneTransformations st;
        st.model_matrix = model_matrix;
        st.view_matrix = camera_state.pose.Rt4x4();
        st.intrinsics = camera_state.intrinsics;
        cache.src_xy.resize(2 * num_matches);
        for (size_t i = 0; i < num_matches; i++) {
            const uint32_t kp_idx = cache.flow.src_kps_indices[i];
            CHECK_LT(kp_idx, cache.keypoints.size());
            cache.src_xy[2 * i] = cache.keypoints[kp_idx][0];
            cache.src_xy[2 * i + 1] = cache.keypoints[kp_idx][1];
        }
        accel_mesh.RayCastPixels(st, cache.src_xy.data(), num_matches, /*check_mask=*/true, cache.hits);  // :76-77
        for (size_t i = 0; i < num_matches; i++) {
            if (!cache.hits[i]) continue;
            const Vec3f& hp = cache.hits[i]->pos;
            for (int r = 0; r < 3; r++)  // model.block<3,3>(0,0) * pos + model.block<3,1>(0,3)  (:80-82)
                cache.object_points_worldspace.push_back(model_matrix[4 * r] * hp[0] + model_matrix[4 * r + 1] * hp[1] +
                                                         model_matrix[4 * r + 2] * hp[2] + model_matrix[4 * r + 3]);
            cache.image_points.push_back(cache.flow.tgt_kps[i][0]);
            cache.image_points.push_back(cache.flow.tgt_kps[i][1]);
        }
    }
    const size_t n = cache.object_points_worldspace.size() / 3;
    if (n < 3) return std::nullopt;  // :95-97

    PnPResult result;
    // The solution should be very close to the previous/next pose (:111-119)
    if (camera_traj.IsFrameFilled(frame_id)) result.camera = *camera_traj.Get(frame_id);
    else if (camera_traj.IsFrameFilled(frame_id - 1)) result.camera = *camera_traj.Get(frame_id - 1);
    else if (camera_traj.IsFrameFilled(frame_id + 1)) result.camera = *camera_traj.Get(frame_id + 1);

    PnPOptions opts;
    opts.bundle_opts = bundle_opts;
    opts.max_inlier_error = 12.0f;  // :123
    opts.optimize_focal_length = optimize_focal_length;
    opts.optimize_principal_point = optimize_principal_point;
    SolvePnPIterative(cache.object_points_worldspace.data(), cache.image_points.data(), nullptr, n, opts, result);
    return result;
}

}  // namespace

void TrackCameraTrajectory(const Database& database, CameraTrajectory& camera_traj, int32_t frame_from,
                           int32_t frame_to_inclusive, const Mat4f& model_matrix, const AcceleratedMesh& accel_mesh,
                           TrackingCallback callback, bool optimize_focal_length, bool optimize_principal_point,
                           const BundleOptions& opts) {
    const int32_t first_frame = std::min(frame_from, frame_to_inclusive);
    const int32_t last_frame = std::max(frame_from, frame_to_inclusive);
    const int32_t dir = (frame_from < frame_to_inclusive) ? 1 : -1;
    CHECK(camera_traj.IsValidFrame(first_frame));
    CHECK(camera_traj.IsValidFrame(last_frame));
    CHECK(camera_traj.IsFrameFilled(frame_from));
    SolveFrameCache cache;
    for (int32_t frame_id = frame_from + dir; frame_id != frame_to_inclusive + dir; frame_id += dir) {
        const std::optional<PnPResult> maybe_result =
            SolveFrame(database, camera_traj, model_matrix, frame_id, accel_mesh, optimize_focal_length,
                       optimize_principal_point, opts, cache);
        if (!maybe_result)
            throw std::runtime_error("Could not track to frame: " + std::to_string(frame_id) + ". Not enough features.");
        const PnPResult& r = *maybe_result;
        if (callback) {
            FrameTrackingResult fr;
            fr.frame = frame_id;
            fr.pose = r.camera.pose;
            fr.intrinsics = r.camera.intrinsics;
            fr.bundle_stats = r.bundle_stats;
            fr.inlier_ratio = r.inlier_ratio;
            if (!callback(fr)) return;  // user requested to stop (:179-183)
        }
        camera_traj.Set(frame_id, r.camera);
    }
}

void TrackSequence(const std::string& database_path, int32_t frame_from, int32_t frame_to_inclusive,
                   const SceneTransformations& scene_transform, const AcceleratedMesh& accel_mesh,
                   TrackingCallback callback, bool optimize_focal_length, bool optimize_principal_point,
                   BundleOptions bundle_opts) {
    const Database database{database_path};
    const size_t num_frames = static_cast<size_t>(std::abs(frame_to_inclusive - frame_from)) + 1;
    CameraTrajectory camera_traj{std::min(frame_from, frame_to_inclusive), num_frames};
    camera_traj.Set(frame_from, CameraState{scene_transform.intrinsics, Pose::FromRt(scene_transform.view_matrix)});
    TrackCameraTrajectory(database, camera_traj, frame_from, frame_to_inclusive, scene_transform.model_matrix,
                          accel_mesh, callback, optimize_focal_length, optimize_principal_point, bundle_opts);
}
