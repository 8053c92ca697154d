// tracker.h -- "Track Sequence" (reference cpp/tracker.h:23-39, cpp/tracker.cc).
#pragma once

#include <functional>
#include <string>

#include "database.h"
#include "ray_casting.h"
#include "types.h"

using TrackingCallback = std::function<bool(const FrameTrackingResult&)>;

void TrackSequence(const std::string& database_path, int32_t frame_from, int32_t frame_to_inclusive,
                   const SceneTransformations& scene_transform, const AcceleratedMesh& accel_mesh,
                   TrackingCallback callback, bool optimize_focal_length, bool optimize_principal_point,
                   BundleOptions opts);

void TrackCameraTrajectory(const Database& database, CameraTrajectory& camera_traj, int32_t frame_from,
                           int32_t frame_to_inclusive, const Mat4f& model_matrix, const AcceleratedMesh& accel_mesh,
                           TrackingCallback callback, bool optimize_focal_length, bool optimize_principal_point,
                           const BundleOptions& opts);
