// track_sequence.h -- "Track Sequence": camera poses frame by frame from the flow database
// (reference cpp/tracker.h:23-39, cpp/tracker.cc:36-213).
#pragma once

#include <functional>
#include <string>

#include "flow_database.h"
#include "ray_casting.h"
#include "types.h"

// called after every solved frame; returning false stops the run (tracker.cc:170-184)
using TrackingCallback = std::function<bool(const FrameTrackingResult&)>;

// Solves frames frame_from+-1 ... frame_to_inclusive in order; camera_traj must hold frame_from.
void TrackCameraTrajectory(const Database& database, CameraTrajectory& camera_traj, int32_t frame_from,
                           int32_t frame_to_inclusive, const Mat4f& model_matrix, const AcceleratedMesh& accel_mesh,
                           TrackingCallback callback, bool optimize_focal_length, bool optimize_principal_point,
                           const BundleOptions& opts);

// Opens the database, seeds the trajectory with scene_transform's view/intrinsics at frame_from.
void TrackSequence(const std::string& database_path, int32_t frame_from, int32_t frame_to_inclusive,
                   const SceneTransformations& scene_transform, const AcceleratedMesh& accel_mesh,
                   TrackingCallback callback, bool optimize_focal_length, bool optimize_principal_point,
                   BundleOptions opts);

// Gives back what a run keeps for the next one: the parked correspondence set (device memory, a stream) and the pool of
// page-locked blocks (track_sequence.cc; POLYCHASE_TRACK_CACHE=0 keeps nothing in the first place).
void ReleaseTrackerCaches();

// Creates, at a moment when nobody waits for a pose (the construction of an AcceleratedMesh: tracking always follows one), what the
// first TrackSequence call of a process would otherwise create while the caller waits for its first pose: the correspondence set
// with its stream, device arrays and page-locked result words (parked like a finished run's), and a handful of page-locked blocks
// in the pool -- 6-8 ms, a tenth of a 300-frame run.  Once per process (again after ReleaseTrackerCaches); POLYCHASE_TRACK_CACHE=0
// or POLYCHASE_TRACK_WARM=0: nothing.
void WarmTrackerCaches();

// The 3D-2D correspondences SolveFrame (tracker.cc:36-97) would hand to PnP for `frame` under the poses in
// camera_traj -- built on the GPU like in TrackCameraTrajectory, then downloaded (tests / debugging).
void FrameCorrespondences(const Database& database, const CameraTrajectory& camera_traj, const Mat4f& model_matrix,
                          int32_t frame, const AcceleratedMesh& accel_mesh, std::vector<float>& world_points,
                          std::vector<float>& image_points);
