// trajectory_refiner.h -- "Refine Sequence": joint refinement of all camera poses of a tracked segment against
// the optical-flow database (reference cpp/refiner.h:13-27, cpp/refiner.cc:506-725).
#pragma once

#include <functional>
#include <string>

#include "ray_casting.h"
#include "types.h"

struct RefineTrajectoryUpdate {  // refiner.h:13-18
    float progress = 0;
    std::string message;
    BundleStats stats;
};

// called after every accepted / rejected LM step and once at the end; false stops the run
using RefineTrajectoryCallback = std::function<bool(RefineTrajectoryUpdate)>;

// `traj` must be filled for every frame of [FirstFrame, LastFrame] and span more than two frames.
// The first and the last camera stay fixed; the others are updated in place.
void RefineTrajectory(const std::string& database_path, CameraTrajectory& traj, const Mat4f& model_matrix,
                      const AcceleratedMesh& mesh, bool optimize_focal_length, bool optimize_principal_point,
                      RefineTrajectoryCallback callback, BundleOptions bundle_opts);

// Not in the reference: cost and dense J^T J / J^T r at `traj` (one TotalCost + one
// BuildNormalEquations), for tests and tools.
struct RefinementSystem {
    float cost = 0;
    int num_params = 0, block_length = 0, num_edges = 0, num_residuals = 0, num_keypoints = 0;
    std::vector<float> JtJ;  // num_params x num_params, symmetric
    std::vector<float> Jtr;
};
RefinementSystem EvaluateRefinementSystem(const std::string& database_path, const CameraTrajectory& traj,
                                          const Mat4f& model_matrix, const AcceleratedMesh& mesh, bool optimize_focal_length,
                                          bool optimize_principal_point, const BundleOptions& opts);
