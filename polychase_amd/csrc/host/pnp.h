// pnp.h -- SolvePnPIterative of the reference (cpp/pnp/solvers.h:22-29): dense Levenberg-Marquardt
// over 9 parameters (so(3) step, translation, fy, cx, cy) with a robust loss.
#pragma once

#include "types.h"

// object_points: n x 3, image_points: n x 2, weights: n or nullptr.
void SolvePnPIterative(const float* object_points, const float* image_points, const float* weights, size_t n,
                       const PnPOptions& opts, PnPResult& result);

struct pc_pnp_problem;
// The same solver over a problem that already lives on the GPU (pc_pnp_problem_create / pc_pnp_problem_from_set);
// n = its number of correspondences.
void SolvePnPIterativeOnGpu(pc_pnp_problem* problem, size_t n, const PnPOptions& opts, PnPResult& result);
