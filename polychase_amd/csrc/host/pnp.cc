// pnp.cc -- LevMarqDenseSolver (cpp/pnp/lev_marq.h:99-389) + PnPProblem::Step
// (cpp/pnp/pnp_problem.h:101-131) on the host; residual sweeps (normal equations, cost, inliers)
// on the GPU through pc_pnp_* (one deterministic two-stage reduction per call).
#include "pnp.h"

#include <cmath>
#include <cstdlib>
#include <stdexcept>

#include "gpu_context.h"
#include "stage_clock.h"

namespace {

struct Params {  // PnPProblem::Parameters
    CameraState cam;
    Mat3f R;
};

pc_pnp_params ToGpu(const Params& p, bool opt_f, bool opt_pp, const BundleOptions& o) {
    pc_pnp_params g;
    for (int i = 0; i < 9; i++) g.R[i] = p.R[i];
    for (int i = 0; i < 3; i++) g.t[i] = p.cam.pose.t[i];
    g.fx = p.cam.intrinsics.fx;
    g.fy = p.cam.intrinsics.fy;
    g.cx = p.cam.intrinsics.cx;
    g.cy = p.cam.intrinsics.cy;
    g.aspect_ratio = p.cam.intrinsics.aspect_ratio;
    g.convention_opencv = p.cam.intrinsics.convention == CameraConvention::OpenCV ? 1 : 0;
    g.optimize_focal_length = opt_f ? 1 : 0;
    g.optimize_principal_point = opt_pp ? 1 : 0;
    g.loss_type = static_cast<int>(o.loss_type);
    g.loss_scale = o.loss_scale;
    return g;
}

struct GpuProblem {
    pc_context* ctx;
    pc_pnp_problem* prob = nullptr;
    ~GpuProblem() { pc_pnp_problem_destroy(prob); }
};

[[noreturn]] void ThrowHip(const char* what) { throw std::runtime_error(std::string(what) + ": " + pc_last_error()); }

}  // namespace

void SolvePnPIterative(const float* object_points, const float* image_points, const float* weights, size_t n,
                       const PnPOptions& opts, PnPResult& result) {
    CHECK_GE(n, static_cast<size_t>(3));  // solvers.cc:54-55
    const int lt = static_cast<int>(opts.bundle_opts.loss_type);
    if (lt < 0 || lt > 2) throw std::runtime_error("Unknown loss type: " + std::to_string(lt));
    GpuSection section;
    GpuProblem gp{SharedGpuContext()};
    {
        StageClock::Scope sc("pnp/create+upload");
        if (pc_pnp_problem_create(gp.ctx, object_points, image_points, weights, static_cast<int>(n), &gp.prob) != PC_OK)
            ThrowHip("pc_pnp_problem_create");
    }
    SolvePnPIterativeOnGpu(gp.prob, n, opts, result);
}

void SolvePnPIterativeOnGpu(pc_pnp_problem* problem, size_t n, const PnPOptions& opts, PnPResult& result) {
    CHECK_GE(n, static_cast<size_t>(3));  // solvers.cc:54-55
    const int lt = static_cast<int>(opts.bundle_opts.loss_type);
    if (lt < 0 || lt > 2) throw std::runtime_error("Unknown loss type: " + std::to_string(lt));
    const BundleOptions& bo = opts.bundle_opts;
    // PnPProblem: intrinsics are only optimised with more than 3 points (pnp_problem.h:34-35)
    const bool opt_f = opts.optimize_focal_length && n > 3;
    const bool opt_pp = opts.optimize_principal_point && n > 3;
    const CameraIntrinsics::Bounds bounds = result.camera.intrinsics.GetBounds();
    GpuSection section;
    struct {
        pc_context* ctx;
        pc_pnp_problem* prob;
    } gp{SharedGpuContext(), problem};

    // The solver runs on the device (pc_pnp_solve: the LM state lives in device memory, a one-lane kernel takes the
    // decisions between the residual sweeps).  POLYCHASE_PNP_HOST_LM=1 runs the same loop on the host instead,
    // one read-back per sweep -- the cross-check of the device state machine.
    const char* host_lm_env = std::getenv("POLYCHASE_PNP_HOST_LM");   // read per call: the tests flip it
    const bool host_lm = host_lm_env && host_lm_env[0] == '1';
    if (!host_lm) {
        pc_pnp_camera init;
        const CameraState& c0 = result.camera;
        init.q_xyzw[0] = c0.pose.q.x;
        init.q_xyzw[1] = c0.pose.q.y;
        init.q_xyzw[2] = c0.pose.q.z;
        init.q_xyzw[3] = c0.pose.q.w;
        for (int i = 0; i < 3; i++) init.t[i] = c0.pose.t[i];
        init.fx = c0.intrinsics.fx;
        init.fy = c0.intrinsics.fy;
        init.cx = c0.intrinsics.cx;
        init.cy = c0.intrinsics.cy;
        init.aspect_ratio = c0.intrinsics.aspect_ratio;
        init.convention_opencv = c0.intrinsics.convention == CameraConvention::OpenCV ? 1 : 0;
        pc_pnp_solve_options so;
        so.max_iterations = static_cast<int>(bo.max_iterations);
        so.initial_lambda = bo.initial_lambda;
        so.min_lambda = bo.min_lambda;
        so.max_lambda = bo.max_lambda;
        so.gradient_tol = bo.gradient_tol;
        so.step_tol = bo.step_tol;
        so.loss_type = lt;
        so.loss_scale = bo.loss_scale;
        so.optimize_focal_length = opt_f ? 1 : 0;
        so.optimize_principal_point = opt_pp ? 1 : 0;
        so.f_low = bounds.f_low;
        so.f_high = bounds.f_high;
        so.cx_low = bounds.cx_low;
        so.cx_high = bounds.cx_high;
        so.cy_low = bounds.cy_low;
        so.cy_high = bounds.cy_high;
        so.max_inlier_error = opts.max_inlier_error;
        so.rounds_hint = 0;
        pc_pnp_solve_result sr;
        if (pc_pnp_solve(gp.ctx, gp.prob, &init, &so, &sr) != PC_OK) ThrowHip("pc_pnp_solve");
        CameraState& c = result.camera;
        c.pose.q.x = sr.camera.q_xyzw[0];
        c.pose.q.y = sr.camera.q_xyzw[1];
        c.pose.q.z = sr.camera.q_xyzw[2];
        c.pose.q.w = sr.camera.q_xyzw[3];
        for (int i = 0; i < 3; i++) c.pose.t[i] = sr.camera.t[i];
        c.intrinsics.fx = sr.camera.fx;
        c.intrinsics.fy = sr.camera.fy;
        c.intrinsics.cx = sr.camera.cx;
        c.intrinsics.cy = sr.camera.cy;
        BundleStats st;
        st.iterations = static_cast<size_t>(sr.iterations);
        st.invalid_steps = static_cast<size_t>(sr.invalid_steps);
        st.initial_cost = sr.initial_cost;
        st.cost = sr.cost;
        st.lambda = sr.lambda;
        st.step_norm = sr.step_norm;
        st.grad_norm = sr.grad_norm;
        result.bundle_stats = st;
        result.inlier_ratio = static_cast<Float>(sr.inliers) / static_cast<Float>(n);
        return;
    }

    Params params{result.camera, result.camera.pose.R()};
    Params params_new = params;

    auto total_cost = [&](const Params& p, int* inliers, float max_err_sq) {
        const pc_pnp_params g = ToGpu(p, opt_f, opt_pp, bo);
        float cost = 0;
        int valid = 0;
        if (pc_pnp_total_cost(gp.ctx, gp.prob, &g, max_err_sq, &cost, &valid, inliers) != PC_OK) ThrowHip("pc_pnp_total_cost");
        return cost;  // kShouldNormalize == false
    };

    // One sweep returns the cost of a parameter set AND its normal equations (pc_pnp_normal_equations_cost): the
    // candidate of every LM step is evaluated that way, so an accepted step already holds the system the
    // reference would build at the top of the next iteration (lev_marq.h:146-160) -- same numbers, half the
    // GPU round trips.
    struct System {
        float lower[45];
        float Jtr[9];
        float cost = 0;
    };
    auto sweep = [&](const Params& p, System& out) {
        const pc_pnp_params g = ToGpu(p, opt_f, opt_pp, bo);
        int valid = 0;
        if (pc_pnp_normal_equations_cost(gp.ctx, gp.prob, &g, out.lower, out.Jtr, &valid, &out.cost) != PC_OK)
            ThrowHip("pc_pnp_normal_equations");
    };

    // ---- LevMarqDenseSolver::Solve (lev_marq.h:132-228) ----
    BundleStats stats;
    System current, candidate;
    sweep(params, current);
    stats.cost = current.cost;
    stats.initial_cost = stats.cost;
    stats.grad_norm = -1;
    stats.step_norm = -1;
    stats.invalid_steps = 0;
    stats.lambda = bo.initial_lambda;

    float JtJ[81];  // row-major, lower triangle meaningful
    float diag[9], Jtr[9], step[9];
    Float v = 2.0f;
    bool rebuild = true;
    for (stats.iterations = 0; stats.iterations < bo.max_iterations; ++stats.iterations) {
        if (rebuild) {
            int o = 0;
            for (int a = 0; a < 9; a++)
                for (int b = 0; b <= a; b++) JtJ[9 * a + b] = current.lower[o++];
            for (int a = 0; a < 9; a++) Jtr[a] = current.Jtr[a];
            // JtJ_diag = diag.cwiseMax(1e-6).cwiseMin(1e32)  (:296)
            for (int a = 0; a < 9; a++) diag[a] = std::min(std::max(JtJ[10 * a], 1e-6f), 1e32f);
            float g2 = 0;
            for (int a = 0; a < 9; a++) g2 += Jtr[a] * Jtr[a];
            stats.grad_norm = std::sqrt(g2);
            if (stats.grad_norm < bo.gradient_tol) break;
        }
        // ComputeStep (:299-314): multiplicative damping, LLT of the lower triangle
        float L[81];
        for (int a = 0; a < 9; a++)
            for (int b = 0; b <= a; b++) L[9 * a + b] = JtJ[9 * a + b];
        for (int a = 0; a < 9; a++) L[10 * a] = diag[a] * (1.0f + stats.lambda);
        for (int a = 0; a < 9; a++) JtJ[10 * a] = diag[a];  // "remove dampening" leaves the clamped diagonal
        const bool ok = CholeskyLower<9>(L);
        if (ok) {
            CholeskySolve<9>(L, Jtr, step);
            for (int a = 0; a < 9; a++) step[a] = -step[a];
        }
        if (!ok) {
            stats.invalid_steps++;
            if (stats.lambda == bo.max_lambda) break;
            stats.lambda = std::min(bo.max_lambda, stats.lambda * v);
            v = 2 * v;
            rebuild = false;
            continue;
        }
        float s2 = 0;
        for (int a = 0; a < 9; a++) s2 += step[a] * step[a];
        stats.step_norm = std::sqrt(s2);
        if (stats.step_norm < bo.step_tol) break;

        // PnPProblem::Step (pnp_problem.h:101-131)
        {
            const CameraState& cam = params.cam;
            CameraState& nw = params_new.cam;
            nw.pose.q = QuatStepPost(cam.pose.q, Vec3f{step[0], step[1], step[2]});
            nw.pose.t = cam.pose.t + Vec3f{step[3], step[4], step[5]};
            if (opt_f) {
                nw.intrinsics.fy = cam.intrinsics.fy + step[6];
                nw.intrinsics.fx = nw.intrinsics.fy * nw.intrinsics.aspect_ratio;
                nw.intrinsics.fy = std::clamp(nw.intrinsics.fy, bounds.f_low, bounds.f_high);
                nw.intrinsics.fx = std::clamp(nw.intrinsics.fx, bounds.f_low, bounds.f_high);
            }
            if (opt_pp) {
                nw.intrinsics.cx = std::clamp(cam.intrinsics.cx + step[7], bounds.cx_low, bounds.cx_high);
                nw.intrinsics.cy = std::clamp(cam.intrinsics.cy + step[8], bounds.cy_low, bounds.cy_high);
            }
            params_new.R = nw.pose.R();
        }
        sweep(params_new, candidate);
        const Float cost_new = candidate.cost;

        if (cost_new < stats.cost) {
            const Float actual = cost_new - stats.cost;
            // step^T (2 Jtr + JtJ_sym step)   (:183-186), fp32 like the reference
            Float expected = 0;
            for (int a = 0; a < 9; a++) {
                Float row = 0;
                for (int b = 0; b < 9; b++) row += (b <= a ? JtJ[9 * a + b] : JtJ[9 * b + a]) * step[b];
                expected += step[a] * (2.0f * Jtr[a] + row);
            }
            const Float rho = actual / expected;
            if (rho > 0) {  // ill-conditioned JtJ can make `expected` positive (:189-197)
                const double factor = std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rho - 1.0, 3));
                stats.lambda = std::clamp(static_cast<Float>(stats.lambda * factor), bo.min_lambda, bo.max_lambda);
            }
            params = params_new;
            current = candidate;
            stats.cost = cost_new;
            v = 2;
            rebuild = true;
        } else {
            stats.invalid_steps++;
            if (stats.lambda == bo.max_lambda) break;
            stats.lambda = std::min(bo.max_lambda, stats.lambda * v);
            v = 2 * v;
            rebuild = false;
        }
    }
    result.bundle_stats = stats;
    result.camera = params.cam;

    // inlier ratio (solvers.cc:31-47)
    int inliers = 0;
    if (opts.max_inlier_error > 0.0f) total_cost(params, &inliers, opts.max_inlier_error * opts.max_inlier_error);
    result.inlier_ratio = static_cast<Float>(inliers) / static_cast<Float>(n);
}
