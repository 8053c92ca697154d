// pnp_lm.hpp -- LevMarqDenseSolver::Solve (reference cpp/pnp/lev_marq.h:132-228) + PnPProblem::Step
// (cpp/pnp/pnp_problem.h:101-131) as a state machine that lives in device memory.
//
// The solver alternates residual sweeps (kernels over all correspondences) with a few hundred flops of 9x9
// algebra that decide what to evaluate next.  Run from the host, every decision costs a launch, a read-back and
// a stream synchronisation (~50 us) -- more than the sweep.  Here the decision step is a one-lane kernel
// (lm_consume): the host enqueues [sweep, reduce, decide] a dozen times without waiting and reads the state
// once; kernels enqueued past convergence see `done` and return at once.
//
// Arithmetic: fp32 like the reference, the same operation order as the former host loop.
#pragma once

#include <cmath>

#include "kernels.hpp"

#define PC_HD __host__ __device__ __forceinline__

namespace pc {

struct LmCamera {
    float qx, qy, qz, qw;     // pose rotation, Eigen order
    float t[3];
    float fx, fy, cx, cy, aspect_ratio;
    int convention_opencv;
};

struct LmConfig {
    int max_iterations;
    float initial_lambda, min_lambda, max_lambda, gradient_tol, step_tol;
    float f_low, f_high, cx_low, cx_high, cy_low, cy_high;   // CameraIntrinsics::GetBounds (types.h:156-192)
    int optimize_focal, optimize_pp, loss_type;
    float loss_scale;
    float max_inlier_err_sq;
};

struct LmState {
    LmConfig cfg;
    LmCamera cam, cam_new;
    PnPParams sweep;          // the parameters the next sweep evaluates (the final ones once done)
    float cur_lower[45], cur_Jtr[9];
    float JtJ[81], diag[9], Jtr[9], step[9];
    float cost, initial_cost, lambda, v, grad_norm, step_norm;
    int iterations, invalid_steps, rebuild, phase, done;
    int n_valid, status;      // track_lm_kernel: correspondences counted by the first sweep; 0 ok, 1 fewer than 3, 3 round limit
};

// track_lm_kernel (kernels_tracker.hip): the launch arguments and what it leaves in pinned host memory
constexpr int kTrackTickPhases = 8;
struct TrackLmOut {
    LmCamera cam;
    int iterations, invalid_steps;
    float initial_cost, cost, lambda, step_norm, grad_norm;
    int n_valid;              // correspondences (matches whose ray hit the mesh)
    int inliers;
    int rounds;               // sweeps evaluated
    int status;               // 0 solved, 1 fewer than 3 correspondences, 2 a grid barrier timed out, 3 round limit
    int bad_index;            // a match named a keypoint past its source's array (tracker.cc:61)
    // 100 MHz ticks of workgroup 0, summed over the rounds: [0] sweep + publish, [1] waiting for the other workgroups, [2] adding
    // the partials, [3] the decision (one lane), [4] publishing it, [5] fetching the next parameters, [6] inlier pass, [7] launch
    uint32_t ticks[kTrackTickPhases];
    unsigned long long begin_tick, end_tick;   // wall_clock64 (100 MHz) when workgroup 0 started / wrote this: idle time between launches
    // written LAST, with a system-scope release: TrackLmArgs::seq of the launch.  The host polls this word in its page-locked
    // memory instead of waiting for the stream (the result is out before the other workgroups have left and the queue has
    // signalled completion)
    uint32_t seq;
};
// What a tracked frame leaves on the device for the launches of the NEXT frame, which are enqueued before the host has seen this
// frame's pose (csrc/host/track_sequence.cc: the frame after the one on the GPU is always queued behind it): the solved camera
// (the next frame's initial guess, tracker.cc:111-119) and the same camera as a ray-cast source (SourceCamera of
// track_sequence.cc = GetRayObjectSpace, ray_casting.h:53-63, computed with the host's operations in the host's order: the
// next frame's world points are the same bits whichever side made the camera).
struct TrackChainSlot {
    LmCamera cam;
    RayCamera ray;
};
struct TrackLmArgs {
    const float4* pts;        // track_cast_kernel's output
    const float2* obs;        // tracked positions, same rows
    int n;
    LmConfig cfg;
    LmCamera cam;             // initial guess
    const TrackChainSlot* chain_in;   // not null: the initial guess is chain_in->cam (left there by the launch in front of this one)
    TrackChainSlot* chain_out;        // not null: the solved camera goes there as well (with `model`: its ray-cast form)
    float model[16];                  // the model matrix (row-major 4x4), for chain_out->ray
    float* partials;          // 56 x track_lm_blocks(n)
    uint32_t* sync;           // kTrackSyncWords
    TrackLmOut* out;          // device-visible host memory
    uint32_t seq;             // this launch's number (TrackLmOut::seq)
    const int* bad_index;     // track_cast_kernel's flag, passed on to `out`
    int max_rounds;
    int serial_decision;      // != 0: the decision of a round on ONE lane (lm_consume; round 5's, the cross-check of lm_consume_wave)
};

PC_HD float lm_clamp(float v, float lo, float hi) { return v < lo ? lo : (hi < v ? hi : v); }   // std::clamp

// Eigen::Quaternion::toRotationMatrix + the intrinsics, as the sweep kernels want them
PC_HD void lm_make_params(const LmCamera& c, const LmConfig& cfg, PnPParams* p) {
    const float tx = 2 * c.qx, ty = 2 * c.qy, tz = 2 * c.qz;
    const float twx = tx * c.qw, twy = ty * c.qw, twz = tz * c.qw;
    const float txx = tx * c.qx, txy = ty * c.qx, txz = tz * c.qx;
    const float tyy = ty * c.qy, tyz = tz * c.qy, tzz = tz * c.qz;
    p->R[0] = 1 - (tyy + tzz); p->R[1] = txy - twz;       p->R[2] = txz + twy;
    p->R[3] = txy + twz;       p->R[4] = 1 - (txx + tzz); p->R[5] = tyz - twx;
    p->R[6] = txz - twy;       p->R[7] = tyz + twx;       p->R[8] = 1 - (txx + tyy);
    for (int i = 0; i < 3; i++) p->t[i] = c.t[i];
    p->fx = c.fx;
    p->fy = c.fy;
    p->cx = c.cx;
    p->cy = c.cy;
    p->aspect_ratio = c.aspect_ratio;
    p->convention_opencv = c.convention_opencv;
    p->optimize_focal = cfg.optimize_focal;
    p->optimize_pp = cfg.optimize_pp;
    p->loss_type = cfg.loss_type;
    p->loss_scale = cfg.loss_scale;
}

// In-place lower Cholesky of a 9x9 row-major matrix, left-looking like Eigen's unblocked llt_inplace
// Every loop of the 9x9 algebra is fully unrolled: all indices are compile-time constants, so the work arrays live in
// registers.  (Round 4's decision kernel kept L[81] in scratch memory -- 116 bytes per lane, every access a trip to
// memory -- and took 15 us per LM round, as long as the residual sweep it follows: profiles/r05_head_c5_timeline.json.)
// A failed pivot does not return from inside the unrolled loop (that would keep the array addressable): the flag is
// carried to the end; the entries computed after a failure are never used.
PC_HD bool lm_cholesky9(float* a) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        float x = a[k * 9 + k];
#pragma unroll
        for (int j = 0; j < k; j++) x -= a[k * 9 + j] * a[k * 9 + j];
        ok = ok && (x > 0.0f);
        x = sqrtf(x);
        a[k * 9 + k] = x;
#pragma unroll
        for (int i = k + 1; i < 9; i++) {
            float s = a[i * 9 + k];
#pragma unroll
            for (int j = 0; j < k; j++) s -= a[i * 9 + j] * a[k * 9 + j];
            a[i * 9 + k] = s / x;
        }
    }
    return ok;
}
PC_HD void lm_cholesky9_solve(const float* l, const float* b, float* x) {
    float y[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        float s = b[i];
#pragma unroll
        for (int j = 0; j < i; j++) s -= l[i * 9 + j] * y[j];
        y[i] = s / l[i * 9 + i];
    }
#pragma unroll
    for (int i = 8; i >= 0; i--) {
        float s = y[i];
#pragma unroll
        for (int j = i + 1; j < 9; j++) s -= l[j * 9 + i] * x[j];
        x[i] = s / l[i * 9 + i];
    }
}

// PnPProblem::Step (pnp_problem.h:101-131): q <- q * AngleAxis(step[0..2]), t += step[3..5], clamped intrinsics
PC_HD void lm_step_camera(const LmCamera& c, const float* step, const LmConfig& cfg, LmCamera* out) {
    *out = c;
    const float angle = sqrtf(step[0] * step[0] + step[1] * step[1] + step[2] * step[2]);
    if (angle > 0) {   // QuatStepPost (cpp/pnp/quaternion.h:11-20)
        const float inv = 1.0f / angle;
        const float ax = step[0] * inv, ay = step[1] * inv, az = step[2] * inv;
        const float s = sinf(0.5f * angle), co = cosf(0.5f * angle);
        const float bx = ax * s, by = ay * s, bz = az * s, bw = co;
        out->qx = c.qw * bx + c.qx * bw + c.qy * bz - c.qz * by;
        out->qy = c.qw * by + c.qy * bw + c.qz * bx - c.qx * bz;
        out->qz = c.qw * bz + c.qz * bw + c.qx * by - c.qy * bx;
        out->qw = c.qw * bw - c.qx * bx - c.qy * by - c.qz * bz;
    }
    for (int i = 0; i < 3; i++) out->t[i] = c.t[i] + step[3 + i];
    if (cfg.optimize_focal) {
        out->fy = c.fy + step[6];
        out->fx = out->fy * out->aspect_ratio;
        out->fy = lm_clamp(out->fy, cfg.f_low, cfg.f_high);
        out->fx = lm_clamp(out->fx, cfg.f_low, cfg.f_high);
    }
    if (cfg.optimize_pp) {
        out->cx = lm_clamp(c.cx + step[7], cfg.cx_low, cfg.cx_high);
        out->cy = lm_clamp(c.cy + step[8], cfg.cy_low, cfg.cy_high);
    }
}

PC_HD void lm_finish(LmState& s) {
    s.done = 1;
    lm_make_params(s.cam, s.cfg, &s.sweep);   // the inlier pass runs on the accepted parameters
}

// The top of the solver's loop (lev_marq.h:146-181) up to the point where a candidate has to be evaluated:
// leaves cam_new / sweep set and phase = 1, or finishes.
PC_HD void lm_advance(LmState& s) {
    while (s.iterations < s.cfg.max_iterations) {
        if (s.rebuild) {
            int o = 0;
#pragma unroll
            for (int a = 0; a < 9; a++)
#pragma unroll
                for (int b = 0; b <= a; b++) s.JtJ[9 * a + b] = s.cur_lower[o++];
#pragma unroll
            for (int a = 0; a < 9; a++) s.Jtr[a] = s.cur_Jtr[a];
            // JtJ_diag = diag.cwiseMax(1e-6).cwiseMin(1e32)  (:296)
#pragma unroll
            for (int a = 0; a < 9; a++) s.diag[a] = fminf(fmaxf(s.JtJ[10 * a], 1e-6f), 1e32f);
            float g2 = 0;
#pragma unroll
            for (int a = 0; a < 9; a++) g2 += s.Jtr[a] * s.Jtr[a];
            s.grad_norm = sqrtf(g2);
            if (s.grad_norm < s.cfg.gradient_tol) break;
        }
        // ComputeStep (:299-314): multiplicative damping, LLT of the lower triangle
        float L[81];
#pragma unroll
        for (int a = 0; a < 9; a++)
#pragma unroll
            for (int b = 0; b <= a; b++) L[9 * a + b] = s.JtJ[9 * a + b];
#pragma unroll
        for (int a = 0; a < 9; a++) L[10 * a] = s.diag[a] * (1.0f + s.lambda);
#pragma unroll
        for (int a = 0; a < 9; a++) s.JtJ[10 * a] = s.diag[a];  // "remove dampening" leaves the clamped diagonal
        if (!lm_cholesky9(L)) {
            s.invalid_steps++;
            if (s.lambda == s.cfg.max_lambda) break;
            s.lambda = fminf(s.cfg.max_lambda, s.lambda * s.v);
            s.v = 2 * s.v;
            s.rebuild = 0;
            s.iterations++;
            continue;
        }
        lm_cholesky9_solve(L, s.Jtr, s.step);
#pragma unroll
        for (int a = 0; a < 9; a++) s.step[a] = -s.step[a];
        float s2 = 0;
#pragma unroll
        for (int a = 0; a < 9; a++) s2 += s.step[a] * s.step[a];
        s.step_norm = sqrtf(s2);
        if (s.step_norm < s.cfg.step_tol) break;
        lm_step_camera(s.cam, s.step, s.cfg, &s.cam_new);
        lm_make_params(s.cam_new, s.cfg, &s.sweep);
        s.phase = 1;
        return;   // evaluate cam_new
    }
    lm_finish(s);
}

// out56 = the sweep of s.sweep: [0..44] JtJ lower, [45..53] Jtr, [54] valid, [55] cost
PC_HD void lm_consume(LmState& s, const float* out56) {
    if (s.done) return;
    if (s.phase == 0) {   // the initial parameters (lev_marq.h:139-144)
#pragma unroll
        for (int k = 0; k < 45; k++) s.cur_lower[k] = out56[k];
#pragma unroll
        for (int k = 0; k < 9; k++) s.cur_Jtr[k] = out56[45 + k];
        s.cost = out56[55];
        s.initial_cost = s.cost;
        s.rebuild = 1;
        lm_advance(s);
        return;
    }
    const float cost_new = out56[55];
    if (cost_new < s.cost) {
        const float actual = cost_new - s.cost;
        // step^T (2 Jtr + JtJ_sym step)   (:183-186), fp32 like the reference
        float expected = 0;
#pragma unroll
        for (int a = 0; a < 9; a++) {
            float row = 0;
#pragma unroll
            for (int b = 0; b < 9; b++) row += (b <= a ? s.JtJ[9 * a + b] : s.JtJ[9 * b + a]) * s.step[b];
            expected += s.step[a] * (2.0f * s.Jtr[a] + row);
        }
        const float rho = actual / expected;
        if (rho > 0) {  // ill-conditioned JtJ can make `expected` positive (:189-197)
            const double d = 2.0 * (double)rho - 1.0;
            const double f = 1.0 - d * d * d;
            const double factor = f > 1.0 / 3.0 ? f : 1.0 / 3.0;
            s.lambda = lm_clamp((float)((double)s.lambda * factor), s.cfg.min_lambda, s.cfg.max_lambda);
        }
        s.cam = s.cam_new;
#pragma unroll
        for (int k = 0; k < 45; k++) s.cur_lower[k] = out56[k];
#pragma unroll
        for (int k = 0; k < 9; k++) s.cur_Jtr[k] = out56[45 + k];
        s.cost = cost_new;
        s.v = 2;
        s.rebuild = 1;
    } else {
        s.invalid_steps++;
        if (s.lambda == s.cfg.max_lambda) {
            lm_finish(s);
            return;
        }
        s.lambda = fminf(s.cfg.max_lambda, s.lambda * s.v);
        s.v = 2 * s.v;
        s.rebuild = 0;
    }
    s.iterations++;
    lm_advance(s);
}

#ifdef __HIPCC__
// ---------------------------------------------------------------------------------------------------------------------------
// The same decision taken by one WAVEFRONT (track_lm_kernel, round 6): lm_consume above is ~1600 dependent instructions of one
// lane -- 4.6 us of every 16-us LM round during which 255 workgroups wait.  Here the 9x9 algebra is dealt out over lanes 0..8
// with EVERY floating-point operation of the serial code kept, on the same operands, in the same order per result -- so the
// bits are the serial code's (tests/test_tracker_gpu.py holds the two against each other; POLYCHASE_TRACK_SERIAL_DECISION=1
// runs the serial one):
//   * Cholesky, left-looking: lane i owns row i of L.  Column k: every lane i >= k forms a[i][k] - sum_{j<k} a[i][j] a[k][j]
//     (row k's entries arrive as scalars: v_readlane), lane k's value is the pivot, one square root, one division per lane --
//     9 dependent divisions instead of 36.
//   * forward substitution by columns: y[j] = s[j] / l[j][j] in lane j, then every lane i > j subtracts l[i][j] y[j] -- for each
//     row the subtractions come in ascending j, as in the serial loop.
//   * back substitution: x[i] needs x[i+1..8] in ascending order, the last one first -- a dependent chain whichever lane runs
//     it: done uniformly (l[j][i] by v_readlane).
//   * step^T (2 J^T r + J^T J step): row a in lane a, the nine terms added in order.
//   * copies of the 45 + 9 + 81 entries: one lane each.
// Scalars of the state are updated by lane 0; the wavefront reads them back from LDS (in-order LDS, fences between the phases).
// Called by all 64 lanes of ONE wavefront; `s` and `out56` live in LDS.
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float lm_lane_value(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ void lm_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// lm_advance by a wavefront
__device__ __forceinline__ void lm_advance_wave(LmState& s, int lane) {
    for (;;) {
        lm_wave_sync();
        if (!(s.iterations < s.cfg.max_iterations)) break;
        if (s.rebuild) {
            // JtJ's lower triangle <- cur_lower, Jtr <- cur_Jtr
            for (int e = lane; e < 81; e += 64) {
                const int a = e / 9, b = e - 9 * a;
                if (b <= a) s.JtJ[e] = s.cur_lower[a * (a + 1) / 2 + b];
            }
            if (lane < 9) s.Jtr[lane] = s.cur_Jtr[lane];
            lm_wave_sync();
            // JtJ_diag = diag.cwiseMax(1e-6).cwiseMin(1e32)  (:296)
            float sq = 0.f;
            if (lane < 9) {
                s.diag[lane] = fminf(fmaxf(s.JtJ[10 * lane], 1e-6f), 1e32f);
                sq = s.Jtr[lane] * s.Jtr[lane];
            }
            float g2 = 0;
#pragma unroll
            for (int a = 0; a < 9; a++) g2 += lm_lane_value(sq, a);
            const float grad_norm = sqrtf(g2);
            if (lane == 0) s.grad_norm = grad_norm;
            lm_wave_sync();
            if (grad_norm < s.cfg.gradient_tol) break;
        }
        // ComputeStep (:299-314): multiplicative damping, LLT of the lower triangle; lane i holds row i
        float Lr[9];
        const int row = lane < 9 ? lane : 0;
        const float lambda = s.lambda;
#pragma unroll
        for (int j = 0; j < 9; j++) Lr[j] = (lane < 9 && j < row) ? s.JtJ[9 * row + j] : 0.0f;
        {
            const float d = (lane < 9) ? s.diag[row] * (1.0f + lambda) : 1.0f;
#pragma unroll
            for (int j = 0; j < 9; j++) Lr[j] = (j == lane) ? d : Lr[j];
            if (lane < 9) s.JtJ[10 * lane] = s.diag[lane];   // "remove dampening" leaves the clamped diagonal
        }
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            float sk = Lr[k];
#pragma unroll
            for (int j = 0; j < k; j++) sk -= Lr[j] * lm_lane_value(Lr[j], k);
            const float pivot = lm_lane_value(sk, k);
            ok = ok && (pivot > 0.0f);
            const float x = sqrtf(pivot);
            Lr[k] = (lane == k) ? x : sk / x;   // (lanes < k: the upper triangle, never read)
        }
        if (!ok) {
            if (lane == 0) {
                s.invalid_steps++;
                if (s.lambda != s.cfg.max_lambda) {
                    s.lambda = fminf(s.cfg.max_lambda, s.lambda * s.v);
                    s.v = 2 * s.v;
                    s.rebuild = 0;
                    s.iterations++;
                }
            }
            if (lambda == s.cfg.max_lambda) break;   // (the value before the update: s.lambda is unchanged in that case)
            continue;
        }
        // L y = J^T r
        float sy = lane < 9 ? s.Jtr[row] : 0.0f;
        float yv[9], xv[9];
#pragma unroll
        for (int j = 0; j < 9; j++) {
            yv[j] = lm_lane_value(sy / Lr[j], j);   // lane j: s[j] / l[j][j]
            sy -= Lr[j] * yv[j];                    // lanes i > j: s[i] -= l[i][j] y[j]
        }
        // L^T x = y
#pragma unroll
        for (int i = 8; i >= 0; i--) {
            float sx = yv[i];
#pragma unroll
            for (int j = i + 1; j < 9; j++) sx -= lm_lane_value(Lr[i], j) * xv[j];
            xv[i] = sx / lm_lane_value(Lr[i], i);
        }
        float s2 = 0;
#pragma unroll
        for (int a = 0; a < 9; a++) {
            const float st = -xv[a];
            if (lane == a) s.step[a] = st;
            s2 += st * st;
        }
        const float step_norm = sqrtf(s2);
        if (lane == 0) s.step_norm = step_norm;
        lm_wave_sync();
        if (step_norm < s.cfg.step_tol) break;
        if (lane == 0) {
            lm_step_camera(s.cam, s.step, s.cfg, &s.cam_new);
            lm_make_params(s.cam_new, s.cfg, &s.sweep);
            s.phase = 1;
        }
        lm_wave_sync();
        return;   // evaluate cam_new
    }
    if (lane == 0) lm_finish(s);
    lm_wave_sync();
}

// lm_consume by a wavefront
__device__ __forceinline__ void lm_consume_wave(LmState& s, const float* out56, int lane) {
    lm_wave_sync();
    if (s.done) return;
    auto take_sums = [&]() {   // cur_lower, cur_Jtr <- the sweep's sums
        if (lane < 45) s.cur_lower[lane] = out56[lane];
        if (lane < 9) s.cur_Jtr[lane] = out56[45 + lane];
    };
    if (s.phase == 0) {   // the initial parameters (lev_marq.h:139-144)
        take_sums();
        if (lane == 0) {
            s.cost = out56[55];
            s.initial_cost = s.cost;
            s.rebuild = 1;
        }
        lm_advance_wave(s, lane);
        return;
    }
    const float cost_new = out56[55];
    if (cost_new < s.cost) {
        const float actual = cost_new - s.cost;
        // step^T (2 Jtr + JtJ_sym step)   (:183-186), fp32 like the reference: row a in lane a, the terms added in order
        float term = 0.f;
        if (lane < 9) {
            float rowsum = 0;
#pragma unroll
            for (int b = 0; b < 9; b++) rowsum += (b <= lane ? s.JtJ[9 * lane + b] : s.JtJ[9 * b + lane]) * s.step[b];
            term = s.step[lane] * (2.0f * s.Jtr[lane] + rowsum);
        }
        float expected = 0;
#pragma unroll
        for (int a = 0; a < 9; a++) expected += lm_lane_value(term, a);
        const float rho = actual / expected;
        lm_wave_sync();   // every lane has read cost, JtJ, step, Jtr
        if (lane == 0) {
            if (rho > 0) {  // ill-conditioned JtJ can make `expected` positive (:189-197)
                const double d = 2.0 * (double)rho - 1.0;
                const double f = 1.0 - d * d * d;
                const double factor = f > 1.0 / 3.0 ? f : 1.0 / 3.0;
                s.lambda = lm_clamp((float)((double)s.lambda * factor), s.cfg.min_lambda, s.cfg.max_lambda);
            }
            s.cam = s.cam_new;
            s.cost = cost_new;
            s.v = 2;
            s.rebuild = 1;
            s.iterations++;
        }
        take_sums();
    } else {
        const bool at_max = s.lambda == s.cfg.max_lambda;
        lm_wave_sync();
        if (lane == 0) {
            s.invalid_steps++;
            if (at_max) {
                lm_finish(s);
            } else {
                s.lambda = fminf(s.cfg.max_lambda, s.lambda * s.v);
                s.v = 2 * s.v;
                s.rebuild = 0;
                s.iterations++;
            }
        }
        if (at_max) {
            lm_wave_sync();
            return;
        }
    }
    lm_advance_wave(s, lane);
}
#endif  // __HIPCC__

}  // namespace pc
