"""pnp_oracle.py -- numpy (float64) restatement of the tracking path's numerics.  TEST
INFRASTRUCTURE ONLY (see oracle/pc_oracle.h for the rules): nothing in polychase_amd/ imports it.

Follows, line by line:
  * PnPProblem::Evaluate / EvaluateWithJacobian / Step     /root/reference/cpp/pnp/pnp_problem.h:52-131
  * CameraIntrinsics::Project[WithJac] / Unproject / IsBehind / GetBounds   cpp/pnp/types.h:65-192
  * Pose::ApplyWithJac, Skew                               cpp/pose.h:60-78, :151-159
  * QuatStepPost                                           cpp/pnp/quaternion.h:11-20
  * TrivialLoss / HuberLoss / CauchyLoss                   cpp/pnp/robust_loss.h:47-104
  * LevMarqDenseSolver::Solve / BuildNormalEquations / ComputeStep / TotalCost   cpp/pnp/lev_marq.h:132-356
  * SolvePnPIterative (inlier ratio)                       cpp/pnp/solvers.cc:11-48
  * ray/triangle intersection                              cpp/ray_casting.h:125-179
  * SolveFrame / TrackCameraTrajectory                     cpp/tracker.cc:36-192
Parity status: the reference computes in float32 with Eigen and TBB (summation order not
reproducible run to run); this restatement is float64, so the GPU path is compared within the
tolerances of SURVEY.md section 8(d) (rotation 1e-4 rad, translation 1e-4 * |t|), not bit-exactly.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field, replace

import numpy as np


def quat_to_R(q_wxyz):
    w, x, y, z = q_wxyz
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=np.float64)


def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx])


def R_to_quat(R):
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        return np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    i = int(np.argmax(np.diag(R)))
    j, k = (i + 1) % 3, (i + 2) % 3
    s = math.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
    q = np.zeros(4)
    q[0] = (R[k, j] - R[j, k]) / s
    q[1 + i] = 0.25 * s
    q[1 + j] = (R[j, i] + R[i, j]) / s
    q[1 + k] = (R[k, i] + R[i, k]) / s
    return q


def quat_step_post(q, w_delta):  # quaternion.h:11-20
    angle = np.linalg.norm(w_delta)
    if angle > 0:
        axis = w_delta / angle
        return quat_mul(q, np.concatenate([[math.cos(angle / 2)], axis * math.sin(angle / 2)]))
    return q


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=np.float64)


@dataclass
class Camera:
    fx: float
    fy: float
    cx: float
    cy: float
    aspect_ratio: float
    width: float
    height: float
    opencv: bool = False          # CameraConvention
    q: np.ndarray = field(default_factory=lambda: np.array([1.0, 0, 0, 0]))  # WXYZ
    t: np.ndarray = field(default_factory=lambda: np.zeros(3))

    def R(self):
        return quat_to_R(self.q)

    def bounds(self, min_fov=15.0, max_fov=160.0):  # types.h:156-192
        tmin, tmax = math.tan(math.radians(min_fov) / 2), math.tan(math.radians(max_fov) / 2)
        if not self.opencv:
            f_low, f_high = -(self.width / 2) / tmin, -(self.width / 2) / tmax
        else:
            f_high, f_low = (self.width / 2) / tmin, (self.width / 2) / tmax
        return f_low, f_high, 0.0, self.width, 0.0, self.height

    def unproject(self, xy):  # types.h:95-98
        s = 1.0 if self.opencv else -1.0
        xy = np.asarray(xy, dtype=np.float64)
        return s * np.stack([(xy[..., 0] - self.cx) / self.fx, (xy[..., 1] - self.cy) / self.fy, np.ones(xy.shape[:-1])], -1)

    def project_world(self, Xw):
        Z = Xw @ self.R().T + self.t
        return np.stack([self.fx * Z[:, 0] / Z[:, 2] + self.cx, self.fy * Z[:, 1] / Z[:, 2] + self.cy], 1), Z


def loss_fns(kind: str, scale: float):
    if kind == "trivial":
        return (lambda r2: r2), (lambda r2: np.ones_like(r2))
    if kind == "huber":
        def L(r2):
            r = np.sqrt(r2)
            return np.where(r2 <= scale * scale, r2, scale * (2 * r - scale))

        def W(r2):
            return np.where(r2 <= scale * scale, 1.0, scale / np.sqrt(np.maximum(r2, 1e-300)))
        return L, W
    sq = scale * scale
    return (lambda r2: sq * np.log1p(r2 / sq)), (lambda r2: np.maximum(np.finfo(np.float32).tiny, 1 / (1 + r2 / sq)))


def residuals(cam: Camera, X, x):
    """Evaluate: (r [n,2], behind [n])."""
    z, Z = cam.project_world(X)
    behind = (Z[:, 2] < 0) if cam.opencv else (Z[:, 2] > 0)
    return z - x, behind


def jacobians(cam: Camera, X, x, opt_f, opt_pp):
    """EvaluateWithJacobian: J [n,2,9], r [n,2]."""
    R = cam.R()
    n = len(X)
    Z = X @ R.T + cam.t
    z = np.stack([cam.fx * Z[:, 0] / Z[:, 2] + cam.cx, cam.fy * Z[:, 1] / Z[:, 2] + cam.cy], 1)
    J = np.zeros((n, 2, 9))
    for i in range(n):
        d = np.array([[cam.fx / Z[i, 2], 0, -cam.fx * Z[i, 0] / Z[i, 2] ** 2],
                      [0, cam.fy / Z[i, 2], -cam.fy * Z[i, 1] / Z[i, 2] ** 2]])
        J[i, :, 0:3] = d @ (R @ skew(-X[i]))
        J[i, :, 3:6] = d
        if opt_f:
            J[i, :, 6] = [cam.aspect_ratio * Z[i, 0] / Z[i, 2], Z[i, 1] / Z[i, 2]]
        if opt_pp:
            J[i, 0, 7] = 1.0
            J[i, 1, 8] = 1.0
    return J, z - x


def normal_equations(cam, X, x, kind, scale, opt_f, opt_pp):
    _, W = loss_fns(kind, scale)
    J, r = jacobians(cam, X, x, opt_f, opt_pp)
    w = W((r * r).sum(1))
    JtJ = np.einsum("n,nia,nib->ab", w, J, J)
    Jtr = np.einsum("n,nia,ni->a", w, J, r)
    return JtJ, Jtr


def total_cost(cam, X, x, kind, scale):
    L, _ = loss_fns(kind, scale)
    r, behind = residuals(cam, X, x)
    r2 = np.where(behind, np.inf, (r * r).sum(1))
    return float(L(r2).sum())


def step_camera(cam: Camera, dp, opt_f, opt_pp):  # pnp_problem.h:101-131
    new = replace(cam, q=quat_step_post(cam.q, dp[0:3]), t=cam.t + dp[3:6])
    f_low, f_high, cx_lo, cx_hi, cy_lo, cy_hi = cam.bounds()
    if opt_f:
        fy = cam.fy + dp[6]
        fx = fy * cam.aspect_ratio
        new.fy, new.fx = float(np.clip(fy, f_low, f_high)), float(np.clip(fx, f_low, f_high))
    if opt_pp:
        new.cx, new.cy = float(np.clip(cam.cx + dp[7], cx_lo, cx_hi)), float(np.clip(cam.cy + dp[8], cy_lo, cy_hi))
    return new


def solve_pnp(X, x, cam: Camera, kind="huber", scale=1.0, opt_f=False, opt_pp=False, max_iterations=100,
              gradient_tol=1e-10, step_tol=1e-8, initial_lambda=1e-5, min_lambda=1e-10, max_lambda=1e10,
              max_inlier_error=12.0):
    """LevMarqDenseSolver::Solve (lev_marq.h:132-228) + inlier ratio. Returns (camera, stats dict)."""
    X, x = np.asarray(X, np.float64), np.asarray(x, np.float64)
    n = len(X)
    opt_f, opt_pp = opt_f and n > 3, opt_pp and n > 3
    cost = total_cost(cam, X, x, kind, scale)
    stats = dict(initial_cost=cost, cost=cost, lam=initial_lambda, invalid_steps=0, iterations=0)
    v, rebuild = 2.0, True
    it = 0
    while it < max_iterations:
        if rebuild:
            JtJ, Jtr = normal_equations(cam, X, x, kind, scale, opt_f, opt_pp)
            diag = np.clip(np.diag(JtJ), 1e-6, 1e32)
            if np.linalg.norm(Jtr) < gradient_tol:
                break
        A = JtJ.copy()
        A[np.diag_indices(9)] = diag * (1 + stats["lam"])
        JtJ[np.diag_indices(9)] = diag
        try:
            Lc = np.linalg.cholesky(A)
            step = -np.linalg.solve(Lc.T, np.linalg.solve(Lc, Jtr))
        except np.linalg.LinAlgError:
            stats["invalid_steps"] += 1
            if stats["lam"] == max_lambda:
                break
            stats["lam"] = min(max_lambda, stats["lam"] * v)
            v, rebuild = 2 * v, False
            it += 1
            continue
        if np.linalg.norm(step) < step_tol:
            break
        new = step_camera(cam, step, opt_f, opt_pp)
        cost_new = total_cost(new, X, x, kind, scale)
        if cost_new < stats["cost"]:
            rho = (cost_new - stats["cost"]) / float(step @ (2 * Jtr + JtJ @ step))
            if rho > 0:
                stats["lam"] = float(np.clip(stats["lam"] * max(1 / 3, 1 - (2 * rho - 1) ** 3), min_lambda, max_lambda))
            cam, stats["cost"], v, rebuild = new, cost_new, 2.0, True
        else:
            stats["invalid_steps"] += 1
            if stats["lam"] == max_lambda:
                break
            stats["lam"] = min(max_lambda, stats["lam"] * v)
            v, rebuild = 2 * v, False
        it += 1
    stats["iterations"] = it
    r, behind = residuals(cam, X, x)
    r2 = np.where(behind, np.inf, (r * r).sum(1))
    stats["inlier_ratio"] = float((r2 < max_inlier_error ** 2).sum()) / n if max_inlier_error > 0 else 0.0
    return cam, stats


def raycast_closest(verts, tris, origin, dirs, mask_bits=None, check_mask=False):
    """Closest hit per ray with the reference's Moeller-Trumbore (ray_casting.h:125-179), float64.
    Returns (hit [n] bool, prim [n], u, v, t, pos [n,3])."""
    verts, dirs = np.asarray(verts, np.float64), np.asarray(dirs, np.float64)
    p1, e1, e2 = verts[tris[:, 0]], verts[tris[:, 1]] - verts[tris[:, 0]], verts[tris[:, 2]] - verts[tris[:, 0]]
    n = len(dirs)
    best_t = np.full(n, np.inf)
    best = np.full(n, -1)
    bu, bv = np.zeros(n), np.zeros(n)
    for k in range(len(tris)):
        c = np.cross(dirs, e2[k])
        det = c @ e1[k]
        ok = ~((det > -1e-10) & (det < 1e-10))
        inv = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
        s = origin - p1[k]
        u = inv * (c @ s)
        ok &= (u >= 0) & (u <= 1)
        qv = np.cross(s, e1[k])
        v = inv * (dirs @ qv)
        ok &= (v >= 0) & (u + v <= 1)
        t = inv * (e2[k] @ qv)
        ok &= (t >= 0) & (t < best_t)
        best_t, best = np.where(ok, t, best_t), np.where(ok, k, best)
        bu, bv = np.where(ok, u, bu), np.where(ok, v, bv)
    hit = best >= 0
    if check_mask and mask_bits is not None:
        masked = np.array([(mask_bits[b >> 5] >> (b & 31)) & 1 if b >= 0 else 0 for b in best], bool)
        hit &= ~masked
    idx = np.maximum(best, 0)
    pos = (1 - bu - bv)[:, None] * verts[tris[idx, 0]] + bu[:, None] * verts[tris[idx, 1]] + bv[:, None] * verts[tris[idx, 2]]
    return hit, best, bu, bv, best_t, pos
