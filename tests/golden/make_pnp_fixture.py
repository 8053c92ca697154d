#!/usr/bin/env python3
"""Writes tests/golden/pnp_small.npz: a frozen PnP problem (400 correspondences with noise and outliers, an offset
initial camera) and what the float64 restatement of SolvePnPIterative (oracle/pnp_oracle.py, following
cpp/pnp/solvers.cc:11-71 and cpp/pnp/lev_marq.h:132-228) makes of it, for the three loss types.  The reference itself
cannot be built in this image (Eigen is absent).  tests/test_golden_cpu.py checks that the oracle still reproduces the
file, tests/test_tracker_gpu.py checks the GPU solver against it.

    python tests/golden/make_pnp_fixture.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pnp_oracle as po  # noqa: E402

W, H, F = 960.0, 540.0, 1100.0


def problem():
    rng = np.random.default_rng(20250928)
    axis = np.array([0.2, 1.0, 0.1]) / np.linalg.norm([0.2, 1.0, 0.1])
    ang = 0.06
    q = np.concatenate([[np.cos(ang / 2)], axis * np.sin(ang / 2)])
    cam = po.Camera(fx=-F, fy=-F, cx=W / 2, cy=H / 2, aspect_ratio=1.0, width=W, height=H, opencv=False, q=q,
                    t=np.array([0.15, -0.1, -5.95]))
    X = rng.uniform([-2, -1.5, -0.3], [2, 1.5, 0.3], (400, 3))
    x, _ = cam.project_world(X)
    x = x + rng.normal(0, 0.5, x.shape)
    x[::23] += rng.uniform(-60, 60, x[::23].shape)
    q0 = po.quat_mul(q, np.concatenate([[np.cos(0.01)], np.array([0.6, -0.3, 0.74]) / np.linalg.norm([0.6, -0.3, 0.74]) * np.sin(0.01)]))
    init = po.Camera(fx=-F, fy=-F, cx=W / 2, cy=H / 2, aspect_ratio=1.0, width=W, height=H, opencv=False, q=q0,
                     t=cam.t + np.array([0.04, -0.03, 0.08]))
    return X.astype(np.float32), x.astype(np.float32), init


def main():
    X, x, init = problem()
    out = dict(X=X, x=x, init_q_wxyz=init.q, init_t=init.t, intrinsics=np.array([-F, -F, W / 2, H / 2, 1.0, W, H]))
    for kind in ("trivial", "huber", "cauchy"):
        cam, st = po.solve_pnp(X, x, init, kind=kind, scale=1.5)
        out[f"{kind}_q_wxyz"], out[f"{kind}_t"] = cam.q, cam.t
        out[f"{kind}_stats"] = np.array([st["initial_cost"], st["cost"], st["iterations"], st["inlier_ratio"]])
        print(kind, st)
    np.savez_compressed(os.path.join(HERE, "pnp_small.npz"), **out)


if __name__ == "__main__":
    main()
