"""CPU: pieces of the tracking path that need no GPU -- the 9x9 float32 Cholesky known-answer
(the reference's only numeric KAT) and the numpy oracle's Jacobians against finite differences."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pnp_oracle as po  # noqa: E402


@pytest.fixture(scope="module")
def core():
    from polychase_amd import build
    build.build_all()
    import torch  # noqa: F401
    sys.path.insert(0, os.path.join(ROOT, "polychase_amd", "core"))
    import polychase_core
    return polychase_core


def test_llt9_known_answer(core):
    """cpp/examples/levmarq_ill_conditioned_float32_issue.cpp: cond(JtJ) ~ 4e10, float32 LLT.  Eigen's
    float32 result has residual 2.9e-3 and an expected cost change of the WRONG sign (+2.4e-4); the
    LM loop guards against that with `rho > 0` (lev_marq.h:189-197).  Our Cholesky must be at least as
    accurate as the reference's printed numbers."""
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "llt9_ill_conditioned.json")))
    A = np.zeros((9, 9), np.float32)
    for i, r in enumerate(d["jtj_lower_rows"]):
        A[i, :len(r)] = r
    b = np.array(d["jtr"], np.float32)
    Ad = A.copy()
    Ad[np.diag_indices(9)] += np.float32(d["lambda"])
    sym = lambda M: np.tril(M) + np.tril(M, -1).T
    step = -core._llt9_solve(Ad, b)
    residual = np.linalg.norm(sym(Ad) @ step + b)
    assert residual <= d["reference_float32_residual_norm"]
    exact = -np.linalg.solve(sym(Ad).astype(np.float64), b.astype(np.float64))
    exp_exact = exact @ (2 * b + sym(A).astype(np.float64) @ exact)
    exp_ours = float(step @ (2 * b + sym(A) @ step))
    assert exp_exact < 0
    assert abs(exp_ours - exp_exact) <= abs(d["reference_float32_expected_cost_change"] - exp_exact)
    with pytest.raises(RuntimeError):
        core._llt9_solve(-np.eye(9, dtype=np.float32), b)


@pytest.mark.parametrize("opencv", [False, True])
def test_oracle_jacobian_matches_finite_differences(opencv):
    rng = np.random.default_rng(0)
    s = 1.0 if opencv else -1.0
    cam = po.Camera(fx=s * 900.0, fy=s * 880.0, cx=320.0, cy=240.0, aspect_ratio=900 / 880, width=640,
                    height=480, opencv=opencv, q=po.R_to_quat(po.quat_to_R(np.array([0.98, 0.1, -0.12, 0.05]) /
                                                                          np.linalg.norm([0.98, 0.1, -0.12, 0.05]))),
                    t=np.array([0.1, -0.2, s * 5.0]))
    X = rng.uniform(-1, 1, (6, 3))
    x = rng.uniform(0, 640, (6, 2))
    J, r = po.jacobians(cam, X, x, True, True)
    eps = 1e-6
    for k in range(9):
        dp = np.zeros(9)
        dp[k] = eps
        c2 = po.step_camera(cam, dp, True, True)
        if k == 6:  # fx follows fy through the aspect ratio
            c2.fx = c2.fy * cam.aspect_ratio
        r2, _ = po.residuals(c2, X, x)
        fd = (r2 - r) / eps
        assert np.allclose(fd, J[:, :, k], rtol=1e-4, atol=1e-4), k


def test_struct_defaults_tracking(core):
    b = core.BundleOptions()
    assert (b.max_iterations, b.max_allowed_parallelism, b.loss_type, b.loss_scale) == (100, 8, core.LossType.Huber, 1.0)
    assert (b.gradient_tol, b.step_tol, b.initial_lambda, b.min_lambda, b.max_lambda, b.verbose) == pytest.approx(
        (1e-10, 1e-8, 1e-5, 1e-10, 1e10, False))
    p = core.Pose()
    assert np.array_equal(p.q, [1, 0, 0, 0]) and np.array_equal(p.t, [0, 0, 0])   # WXYZ
    p.q = np.array([0.5, 0.5, -0.5, 0.5], np.float32)
    assert np.array_equal(p.q, [0.5, 0.5, -0.5, 0.5])
    t = core.CameraTrajectory(first_frame_id=10, count=5)
    assert (t.first_frame(), t.last_frame(), t.count()) == (10, 14, 5)
    assert t.is_valid_frame(12) and not t.is_valid_frame(15) and not t.is_frame_filled(12)
    t.set(12, core.CameraState())
    assert t.is_frame_filled(12) and t.get(12) is not None and t.get(11) is None
    with pytest.raises(RuntimeError, match="not part of the MI355X hot-path build"):
        core.find_transformation()
