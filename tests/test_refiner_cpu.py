"""CPU: the refiner's oracle (oracle/refine_oracle.py) is self-consistent -- analytic Jacobians of
EvaluateWithJacobian (reference cpp/refiner.cc:363-506) against finite differences of Evaluate
(:274-361) -- and the host-side pieces that need no GPU: banded Cholesky, Python surface."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refine_scene as S  # noqa: E402
from refine_scene import po, ro  # noqa: E402


@pytest.fixture(scope="module")
def core():
    sys.path.insert(0, os.path.join(S.ROOT, "polychase_amd", "core"))
    try:
        import torch  # noqa: F401  (one HIP runtime per process)
    except Exception:
        pass
    import polychase_core
    return polychase_core


def _scene(opencv, n=6, seed=5):
    verts, tris = S.grid_mesh()
    model = np.diag([1.5, 1.2, 1.3, 1.0])
    model[:3, 3] = [0.1, -0.05, 0.2]
    truth = [S.true_camera(t, opencv) for t in range(1, n + 1)]
    kps, flows = S.make_flows(verts, tris, model, truth, 1, n_kp=60, noise=0.3, seed=seed)
    cams = S.perturbed(truth, np.random.default_rng(seed))
    seg = ro.load_segment(kps, flows, cams, 1, verts, model)
    return verts, tris, model, cams, seg


def _local_step(cam, k, eps):
    """camera with local parameter k (0-2 rotation, 3-5 translation, 6 fy, 7 cx, 8 cy) moved by eps"""
    d = np.zeros(9)
    d[k] = eps
    fy = cam.fy + d[6]
    return po.Camera(fx=fy * cam.aspect_ratio, fy=fy, cx=cam.cx + d[7], cy=cam.cy + d[8], aspect_ratio=cam.aspect_ratio,
                     width=cam.width, height=cam.height, opencv=cam.opencv, q=po.quat_step_post(cam.q, d[0:3]), t=cam.t + d[3:6])


@pytest.mark.parametrize("opencv", [False, True])
def test_oracle_jacobian_matches_finite_differences(opencv):
    verts, tris, model, cams, seg = _scene(opencv)
    mask = np.zeros(16, np.uint32)
    ro.total_cost(seg, cams, verts, tris, mask, model, "cauchy", 1.0)  # fills the triangle cache
    checked = 0
    for e, (i, j, kp_idx, _, _) in enumerate(seg.edges):
        if i in (0, seg.n_frames - 1) or j in (0, seg.n_frames - 1):
            continue
        J, r, valid = ro.edge_jacobians(seg, cams, e, verts, tris, model, True, True)
        cache0 = [c.copy() for c in seg.cache]
        for which, frame in ((0, i), (1, j)):
            for k in range(9):
                eps = 1e-6 if k < 6 else 1e-4
                rs = []
                same = np.ones(len(kp_idx), bool)
                for sgn in (1, -1):
                    moved = list(cams)
                    moved[frame] = _local_step(cams[frame], k, sgn * eps)
                    rr, vv = ro.edge_residuals(seg, moved, e, verts, tris, mask, model)
                    same &= vv & (seg.cache[i][kp_idx] == cache0[i][kp_idx])
                    seg.cache = [c.copy() for c in cache0]
                    rs.append(rr)
                fd = (rs[0] - rs[1]) / (2 * eps)
                ok = valid & same
                assert ok.sum() > 0.8 * valid.sum()
                scale = np.abs(J[ok][:, :, which * 9 + k]).max() + 1e-9
                assert np.abs(fd[ok] - J[ok][:, :, which * 9 + k]).max() <= 2e-4 * scale + 1e-5, (e, which, k)
                checked += 1
        if checked >= 72:
            break
    assert checked >= 72


def test_fixed_end_frames_have_no_jacobian():
    verts, tris, model, cams, seg = _scene(False)
    ro.total_cost(seg, cams, verts, tris, np.zeros(16, np.uint32), model, "cauchy", 1.0)
    JtJ, Jtr = ro.normal_equations(seg, cams, verts, tris, model, "cauchy", 1.0, False, False)
    assert not JtJ[:6].any() and not JtJ[-6:].any() and not Jtr[:6].any() and not Jtr[-6:].any()
    assert np.linalg.norm(Jtr[6:-6]) > 0
    assert np.allclose(JtJ, JtJ.T)
    # frames only connect to frames +-{1,2,4,8} away: block band structure
    B, n = 6, seg.n_frames
    for a in range(n):
        for b in range(n):
            if abs(a - b) not in (0, 1, 2, 4, 8):
                assert not JtJ[a * B:(a + 1) * B, b * B:(b + 1) * B].any()


def test_oracle_refinement_recovers_the_trajectory():
    verts, tris = S.grid_mesh()
    model = np.eye(4)
    truth = [S.true_camera(t) for t in range(1, 7)]
    kps, flows = S.make_flows(verts, tris, model, truth, 1, n_kp=80, noise=0.0)
    cams = S.perturbed(truth, np.random.default_rng(1))
    seg = ro.load_segment(kps, flows, cams, 1, verts, model)
    out, stats = ro.refine(seg, cams, verts, tris, np.zeros(16, np.uint32), model, max_iterations=30)
    assert stats["cost"] < 1e-3 * stats["initial_cost"]
    for c, t in zip(out, truth):
        assert S.angle(c.R(), t.R()) < 2e-4 and np.linalg.norm(c.t - t.t) < 2e-3


def test_banded_cholesky_matches_dense(core):
    rng = np.random.default_rng(0)
    n, bw = 90, 17
    A = np.zeros((n, n))
    for r in range(n):
        for c in range(max(0, r - bw), r + 1):
            A[r, c] = A[c, r] = rng.normal()
    A += np.eye(n) * (np.abs(A).sum(1).max() + 1.0)          # diagonally dominant -> SPD
    b = rng.normal(size=n)
    x = core._banded_llt_solve(A.astype(np.float32), bw, b.astype(np.float32))
    ref = np.linalg.solve(A.astype(np.float32).astype(np.float64), b.astype(np.float32).astype(np.float64))
    assert np.abs(x - ref).max() <= 2e-5 * np.abs(ref).max()
    A[40, 40] = -1.0                                          # not positive definite -> Eigen::NumericalIssue
    assert core._banded_llt_solve(A.astype(np.float32), bw, b.astype(np.float32)) is None


def test_python_surface(core):
    for field in ("progress", "message", "stats"):         # polychase_pybind.cc:305-308 (no constructor there either)
        assert hasattr(core.RefineTrajectoryUpdate, field)
    for name in ("request_stop", "join", "try_pop", "empty"):
        assert hasattr(core.RefinerThread, name)
    doc = core.refine_trajectory.__doc__
    for kw in ("database_path", "camera_trajectory", "model_matrix", "mesh", "optimize_focal_length",
               "optimize_principal_point", "callback", "bundle_opts"):
        assert kw in doc
