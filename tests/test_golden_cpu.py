"""CPU: the oracle against the committed golden vectors (tests/golden/oracle_small.npz, made by
tests/golden/make_oracle_fixtures.py).  A change in oracle/pc_oracle.c that alters any output bit
fails here before it can silently move the parity target."""
import os

import numpy as np

import oracle

import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["oracle_small.npz", "oracle_x86_small.npz"])
def test_oracle_reproduces_golden_vectors(name):
    G = np.load(os.path.join(GOLDEN, name))
    with oracle.emulation(int(G["emulation"]) if "emulation" in G else oracle.EMU_CANONICAL):
        _check_against(G)


def test_the_two_golden_files_pin_two_different_executions():
    """on the step-edge fixture the canonical order must NOT reproduce the x86 vectors (else the file pins nothing new)"""
    G = np.load(os.path.join(GOLDEN, "oracle_x86_small.npz"))
    g0 = oracle.rgb2gray(G["frames"][0])
    with oracle.emulation(oracle.EMU_CANONICAL):
        eig = oracle.min_eigen_val(g0)
        p = [oracle.Pyramid(oracle.rgb2gray(f), 10, 2) for f in G["frames"]]
        xy, st, err = oracle.lk(p[0], p[1], G["keypoints0"], oracle.flow_options(max_level=2))
    assert not np.array_equal(eig.view(np.uint32), G["min_eig0"].view(np.uint32))
    assert not np.array_equal(xy.view(np.uint32), G["lk_xy_1"].view(np.uint32))


def _check_against(G):
    frames = G["frames"]
    grays = [oracle.rgb2gray(f) for f in frames]
    assert np.array_equal(grays[0], G["gray0"])
    assert np.array_equal(oracle.min_eigen_val(grays[0]).view(np.uint32), G["min_eig0"].view(np.uint32))
    kps, _, ncand = oracle.gftt(grays[0], want_eig=True)
    assert ncand == int(G["n_candidates0"]) and np.array_equal(kps, G["keypoints0"])
    p = [oracle.Pyramid(g, 10, 2) for g in grays]
    assert p[0].num_levels == int(G["num_levels"])
    for l in range(p[0].num_levels):
        assert np.array_equal(p[0].image(l), G[f"level{l}"]) and np.array_equal(p[0].deriv(l), G[f"deriv{l}"])
    for k in (1, 2):
        xy, st, err = oracle.lk(p[0], p[k], kps, oracle.flow_options(max_level=2))
        assert np.array_equal(st, G[f"lk_status_{k}"])
        assert np.array_equal(xy.view(np.uint32), G[f"lk_xy_{k}"].view(np.uint32))
        assert np.array_equal(err.view(np.uint32), G[f"lk_err_{k}"].view(np.uint32))


def test_pnp_oracle_reproduces_golden_vector():
    """tests/golden/pnp_small.npz (make_pnp_fixture.py): the float64 restatement of SolvePnPIterative on a frozen
    problem.  float64 numpy on another machine may differ in the last bits: 1e-9 is far below anything the GPU
    comparison (1e-4) could notice, far above such noise."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pnp_oracle as po

    P = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pnp_small.npz"))
    fx, fy, cx, cy, ar, w, h = P["intrinsics"]
    init = po.Camera(fx=fx, fy=fy, cx=cx, cy=cy, aspect_ratio=ar, width=w, height=h, opencv=False, q=P["init_q_wxyz"],
                     t=P["init_t"])
    for kind in ("trivial", "huber", "cauchy"):
        cam, st = po.solve_pnp(P["X"], P["x"], init, kind=kind, scale=1.5)
        assert np.allclose(cam.q, P[f"{kind}_q_wxyz"], atol=1e-9, rtol=0)
        assert np.allclose(cam.t, P[f"{kind}_t"], atol=1e-9, rtol=0)
        want = P[f"{kind}_stats"]
        assert abs(st["cost"] - want[1]) <= 1e-9 * want[1] and st["iterations"] == int(want[2])
        assert st["inlier_ratio"] == want[3]
