"""GPU: the tracking entry points of include/polychase_hip.h called directly through ctypes -- argument checks,
error codes and the small cases the polychase_core path never produces (empty set, index out of range, growth of the
set across appends)."""
import ctypes as C

import numpy as np
import pytest

from polychase_amd import hip

pytestmark = pytest.mark.gpu
VP = C.c_void_p


class RayCamera(C.Structure):
    _fields_ = [("dir_matrix", C.c_float * 9), ("origin", C.c_float * 3), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float), ("unproject_sign", C.c_float)]


class PnPCamera(C.Structure):
    _fields_ = [("q_xyzw", C.c_float * 4), ("t", C.c_float * 3), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("aspect_ratio", C.c_float), ("convention_opencv", C.c_int)]


class SolveOptions(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("initial_lambda", C.c_float), ("min_lambda", C.c_float),
                ("max_lambda", C.c_float), ("gradient_tol", C.c_float), ("step_tol", C.c_float), ("loss_type", C.c_int),
                ("loss_scale", C.c_float), ("optimize_focal_length", C.c_int), ("optimize_principal_point", C.c_int),
                ("f_low", C.c_float), ("f_high", C.c_float), ("cx_low", C.c_float), ("cx_high", C.c_float),
                ("cy_low", C.c_float), ("cy_high", C.c_float), ("max_inlier_error", C.c_float), ("rounds_hint", C.c_int)]


class SolveResult(C.Structure):
    _fields_ = [("camera", PnPCamera), ("iterations", C.c_int), ("invalid_steps", C.c_int), ("initial_cost", C.c_float),
                ("cost", C.c_float), ("lambda_", C.c_float), ("step_norm", C.c_float), ("grad_norm", C.c_float),
                ("inliers", C.c_int)]


def _p(a):
    return a.ctypes.data_as(VP)


@pytest.fixture(scope="module")
def env():
    L = hip.load()
    ctx = hip.Context(0)
    # one big quad in the plane z = 0, seen by an OpenCV-convention camera at z = -5 looking along +z
    verts = np.array([[-4, -4, 0], [4, -4, 0], [4, 4, 0], [-4, 4, 0]], np.float32)
    tris = np.array([[0, 1, 2], [0, 2, 3]], np.uint32)
    mesh = VP()
    assert L.pc_mesh_create(ctx._h, _p(verts), 4, _p(tris), 2, C.byref(mesh)) == 0
    cam = RayCamera()
    cam.dir_matrix[:] = [1, 0, 0, 0, 1, 0, 0, 0, 1]
    cam.origin[:] = [0, 0, -5]
    cam.fx = cam.fy = 500.0
    cam.cx, cam.cy, cam.unproject_sign = 320.0, 240.0, 1.0
    yield L, ctx, mesh, cam
    L.pc_mesh_destroy.argtypes = [VP]
    L.pc_mesh_destroy(mesh)
    ctx.close()


def _append(L, ctx, s, mesh, cam, kps, idx, tgt, key=-1, model=None):
    model = np.eye(4, dtype=np.float32) if model is None else model
    L.pc_corr_set_append.argtypes = [VP, VP, VP, C.POINTER(RayCamera), VP, C.c_longlong, VP, C.c_int, VP, VP, C.c_int, C.c_int]
    return L.pc_corr_set_append(ctx._h, s, mesh, C.byref(cam), _p(model), key, _p(kps), len(kps), _p(idx), _p(tgt), len(idx), 1)


def test_correspondence_set_small_cases_and_errors(env):
    L, ctx, mesh, cam = env
    s = VP()
    assert L.pc_corr_set_create(ctx._h, C.byref(s)) == 0
    n = C.c_int(-1)
    assert L.pc_corr_set_size(ctx._h, s, C.byref(n)) == 0 and n.value == 0
    prob = VP()
    assert L.pc_pnp_problem_from_set(ctx._h, s, C.byref(prob)) != 0          # empty set: no problem
    assert b"empty" in L.pc_last_error()
    # pixel (320, 240) hits the quad at the origin; pixel (5000, 240) flies past it
    kps = np.array([[320, 240], [5000, 240], [420, 240]], np.float32)
    idx = np.array([0, 1, 2, 0], np.uint32)
    tgt = np.array([[1, 2], [3, 4], [5, 6], [7, 8]], np.float32)
    model = np.array([[2, 0, 0, 10], [0, 1, 0, 20], [0, 0, 1, 30], [0, 0, 0, 1]], np.float32)
    assert _append(L, ctx, s, mesh, cam, kps, idx, tgt, key=7, model=model) == 0
    assert L.pc_corr_set_size(ctx._h, s, C.byref(n)) == 0 and n.value == 3          # the miss is dropped
    w, x = np.zeros((3, 3), np.float32), np.zeros((3, 2), np.float32)
    L.pc_corr_set_download.argtypes = [VP, VP, VP, VP]
    assert L.pc_corr_set_download(ctx._h, s, _p(w), _p(x)) == 0
    assert np.array_equal(x, tgt[[0, 2, 3]])                                         # match order kept
    assert np.allclose(w[0], [10, 20, 30]) and np.allclose(w[2], [10, 20, 30])       # model * (0, 0, 0)
    assert np.allclose(w[1], [2 * 1.0 + 10, 20, 30], atol=1e-5)                      # (100 px / 500) * 5 = 1 unit along x
    # many appends: the set grows and keeps what it holds; the cached key skips the keypoint upload (same result)
    big_idx = np.tile(np.array([0, 2], np.uint32), 40000)
    big_tgt = np.arange(2 * len(big_idx), dtype=np.float32).reshape(-1, 2)
    for _ in range(3):
        assert _append(L, ctx, s, mesh, cam, kps, big_idx, big_tgt, key=7, model=model) == 0
    assert L.pc_corr_set_size(ctx._h, s, C.byref(n)) == 0 and n.value == 3 + 3 * len(big_idx)
    w2, x2 = np.zeros((n.value, 3), np.float32), np.zeros((n.value, 2), np.float32)
    assert L.pc_corr_set_download(ctx._h, s, _p(w2), _p(x2)) == 0
    assert np.array_equal(x2[:3], x) and np.array_equal(w2[:3], w)
    assert np.array_equal(x2[3:3 + len(big_idx)], big_tgt) and np.array_equal(x2[-len(big_idx):], big_tgt)
    # an index past the keypoints is reported (tracker.cc:61 CHECK_LT), not read
    assert L.pc_corr_set_clear(ctx._h, s) == 0
    assert _append(L, ctx, s, mesh, cam, kps, np.array([0, 3], np.uint32), tgt[:2]) == 0
    assert L.pc_corr_set_size(ctx._h, s, C.byref(n)) != 0 and b"out of range" in L.pc_last_error()
    # argument checks
    assert L.pc_corr_set_clear(ctx._h, s) == 0
    assert _append(L, ctx, s, mesh, cam, kps, idx[:0], tgt[:0]) == 0                 # nothing to append is fine
    assert L.pc_corr_set_append(ctx._h, s, mesh, None, None, -1, None, 0, None, None, 1, 1) != 0
    L.pc_corr_set_destroy.argtypes = [VP]
    L.pc_corr_set_destroy(s)


def test_pnp_solve_on_a_set_and_bad_options(env):
    L, ctx, mesh, cam = env
    s = VP()
    assert L.pc_corr_set_create(ctx._h, C.byref(s)) == 0
    rng = np.random.default_rng(3)
    kps = rng.uniform([60, 40], [580, 440], (600, 2)).astype(np.float32)
    idx = np.arange(600, dtype=np.uint32)
    # the camera that cast the rays sees every world point at its own pixel: PnP from a nearby start must come back to it
    assert _append(L, ctx, s, mesh, cam, kps, idx, kps) == 0
    prob = VP()
    assert L.pc_pnp_problem_from_set(ctx._h, s, C.byref(prob)) == 0
    init = PnPCamera()
    init.q_xyzw[:] = [0.004, -0.003, 0.002, 1.0]
    init.t[:] = [0.03, -0.02, 5.05]          # view = inverse of the ray camera: R = I, t = (0, 0, 5)
    init.fx = init.fy = 500.0
    init.cx, init.cy, init.aspect_ratio, init.convention_opencv = 320.0, 240.0, 1.0, 1
    o = SolveOptions(max_iterations=100, initial_lambda=1e-5, min_lambda=1e-10, max_lambda=1e10, gradient_tol=1e-10,
                     step_tol=1e-8, loss_type=0, loss_scale=1.0, optimize_focal_length=0, optimize_principal_point=0,
                     f_low=10, f_high=5000, cx_low=0, cx_high=640, cy_low=0, cy_high=480, max_inlier_error=2.0, rounds_hint=0)
    r = SolveResult()
    L.pc_pnp_solve.argtypes = [VP, VP, C.POINTER(PnPCamera), C.POINTER(SolveOptions), C.POINTER(SolveResult)]
    assert L.pc_pnp_solve(ctx._h, prob, C.byref(init), C.byref(o), C.byref(r)) == 0
    assert r.cost < 1e-3 * r.initial_cost and r.iterations >= 2 and r.inliers == 600
    assert np.allclose(list(r.camera.t), [0, 0, 5], atol=2e-3) and abs(r.camera.q_xyzw[3]) > 0.99999
    # a small first batch of rounds forces the solver to continue after a read-back: same answer
    o.rounds_hint = 2
    r2 = SolveResult()
    assert L.pc_pnp_solve(ctx._h, prob, C.byref(init), C.byref(o), C.byref(r2)) == 0
    assert (r2.iterations, r2.cost, list(r2.camera.t), list(r2.camera.q_xyzw)) == (r.iterations, r.cost, list(r.camera.t),
                                                                                   list(r.camera.q_xyzw))
    o.loss_type = 7
    assert L.pc_pnp_solve(ctx._h, prob, C.byref(init), C.byref(o), C.byref(r2)) != 0 and b"loss type" in L.pc_last_error()
    L.pc_pnp_problem_destroy.argtypes = [VP]
    L.pc_pnp_problem_destroy(prob)
    L.pc_corr_set_destroy.argtypes = [VP]
    L.pc_corr_set_destroy(s)
