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


class TrackSource(C.Structure):
    _fields_ = [("cam", RayCamera), ("keypoints_key", C.c_longlong), ("keypoints_xy", VP), ("n_keypoints", C.c_int),
                ("n_matches", C.c_int), ("idx_offset", C.c_size_t), ("tgt_offset", C.c_size_t)]


class TrackResult(C.Structure):
    _fields_ = [("pnp", SolveResult), ("n_matches", C.c_int), ("n_correspondences", C.c_int), ("rounds", C.c_int), ("lm_ticks", C.c_uint * 8),
                ("lm_begin_tick", C.c_ulonglong), ("lm_end_tick", C.c_ulonglong)]


def _block(flows):
    """[(idx, tgt), ...] -> one byte block with 16-byte aligned columns + the offsets (what csrc/host/track_sequence.cc builds)"""
    parts, offs, at = [], [], 0
    for idx, tgt in flows:
        io = (at + 15) // 16 * 16
        to = (io + idx.nbytes + 15) // 16 * 16
        offs.append((io, to))
        at = to + tgt.nbytes
        parts.append((io, idx.tobytes()))
        parts.append((to, tgt.tobytes()))
    blob = np.zeros(max(at, 16), np.uint8)
    for o, b in parts:
        blob[o:o + len(b)] = np.frombuffer(b, np.uint8)
    return blob, offs


def _solve_frame(L, ctx, s, mesh, cams, kps_list, flows, init, o, keys=None, model=None):
    model = np.eye(4, dtype=np.float32) if model is None else model
    blob, offs = _block(flows)
    src = (TrackSource * len(flows))()
    for k, ((idx, tgt), (io, to)) in enumerate(zip(flows, offs)):
        src[k].cam = cams[k]
        src[k].keypoints_key = -1 if keys is None else keys[k]
        src[k].keypoints_xy = _p(kps_list[k])
        src[k].n_keypoints = len(kps_list[k])
        src[k].n_matches = len(idx)
        src[k].idx_offset, src[k].tgt_offset = io, to
    r = TrackResult()
    L.pc_track_solve_frame.argtypes = [VP, VP, VP, VP, C.c_int, C.POINTER(TrackSource), C.c_int, VP, C.c_size_t, C.POINTER(PnPCamera),
                                       C.POINTER(SolveOptions), C.POINTER(TrackResult)]
    rc = L.pc_track_solve_frame(ctx._h, s, mesh, _p(model), 1, src, len(flows), _p(blob), blob.nbytes, C.byref(init), C.byref(o), C.byref(r))
    return rc, r


def test_solve_frame_in_one_call_against_its_building_blocks(env):
    """pc_track_solve_frame (one transfer, one ray-cast launch over all sources, the LM loop as ONE persistent launch) against
    pc_corr_set_append per source + pc_pnp_solve (round 4's 40 launches): the same world points bit for bit, the same
    correspondence count / inliers / iteration count, poses equal up to the order of the fp32 sums; misses, a 3-point frame,
    too few points, a bad index, loss types, repeated calls on one set (the barrier words come back zero)."""
    L, ctx, mesh, cam = env
    s = VP()
    assert L.pc_corr_set_create(ctx._h, C.byref(s)) == 0
    rng = np.random.default_rng(5)
    # three "source frames" with their own keypoints, two seen by a second camera shifted along x
    cam2 = RayCamera()
    cam2.dir_matrix[:] = [1, 0, 0, 0, 1, 0, 0, 0, 1]
    cam2.origin[:] = [0.4, -0.1, -5.2]
    cam2.fx = cam2.fy = 500.0
    cam2.cx, cam2.cy, cam2.unproject_sign = 320.0, 240.0, 1.0
    cams = [cam, cam2, cam]
    kps_list, flows, truth_X = [], [], []
    for k, n in enumerate((700, 133, 4099)):
        kps = rng.uniform([60, 40], [580, 440], (n, 2)).astype(np.float32)
        kps[::97] = [5000, 240]                                   # rays that miss the quad
        idx = rng.permutation(n).astype(np.uint32)[: n - 7]
        c = cams[k]
        # the target frame's camera is `cam` at origin (0, 0, -5): observation of the world point the source ray hits
        d = np.stack([(kps[idx, 0] - 320) / 500, (kps[idx, 1] - 240) / 500, np.ones(len(idx))], 1)
        o_ = np.array(list(c.origin), np.float64)
        X = o_ + d * (-o_[2] / d[:, 2:3])
        tgt = np.stack([500 * X[:, 0] / (X[:, 2] + 5) + 320, 500 * X[:, 1] / (X[:, 2] + 5) + 240], 1) + rng.normal(0, 0.2, (len(idx), 2))
        kps_list.append(kps)
        flows.append((idx, tgt.astype(np.float32)))
    init = PnPCamera()
    init.q_xyzw[:] = [0.004, -0.003, 0.002, 1.0]
    init.t[:] = [0.03, -0.02, 5.05]
    init.fx = init.fy = 500.0
    init.cx, init.cy, init.aspect_ratio, init.convention_opencv = 320.0, 240.0, 1.0, 1
    L.pc_pnp_solve.argtypes = [VP, VP, C.POINTER(PnPCamera), C.POINTER(SolveOptions), C.POINTER(SolveResult)]
    L.pc_pnp_problem_destroy.argtypes = [VP]
    L.pc_corr_set_download.argtypes = [VP, VP, VP, VP]
    L.pc_track_download_points.argtypes = [VP, VP, C.c_int, VP]
    for loss, opt_f in ((0, 0), (2, 0), (1, 1)):
        o = SolveOptions(max_iterations=100, initial_lambda=1e-5, min_lambda=1e-10, max_lambda=1e10, gradient_tol=1e-10,
                         step_tol=1e-8, loss_type=loss, loss_scale=1.0, optimize_focal_length=opt_f, optimize_principal_point=opt_f,
                         f_low=10, f_high=5000, cx_low=0, cx_high=640, cy_low=0, cy_high=480, max_inlier_error=2.0, rounds_hint=0)
        rc, r = _solve_frame(L, ctx, s, mesh, cams, kps_list, flows, init, o, keys=[11, 12, 13])
        assert rc == 0, L.pc_last_error()
        total = sum(len(i) for i, _ in flows)
        pts = np.zeros((total, 4), np.float32)
        assert L.pc_track_download_points(ctx._h, s, total, _p(pts)) == 0
        # the building blocks on a second set
        s2 = VP()
        assert L.pc_corr_set_create(ctx._h, C.byref(s2)) == 0
        for k, (idx, tgt) in enumerate(flows):
            assert _append(L, ctx, s2, mesh, cams[k], kps_list[k], idx, tgt, key=11 + k) == 0
        n = C.c_int()
        assert L.pc_corr_set_size(ctx._h, s2, C.byref(n)) == 0
        assert r.n_matches == total and r.n_correspondences == n.value == int(pts[:, 3].sum()) and 0 < total - n.value < 200
        w, x = np.zeros((n.value, 3), np.float32), np.zeros((n.value, 2), np.float32)
        assert L.pc_corr_set_download(ctx._h, s2, _p(w), _p(x)) == 0
        assert np.array_equal(w, pts[pts[:, 3] == 1, :3])                          # the same rays, the same arithmetic
        assert np.array_equal(x, np.concatenate([t for _, t in flows])[pts[:, 3] == 1])
        prob = VP()
        assert L.pc_pnp_problem_from_set(ctx._h, s2, C.byref(prob)) == 0
        r2 = SolveResult()
        assert L.pc_pnp_solve(ctx._h, prob, C.byref(init), C.byref(o), C.byref(r2)) == 0
        L.pc_pnp_problem_destroy(prob)
        L.pc_corr_set_destroy.argtypes = [VP]
        L.pc_corr_set_destroy(s2)
        a, b = r.pnp, r2
        # the two paths add the same terms in different orders: near convergence a step may be accepted in one and rejected in
        # the other, so the iteration counts may differ by a few -- the minimum they reach may not
        assert abs(a.iterations - b.iterations) <= 4 and a.inliers == b.inliers, (loss, a.iterations, b.iterations)
        assert abs(a.cost - b.cost) <= 2e-5 * b.cost and abs(a.initial_cost - b.initial_cost) <= 2e-5 * b.initial_cost
        # (a plane seen head-on leaves focal length and distance nearly interchangeable: with the intrinsics free the two
        # summation orders stop at different points of that flat valley -- same cost, poses within its width)
        tol = 100.0 if opt_f else 1.0
        assert np.allclose(list(a.camera.t), list(b.camera.t), atol=2e-5 * tol) and np.allclose(list(a.camera.q_xyzw), list(b.camera.q_xyzw), atol=2e-6 * tol)
        assert np.allclose([a.camera.fx, a.camera.fy, a.camera.cx, a.camera.cy], [b.camera.fx, b.camera.fy, b.camera.cx, b.camera.cy], rtol=2e-5 * tol)
        assert 2 <= r.rounds <= a.iterations + 1                                   # one sweep per evaluated parameter set, none after `done`
        assert np.allclose(list(a.camera.t), [0, 0, 5], atol=5e-2 if opt_f else 5e-3)
        # the same call again on the same set: bit-identical (the barrier words came back zero, sums in a fixed order)
        rc, rr = _solve_frame(L, ctx, s, mesh, cams, kps_list, flows, init, o, keys=[11, 12, 13])
        assert rc == 0 and bytes(rr)[:C.sizeof(TrackResult) - 48] == bytes(r)[:C.sizeof(TrackResult) - 48]   # all but the tick counters
    o = SolveOptions(max_iterations=100, initial_lambda=1e-5, min_lambda=1e-10, max_lambda=1e10, gradient_tol=1e-10, step_tol=1e-8,
                     loss_type=0, loss_scale=1.0, optimize_focal_length=1, optimize_principal_point=1, f_low=10, f_high=5000, cx_low=0,
                     cx_high=640, cy_low=0, cy_high=480, max_inlier_error=2.0, rounds_hint=0)
    # exactly 3 correspondences: solved, intrinsics NOT optimised although asked for (pnp_problem.h:34-35)
    kp3 = np.array([[100, 100], [500, 120], [300, 400], [5000, 5000]], np.float32)
    rc, r = _solve_frame(L, ctx, s, mesh, [cam], [kp3], [(np.array([0, 1, 2, 3], np.uint32), kp3.copy())], init, o)
    assert rc == 0 and r.n_matches == 4 and r.n_correspondences == 3
    assert (r.pnp.camera.fx, r.pnp.camera.cx, r.pnp.camera.cy) == (500.0, 320.0, 240.0) and r.pnp.cost < 1e-6
    # 2 correspondences / none / no sources: nothing solved, no error
    rc, r = _solve_frame(L, ctx, s, mesh, [cam], [kp3], [(np.array([0, 3, 1], np.uint32), kp3[:3].copy())], init, o)
    assert rc == 0 and r.n_correspondences == 2 and r.pnp.iterations == 0
    rc, r = _solve_frame(L, ctx, s, mesh, [cam], [kp3], [(np.array([3], np.uint32), kp3[:1].copy())], init, o)
    assert rc == 0 and r.n_correspondences == 0
    rc, r = _solve_frame(L, ctx, s, mesh, [], [], [], init, o)
    assert rc == 0 and r.n_correspondences == 0 and r.n_matches == 0
    # an index past the keypoints is reported (tracker.cc:61 CHECK_LT), and the next call is clean again
    rc, r = _solve_frame(L, ctx, s, mesh, [cam], [kp3], [(np.array([0, 1, 4], np.uint32), kp3[:3].copy())], init, o)
    assert rc != 0 and b"out of range" in L.pc_last_error()
    rc, r = _solve_frame(L, ctx, s, mesh, [cam], [kp3], [(np.array([0, 1, 2], np.uint32), kp3[:3].copy())], init, o)
    assert rc == 0 and r.n_correspondences == 3
    # argument checks: offsets outside the block, bad loss type, too many sources
    src = (TrackSource * 1)()
    src[0].cam, src[0].keypoints_key, src[0].keypoints_xy, src[0].n_keypoints, src[0].n_matches = cam, -1, _p(kp3), 4, 3
    src[0].idx_offset, src[0].tgt_offset = 0, 16
    blob = np.zeros(32, np.uint8)
    r = TrackResult()
    assert L.pc_track_solve_frame(ctx._h, s, mesh, _p(np.eye(4, dtype=np.float32)), 1, src, 1, _p(blob), 32, C.byref(init), C.byref(o), C.byref(r)) != 0
    assert b"inside the block" in L.pc_last_error()
    o.loss_type = 9
    rc, r = _solve_frame(L, ctx, s, mesh, [cam], [kp3], [(np.array([0, 1, 2], np.uint32), kp3[:3].copy())], init, o)
    assert rc != 0 and b"loss type" in L.pc_last_error()
    o.loss_type = 0
    rc, r = _solve_frame(L, ctx, s, mesh, [cam] * 9, [kp3] * 9, [(np.array([0], np.uint32), kp3[:1].copy())] * 9, init, o)
    assert rc != 0 and b"at most" in L.pc_last_error()
    L.pc_corr_set_destroy.argtypes = [VP]
    L.pc_corr_set_destroy(s)


# ---- "Refine Sequence" problem handed over in parts ------------------------------------------------------------------------
class RefineDesc(C.Structure):
    _fields_ = [("n_frames", C.c_int), ("n_edges", C.c_int), ("kp_offset", VP), ("kp_xy", VP), ("edge_src", VP), ("edge_tgt", VP),
                ("edge_offset", VP), ("res_src_kp", VP), ("res_tgt_xy", VP), ("edge_weight", VP), ("model_matrix", C.c_float * 16),
                ("model_matrix_inv", C.c_float * 16), ("block_len", C.c_int), ("optimize_focal_length", C.c_int),
                ("optimize_principal_point", C.c_int)]


class RefinePart(C.Structure):
    _fields_ = [("kp_xy", VP), ("n_keypoints", C.c_int64), ("res_src_kp", VP), ("res_tgt_xy", VP), ("n_residuals", C.c_int64)]


class RefineCamera(C.Structure):
    _fields_ = [("R", C.c_float * 9), ("t", C.c_float * 3), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("aspect_ratio", C.c_float), ("unproject_sign", C.c_float), ("reserved", C.c_float * 2)]


def _refine_problem(rng, n_frames=5, n_kp=40):
    """keypoints over the quad of `env` seen from z = -5 (OpenCV), every frame tracked into its two neighbours"""
    kp_offset = np.arange(n_frames + 1, dtype=np.int32) * n_kp
    kp_xy = (rng.uniform(-150, 150, (n_frames * n_kp, 2)) + [320, 240]).astype(np.float32)
    src, tgt, off, res_kp, res_xy = [], [], [0], [], []
    for f in range(n_frames):
        for g in (f - 1, f + 1):
            if not 0 <= g < n_frames:
                continue
            take = np.sort(rng.choice(n_kp, n_kp - 7, replace=False)).astype(np.uint32)
            src.append(f)
            tgt.append(g)
            res_kp.append(take)
            res_xy.append(kp_xy[f * n_kp + take] + rng.normal(0, 0.5, (len(take), 2)).astype(np.float32))
            off.append(off[-1] + len(take))
    arrays = dict(kp_offset=kp_offset, kp_xy=kp_xy, edge_src=np.array(src, np.int32), edge_tgt=np.array(tgt, np.int32),
                  edge_offset=np.array(off, np.int32), res_src_kp=np.concatenate(res_kp), res_tgt_xy=np.concatenate(res_xy).astype(np.float32),
                  edge_weight=np.ones(len(src), np.float32))
    cams = (RefineCamera * n_frames)()
    for f in range(n_frames):
        cams[f].R[:] = [1, 0, 0, 0, 1, 0, 0, 0, 1]
        cams[f].t[:] = [0.01 * f, 0, 5]
        cams[f].fx = cams[f].fy = 500.0
        cams[f].cx, cams[f].cy, cams[f].aspect_ratio, cams[f].unproject_sign = 320.0, 240.0, 1.0, 1.0
    return arrays, cams


def _refine_desc(a, with_large_arrays):
    d = RefineDesc()
    d.n_frames, d.n_edges = len(a["kp_offset"]) - 1, len(a["edge_src"])
    for name in ("kp_offset", "edge_src", "edge_tgt", "edge_offset", "edge_weight"):
        setattr(d, name, _p(a[name]))
    if with_large_arrays:
        for name in ("kp_xy", "res_src_kp", "res_tgt_xy"):
            setattr(d, name, _p(a[name]))
    eye = np.eye(4, dtype=np.float32).ravel()
    d.model_matrix[:] = eye
    d.model_matrix_inv[:] = eye
    d.block_len = 6
    return d


def _refine_parts(a, cut_frames):
    """the large arrays cut after the given frames (a part = a run of frames with the edges that start in them)"""
    bounds = [0] + list(cut_frames) + [len(a["kp_offset"]) - 1]
    parts = (RefinePart * (len(bounds) - 1))()
    keep = []
    for k in range(len(bounds) - 1):
        f0, f1 = bounds[k], bounds[k + 1]
        edges = np.nonzero((a["edge_src"] >= f0) & (a["edge_src"] < f1))[0]
        r0, r1 = (a["edge_offset"][edges[0]], a["edge_offset"][edges[-1] + 1]) if len(edges) else (0, 0)
        kp = np.ascontiguousarray(a["kp_xy"][a["kp_offset"][f0]:a["kp_offset"][f1]])
        rk = np.ascontiguousarray(a["res_src_kp"][r0:r1])
        rx = np.ascontiguousarray(a["res_tgt_xy"][r0:r1])
        keep += [kp, rk, rx]
        parts[k].kp_xy, parts[k].n_keypoints = _p(kp), len(kp)
        parts[k].res_src_kp, parts[k].res_tgt_xy, parts[k].n_residuals = _p(rk), _p(rx), len(rk)
    return parts, keep


def test_refine_problem_in_parts_equals_the_whole_and_checks_its_indices(env):
    L, ctx, mesh, _cam = env
    L.pc_refine_problem_create.argtypes = [VP, VP, C.POINTER(RefineDesc), C.POINTER(VP)]
    L.pc_refine_problem_create_parts.argtypes = [VP, VP, C.POINTER(RefineDesc), C.POINTER(RefinePart), C.c_int, C.POINTER(VP)]
    L.pc_refine_total_cost.argtypes = [VP, VP, C.POINTER(RefineCamera), C.c_int, C.c_float, C.POINTER(C.c_double)]
    L.pc_refine_normal_equations.argtypes = [VP, VP, C.POINTER(RefineCamera), C.c_int, C.c_float, VP, VP]
    L.pc_refine_problem_destroy.argtypes = [VP]
    a, cams = _refine_problem(np.random.default_rng(3))
    n_edges = len(a["edge_src"])

    def evaluate(prob):
        cost = C.c_double()
        assert L.pc_refine_total_cost(ctx._h, prob, cams, 2, 1.0, C.byref(cost)) == 0
        blocks = np.zeros(n_edges * (12 * 13 // 2 + 12))
        valid = np.zeros(n_edges, np.int32)
        assert L.pc_refine_normal_equations(ctx._h, prob, cams, 2, 1.0, _p(blocks), _p(valid)) == 0
        return cost.value, blocks, valid

    whole = VP()
    d = _refine_desc(a, True)
    assert L.pc_refine_problem_create(ctx._h, mesh, C.byref(d), C.byref(whole)) == 0, L.pc_last_error()
    want = evaluate(whole)
    L.pc_refine_problem_destroy(whole)
    assert want[0] > 0 and want[2].min() > 20
    d = _refine_desc(a, False)
    for cuts in ((), (2,), (1, 2, 4), (0, 3, 3)):          # also empty parts
        parts, keep = _refine_parts(a, cuts)
        prob = VP()
        assert L.pc_refine_problem_create_parts(ctx._h, mesh, C.byref(d), parts, len(parts), C.byref(prob)) == 0, L.pc_last_error()
        got = evaluate(prob)
        L.pc_refine_problem_destroy(prob)
        assert got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
        del keep

    # refused: large arrays in the description, parts that do not add up, a residual naming a keypoint its frame does not have
    prob = VP()
    parts, keep = _refine_parts(a, (2,))
    d_bad = _refine_desc(a, True)
    assert L.pc_refine_problem_create_parts(ctx._h, mesh, C.byref(d_bad), parts, 2, C.byref(prob)) != 0 and not prob.value
    assert L.pc_refine_problem_create_parts(ctx._h, mesh, C.byref(d), parts, 1, C.byref(prob)) != 0 and not prob.value
    assert b"parts hold" in L.pc_last_error()
    b = dict(a)
    b["res_src_kp"] = a["res_src_kp"].copy()
    edge = 3
    b["res_src_kp"][a["edge_offset"][edge] + 5] = 40          # frames have keypoints 0..39
    for make in ("whole", "parts"):
        if make == "whole":
            d_b = _refine_desc(b, True)
            rc = L.pc_refine_problem_create(ctx._h, mesh, C.byref(d_b), C.byref(prob))
        else:
            parts_b, keep_b = _refine_parts(b, (1, 3))
            rc = L.pc_refine_problem_create_parts(ctx._h, mesh, C.byref(d), parts_b, 3, C.byref(prob))
        assert rc != 0 and not prob.value
        assert b"edge 3 references a keypoint" in L.pc_last_error(), L.pc_last_error()
