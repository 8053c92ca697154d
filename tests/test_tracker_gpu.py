"""GPU: the "Track Sequence" path (reference cpp/tracker.cc) -- batched ray casting, PnP accumulation
and the LM loop -- against the float64 numpy oracle (oracle/pnp_oracle.py) and analytic ground truth.
Tolerances: rotation angle <= 1e-4 rad, translation <= 1e-4 * |t| vs the oracle."""
import math
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pnp_oracle as po  # noqa: E402

pytestmark = pytest.mark.gpu
W, H, F = 960.0, 540.0, 1100.0


@pytest.fixture(scope="module")
def core():
    import torch  # noqa: F401
    sys.path.insert(0, os.path.join(ROOT, "polychase_amd", "core"))
    import polychase_core
    return polychase_core


def grid_mesh(n=20, size=4.0):
    xs = np.linspace(-size / 2, size / 2, n + 1)
    X, Y = np.meshgrid(xs, xs)
    Z = 0.3 * np.sin(1.3 * X) * np.cos(1.1 * Y)
    verts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1).astype(np.float32)
    tris = []
    for j in range(n):
        for i in range(n):
            a = j * (n + 1) + i
            tris += [[a, a + 1, a + n + 2], [a, a + n + 2, a + n + 1]]
    return verts, np.array(tris, np.uint32)


def rot(axis, ang):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    return po.quat_to_R(np.concatenate([[math.cos(ang / 2)], axis * math.sin(ang / 2)]))


def true_pose(t):
    """view (world->camera) of frame t: OpenGL camera 6 units in front of the mesh, orbiting slightly."""
    R = rot([0.2, 1.0, 0.1], 0.012 * t) @ rot([1, 0, 0], 0.004 * t)
    return R, np.array([0.03 * t, -0.02 * t, -6.0 + 0.01 * t])


def intr(core):
    return core.CameraIntrinsics(fx=-F, fy=-F, cx=W / 2, cy=H / 2, aspect_ratio=1.0, width=W, height=H,
                                 convention=core.CameraConvention.OpenGL)


def ocam(R, t):
    return po.Camera(fx=-F, fy=-F, cx=W / 2, cy=H / 2, aspect_ratio=1.0, width=W, height=H, opencv=False,
                     q=po.R_to_quat(R), t=np.asarray(t, float))


def view4(R, t):
    m = np.eye(4, dtype=np.float32)
    m[:3, :3], m[:3, 3] = R, t
    return m


def rays_object_space(cam: po.Camera, model, xy):
    inv = np.linalg.inv(view4(cam.R(), cam.t).astype(np.float64) @ model.astype(np.float64))
    return inv[:3, 3], cam.unproject(xy) @ inv[:3, :3].T


def test_ray_casting_matches_oracle(core):
    verts, tris = grid_mesh()
    mesh = core.AcceleratedMesh(verts, tris)
    model = np.diag([1.5, 1.5, 1.5, 1.0]).astype(np.float32)
    R, t = true_pose(3)
    st = core.SceneTransformations(model, view4(R, t), intr(core))
    rng = np.random.default_rng(0)
    xy = rng.uniform([0, 0], [W, H], (3000, 2)).astype(np.float32)
    hits = core._ray_cast_pixels(mesh, st, xy, True)
    origin, dirs = rays_object_space(ocam(R, t), model, xy)
    hit, prim, u, v, tt, pos = po.raycast_closest(verts, tris, origin, dirs)
    got = np.array([h is not None for h in hits])
    # rays grazing an edge may flip between float32 and float64: allow a handful
    assert (got != hit).sum() <= 3
    both = got & hit
    assert both.sum() > 1000
    gp = np.array([hits[i].pos for i in np.nonzero(both)[0]])
    assert np.abs(gp - pos[both]).max() < 2e-4
    same = np.array([hits[i].primitive_id for i in np.nonzero(both)[0]]) == prim[both]
    assert same.mean() > 0.995
    # single-ray API and the mask: a masked closest triangle is a miss (ray_casting.cc:104-106)
    i0 = int(np.nonzero(both)[0][0])
    h0 = core.ray_cast(mesh, st, xy[i0], True)
    assert h0 is not None and h0.primitive_id == hits[i0].primitive_id
    assert abs(np.linalg.norm(h0.normal) - 1) < 1e-5
    mesh.inner_mut().mask_triangle(h0.primitive_id)
    assert mesh.inner().is_triangle_masked(h0.primitive_id)
    assert core.ray_cast(mesh, st, xy[i0], True) is None
    assert core.ray_cast(mesh, st, xy[i0], False) is not None
    mesh.inner_mut().unmask_triangle(h0.primitive_id)
    assert core.ray_cast(mesh, st, xy[i0], True) is not None


def _angle(Ra, Rb):
    c = (np.trace(Ra.T @ Rb) - 1) / 2
    return math.acos(max(-1.0, min(1.0, c)))


@pytest.mark.parametrize("loss", ["Cauchy", "Huber", "Trivial"])
def test_pnp_matches_oracle_and_truth(core, loss):
    rng = np.random.default_rng(1)
    n = 2000
    Xw = rng.uniform(-2, 2, (n, 3))
    R, t = true_pose(5)
    cam_true = ocam(R, t)
    x, _ = cam_true.project_world(Xw)
    x += rng.normal(0, 0.3, x.shape)
    x[:50] += rng.uniform(-80, 80, (50, 2))          # outliers
    R0, t0 = true_pose(4)
    init = core.CameraState(intr(core), core.Pose())
    p = core.Pose()
    p.q = po.R_to_quat(R0).astype(np.float32)
    p.t = t0.astype(np.float32)
    init.pose = p
    bo = core.BundleOptions()
    bo.loss_type = getattr(core.LossType, loss)
    res = core._solve_pnp_iterative(Xw.astype(np.float32), x.astype(np.float32), init, bo, 12.0, False, False)
    ocam_res, ostats = po.solve_pnp(Xw.astype(np.float32), x.astype(np.float32), ocam(R0, t0), kind=loss.lower())
    Rg = po.quat_to_R(np.array(res.camera.pose.q, float))
    tg = np.array(res.camera.pose.t, float)
    assert _angle(Rg, ocam_res.R()) <= 1e-4
    assert np.linalg.norm(tg - ocam_res.t) <= 1e-4 * np.linalg.norm(ocam_res.t)
    assert abs(res.inlier_ratio - ostats["inlier_ratio"]) <= 2.0 / n
    assert res.bundle_stats.cost == pytest.approx(ostats["cost"], rel=2e-3)
    assert res.bundle_stats.initial_cost == pytest.approx(ostats["initial_cost"], rel=1e-3)
    if loss != "Trivial":   # robust losses recover the true pose despite the outliers
        assert _angle(Rg, R) < 2e-4 and np.linalg.norm(tg - t) < 2e-3


def test_pnp_intrinsics_refinement(core):
    rng = np.random.default_rng(2)
    Xw = rng.uniform(-2, 2, (1500, 3))
    R, t = true_pose(2)
    x, _ = ocam(R, t).project_world(Xw)
    k = intr(core)
    k.fx = k.fy = -F * 0.97          # wrong focal length to start from
    init = core.CameraState(k, core.Pose())
    p = core.Pose()
    p.q, p.t = po.R_to_quat(R).astype(np.float32), t.astype(np.float32)
    init.pose = p
    res = core._solve_pnp_iterative(Xw.astype(np.float32), x.astype(np.float32), init, core.BundleOptions(), 12.0, True, False)
    assert abs(res.camera.intrinsics.fy + F) < 0.5 and abs(res.camera.intrinsics.fx + F) < 0.5
    with pytest.raises(Exception, match="Assertion failed"):   # CHECK_GE(rows, 3), solvers.cc:55
        core._solve_pnp_iterative(Xw[:2].astype(np.float32), x[:2].astype(np.float32), init, core.BundleOptions(), 12.0, False, False)


def _build_flow_db(core, path, verts, tris, model, n_frames, n_kp=300, noise=0.0, seed=3, skips=(-8, -4, -2, -1, 1, 2, 4, 8)):
    """keypoints at random pixels; flow f -> f+s = analytic reprojection of the mesh point under the
    keypoint (so the true trajectory explains every match exactly when noise == 0)."""
    rng = np.random.default_rng(seed)
    db = core.Database(path)
    wverts = verts.astype(np.float64) * np.diag(model)[:3]
    kp = {}
    world = {}
    for f in range(1, n_frames + 1):
        R, t = true_pose(f)
        xy = np.floor(rng.uniform([20, 20], [W - 20, H - 20], (n_kp, 2))).astype(np.float32)
        origin, dirs = rays_object_space(ocam(R, t), model, xy)
        hit, _, _, _, _, pos = po.raycast_closest(verts, tris, origin, dirs)
        kp[f], world[f] = xy, (pos * np.diag(model)[:3], hit)
        db.write_keypoints(f, xy)
    for f in range(1, n_frames + 1):
        pw, hit = world[f]
        for s in skips:
            g = f + s
            if g < 1 or g > n_frames:
                continue
            R, t = true_pose(g)
            x2, Z = ocam(R, t).project_world(pw)
            ok = hit & (Z[:, 2] < 0) & (x2[:, 0] > 0) & (x2[:, 0] < W) & (x2[:, 1] > 0) & (x2[:, 1] < H)
            # keypoints that miss the mesh still produce (wrong) flows, as LK would
            idx = np.nonzero(ok | ~hit)[0].astype(np.uint32)
            tgt = np.where(hit[idx, None], x2[idx], kp[f][idx] + 1.0) + rng.normal(0, noise, (len(idx), 2))
            db.write_image_pair_flow(f, g, idx, tgt.astype(np.float32), np.zeros(len(idx), np.float32))
    db.close()
    return kp


def _oracle_track(path, core, verts, tris, model, frame_from, frame_to):
    """tracker.cc:36-192 in numpy float64."""
    db = core.Database(path)
    traj = {frame_from: ocam(*true_pose(frame_from))}
    step = 1 if frame_to > frame_from else -1
    for f in range(frame_from + step, frame_to + step, step):
        Xs, xs = [], []
        for src in db.find_optical_flows_to_image(f):
            if src not in traj:
                continue
            kps = db.read_keypoints(src)
            fl = db.read_image_pair_flow(src, f)
            origin, dirs = rays_object_space(traj[src], model, kps[fl.src_kps_indices])
            hit, _, _, _, _, pos = po.raycast_closest(verts, tris, origin, dirs)
            Xs.append((pos * np.diag(model)[:3])[hit])
            xs.append(fl.tgt_kps[hit])
        init = traj.get(f) or traj.get(f - 1) or traj.get(f + 1)
        cam, _ = po.solve_pnp(np.concatenate(Xs).astype(np.float32), np.concatenate(xs).astype(np.float32), init,
                              kind="cauchy", scale=1.0)
        traj[f] = cam
    db.close()
    return traj


@pytest.mark.parametrize("direction", ["forward", "backward"])
def test_track_sequence_on_synthetic_database(core, tmp_path, direction):
    verts, tris = grid_mesh()
    model = np.diag([1.5, 1.5, 1.5, 1.0]).astype(np.float32)
    n_frames = 14
    path = str(tmp_path / "flow.db")
    _build_flow_db(core, path, verts, tris, model, n_frames, noise=0.05)
    a, b = (1, n_frames) if direction == "forward" else (n_frames, 1)
    R0, t0 = true_pose(a)
    st = core.SceneTransformations(model, view4(R0, t0), intr(core))
    mesh = core.AcceleratedMesh(verts, tris)
    bo = core.BundleOptions()
    bo.loss_type = core.LossType.Cauchy          # what the addon uses (tracking.py:208-210)
    got = {}

    def cb(r):
        got[r.frame] = (po.quat_to_R(np.array(r.pose.q, float)), np.array(r.pose.t, float), r.inlier_ratio)
        return True

    core.track_sequence(path, a, b, st, mesh, cb, False, False, bo)
    oracle = _oracle_track(path, core, verts, tris, model, a, b)
    frames = range(2, n_frames + 1) if direction == "forward" else range(1, n_frames)
    assert sorted(got) == list(frames)
    for f in frames:
        Rg, tg, inl = got[f]
        Rt, tt = true_pose(f)
        assert _angle(Rg, oracle[f].R()) <= 1e-4, f                          # vs oracle: stated tolerance
        assert np.linalg.norm(tg - oracle[f].t) <= 1e-4 * np.linalg.norm(oracle[f].t), f
        assert _angle(Rg, Rt) < 1e-3 and np.linalg.norm(tg - tt) < 1e-2, f     # vs truth (0.05 px noise)
        assert inl > 0.9


def test_device_built_correspondences_equal_the_reference_loop(core, tmp_path):
    """pc_corr_set_append (gather + ray cast + model transform + ordered append on the GPU) against the loop of
    tracker.cc:52-92 done on the host with the batched ray cast: same points, same order, bit for bit -- with
    misses, masked triangles, several sources and a source frame that has no pose yet."""
    verts, tris = grid_mesh(n=16)
    model = np.array([[1.5, 0, 0, 0.1], [0, 1.4, 0, -0.2], [0, 0, 1.6, 0.05], [0, 0, 0, 1]], np.float32)
    n_frames, target = 12, 9
    path = str(tmp_path / "corr.db")
    _build_flow_db(core, path, verts, tris, np.diag([1.5, 1.4, 1.6, 1.0]).astype(np.float32), n_frames, n_kp=700, noise=0.3)
    masked = np.zeros((len(tris) + 31) // 32, np.uint32)
    for t in (3, 40, 41, 200, 333):
        masked[t >> 5] |= np.uint32(1 << (t & 31))
    mesh = core.AcceleratedMesh(verts, tris, masked)
    traj = core.CameraTrajectory(1, n_frames)
    filled = [f for f in range(1, n_frames + 1) if f not in (target, 11)]        # 11 -> 9 exists in the database: skipped
    for f in filled:
        R, t = true_pose(f)
        pose = core.Pose()
        pose.q, pose.t = po.R_to_quat(R).astype(np.float32), t.astype(np.float32)
        traj.set(f, core.CameraState(intr(core), pose))
    got_w, got_x = core._frame_correspondences(path, traj, model, target, mesh)

    db = core.Database(path)
    want_w, want_x = [], []
    for src in db.find_optical_flows_to_image(target):
        if src not in filled:
            continue
        kps = db.read_keypoints(src)
        fl = db.read_image_pair_flow(src, target)
        st = core.SceneTransformations(model, traj.get(src).pose._Rt4x4(), intr(core))
        hits = core._ray_cast_pixels(mesh, st, np.ascontiguousarray(kps[fl.src_kps_indices]), True)
        for h, x in zip(hits, fl.tgt_kps):
            if h is None:
                continue
            p = np.asarray(h.pos, np.float32)
            w = [np.float32(np.float32(np.float32(np.float32(model[r, 0] * p[0]) + np.float32(model[r, 1] * p[1])) +
                                       np.float32(model[r, 2] * p[2])) + model[r, 3]) for r in range(3)]
            want_w.append(w)
            want_x.append(x)
    db.close()
    want_w, want_x = np.array(want_w, np.float32), np.array(want_x, np.float32)
    assert len(want_w) > 1500 and len(want_w) < 700 * 7          # several sources; misses and masked hits dropped
    assert got_w.shape == want_w.shape and got_x.shape == want_x.shape
    assert np.array_equal(got_x.view(np.uint32), want_x.view(np.uint32))
    assert np.array_equal(got_w.view(np.uint32), want_w.view(np.uint32))


@pytest.mark.parametrize("loss,opt_f,opt_pp", [("Cauchy", False, False), ("Huber", True, True), ("Trivial", True, False)])
def test_device_resident_lm_equals_the_host_driven_loop(core, monkeypatch, loss, opt_f, opt_pp):
    """pc_pnp_solve keeps the LM state on the GPU and decides between sweeps in a one-lane kernel; with
    POLYCHASE_PNP_HOST_LM=1 the same loop runs on the host with one read-back per sweep.  Same algorithm, same fp32
    operations (only sin/cos of the rotation step and one cube come from different libraries): the trajectories of
    the two solvers agree step for step."""
    rng = np.random.default_rng(17)
    verts, tris = grid_mesh()
    R, t = true_pose(5)
    cam = ocam(R, t)
    Xw = rng.uniform([-2, -2, -0.3], [2, 2, 0.3], (4000, 3))
    x, Z = cam.project_world(Xw)
    ok = Z[:, 2] < 0
    Xw, x = Xw[ok], x[ok] + rng.normal(0, 0.4, (ok.sum(), 2))
    x[::37] += rng.uniform(-80, 80, x[::37].shape)                      # outliers for the robust losses
    init = core.CameraState()
    init.intrinsics = intr(core)
    if opt_f:
        k = init.intrinsics
        init.intrinsics = core.CameraIntrinsics(fx=k.fx * 1.02, fy=k.fy * 1.02, cx=k.cx + 3, cy=k.cy - 2, aspect_ratio=1.0,
                                                width=W, height=H, convention=core.CameraConvention.OpenGL)
    p = core.Pose()
    R0 = rot([0.3, 1, 0.2], 0.02) @ R
    p.q, p.t = po.R_to_quat(R0).astype(np.float32), (t + [0.05, -0.04, 0.1]).astype(np.float32)
    init.pose = p
    bo = core.BundleOptions()
    bo.loss_type = getattr(core.LossType, loss)
    bo.loss_scale = 1.5
    args = (Xw.astype(np.float32), x.astype(np.float32), init, bo, 12.0, opt_f, opt_pp)
    monkeypatch.delenv("POLYCHASE_PNP_HOST_LM", raising=False)
    dev = core._solve_pnp_iterative(*args)
    monkeypatch.setenv("POLYCHASE_PNP_HOST_LM", "1")
    host = core._solve_pnp_iterative(*args)
    sd, sh = dev.bundle_stats, host.bundle_stats
    assert sd.iterations == sh.iterations and sd.invalid_steps == sh.invalid_steps and sd.iterations >= 4
    assert sd.initial_cost == sh.initial_cost                          # the first sweep is the same kernel on the same input
    assert abs(sd.cost - sh.cost) <= 1e-5 * abs(sh.cost)
    assert np.allclose(np.array(dev.camera.pose.q), np.array(host.camera.pose.q), atol=2e-6)
    assert np.allclose(np.array(dev.camera.pose.t), np.array(host.camera.pose.t), atol=2e-5)
    assert abs(dev.camera.intrinsics.fy - host.camera.intrinsics.fy) <= 1e-4 * abs(host.camera.intrinsics.fy)
    assert dev.inlier_ratio == host.inlier_ratio


@pytest.mark.parametrize("loss", ["Trivial", "Huber", "Cauchy"])
def test_pnp_against_the_committed_golden_vector(core, loss):
    """tests/golden/pnp_small.npz: frozen inputs + the float64 restatement's result (committed, so the target cannot
    drift with the oracle).  Rotation <= 1e-4 rad, translation <= 1e-4 |t|."""
    P = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pnp_small.npz"))
    fx, fy, cx, cy, ar, w, h = (float(v) for v in P["intrinsics"])
    init = core.CameraState()
    init.intrinsics = core.CameraIntrinsics(fx=fx, fy=fy, cx=cx, cy=cy, aspect_ratio=ar, width=w, height=h,
                                            convention=core.CameraConvention.OpenGL)
    p = core.Pose()
    p.q, p.t = P["init_q_wxyz"].astype(np.float32), P["init_t"].astype(np.float32)
    init.pose = p
    bo = core.BundleOptions()
    bo.loss_type = getattr(core.LossType, loss)
    bo.loss_scale = 1.5
    res = core._solve_pnp_iterative(P["X"], P["x"], init, bo, 12.0, False, False)
    kind = loss.lower()
    Rw, tw = po.quat_to_R(P[f"{kind}_q_wxyz"]), P[f"{kind}_t"]
    assert _angle(po.quat_to_R(np.array(res.camera.pose.q, float)), Rw) <= 1e-4
    assert np.linalg.norm(np.array(res.camera.pose.t, float) - tw) <= 1e-4 * np.linalg.norm(tw)
    want = P[f"{kind}_stats"]
    assert abs(res.bundle_stats.cost - want[1]) <= 1e-3 * want[1]
    assert abs(res.inlier_ratio - want[3]) <= 1.0 / len(P["X"]) + 1e-6


def test_read_ahead_does_not_change_the_poses(core, tmp_path, monkeypatch):
    """The tracker reads the next frame's blobs ahead on a second connection (POLYCHASE_TRACK_PREFETCH=0 switches
    that off): same database, same poses bit for bit, forward and backward."""
    verts, tris = grid_mesh()
    model = np.diag([1.5, 1.5, 1.5, 1.0]).astype(np.float32)
    n_frames = 20
    path = str(tmp_path / "flow.db")
    _build_flow_db(core, path, verts, tris, model, n_frames, noise=0.05)
    mesh = core.AcceleratedMesh(verts, tris)
    bo = core.BundleOptions()
    bo.loss_type = core.LossType.Cauchy

    def run(a, b):
        R0, t0 = true_pose(a)
        st = core.SceneTransformations(model, view4(R0, t0), intr(core))
        got = {}
        core.track_sequence(path, a, b, st, mesh,
                            lambda r: got.update({r.frame: (np.array(r.pose.q), np.array(r.pose.t), r.inlier_ratio,
                                                            r.bundle_stats.iterations)}) or True, False, False, bo)
        return got

    for a, b in ((1, n_frames), (n_frames, 1)):
        monkeypatch.delenv("POLYCHASE_TRACK_PREFETCH", raising=False)
        ahead = run(a, b)
        monkeypatch.setenv("POLYCHASE_TRACK_PREFETCH", "0")
        plain = run(a, b)
        assert sorted(ahead) == sorted(plain) and len(ahead) == n_frames - 1
        for f in ahead:
            assert np.array_equal(ahead[f][0], plain[f][0]) and np.array_equal(ahead[f][1], plain[f][1]), f
            assert ahead[f][2:] == plain[f][2:], f


def test_tracker_thread_protocol_and_errors(core, tmp_path):
    import time
    verts, tris = grid_mesh()
    model = np.eye(4, dtype=np.float32)
    path = str(tmp_path / "flow.db")
    _build_flow_db(core, path, verts, tris, model, 6, n_kp=120)
    R0, t0 = true_pose(1)
    st = core.SceneTransformations(model, view4(R0, t0), intr(core))
    mesh = core.AcceleratedMesh(verts, tris)
    th = core.TrackerThread(path, 1, 6, st, mesh, False, False, core.BundleOptions())
    msgs, t_start = [], time.time()
    while time.time() - t_start < 60:
        m = th.try_pop()
        if m is None:
            time.sleep(0.001)
            continue
        msgs.append(m)
        if m is True:
            break
    th.join()
    assert [m.frame for m in msgs[:-1]] == [2, 3, 4, 5, 6] and msgs[-1] is True
    # frames beyond the database: "Could not track to frame" (tracker.cc:162-166)
    th = core.TrackerThread(path, 6, 9, st, mesh, False, False, core.BundleOptions())
    th.join()
    out = []
    while not th.empty():
        out.append(th.try_pop())
    assert isinstance(out[0], core.CppException) and "Not enough features" in out[0].what() and out[-1] is True


def test_c5_end_to_end_rendered_plane(core, tmp_path):
    """BASELINE config C5 in miniature: frames RENDERED from a textured mesh under a known camera
    trajectory -> GFTT + LK on the GPU -> SQLite -> ray casting + PnP on the GPU -> poses vs truth."""
    import torch
    import torch.nn.functional as Fn
    from polychase_amd import synth

    n_frames, sx, sy = 12, 9.0, 5.0
    tex = torch.from_numpy(synth.noise_canvas(1400, 800, margin=0, sigma=2.5))[None]      # 1,3,800,1400
    ys, xs = torch.meshgrid(torch.arange(int(H), dtype=torch.float64), torch.arange(int(W), dtype=torch.float64),
                            indexing="ij")
    d_cam = torch.stack([(xs - W / 2) / F, (ys - H / 2) / F, -torch.ones_like(xs)], -1)     # Unproject, OpenGL
    frames = []
    for f in range(1, n_frames + 1):
        R, t = true_pose(f)
        Rt = torch.from_numpy(R.T.copy())
        o = -(Rt @ torch.from_numpy(t))
        d = d_cam @ Rt.T
        s = -o[2] / d[..., 2]
        P = o + s[..., None] * d                                                        # hit on the plane z = 0
        grid = torch.stack([P[..., 0] / (sx / 2), P[..., 1] / (sy / 2)], -1)[None].float()
        img = Fn.grid_sample(tex, grid, mode="bicubic", padding_mode="border", align_corners=True)[0]
        frames.append(img.clamp(0, 255).round().to(torch.uint8).permute(1, 2, 0).contiguous().numpy())
    path = str(tmp_path / "c5.db")
    core.generate_optical_flow_database(core.VideoInfo(int(W), int(H), 1, n_frames), lambda fid: frames[fid - 1], None, path)
    verts = np.array([[-sx / 2, -sy / 2, 0], [sx / 2, -sy / 2, 0], [sx / 2, sy / 2, 0], [-sx / 2, sy / 2, 0]], np.float32)
    tris = np.array([[0, 1, 2], [0, 2, 3]], np.uint32)
    mesh = core.AcceleratedMesh(verts, tris)
    R1, t1 = true_pose(1)
    st = core.SceneTransformations(np.eye(4, dtype=np.float32), view4(R1, t1), intr(core))
    bo = core.BundleOptions()
    bo.loss_type = core.LossType.Cauchy
    got = {}
    core.track_sequence(path, 1, n_frames, st, mesh,
                        lambda r: got.update({r.frame: (po.quat_to_R(np.array(r.pose.q, float)), np.array(r.pose.t, float),
                                                        r.inlier_ratio)}) or True, False, False, bo)
    assert sorted(got) == list(range(2, n_frames + 1))
    for f, (Rg, tg, inl) in got.items():
        Rt_, tt = true_pose(f)
        assert _angle(Rg, Rt_) < 2e-3, (f, _angle(Rg, Rt_))
        assert np.linalg.norm(tg - tt) < 0.05, (f, np.linalg.norm(tg - tt))
        assert inl > 0.5    # keypoints on the smeared border outside the textured plane are outliers; the
                            # addon aborts below 0.25 (blender_addon/operators/tracking.py:286-289)


def _random_soup(n_tris, rng):
    """triangle soup with wildly different sizes, duplicates and degenerate (zero-area) triangles"""
    c = rng.uniform(-2, 2, (n_tris, 1, 3))
    size = 10 ** rng.uniform(-2.5, -0.3, (n_tris, 1, 1))
    verts = (c + rng.normal(0, 1, (n_tris, 3, 3)) * size).reshape(-1, 3).astype(np.float32)
    tris = np.arange(3 * n_tris, dtype=np.uint32).reshape(-1, 3)
    tris[5] = tris[4]                        # duplicate triangle: equal t, lowest index must win
    verts[tris[7]] = verts[tris[7][0]]       # degenerate
    return verts, tris


@pytest.mark.parametrize("kind,n_tris", [("grid", 2 * 150 * 150), ("soup", 30000), ("one", 1), ("two", 2)])
def test_bvh_ray_casting_equals_exhaustive_sweep(core, kind, n_tris):
    """The LBVH (Embree's role, ray_casting.cc:23-121) must return exactly what a sweep over every triangle
    returns: same hit flags, triangles, barycentrics and distances, bit for bit."""
    rng = np.random.default_rng(5)
    if kind == "grid":
        verts, tris = grid_mesh(150)
    else:
        verts, tris = _random_soup(max(n_tris, 8), rng)
        tris = tris[:n_tris] if kind in ("one", "two") else tris
    mesh = core.AcceleratedMesh(verts, tris)
    for t in range(0, len(tris), 7):
        mesh.inner_mut().mask_triangle(t)
    model = np.diag([1.5, 1.2, 1.3, 1.0]).astype(np.float32)
    for frame, opencv in ((3, False), (40, False)):
        R, t = true_pose(frame)
        st = core.SceneTransformations(model, view4(R, t), intr(core))
        xy = rng.uniform([-50, -50], [W + 50, H + 50], (60000, 2)).astype(np.float32)
        xy[:64] = np.floor(xy[:64])                     # integer pixels, like detected keypoints
        for check_mask in (False, True):
            a = core._ray_cast_pixels(mesh, st, xy, check_mask)
            b = core._ray_cast_pixels(mesh, st, xy, check_mask, exhaustive=True)
            hit_a = np.array([h is not None for h in a])
            hit_b = np.array([h is not None for h in b])
            assert np.array_equal(hit_a, hit_b)
            if kind in ("grid", "soup") and frame == 3:
                assert 0.05 < hit_a.mean() <= 1.0
            for ha, hb in zip(a, b):
                if ha is None:
                    continue
                assert ha.primitive_id == hb.primitive_id and ha.t == hb.t
                assert np.array_equal(ha.barycentric_coordinate, hb.barycentric_coordinate) and np.array_equal(ha.pos, hb.pos)


def test_llt9_known_answer_on_the_device_solver(core):
    """The reference's float32 LLT known-answer case (cpp/examples/levmarq_ill_conditioned_float32_issue.cpp:16-63,
    tests/golden/llt9_ill_conditioned.json) through the factorisation the device-resident LM actually uses
    (lm_cholesky9 in csrc/hip/pnp_lm.hpp, via pc_debug_llt9) -- the CPU test runs the same case on the host copy
    (tests/test_tracker_cpu.py); both must be at least as accurate as the numbers the reference prints, and equal."""
    import json
    from polychase_amd import hip
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "llt9_ill_conditioned.json")))
    A = np.zeros((9, 9), np.float32)
    for i, r in enumerate(d["jtj_lower_rows"]):
        A[i, :len(r)] = r
    b = np.array(d["jtr"], np.float32)
    Ad = A.copy()
    Ad[np.diag_indices(9)] += np.float32(d["lambda"])
    sym = lambda M: np.tril(M) + np.tril(M, -1).T
    ctx = hip.Context(0)
    L, x, ok = ctx.llt9(Ad, b)
    assert ok
    step = -x
    residual = np.linalg.norm(sym(Ad) @ step + b)
    assert residual <= d["reference_float32_residual_norm"]
    exact = -np.linalg.solve(sym(Ad).astype(np.float64), b.astype(np.float64))
    exp_exact = exact @ (2 * b + sym(A).astype(np.float64) @ exact)
    exp_dev = float(step @ (2 * b + sym(A) @ step))
    assert exp_exact < 0
    assert abs(exp_dev - exp_exact) <= abs(d["reference_float32_expected_cost_change"] - exp_exact)
    # the host copy (linalg.h CholeskyLower<9>) takes the same steps: same bits
    assert np.array_equal(x, core._llt9_solve(Ad, b))
    # a matrix that is not positive definite is reported, not factorised
    _, x0, ok0 = ctx.llt9(-np.eye(9, dtype=np.float32), b)
    assert not ok0 and not x0.any()
    ctx.close()


def test_ray_cast_from_python_while_a_tracker_thread_runs(core, tmp_path):
    """The reference's Embree ray_cast is thread safe, so the addon may cast rays (pin mode, mask painting) while a
    TrackerThread works.  Here both go through one shared GPU context: the host sections that use it are serialised
    (csrc/host/gpu_context.h).  Hammer ray_cast from this thread while tracking runs on the worker; the poses must equal
    those of an undisturbed run and every ray cast must give the undisturbed answer."""
    import time
    verts, tris = grid_mesh()
    model = np.eye(4, dtype=np.float32)
    path = str(tmp_path / "flow.db")
    n_frames = 16
    _build_flow_db(core, path, verts, tris, model, n_frames, n_kp=2000)
    R0, t0 = true_pose(1)
    st = core.SceneTransformations(model, view4(R0, t0), intr(core))
    mesh = core.AcceleratedMesh(verts, tris)
    quiet = {}
    core.track_sequence(path, 1, n_frames, st, mesh, lambda r: quiet.__setitem__(r.frame, (np.array(r.pose.q), np.array(r.pose.t))) or True,
                        False, False, core.BundleOptions())
    px = [np.array([W * (0.2 + 0.6 * k / 9), H * (0.3 + 0.4 * ((k * 7) % 10) / 9)], np.float32) for k in range(10)]
    want = [core.ray_cast(mesh, st, p, False) for p in px]
    assert any(h is not None for h in want)
    th = core.TrackerThread(path, 1, n_frames, st, mesh, False, False, core.BundleOptions())
    got, casts, t_start = {}, 0, time.time()
    done = False
    while not done and time.time() - t_start < 120:
        for p, w_ in zip(px, want):                     # ray casts between (and during) the worker's frames
            h = core.ray_cast(mesh, st, p, False)
            assert (h is None) == (w_ is None)
            if h is not None:
                assert h.primitive_id == w_.primitive_id and np.array_equal(np.array(h.pos), np.array(w_.pos))
            casts += 1
        while True:
            m = th.try_pop()
            if m is None:
                break
            if m is True:
                done = True
                break
            assert not isinstance(m, core.CppException), m.what() if isinstance(m, core.CppException) else ""
            got[m.frame] = (np.array(m.pose.q), np.array(m.pose.t))
    th.join()
    assert done and sorted(got) == sorted(quiet) and casts >= 10
    for f in quiet:
        assert np.array_equal(got[f][0], quiet[f][0]) and np.array_equal(got[f][1], quiet[f][1]), f


def test_pipelined_solve_frame_semantics(core, tmp_path, monkeypatch):
    """Round 5: SolveFrame is one transfer + two launches + one wait, and the host runs one frame ahead of the GPU (the launches of
    frame f + 1 are enqueued before the callback of frame f).  What the caller sees must be what the frame-by-frame loop of
    cpp/tracker.cc:133-192 shows: the same poses as the per-source building blocks (POLYCHASE_TRACK_FUSED=0: the cross-check path),
    a callback that stops the run at frame k (nothing after k is reported, the speculative launches of k + 1 are dropped, the next
    run is clean), an exception out of the callback, and "Not enough features" at the frame that has no flows."""
    verts, tris = grid_mesh()
    model = np.diag([1.5, 1.5, 1.5, 1.0]).astype(np.float32)
    n_frames = 16
    path = str(tmp_path / "flow.db")
    _build_flow_db(core, path, verts, tris, model, n_frames, noise=0.05)
    mesh = core.AcceleratedMesh(verts, tris)
    bo = core.BundleOptions()
    bo.loss_type = core.LossType.Cauchy

    def run(a, b, cb=None):
        R0, t0 = true_pose(a)
        st = core.SceneTransformations(model, view4(R0, t0), intr(core))
        got = {}

        def default_cb(r):
            got[r.frame] = (np.array(r.pose.q, float), np.array(r.pose.t, float), r.inlier_ratio, r.bundle_stats.iterations)
            return True if cb is None else cb(r)
        core.track_sequence(path, a, b, st, mesh, default_cb, False, False, bo)
        return got

    for a, b in ((1, n_frames), (n_frames, 1)):
        monkeypatch.delenv("POLYCHASE_TRACK_FUSED", raising=False)
        fused = run(a, b)
        again = run(a, b)
        monkeypatch.setenv("POLYCHASE_TRACK_FUSED", "0")
        blocks = run(a, b)
        monkeypatch.delenv("POLYCHASE_TRACK_FUSED", raising=False)
        # round 6: by default the frame after the one on the GPU is enqueued behind it and takes that frame's pose -- a source camera
        # and the initial guess -- from the device (pc_track_frame_launch_chained); POLYCHASE_TRACK_CHAIN=0 is round 5's schedule
        # (every camera made on the host): the same bits, forward and backward
        monkeypatch.setenv("POLYCHASE_TRACK_CHAIN", "0")
        unchained = run(a, b)
        monkeypatch.delenv("POLYCHASE_TRACK_CHAIN", raising=False)
        assert sorted(fused) == sorted(unchained)
        for f in fused:
            assert np.array_equal(fused[f][0], unchained[f][0]) and np.array_equal(fused[f][1], unchained[f][1]), f
            assert fused[f][2] == unchained[f][2] and fused[f][3] == unchained[f][3], f
        # round 6: the decision of an LM round is taken by a wavefront (pnp_lm.hpp: lm_consume_wave -- the 9x9 Cholesky, the
        # substitutions and the gain ratio dealt out over nine lanes, every operation of the serial code on the same operands in the
        # same order); POLYCHASE_TRACK_SERIAL_DECISION=1 is round 5's one-lane decision: the same bits, iteration counts included
        monkeypatch.setenv("POLYCHASE_TRACK_SERIAL_DECISION", "1")
        serial = run(a, b)
        monkeypatch.delenv("POLYCHASE_TRACK_SERIAL_DECISION", raising=False)
        assert sorted(fused) == sorted(serial)
        for f in fused:
            assert np.array_equal(fused[f][0], serial[f][0]) and np.array_equal(fused[f][1], serial[f][1]), f
            assert fused[f][2] == serial[f][2] and fused[f][3] == serial[f][3], f
        assert sorted(fused) == sorted(blocks) and len(fused) == n_frames - 1
        for f in fused:
            assert np.array_equal(fused[f][0], again[f][0]) and np.array_equal(fused[f][1], again[f][1]), f      # deterministic
            assert np.abs(fused[f][0] - blocks[f][0]).max() < 2e-6 and np.abs(fused[f][1] - blocks[f][1]).max() < 2e-5, f
            assert abs(fused[f][2] - blocks[f][2]) < 1e-3
    # the callback stops the run at frame 7: frames 2 .. 7 reported, nothing later (frame 8 was launched speculatively)
    seen = run(1, n_frames, cb=lambda r: r.frame < 7)
    assert sorted(seen) == [2, 3, 4, 5, 6, 7]
    full = run(1, n_frames)
    assert sorted(full) == list(range(2, n_frames + 1))
    for f in seen:
        assert np.array_equal(seen[f][0], full[f][0]) and np.array_equal(seen[f][1], full[f][1])
    # an exception out of the callback travels to the caller; the next run is clean

    class Boom(Exception):
        pass

    def boom(r):
        if r.frame == 5:
            raise Boom("stop")
        return True
    with pytest.raises(Exception, match="stop"):
        run(1, n_frames, cb=boom)
    after = run(1, n_frames)
    assert all(np.array_equal(after[f][0], full[f][0]) for f in full)
    # the persistent launch could not get its workgroups resident (another tenant holds the GPU; it gives the CUs back after 100 ms):
    # ONE retry of the frame (POLYCHASE_TRACK_TEST_LOSE_AT: the 4th finished frame behaves as if its launch had timed out) -- what
    # was queued behind it is dropped, the pipeline starts again: every frame reported once, the same bits
    monkeypatch.setenv("POLYCHASE_TRACK_TEST_LOSE_AT", "4")
    retried = run(1, n_frames)
    assert sorted(retried) == sorted(full)
    for f in full:
        assert np.array_equal(retried[f][0], full[f][0]) and np.array_equal(retried[f][1], full[f][1]), f
    # ... and if the retry is lost as well (POLYCHASE_TRACK_TEST_LOSE_TWICE), the frame and the rest of the run are solved with the
    # per-source building blocks -- every frame reported once, same poses --, and the runs of the next
    # POLYCHASE_TRACK_FUSED_BACKOFF_S seconds start there
    monkeypatch.setenv("POLYCHASE_TRACK_TEST_LOSE_TWICE", "1")
    monkeypatch.setenv("POLYCHASE_TRACK_FUSED_BACKOFF_S", "600")
    lost = run(1, n_frames)
    monkeypatch.delenv("POLYCHASE_TRACK_TEST_LOSE_AT", raising=False)
    monkeypatch.delenv("POLYCHASE_TRACK_TEST_LOSE_TWICE", raising=False)
    backed_off = run(1, n_frames)                      # no injected loss, but inside the back-off: the building blocks from frame 2 on
    monkeypatch.setenv("POLYCHASE_TRACK_FUSED", "0")
    blocks = run(1, n_frames)
    monkeypatch.delenv("POLYCHASE_TRACK_FUSED", raising=False)
    for f in full:
        assert np.array_equal(backed_off[f][0], blocks[f][0]) and np.array_equal(backed_off[f][1], blocks[f][1]), f
    core.release_cached_engine()                       # forgets the back-off
    monkeypatch.delenv("POLYCHASE_TRACK_FUSED_BACKOFF_S", raising=False)
    fused_again = run(1, n_frames)
    for f in full:
        assert np.array_equal(fused_again[f][0], full[f][0]) and np.array_equal(fused_again[f][1], full[f][1]), f
    assert sorted(lost) == sorted(full)
    for f in full:
        assert np.abs(lost[f][0] - full[f][0]).max() < 2e-6 and np.abs(lost[f][1] - full[f][1]).max() < 2e-5, f
    for f in (2, 3, 4):
        assert np.array_equal(lost[f][0], full[f][0])      # before the loss: the same path
    # a frame without any flow into it: reported as in the reference, after the callbacks of the frames before it
    db = core.Database(path)
    import sqlite3
    db.close()
    con = sqlite3.connect(path)
    con.execute("DELETE FROM optical_flow WHERE image_id_to = 9")
    con.commit()
    con.close()
    seen = {}
    with pytest.raises(RuntimeError, match="Could not track to frame: 9. Not enough features"):
        R0, t0 = true_pose(1)
        core.track_sequence(path, 1, n_frames, core.SceneTransformations(model, view4(R0, t0), intr(core)), mesh,
                            lambda r: seen.update({r.frame: 1}) or True, False, False, bo)
    assert sorted(seen) == [2, 3, 4, 5, 6, 7, 8]


def test_more_flows_into_a_frame_than_the_reference_skips_make(core, tmp_path, monkeypatch):
    """SolveFrame takes ANY number of flows into a frame (tracker.cc:43-50); the fused path holds eight sources (the skips
    of cpp/opticalflow.cc:76-77) and used to CHECK-fail on a database written with another skip set.  Eighteen skips: forward
    tracking reaches nine filled sources at frame 13 -- from there on the per-source building blocks solve (no abort), every frame is
    reported once, and the poses are those of a run that used the building blocks throughout (to the order of the fp32 sums)."""
    verts, tris = grid_mesh()
    model = np.diag([1.5, 1.5, 1.5, 1.0]).astype(np.float32)
    n_frames = 18
    path = str(tmp_path / "dense.db")
    _build_flow_db(core, path, verts, tris, model, n_frames, noise=0.05, skips=(-12, -10, -8, -6, -5, -4, -3, -2, -1, 1, 2, 3, 4, 5, 6, 8, 10, 12))
    mesh = core.AcceleratedMesh(verts, tris)
    bo = core.BundleOptions()

    def run():
        R0, t0 = true_pose(1)
        got = {}
        core.track_sequence(path, 1, n_frames, core.SceneTransformations(model, view4(R0, t0), intr(core)), mesh,
                            lambda r: got.update({r.frame: (np.array(r.pose.q, float), np.array(r.pose.t, float))}) or True, False, False, bo)
        return got
    fused = run()
    monkeypatch.setenv("POLYCHASE_TRACK_FUSED", "0")
    blocks = run()
    monkeypatch.delenv("POLYCHASE_TRACK_FUSED", raising=False)
    assert sorted(fused) == sorted(blocks) == list(range(2, n_frames + 1))
    for f in fused:
        assert np.abs(fused[f][0] - blocks[f][0]).max() < 5e-6 and np.abs(fused[f][1] - blocks[f][1]).max() < 5e-5, f


def test_what_a_run_keeps_for_the_next_one_does_not_leak_into_it(core, tmp_path, monkeypatch):
    """A TrackSequence call that ends normally parks its correspondence set (device arrays, a stream, and -- inside the set -- the
    keypoint arrays of its source frames CACHED BY FRAME ID) and its page-locked blocks for the next call (track_sequence.cc:
    Scratch::ParkedSet, PinnedPool; pc_corr_set_recycle).  The next call may read ANOTHER database with the same frame ids:
    its poses must be those of a process that kept nothing -- after a run on a different database, after a stopped run, after an
    exception -- bit for bit, and POLYCHASE_TRACK_CACHE=0 / release_cached_engine() must give the same again."""
    verts, tris = grid_mesh()
    model = np.diag([1.5, 1.5, 1.5, 1.0]).astype(np.float32)
    n_frames = 12
    path_a, path_b = str(tmp_path / "a.db"), str(tmp_path / "b.db")
    _build_flow_db(core, path_a, verts, tris, model, n_frames, n_kp=300, noise=0.05, seed=3)
    _build_flow_db(core, path_b, verts, tris, model, n_frames, n_kp=340, noise=0.05, seed=11)   # other keypoints under the same ids
    mesh = core.AcceleratedMesh(verts, tris)
    bo = core.BundleOptions()
    bo.loss_type = core.LossType.Cauchy

    def run(path, cb=None):
        R0, t0 = true_pose(1)
        st = core.SceneTransformations(model, view4(R0, t0), intr(core))
        got = {}

        def default_cb(r):
            got[r.frame] = (np.array(r.pose.q, float), np.array(r.pose.t, float), r.inlier_ratio, r.bundle_stats.iterations)
            return True if cb is None else cb(r)
        core.track_sequence(path, 1, n_frames, st, mesh, default_cb, False, False, bo)
        return got

    def same(x, y):
        return sorted(x) == sorted(y) and all(np.array_equal(x[f][0], y[f][0]) and np.array_equal(x[f][1], y[f][1]) and x[f][2:] == y[f][2:] for f in x)

    core.release_cached_engine()
    fresh_b = run(path_b)
    core.release_cached_engine()
    fresh_a = run(path_a)                       # parks a set that holds A's keypoints under the frame ids 1 .. 12
    assert not same(fresh_a, fresh_b)
    assert same(run(path_b), fresh_b)           # ... which B must not see
    assert same(run(path_a), fresh_a)
    run(path_a, cb=lambda r: r.frame < 6)       # a stopped run (its speculative launch dropped) parks too
    assert same(run(path_b), fresh_b)

    class Boom(Exception):
        pass

    def boom(r):
        raise Boom("stop")
    with pytest.raises(Exception, match="stop"):
        run(path_a, cb=boom)                    # an exception: the set is destroyed, not parked
    assert same(run(path_b), fresh_b)
    monkeypatch.setenv("POLYCHASE_TRACK_CACHE", "0")
    assert same(run(path_b), fresh_b) and same(run(path_a), fresh_a)
    monkeypatch.delenv("POLYCHASE_TRACK_CACHE")
    assert same(run(path_b), fresh_b)
    core.release_cached_engine()
    assert same(run(path_a), fresh_a)
