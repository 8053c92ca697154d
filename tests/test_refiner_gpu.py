"""GPU: "Refine Sequence" (reference cpp/refiner.cc + the sparse half of cpp/pnp/lev_marq.h) -- the
per-edge cost / normal-equation kernels and the LM driver -- against the float64 numpy oracle
(oracle/refine_oracle.py) and analytic ground truth.  The reference computes in float32 with
unordered atomics, so the comparison is toleranced: cost 1e-4 relative, J^T J / J^T r 2e-3 of their
norm, refined poses 5e-4 rad / 5e-4 |t| against the oracle's own LM run."""
import os
import sys
import time

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refine_scene as S  # noqa: E402
from refine_scene import po, ro  # noqa: E402

pytestmark = pytest.mark.gpu
LOSS = {"Trivial": "trivial", "Huber": "huber", "Cauchy": "cauchy"}


@pytest.fixture(scope="module")
def core():
    import torch  # noqa: F401
    sys.path.insert(0, os.path.join(S.ROOT, "polychase_amd", "core"))
    import polychase_core
    return polychase_core


def _scene(core, tmp_path, n=12, opencv=False, noise=0.3, n_kp=300, seed=5, f_scale=1.0, model=None, rate=1.0):
    verts, tris = S.grid_mesh()
    if model is None:
        model = np.diag([0.7, 0.5, 0.6, 1.0])     # the mesh covers about a third of the image
        model[:3, 3] = [0.1, -0.05, 0.2]
    truth = [S.true_camera(t, opencv, rate) for t in range(1, n + 1)]
    kps, flows = S.make_flows(verts, tris, model, truth, 1, n_kp=n_kp, noise=noise, seed=seed)
    path = str(tmp_path / "flow.db")
    S.write_database(core, path, kps, flows)
    cams = S.perturbed(truth, np.random.default_rng(seed), f_scale=f_scale)
    kps, flows = S.read_database(core, path, 1, n)
    seg = ro.load_segment(kps, flows, cams, 1, verts, model)
    return dict(verts=verts, tris=tris, model=model, truth=truth, cams=cams, seg=seg, path=path)


def _opts(core, loss="Cauchy", scale=1.0, max_iterations=100):
    bo = core.BundleOptions()
    bo.loss_type = getattr(core.LossType, loss)
    bo.loss_scale = scale
    bo.max_iterations = max_iterations
    return bo


@pytest.mark.parametrize("loss,opt_f,opt_pp,opencv", [("Cauchy", False, False, False), ("Huber", True, True, False),
                                                       ("Trivial", True, False, True), ("Cauchy", False, True, True)])
def test_cost_and_normal_equations_match_oracle(core, tmp_path, loss, opt_f, opt_pp, opencv):
    sc = _scene(core, tmp_path, opencv=opencv)
    mesh = core.AcceleratedMesh(sc["verts"], sc["tris"])
    traj = S.to_core_trajectory(core, sc["cams"], 1)
    got = core._refinement_system(sc["path"], traj, sc["model"].astype(np.float32), mesh, opt_f, opt_pp, _opts(core, loss, 2.0))
    seg = sc["seg"]
    assert got["num_edges"] == len(seg.edges)
    assert got["num_keypoints"] == sum(len(k) for k in seg.kps)
    assert got["num_residuals"] == sum(len(e[2]) for e in seg.edges)
    assert got["num_keypoints"] < 12 * 300                      # the bbox filter dropped something
    mask = np.asarray(mesh.inner().masked_triangles)
    cost = ro.total_cost(seg, sc["cams"], sc["verts"], sc["tris"], mask, sc["model"], LOSS[loss], 2.0)
    JtJ, Jtr = ro.normal_equations(seg, sc["cams"], sc["verts"], sc["tris"], sc["model"], LOSS[loss], 2.0, opt_f, opt_pp)
    assert abs(got["cost"] - cost) <= 1e-4 * cost
    assert got["block_length"] == (9 if (opt_f or opt_pp) else 6)
    assert np.array_equal(got["JtJ"], got["JtJ"].T)
    assert np.linalg.norm(got["Jtr"] - Jtr) <= 2e-3 * np.linalg.norm(Jtr)
    # block by block: a relative Frobenius bound on the whole matrix would hide the small translation blocks
    B = got["block_length"]
    for a in range(seg.n_frames):
        for b in range(seg.n_frames):
            blk, ref = got["JtJ"][a * B:(a + 1) * B, b * B:(b + 1) * B], JtJ[a * B:(a + 1) * B, b * B:(b + 1) * B]
            assert np.linalg.norm(blk - ref) <= 2e-3 * np.linalg.norm(ref) + 1e-12, (a, b)
    # the same call twice gives the same bits (fixed reduction order, unlike the reference's atomics)
    again = core._refinement_system(sc["path"], traj, sc["model"].astype(np.float32), mesh, opt_f, opt_pp, _opts(core, loss, 2.0))
    assert again["cost"] == got["cost"] and np.array_equal(again["JtJ"], got["JtJ"]) and np.array_equal(again["Jtr"], got["Jtr"])


def test_masked_triangles_drop_their_residuals(core, tmp_path):
    sc = _scene(core, tmp_path, n=6, noise=0.0)
    mesh = core.AcceleratedMesh(sc["verts"], sc["tris"])
    traj = S.to_core_trajectory(core, sc["cams"], 1)
    for t in range(0, len(sc["tris"]), 3):
        mesh.inner_mut().mask_triangle(t)
    got = core._refinement_system(sc["path"], traj, sc["model"].astype(np.float32), mesh, False, False, _opts(core))
    mask = np.asarray(mesh.inner().masked_triangles)
    assert mask.any()
    seg = sc["seg"]
    cost = ro.total_cost(seg, sc["cams"], sc["verts"], sc["tris"], mask, sc["model"], "cauchy", 1.0)
    assert abs(got["cost"] - cost) <= 1e-4 * cost
    JtJ, Jtr = ro.normal_equations(seg, sc["cams"], sc["verts"], sc["tris"], sc["model"], "cauchy", 1.0, False, False)
    assert np.linalg.norm(got["Jtr"] - Jtr) <= 2e-3 * np.linalg.norm(Jtr)
    # everything masked: every ray cast misses (ray_casting.cc:106), no residual is valid
    for t in range(len(sc["tris"])):
        mesh.inner_mut().mask_triangle(t)
    got = core._refinement_system(sc["path"], traj, sc["model"].astype(np.float32), mesh, False, False, _opts(core))
    assert got["cost"] == 0.0 and not got["Jtr"].any() and not got["JtJ"].any()


@pytest.mark.parametrize("opt_f", [False, True])
def test_refine_trajectory_matches_oracle_and_truth(core, tmp_path, opt_f):
    sc = _scene(core, tmp_path, n=12, noise=0.05, f_scale=1.01 if opt_f else 1.0)
    mesh = core.AcceleratedMesh(sc["verts"], sc["tris"])
    traj = S.to_core_trajectory(core, sc["cams"], 1)
    updates = []

    def cb(u):
        updates.append((u.progress, u.message, u.stats.cost, u.stats.initial_cost, u.stats.iterations))
        return True

    core.refine_trajectory(sc["path"], traj, sc["model"].astype(np.float32), mesh, opt_f, False, cb, _opts(core, max_iterations=40))
    mask = np.asarray(mesh.inner().masked_triangles)
    want, stats = ro.refine(sc["seg"], sc["cams"], sc["verts"], sc["tris"], mask, sc["model"], opt_f=opt_f, max_iterations=40)
    got = S.from_core_trajectory(traj, sc["cams"])
    # callback protocol (refiner.cc:673-681): one update per LM step + a final one, message format
    assert len(updates) >= 3 and updates[-1][1].startswith("Cost: ") and "(Initial: " in updates[-1][1]
    assert updates[-1][1] == "Cost: %.2f (Initial: %.2f)" % (updates[-1][2], updates[-1][3])
    assert all(abs(p - it / 40) < 1e-6 for p, _, _, _, it in updates)
    costs = [u[2] for u in updates]
    assert all(b <= a for a, b in zip(costs, costs[1:]))
    assert abs(updates[0][3] - stats["initial_cost"]) <= 1e-4 * stats["initial_cost"]
    assert updates[-1][2] < 0.02 * updates[-1][3]
    assert abs(updates[-1][2] - stats["cost"]) <= 0.02 * stats["cost"] + 1e-6
    # end cameras untouched, the others close to the oracle's result and to the truth
    for k in (0, -1):
        assert np.allclose(got[k].q, sc["cams"][k].q, atol=1e-7) and np.allclose(got[k].t, sc["cams"][k].t, atol=1e-7)
    for g, w, t in zip(got, want, sc["truth"]):
        assert S.angle(g.R(), w.R()) <= 5e-4 and np.linalg.norm(g.t - w.t) <= 5e-4 * np.linalg.norm(w.t)
        assert S.angle(g.R(), t.R()) <= 2e-3 and np.linalg.norm(g.t - t.t) <= 2e-2
        if opt_f:
            assert abs(g.fy - w.fy) <= 2e-3 * abs(w.fy) and abs(g.fx - g.fy * g.aspect_ratio) < 1e-3


def test_refiner_thread_protocol_cancel_and_errors(core, tmp_path):
    sc = _scene(core, tmp_path, n=10, noise=0.05)
    mesh = core.AcceleratedMesh(sc["verts"], sc["tris"])
    model = sc["model"].astype(np.float32)

    def drain(th, limit=120.0):
        msgs, t0 = [], time.time()
        while time.time() - t0 < limit:
            m = th.try_pop()
            if m is None:
                time.sleep(0.001)
                continue
            msgs.append(m)
            if m is True:
                break
        th.join()
        assert th.empty()
        return msgs

    traj = S.to_core_trajectory(core, sc["cams"], 1)
    msgs = drain(core.RefinerThread(database_path=sc["path"], camera_trajectory=traj, model_matrix=model, mesh=mesh,
                                    optimize_focal_length=False, optimize_principal_point=False, bundle_opts=_opts(core)))
    assert msgs[-1] is True and all(isinstance(m, core.RefineTrajectoryUpdate) for m in msgs[:-1]) and len(msgs) > 3
    full = msgs[-2].stats
    assert full.cost < 0.05 * full.initial_cost
    got = S.from_core_trajectory(traj, sc["cams"])                 # the shared trajectory was refined in place
    assert S.angle(got[4].R(), sc["truth"][4].R()) < 2e-3

    # request_stop right away: the solver stops at its first callback (refiner_thread.h:62-67), which is followed
    # by the final report -- the whole LM run above takes milliseconds, so stopping "mid-way" would be a race
    traj2 = S.to_core_trajectory(core, sc["cams"], 1)
    th = core.RefinerThread(sc["path"], traj2, model, mesh, False, False, _opts(core))
    th.request_stop()
    msgs = drain(th)
    assert msgs[-1] is True and len(msgs) - 1 == 2 and msgs[-2].stats.iterations < full.iterations

    # a trajectory with a hole: CHECK(traj.IsFrameFilled(frame)) -> CppException, then True
    bad = core.CameraTrajectory(1, 10)
    for i, c in enumerate(sc["cams"]):
        if i != 3:
            bad.set(1 + i, traj.get(1 + i))
    msgs = drain(core.RefinerThread(sc["path"], bad, model, mesh, False, False, _opts(core)))
    assert isinstance(msgs[0], core.CppException) and "IsFrameFilled" in msgs[0].what() and msgs[-1] is True
    # two frames only: CHECK(traj.Count() > 2)
    short = core.CameraTrajectory(1, 2)
    short.set(1, traj.get(1))
    short.set(2, traj.get(2))
    with pytest.raises(RuntimeError, match="Count"):
        core.refine_trajectory(sc["path"], short, model, mesh, False, False, None, _opts(core))


def test_reader_thread_count_does_not_change_the_result(core, tmp_path, monkeypatch):
    """The segment is loaded by several read connections (POLYCHASE_DB_READERS) and joined in frame order: the
    system the GPU sees, and therefore the refined trajectory, is identical for 1, 3 and 8 readers."""
    sc = _scene(core, tmp_path, n=40, noise=0.1, n_kp=200, seed=5, model=np.eye(4), rate=0.3)
    mesh = core.AcceleratedMesh(sc["verts"], sc["tris"])
    results = []
    for readers in ("1", "3", "8"):
        monkeypatch.setenv("POLYCHASE_DB_READERS", readers)
        traj = S.to_core_trajectory(core, sc["cams"], 1)
        sysm = core._refinement_system(sc["path"], traj, np.eye(4, dtype=np.float32), mesh, False, False, _opts(core))
        core.refine_trajectory(sc["path"], traj, np.eye(4, dtype=np.float32), mesh, False, False, None, _opts(core, max_iterations=10))
        poses = np.array([np.concatenate([np.array(traj.get(f).pose.q), np.array(traj.get(f).pose.t)]) for f in range(1, 41)])
        results.append((sysm["num_edges"], sysm["num_keypoints"], sysm["num_residuals"], float(sysm["cost"]), poses))
    for r in results[1:]:
        assert r[:4] == results[0][:4]
        assert np.array_equal(r[4], results[0][4])


def test_refine_a_long_segment(core, tmp_path):
    """120 frames x 400 keypoints: ~0.3M residuals per sweep, block-banded system of 720 unknowns."""
    sc = _scene(core, tmp_path, n=120, noise=0.1, n_kp=400, seed=11, model=np.eye(4), rate=0.2)
    mesh = core.AcceleratedMesh(sc["verts"], sc["tris"])
    traj = S.to_core_trajectory(core, sc["cams"], 1)
    last = []
    t0 = time.time()
    core.refine_trajectory(sc["path"], traj, np.eye(4, dtype=np.float32), mesh, False, False,
                           lambda u: last.append(u.stats) or True, _opts(core, max_iterations=30))
    dt = time.time() - t0
    st = last[-1]
    print(f"refined 120 frames in {dt:.2f} s, {st.iterations} iterations, cost {st.initial_cost:.3f} -> {st.cost:.4f}")
    assert st.cost < 0.05 * st.initial_cost
    got = S.from_core_trajectory(traj, sc["cams"])
    err0 = max(S.angle(c.R(), t.R()) for c, t in zip(sc["cams"], sc["truth"]))
    err1 = max(S.angle(c.R(), t.R()) for c, t in zip(got, sc["truth"]))
    assert err1 < 0.2 * err0
