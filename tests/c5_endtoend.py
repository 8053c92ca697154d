#!/usr/bin/env python3
"""c5_endtoend.py -- BASELINE config C5 at full size, on a GPU box:

    frames RENDERED from a textured plane under a known camera trajectory
      -> generate_optical_flow_database   (GFTT + pyramidal LK on the GPU, SQLite)
      -> track_sequence                   (ray casting + PnP on the GPU; tracker.cc)
      -> refine_trajectory                (refiner.cc on the GPU)
    timings of every stage, pose error against the ground truth, and -- for the first frames -- against the CPU
    reference of the tracking step (float64 numpy restatement of tracker.cc, oracle/pnp_oracle.py) run on the same
    database.

    python tests/c5_endtoend.py [--width 1920 --height 1080 --frames 300 --oracle-frames 6 --out gpurun_out/c5.json]

Test infrastructure (it uses the oracle as the checker); tests/test_c5_gpu.py runs it at 1920x1080 with 40 frames
inside `pytest -m gpu` and asserts the pose tolerances; the miniature is
tests/test_tracker_gpu.py::test_c5_end_to_end_rendered_plane.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--oracle-frames", type=int, default=6, help="frames solved by the CPU reference as well")
    ap.add_argument("--oracle-stride", type=int, default=0,
                    help="> 0: every stride-th frame of the WHOLE clip is also solved by the CPU reference, from the GPU's poses of "
                         "its source frames and the GPU's pose of the frame before it as the initial guess")
    ap.add_argument("--oracle-workers", type=int, default=1, help="threads over the sampled frames of --oracle-stride")
    ap.add_argument("--refine-iterations", type=int, default=30)
    ap.add_argument("--refine-intrinsics", action="store_true", help="the refinement also optimises focal length and principal point (9 parameters per camera)")
    ap.add_argument("--refine-oracle-frames", type=int, default=0,
                    help="> 2: the float64 CPU restatement of the refinement sweeps (oracle/refine_oracle.py) is timed on a "
                         "sub-segment of that many frames in the middle of the clip (a CPU baseline; checker leg)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args(argv)
    out = run(a.width, a.height, a.frames, a.oracle_frames, a.oracle_stride, a.refine_iterations, a.oracle_workers, a.refine_oracle_frames,
              a.refine_intrinsics)
    print(json.dumps(out))
    sys.path.insert(0, os.path.join(ROOT, "polychase_amd", "core"))
    import polychase_core
    polychase_core.release_cached_engine()   # nothing of the library is alive when the interpreter (and a profiler) shut down
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(out, open(a.out, "w"), indent=1)
    return 0


def _oracle_pose_check(job):
    """one sampled frame of the C5 check, in a worker process: the float64 CPU reference of the tracking step (cpp/tracker.cc:36-131
    restated in oracle/pnp_oracle.py) from the GPU's poses of the frame's sources -> (rotation angle, relative translation error)
    against the GPU's pose of the frame"""
    path, f, src_cams, guess, q_gpu, t_gpu, verts, tris = job
    import pnp_oracle as po
    sys.path.insert(0, os.path.join(ROOT, "polychase_amd", "core"))
    import polychase_core as core
    db = core.Database(path)
    model = np.eye(4)
    Xs, xs_ = [], []
    for src, cam in src_cams.items():
        kps = db.read_keypoints(src)
        fl = db.read_image_pair_flow(src, f)
        view = np.eye(4, dtype=np.float32)                                  # the view matrix is float32 in the reference
        view[:3, :3], view[:3, 3] = cam.R(), cam.t
        inv = np.linalg.inv(view.astype(np.float64) @ model)                # GetRayObjectSpace, ray_casting.h:53-63
        origin, dirs = inv[:3, 3], cam.unproject(kps[fl.src_kps_indices]) @ inv[:3, :3].T
        hit, _, _, _, _, pos = po.raycast_closest(verts, tris, origin, dirs)
        Xs.append(pos[hit])
        xs_.append(fl.tgt_kps[hit])
    db.close()
    cam, _ = po.solve_pnp(np.concatenate(Xs).astype(np.float32), np.concatenate(xs_).astype(np.float32), guess, kind="cauchy", scale=1.0)
    # rotation between the two poses from the NORMALISED quaternions: QuatStepPost (cpp/pnp/quaternion.h:11-20) never
    # renormalises, so after hundreds of fp32 updates |q| is 1 + 1e-7, and the arccos-of-trace angle of the matrices
    # turns that into 4e-4 "rad" (it grew linearly with the frame number) although costs and translations agree
    qa, qb = np.asarray(q_gpu, float), np.asarray(cam.q, float)
    qa, qb = qa / np.linalg.norm(qa), qb / np.linalg.norm(qb)
    rel = po.quat_mul(np.array([qa[0], -qa[1], -qa[2], -qa[3]]), qb)
    return (2.0 * float(np.arctan2(np.linalg.norm(rel[1:]), abs(rel[0]))), float(np.linalg.norm(np.asarray(t_gpu, float) - cam.t) / np.linalg.norm(cam.t)))


def run(width=1920, height=1080, frames=300, oracle_frames=6, oracle_stride=0, refine_iterations=30, oracle_workers=1,
        refine_oracle_frames=0, refine_intrinsics=False) -> dict:
    """the whole of C5 -> the result object (bench.py's "c5" block calls this; main() prints it)"""
    import types
    a = types.SimpleNamespace(width=width, height=height, frames=frames, oracle_frames=oracle_frames, oracle_stride=oracle_stride,
                              refine_iterations=refine_iterations, oracle_workers=oracle_workers, refine_oracle_frames=refine_oracle_frames,
                              refine_intrinsics=refine_intrinsics)

    import torch
    import torch.nn.functional as Fn
    import pnp_oracle as po
    import test_tracker_gpu as T
    from polychase_amd import synth
    sys.path.insert(0, os.path.join(ROOT, "polychase_amd", "core"))
    import polychase_core as core

    W, H, n = a.width, a.height, a.frames
    F = 1100.0 * W / 960.0
    sx, sy = 9.0, 5.0
    dev = torch.device("cuda")

    def true_pose(f):
        """the trajectory of the miniature test, slowed down so that n frames stay on the plane"""
        s = 12.0 / max(n, 12)
        R = T.rot([0.2, 1.0, 0.1], 0.012 * s * f) @ T.rot([1, 0, 0], 0.004 * s * f)
        return R, np.array([0.03 * s * f, -0.02 * s * f, -6.0 + 0.01 * s * f])

    def intr():
        return core.CameraIntrinsics(fx=-F, fy=-F, cx=W / 2, cy=H / 2, aspect_ratio=1.0, width=float(W), height=float(H),
                                     convention=core.CameraConvention.OpenGL)

    def ocam(R, t):
        return po.Camera(fx=-F, fy=-F, cx=W / 2, cy=H / 2, aspect_ratio=1.0, width=float(W), height=float(H), opencv=False,
                         q=po.R_to_quat(R), t=np.asarray(t, float))

    # ---- render ----
    t0 = time.time()
    tex = torch.from_numpy(synth.noise_canvas(int(1400 * W / 960), int(800 * W / 960), margin=0, sigma=2.5))[None].to(dev)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64, device=dev), torch.arange(W, dtype=torch.float64, device=dev),
                            indexing="ij")
    d_cam = torch.stack([(xs - W / 2) / F, (ys - H / 2) / F, -torch.ones_like(xs)], -1)
    frames = []
    for f in range(1, n + 1):
        R, t = true_pose(f)
        Rt = torch.from_numpy(R.T.copy()).to(dev)
        o = -(Rt @ torch.from_numpy(t).to(dev))
        d = d_cam @ Rt.T
        s = -o[2] / d[..., 2]
        P = o + s[..., None] * d
        grid = torch.stack([P[..., 0] / (sx / 2), P[..., 1] / (sy / 2)], -1)[None].float()
        img = Fn.grid_sample(tex, grid, mode="bicubic", padding_mode="border", align_corners=True)[0]
        frames.append(img.clamp(0, 255).round().to(torch.uint8).permute(1, 2, 0).contiguous())
    torch.cuda.synchronize()
    t_render = time.time() - t0

    out = {"config": f"C5 {W}x{H} {n} frames, rendered textured plane, OpenGL camera f={F:.0f}px", "render_s": round(t_render, 2)}
    td = tempfile.mkdtemp(prefix="c5_", dir="/tmp")
    path = os.path.join(td, "c5.db")

    # ---- analysis ----
    fo = core.OpticalFlowOptions()
    core.generate_optical_flow_database(core.VideoInfo(W, H, 1, 12), lambda f: frames[f - 1], None, "", core.GFTTOptions(), fo)
    t0 = time.time()
    st = core.generate_optical_flow_database(core.VideoInfo(W, H, 1, n), lambda f: frames[f - 1], None, path, core.GFTTOptions(), fo)
    dt = time.time() - t0
    out["analysis"] = {"seconds": round(dt, 3), "fps": n / dt, "db_bytes": os.path.getsize(path), "seconds_db": st.seconds_db}
    del frames
    torch.cuda.empty_cache()

    # ---- tracking ----
    verts = np.array([[-sx / 2, -sy / 2, 0], [sx / 2, -sy / 2, 0], [sx / 2, sy / 2, 0], [-sx / 2, sy / 2, 0]], np.float32)
    tris = np.array([[0, 1, 2], [0, 2, 3]], np.uint32)
    mesh = core.AcceleratedMesh(verts, tris)
    R1, t1 = true_pose(1)
    scene = core.SceneTransformations(np.eye(4, dtype=np.float32), T.view4(R1, t1), intr())
    bo = core.BundleOptions()
    bo.loss_type = core.LossType.Cauchy
    got, lm_iters = {}, []

    def cb(r):
        got[r.frame] = (np.array(r.pose.q, float), np.array(r.pose.t, float), r.inlier_ratio)
        lm_iters.append(r.bundle_stats.iterations)
        return True

    t0 = time.time()
    core.track_sequence(path, 1, n, scene, mesh, cb, False, False, bo)
    dt = time.time() - t0
    first_stages = {k: [round(v[0], 3), v[1]] for k, v in core._stage_report("TrackCameraTrajectory").items()}
    # the same call again: what a run keeps for the next one (the parked correspondence set, page-locked blocks) is there now
    first_got = dict(got)
    got.clear()
    lm_iters_first = list(lm_iters)
    lm_iters.clear()
    t0 = time.time()
    core.track_sequence(path, 1, n, scene, mesh, cb, False, False, bo)
    dt_again = time.time() - t0
    again_stages = {k: [round(v[0], 3), v[1]] for k, v in core._stage_report("TrackCameraTrajectory").items()}
    same_again = sorted(got) == sorted(first_got) and all(np.array_equal(got[f][0], first_got[f][0]) and np.array_equal(got[f][1], first_got[f][1]) for f in got)
    lm_iters[:] = lm_iters_first

    def errors(poses):
        ang, tr = [], []
        for f, (q, t) in poses.items():
            Rt_, tt = true_pose(f)
            ang.append(T._angle(po.quat_to_R(q), Rt_))
            tr.append(float(np.linalg.norm(t - tt)))
        return {"rotation_rad_max": max(ang), "rotation_rad_mean": float(np.mean(ang)), "translation_max": max(tr),
                "translation_mean": float(np.mean(tr))}

    db = core.Database(path)
    n_kp = [len(db.read_keypoints(f)) for f in (1, n // 2, n)]
    db.close()
    out["tracking"] = {"seconds": round(dt, 3), "frames_per_s": (n - 1) / dt, "mean_lm_iterations": float(np.mean(lm_iters)),
                       "min_inlier_ratio": min(v[2] for v in got.values()), "keypoints_per_frame": n_kp,
                       "vs_truth": errors({f: (q, t) for f, (q, t, _) in got.items()}),
                       # where the call's time went (csrc/host/stage_clock.h): host stages in ms, the LM kernel's own phases, counts
                       "stages": first_stages,
                       # the same call a second time in this process (csrc/host/track_sequence.cc: what a run keeps for the next)
                       "second_call": {"seconds": round(dt_again, 3), "frames_per_s": (n - 1) / dt_again, "same_poses_bit_for_bit": bool(same_again),
                                       "start_up_ms": {k.split(": ", 1)[1]: v[0] for k, v in again_stages.items() if "start-up" in k}}}

    # ---- the CPU reference of the tracking step on the same database (first frames) ----
    k = min(a.oracle_frames, n - 1)
    if k > 0:
        t0 = time.time()
        db = core.Database(path)
        traj = {1: ocam(*true_pose(1))}
        model = np.eye(4)
        for f in range(2, 2 + k):
            Xs, xs_ = [], []
            for src in db.find_optical_flows_to_image(f):
                if src not in traj:
                    continue
                kps = db.read_keypoints(src)
                fl = db.read_image_pair_flow(src, f)
                origin, dirs = T.rays_object_space(traj[src], model, kps[fl.src_kps_indices])
                hit, _, _, _, _, pos = po.raycast_closest(verts, tris, origin, dirs)
                Xs.append(pos[hit])
                xs_.append(fl.tgt_kps[hit])
            cam, _ = po.solve_pnp(np.concatenate(Xs).astype(np.float32), np.concatenate(xs_).astype(np.float32),
                                  traj.get(f) or traj[f - 1], kind="cauchy", scale=1.0)
            traj[f] = cam
        db.close()
        dt_o = time.time() - t0
        ang = [T._angle(po.quat_to_R(got[f][0]), traj[f].R()) for f in range(2, 2 + k)]
        tr = [float(np.linalg.norm(got[f][1] - traj[f].t) / np.linalg.norm(traj[f].t)) for f in range(2, 2 + k)]
        out["tracking"]["vs_cpu_reference"] = {"frames": k, "rotation_rad_max": max(ang), "translation_rel_max": max(tr),
                                               "cpu_seconds_per_frame": dt_o / k,
                                               "what": "float64 numpy restatement of tracker.cc (oracle/pnp_oracle.py)"}

    # ---- the same check spread over the whole clip: every stride-th frame, each from the GPU's own source poses ----
    if a.oracle_stride > 0:
        t0 = time.time()
        db = core.Database(path)
        model = np.eye(4)
        gpu_cam = {1: ocam(*true_pose(1))}
        for f, (q, t, _) in got.items():
            gpu_cam[f] = po.Camera(fx=-F, fy=-F, cx=W / 2, cy=H / 2, aspect_ratio=1.0, width=float(W), height=float(H), opencv=False,
                                   q=np.asarray(q, float), t=np.asarray(t, float))
        from concurrent.futures import ProcessPoolExecutor
        import multiprocessing as mp

        checked = list(range(2 + a.oracle_stride - 1, n + 1, a.oracle_stride))
        jobs = []
        for f in checked:
            srcs = [s_ for s_ in db.find_optical_flows_to_image(f) if s_ < f]   # forward tracking: only frames solved before f were filled (tracker.cc:43-50)
            jobs.append((path, f, {s_: gpu_cam[s_] for s_ in srcs}, gpu_cam[f - 1], got[f][0], got[f][1], verts, tris))
        if a.oracle_workers > 1:
            # processes, not threads: the restatement is numpy + Python loops (the GIL serialises threads: 10 samples took 15 s);
            # spawned, so that no child inherits this process's HIP state -- they only read the SQLite file
            with ProcessPoolExecutor(max_workers=min(a.oracle_workers, len(jobs)), mp_context=mp.get_context("spawn")) as ex:
                res = list(ex.map(_oracle_pose_check, jobs))
        else:
            res = [_oracle_pose_check(j) for j in jobs]
        ang, tr = [r[0] for r in res], [r[1] for r in res]
        db.close()
        out["tracking"]["vs_cpu_reference_sampled"] = {"frames": checked, "rotation_rad_max": max(ang), "translation_rel_max": max(tr),
                                                       "cpu_seconds_per_frame": (time.time() - t0) / max(1, len(checked)),
                                                       "cpu_wall_seconds": time.time() - t0, "cpu_processes": max(1, min(a.oracle_workers, len(jobs)))}

    # ---- refinement ----
    traj_c = core.CameraTrajectory(1, n)
    R, t = true_pose(1)
    for f in range(1, n + 1):
        pose = core.Pose()
        if f == 1:
            pose.q, pose.t = po.R_to_quat(R).astype(np.float32), t.astype(np.float32)
        else:
            pose.q, pose.t = got[f][0].astype(np.float32), got[f][1].astype(np.float32)
        traj_c.set(f, core.CameraState(intr(), pose))
    bo2 = core.BundleOptions()
    bo2.loss_type = core.LossType.Cauchy
    bo2.max_iterations = a.refine_iterations
    last = []
    t0 = time.time()
    core.refine_trajectory(path, traj_c, np.eye(4, dtype=np.float32), mesh, a.refine_intrinsics, a.refine_intrinsics,
                           lambda u: last.append(u.stats) or True, bo2)
    dt = time.time() - t0
    refined = {f: (np.array(traj_c.get(f).pose.q, float), np.array(traj_c.get(f).pose.t, float)) for f in range(2, n + 1)}
    out["refinement"] = {"seconds": round(dt, 3), "iterations": last[-1].iterations if last else 0,
                         "cost": [last[-1].initial_cost, last[-1].cost] if last else None, "vs_truth": errors(refined),
                         "stages": {k: [round(v[0], 3), v[1]] for k, v in core._stage_report("RefineTrajectory").items()}}
    # ---- CPU baseline of the refinement sweeps: the float64 restatement on a sub-segment (checker leg) ----
    if a.refine_oracle_frames > 2:
        import refine_oracle as ro
        m = min(a.refine_oracle_frames, n)
        f0 = max(1, n // 2 - m // 2)
        db = core.Database(path)
        kps_d, flows_d = {}, {}
        for f in range(f0, f0 + m):
            kps_d[f] = db.read_keypoints(f)
            fl = []
            for to in db.find_optical_flows_from_image(f):
                if f0 <= to < f0 + m:
                    pf = db.read_image_pair_flow(f, to)
                    fl.append((to, pf.src_kps_indices, pf.tgt_kps))
            flows_d[f] = fl
        db.close()
        cams_o = [po.Camera(fx=-F, fy=-F, cx=W / 2, cy=H / 2, aspect_ratio=1.0, width=float(W), height=float(H), opencv=False,
                            q=refined[f][0] if f in refined else po.R_to_quat(true_pose(f)[0]), t=refined[f][1] if f in refined else true_pose(f)[1])
                  for f in range(f0, f0 + m)]
        model64 = np.eye(4)
        seg = ro.load_segment(kps_d, flows_d, cams_o, f0, verts, model64)
        n_res = int(sum(len(e[2]) for e in seg.edges))
        t0 = time.time()
        ro.total_cost(seg, cams_o, verts.astype(np.float64), tris, None, model64, "cauchy", 1.0)
        t_cost = time.time() - t0
        t0 = time.time()
        ro.normal_equations(seg, cams_o, verts.astype(np.float64), tris, model64, "cauchy", 1.0, False, False)
        t_ne = time.time() - t0
        out["refinement"]["cpu_reference"] = {"frames": m, "edges": len(seg.edges), "residuals": n_res, "cost_sweep_seconds": t_cost,
                                              "normal_equations_seconds": t_ne, "processes": 1,
                                              "what": "float64 numpy restatement of refiner.cc's sweeps (oracle/refine_oracle.py)"}
    for fn in os.listdir(td):
        os.remove(os.path.join(td, fn))
    os.rmdir(td)
    return out


if __name__ == "__main__":
    sys.exit(main())
