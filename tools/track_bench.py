#!/usr/bin/env python3
"""track_bench.py -- "Track Sequence" (polychase_core.track_sequence: batched ray casting on the LBVH + PnP
LM with GPU residual sweeps, reference cpp/tracker.cc) on a synthetic flow database.

    python tools/track_bench.py [--frames 120] [--keypoints 2000] [--grid 48]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "polychase_amd", "core"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=120)
    ap.add_argument("--keypoints", type=int, default=2000)
    ap.add_argument("--grid", type=int, default=48)
    a = ap.parse_args()
    import numpy as np
    import torch  # noqa: F401
    import polychase_core as core
    import refine_scene as S

    n = a.frames
    verts, tris = S.grid_mesh(a.grid)
    model = np.eye(4)
    truth = [S.true_camera(t, False, 24.0 / n) for t in range(1, n + 1)]
    kps, flows = S.make_flows(verts, tris, model, truth, 1, n_kp=a.keypoints, noise=0.1, seed=11)
    path = "/tmp/track_bench.db"
    if os.path.exists(path):
        os.remove(path)
    S.write_database(core, path, kps, flows)
    mesh = core.AcceleratedMesh(verts, tris)
    c0 = truth[0]
    view = np.eye(4, dtype=np.float32)
    view[:3, :3], view[:3, 3] = c0.R(), c0.t
    intr = core.CameraIntrinsics(fx=c0.fx, fy=c0.fy, cx=c0.cx, cy=c0.cy, aspect_ratio=1.0, width=S.W, height=S.H,
                                 convention=core.CameraConvention.OpenGL)
    st = core.SceneTransformations(np.eye(4, dtype=np.float32), view, intr)
    bo = core.BundleOptions()
    bo.loss_type = core.LossType.Cauchy
    for rep in range(2):
        got = []
        t0 = time.time()
        core.track_sequence(path, 1, n, st, mesh, lambda r: got.append(r) or True, False, False, bo)
        dt = time.time() - t0
        err = max(S.angle(S.po.quat_to_R(np.array(r.pose.q, float)), truth[r.frame - 1].R()) for r in got)
        its = np.mean([r.bundle_stats.iterations for r in got])
        print(f"run {rep}: {len(got)} frames in {dt * 1e3:.0f} ms = {len(got) / dt:.0f} frames/s "
              f"({a.keypoints} keypoints/frame, {len(tris)} triangles, mean {its:.1f} LM iterations, "
              f"mean inlier ratio {np.mean([r.inlier_ratio for r in got]):.3f}, max rotation error {err:.1e} rad)")


if __name__ == "__main__":
    main()
