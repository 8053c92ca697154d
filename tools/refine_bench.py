#!/usr/bin/env python3
"""refine_bench.py -- wall time of "Refine Sequence" (polychase_core.refine_trajectory) on a synthetic
segment: F frames x K keypoints, flows to +-{1,2,4,8}, a wavy grid mesh of T triangles.

    python tools/refine_bench.py [--frames 300] [--keypoints 1000] [--grid 48] [--iters 30]
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
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--keypoints", type=int, default=1000)
    ap.add_argument("--grid", type=int, default=48)
    ap.add_argument("--iters", type=int, default=30)
    args = ap.parse_args()

    import numpy as np
    import torch  # noqa: F401  (one HIP runtime per process)
    import polychase_core as core
    import refine_scene as S

    n = args.frames
    verts, tris = S.grid_mesh(args.grid)
    model = np.eye(4)
    truth = [S.true_camera(t, False, 24.0 / n) for t in range(1, n + 1)]
    t0 = time.time()
    kps, flows = S.make_flows(verts, tris, model, truth, 1, n_kp=args.keypoints, noise=0.1, seed=11)
    cams = S.perturbed(truth, np.random.default_rng(11))
    path = "/tmp/refine_bench.db"
    if os.path.exists(path):
        os.remove(path)
    S.write_database(core, path, kps, flows)
    print(f"scene: {n} frames x {args.keypoints} keypoints, {len(tris)} triangles, "
          f"{sum(len(r[1]) for rows in flows.values() for r in rows)} flow rows ({time.time() - t0:.1f} s to synthesise)")
    mesh = core.AcceleratedMesh(verts, tris)
    bo = core.BundleOptions()
    bo.loss_type = core.LossType.Cauchy
    bo.max_iterations = args.iters
    for rep in range(2):
        traj = S.to_core_trajectory(core, cams, 1)
        last = []
        t0 = time.time()
        core.refine_trajectory(path, traj, np.eye(4, dtype=np.float32), mesh, False, False,
                               lambda u: last.append(u.stats) or True, bo)
        dt = time.time() - t0
        st = last[-1]
        got = S.from_core_trajectory(traj, cams)
        e0 = max(S.angle(c.R(), t.R()) for c, t in zip(cams, truth))
        e1 = max(S.angle(c.R(), t.R()) for c, t in zip(got, truth))
        print(f"run {rep}: {dt * 1e3:.1f} ms total ({len(last) - 1} LM callbacks, {st.iterations} iterations, "
              f"{st.invalid_steps} rejected), cost {st.initial_cost:.3f} -> {st.cost:.4f}, max rotation error {e0:.2e} -> {e1:.2e} rad")
    t0 = time.time()
    sysd = core._refinement_system(path, S.to_core_trajectory(core, cams, 1), np.eye(4, dtype=np.float32), mesh, False, False, bo)
    print(f"load + 1 cost sweep + 1 normal-equation sweep: {(time.time() - t0) * 1e3:.1f} ms "
          f"({sysd['num_residuals']} residuals, {sysd['num_edges']} edges)")


if __name__ == "__main__":
    main()
