#!/usr/bin/env python3
"""lk_bench.py -- the pyramidal LK launch alone (run ON the GPU box).

One interior frame1 of the synthetic clip and its 8 targets are made resident through the stage-level
C ABI, then `pc_lk_track` is repeated: the LK class is timed with HIP events by the library
(pc_context_enable_timing), launches do not overlap, nothing else runs.  Prints one JSON line with
the average launch time and a checksum of the raw outputs (equal checksums <=> bit-identical
results, the way kernel variants are compared: POLYCHASE_HIP_LIB selects the library).

    python tools/lk_bench.py [--config c2|c3] [--reps 20]

(Parity against the CPU oracle is the test suite's business: tests/test_gpu_parity.py, tests/test_fullsize_gpu.py.)
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {"c1": (640, 480, 3), "c2": (1920, 1080, 3), "c3": (3840, 2160, 4)}
SKIPS = (-8, -4, -2, -1, 1, 2, 4, 8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--frame", type=int, default=100)
    ap.add_argument("--window", type=int, default=10)
    ap.add_argument("--arith", default="canonical", choices=["canonical", "lk_x86", "sobel_fma", "opencv_x86"])
    ap.add_argument("--max-iters", type=int, default=30, help="term_max_iters (30 = the reference's default); smaller values CUT the "
                    "stragglers' iterations off -- wrong results, but the time saved is the ceiling of what deferring them could gain")
    args = ap.parse_args()

    import torch
    from polychase_amd import hip, synth

    w, h, max_level = CONFIGS[args.config]
    dev = torch.device("cuda", 0)
    clip = synth.NoiseClip(w, h, 300, device=str(dev))
    ctx = hip.Context(0)
    ctx.set_arithmetic({"canonical": hip.ARITH_CANONICAL, "lk_x86": hip.ARITH_LK_X86_ORDER, "sobel_fma": hip.ARITH_SOBEL_FMA,
                        "opencv_x86": hip.ARITH_OPENCV_X86}[args.arith])
    fopt = hip.flow_options(max_level=max_level, window_size=args.window, term_max_iters=args.max_iters)
    frames = {}
    for s in (0,) + SKIPS:
        f = hip.Frame(ctx, w, h, args.window, max_level)
        f.set_rgb(clip.frame_torch(args.frame + s))
        frames[s] = f
    f1 = frames[0]
    f1.detect()
    n = f1.num_keypoints
    targets = [frames[s] for s in SKIPS]
    xy, st, err = hip.lk_track(ctx, f1, targets, fopt)          # warm-up + reference output
    for _ in range(3):
        hip.lk_track(ctx, f1, targets, fopt)
    ctx.enable_timing(["lk"])
    ctx.reset_timing()
    for _ in range(args.reps):
        xy2, st2, err2 = hip.lk_track(ctx, f1, targets, fopt)
    launches, ms = ctx.timing()["lk"]
    ctx.enable_timing(False)
    ctx.lk_profile()   # discard: the timed launches below are profiled on their own
    hip.lk_track(ctx, f1, targets, fopt)
    assert np.array_equal(xy, xy2) and np.array_equal(st, st2) and np.array_equal(err, err2), "results vary between launches"
    filt = hip.lk_track_filtered(ctx, f1, targets, fopt)
    hsh = hashlib.sha256()
    for a in (xy, st, err):
        hsh.update(np.ascontiguousarray(a).tobytes())
    for idx, fxy, ferr in filt:
        for a in (idx, fxy, ferr):
            hsh.update(np.ascontiguousarray(a).tobytes())
    # the filtered path must be the status==1 rows of the raw one
    for t, (idx, fxy, ferr) in enumerate(filt):
        keep = np.nonzero(st[t] == 1)[0]
        assert np.array_equal(idx, keep.astype(np.uint32)) and np.array_equal(fxy, xy[t][keep]) and np.array_equal(ferr, err[t][keep])
    x86_stats = None
    if ctx.arithmetic & hip.ARITH_LK_X86_ORDER:   # one more launch with the diagnostics counters on
        ctx.lk_x86_stats(True)
        hip.lk_track(ctx, f1, targets, fopt)
        x86_stats = ctx.lk_x86_stats(False)
    out = {"config": args.config, "window": args.window, "arith": args.arith, "max_iters": args.max_iters, "keypoints": n, "launches": launches,
           "lk_ms_per_launch": ms / max(1, launches), "tracked_rows": int((st == 1).sum()),
           "sha256": hsh.hexdigest()[:16], "lib": os.environ.get("POLYCHASE_HIP_LIB", "default")}
    if x86_stats is not None:
        out["x86_stats_one_launch"] = x86_stats
    prof = ctx.lk_profile()
    if any(prof):
        names = ["i_stage", "i_eval", "pickup", "j_stage", "iterate", "err", "life", "waves", "wave_iters", "stagings"]
        w = max(1, prof[7])
        out["profile_cycles_per_wave"] = {n: round(v / w, 1) for n, v in zip(names, prof)}
    print(json.dumps(out), flush=True)
    for f in frames.values():
        f.close()
    ctx.close()


if __name__ == "__main__":
    main()
