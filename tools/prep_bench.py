#!/usr/bin/env python3
"""prep_bench.py -- the frame-preparation kernels alone (run ON the GPU box): gray + pyramid + Scharr of one frame
(`pc_frame_set_rgb`) and dense detection (`pc_frame_detect`), nothing else on the GPU.

    python tools/prep_bench.py [--config c2|c3] [--reps 50]

Prints one JSON line: per-class milliseconds per call (HIP events inside the library) and the algorithmic bytes per
second they correspond to (level kernels 4P + 7S, detection 9P).  Run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel durations kept in profiles/.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CONFIGS = {"c1": (640, 480, 3), "c2": (1920, 1080, 3), "c3": (3840, 2160, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--reps", type=int, default=50)
    args = ap.parse_args()
    import torch
    from polychase_amd import hip, synth

    w, h, max_level = CONFIGS[args.config]
    clip = synth.NoiseClip(w, h, 300, device="cuda:0")
    rgb = [clip.frame_torch(100 + i) for i in range(4)]
    ctx = hip.Context(0)
    f = hip.Frame(ctx, w, h, 10, max_level)
    for i in range(4):
        f.set_rgb(rgb[i])
        f.detect()
    ctx.synchronize()
    ctx.enable_timing(True)
    ctx.reset_timing()
    for i in range(args.reps):
        f.set_rgb(rgb[i & 3])
    ctx.synchronize()
    t_img = ctx.timing()
    ctx.reset_timing()
    for i in range(args.reps):
        f.detect()
    t_det = ctx.timing()
    ctx.enable_timing(False)
    P = w * h
    S, lw, lh = 0, w, h
    for _ in range(max_level + 1):
        S += lw * lh
        lw, lh = (lw + 1) // 2, (lh + 1) // 2
    img_ms = (t_img["gray"][1] + t_img["pyramid"][1]) / args.reps
    det = {k: t_det[k][1] / args.reps for k in ("min_eig", "nms", "suppress", "sort")}
    out = {"config": args.config, "keypoints": f.num_keypoints, "candidates": f.num_candidates,
           "image_ms": img_ms, "image_algorithmic_GBs": (4 * P + 7 * S) / (img_ms * 1e-3) / 1e9 if img_ms else None,
           "detect_ms": det, "min_eig_nms_algorithmic_GBs": 9 * P / ((det["min_eig"] + det["nms"]) * 1e-3) / 1e9}
    print(json.dumps(out), flush=True)
    f.close()
    ctx.close()


if __name__ == "__main__":
    main()
