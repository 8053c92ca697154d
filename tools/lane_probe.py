#!/usr/bin/env python3
"""lane_probe.py -- where a pipeline step goes (run ON the GPU box): the job lanes of pc_analyzer with and without the
preparation stream beside them.

Frames 1..17 of the synthetic clip are made resident and detected once.  Then frame1 = 9 (all eight targets resident) is
submitted over and over -- the LK launch, the gate, compaction, the download: everything a job lane does -- while the
preparation stream is given, per step,
    mode lk        nothing
    mode pyramid   one new frame without detection (level kernels only), into the ring's spare slots
    mode full      one new frame with detection (= the benchmark's step)
Prints one JSON line per mode: ms per step, the LK launch's start-to-end and busy time (HIP events in the library).

    python tools/lane_probe.py [--config c2|c3] [--steps 200]        (POLYCHASE_LK_GATE=0 etc. apply as usual)
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CONFIGS = {"c1": (640, 480, 3), "c2": (1920, 1080, 3), "c3": (3840, 2160, 4)}
SKIPS = (-8, -4, -2, -1, 1, 2, 4, 8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--modes", default="lk,pyramid,full")
    ap.add_argument("--jobs", type=int, default=3)
    args = ap.parse_args()
    import torch
    from polychase_amd import hip, synth

    w, h, ml = CONFIGS[args.config]
    clip = synth.NoiseClip(w, h, 60, device="cuda:0")
    frames = [clip.frame_torch(t) for t in range(60)]
    torch.cuda.synchronize()
    ctx = hip.Context(0)
    for mode in args.modes.split(","):
        an = hip.Analyzer(ctx, w, h, hip.gftt_options(), hip.flow_options(max_level=ml), 17, args.jobs)
        for f in range(1, 18):
            an.put_frame(f, frames[f], will_detect=True)
        targets = [9 + s for s in SKIPS]
        spare = [18, 19, 20]          # ring of 17 + 3 slots: ids 18, 19, 20 (mod 20) never evict frames 1..17
        k = 0

        def step():
            nonlocal k
            if mode != "lk":
                fid = spare[k % 3] + 20 * (k // 3)
                an.put_frame(fid, frames[20 + k % 40], will_detect=(mode == "full"))
                k += 1
            if an.pending == args.jobs:
                an.collect_raw()
            an.submit(9, targets)

        for _ in range(30):
            step()
        while an.pending:
            an.collect_raw()
        ctx.synchronize()
        gc.collect()
        gc.disable()
        ctx.enable_timing(["lk"])
        ctx.reset_timing()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        while an.pending:
            r = an.collect_raw()
        ctx.synchronize()
        dt = time.perf_counter() - t0
        gc.enable()
        n, ms = ctx.timing()["lk"]
        busy = ctx.busy_ms("lk")
        ctx.enable_timing(False)
        print(json.dumps({"config": args.config, "mode": mode, "steps": args.steps, "keypoints": r.n_keypoints,
                          "ms_per_step": dt / args.steps * 1e3, "lk_launches": n, "lk_start_to_end_ms": ms / max(1, n),
                          "lk_busy_ms_per_launch": busy / max(1, n), "gate": os.environ.get("POLYCHASE_LK_GATE", "1")}), flush=True)
        an.close()
    ctx.close()


if __name__ == "__main__":
    main()
