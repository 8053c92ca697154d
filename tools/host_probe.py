#!/usr/bin/env python3
"""Where does the HOST spend a step of the analyzer pipeline?  Runs the bench loop of one configuration and
accumulates wall time inside put_frame / submit / collect.  A pipeline whose `collect` time is ~0 is limited by the
host's enqueue rate, not by the GPU.

    python tools/host_probe.py --config c2 [--steps 400]
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--steps", type=int, default=400)
    args = ap.parse_args()
    import torch

    import bench
    from polychase_amd import hip, synth
    from polychase_amd.pipeline import ClipAnalyzer

    w, h, max_level, _ = bench.CONFIGS[args.config]
    n = 60
    clip = synth.NoiseClip(w, h, n, device="cuda:0")
    frames = [clip.frame_torch(t) for t in range(n)]
    torch.cuda.synchronize()

    def source(fid):
        t = fid % (2 * n - 2)
        return frames[t if t < n else 2 * n - 2 - t]

    ctx = hip.Context(0)
    an = ClipAnalyzer(ctx, w, h, 1, 1 << 30, source, hip.gftt_options(), hip.flow_options(max_level=max_level), max_jobs=3)
    acc = {"put": 0.0, "submit": 0.0, "collect": 0.0}
    raw = an.an

    def wrap(name, key):
        fn = getattr(raw, name)

        def inner(*a, **k):
            t = time.perf_counter()
            r = fn(*a, **k)
            acc[key] += time.perf_counter() - t
            return r
        setattr(raw, name, inner)

    wrap("put_frame", "put")
    wrap("submit", "submit")
    wrap("collect", "collect")
    gc.collect()
    gc.disable()
    an.run(range(9, 9 + 40), None, copy=False)
    ctx.synchronize()
    for k in acc:
        acc[k] = 0.0
    t0 = time.perf_counter()
    an.run(range(49, 49 + args.steps), None, copy=False)
    ctx.synchronize()
    total = time.perf_counter() - t0
    out = {"config": args.config, "ms_per_step": 1e3 * total / args.steps}
    out.update({f"{k}_ms": 1e3 * v / args.steps for k, v in acc.items()})
    out["other_ms"] = out["ms_per_step"] - sum(1e3 * v / args.steps for v in acc.values())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
