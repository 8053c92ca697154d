#!/usr/bin/env python3
"""coresidency_probe.py -- does a copy get through while the LK launches run?  (run ON the GPU box)

pc_analyzer's pipeline runs as in the benchmark (one new frame with detection per step, frame1 = 9 submitted over and over,
tools/lane_probe.py's "full" mode).  Every second step a copy of --mbytes is issued on the null stream, done by
    fat    a kernel with the resource shape of rcclGenericKernel (256 lanes, 280 registers, 19.7 KB LDS), --workgroups of them
    slim   the same loop in 10 registers
    dma    hipMemcpyAsync device to device
    d2h    hipMemcpyAsync device to pinned host in --parts parts, one after the other on the null stream
    d2h-streams   the same parts, each on a copy-only stream of its own (concurrent SDMA transfers)
(tools/coresidency.hip).  Reported per kind: the step time, the copy's duration alone and beside LK (event to event, i.e.
from the moment the copy ahead of it finished), and the BACKLOG: how long the null stream still runs after the last job has
been collected -- copies that could not run beside LK pile up there.

    hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/coresidency.hip -o tools/bin/libcoresidency.so
    python tools/coresidency_probe.py [--config c2] [--steps 200] [--mbytes 11] [--workgroups 16]
"""
import argparse
import ctypes as C
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
    ap.add_argument("--mbytes", type=float, default=11.0, help="bytes per copy (two 1080p frames of records = 11 MB)")
    ap.add_argument("--workgroups", type=int, default=16)
    ap.add_argument("--every", type=int, default=2, help="one copy every N steps")
    ap.add_argument("--parts", type=int, default=7, help="parts of a d2h copy (the pushes to seven peers)")
    ap.add_argument("--kinds", default="fat,slim,dma,d2h,d2h-streams")
    args = ap.parse_args()
    import torch
    from polychase_amd import hip, synth

    L = C.CDLL(os.path.join(ROOT, "tools", "bin", "libcoresidency.so"))
    L.cr_init.argtypes = [C.c_size_t, C.c_int]
    L.cr_collect.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    w, h, ml = CONFIGS[args.config]
    clip = synth.NoiseClip(w, h, 60, device="cuda:0")
    frames = [clip.frame_torch(t) for t in range(60)]
    torch.cuda.synchronize()
    assert L.cr_init(int(args.mbytes * 1e6), 4096) == 0
    ctx = hip.Context(0)

    def collect():
        mean, mx, n = C.c_double(), C.c_double(), C.c_int()
        assert L.cr_collect(C.byref(mean), C.byref(mx), C.byref(n)) == 0
        return mean.value, mx.value, n.value

    alone = {}
    KINDS = [(k, n) for k, n in [(0, "fat"), (1, "slim"), (2, "dma"), (3, "d2h"), (4, "d2h-streams")] if n in args.kinds.split(",")]
    arg = lambda kind: args.workgroups if kind < 3 else args.parts
    for kind, name in KINDS:
        for _ in range(4):
            L.cr_launch(kind, arg(kind))
        collect()
        for _ in range(20):
            L.cr_launch(kind, arg(kind))
        alone[name] = collect()[0]

    for kind, name in [(-1, "none")] + KINDS:
        an = hip.Analyzer(ctx, w, h, hip.gftt_options(), hip.flow_options(max_level=ml), 17, 3)
        for f in range(1, 18):
            an.put_frame(f, frames[f], will_detect=True)
        targets = [9 + s for s in SKIPS]
        spare = [18, 19, 20]
        k = 0

        def step(side):
            nonlocal k
            fid = spare[k % 3] + 20 * (k // 3)
            an.put_frame(fid, frames[20 + k % 40], will_detect=True)
            k += 1
            if an.pending == 3:
                an.collect_raw()
            an.submit(9, targets)
            if side and kind >= 0 and k % args.every == 0:
                L.cr_launch(kind, arg(kind))

        for _ in range(30):
            step(False)
        while an.pending:
            an.collect_raw()
        ctx.synchronize()
        gc.collect()
        gc.disable()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(True)
        while an.pending:
            an.collect_raw()
        ctx.synchronize()
        t1 = time.perf_counter()
        idle_at_end = bool(L.cr_idle())
        mean, mx, n = collect()          # waits for the null stream
        t2 = time.perf_counter()
        gc.enable()
        out = {"config": args.config, "copy": name, "steps": args.steps, "ms_per_step": (t1 - t0) / args.steps * 1e3}
        if kind >= 0:
            out.update(copies=n, mbytes=args.mbytes, workgroups=args.workgroups if kind < 2 else None, parts=args.parts if kind >= 3 else None,
                       copy_ms_alone=alone[name], copy_ms_beside_lk_mean=mean, copy_ms_beside_lk_max=mx,
                       copies_done_when_the_lanes_drained=idle_at_end, backlog_ms=(t2 - t1) * 1e3)
        print(json.dumps(out), flush=True)
        an.close()
    ctx.close()


if __name__ == "__main__":
    main()
