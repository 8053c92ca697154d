#!/usr/bin/env python3
"""Host-side time per analyzer call (put_frame / submit / collect) on the bench workload, with the
slowest calls listed (looking for stalls: reallocations, blocking waits)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from polychase_amd import hip, synth
from polychase_amd.pipeline import ClipAnalyzer

w, h, ml = (1920, 1080, 3) if len(sys.argv) < 2 or sys.argv[1] == "c2" else (3840, 2160, 4)
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
clip = synth.NoiseClip(w, h, 300, device="cuda:0")
frames = {1 + i: clip.frame_torch(i) for i in range(K + 30)}
torch.cuda.synchronize()
ctx = hip.Context(0)
an = ClipAnalyzer(ctx, w, h, 1, K + 30, lambda f: frames[f], hip.gftt_options(), hip.flow_options(max_level=ml), max_jobs=3)
log = []
import gc
if os.environ.get("NOGC"): gc.disable()
def wrap(name, fn):
    def g(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); log.append((name, a[0] if a and isinstance(a[0], int) else -1, time.perf_counter() - t, t)); return r
    return g
an.an.put_frame, an.an.submit, an.an.collect = wrap("put", an.an.put_frame), wrap("submit", an.an.submit), wrap("collect", an.an.collect)
an.run(range(9, 19), None)
ctx.synchronize()
log.clear()
t0 = time.perf_counter()
an.run(range(19, 19 + K), None, copy=False)
ctx.synchronize()
dt = time.perf_counter() - t0
print(f"{K / dt:.1f} fps, {dt / K * 1e3:.3f} ms/step")
print("  first calls:", " ".join(f"{n[0]}{f}:{ms * 1e3:.2f}" for n, f, ms, _ in log[:24]))
for name in ("put", "submit", "collect"):
    v = np.array([x[2] for x in log if x[0] == name]) * 1e3
    print(f"  {name:8s} mean {v.mean():.4f} ms  p50 {np.percentile(v, 50):.4f}  p99 {np.percentile(v, 99):.4f}  max {v.max():.4f}")
ends = np.array([x[3] + x[2] for x in log if x[0] == "collect"])
per = np.diff(ends) * 1e3
print(f"  period between collects: p10 {np.percentile(per, 10):.3f} p50 {np.percentile(per, 50):.3f} p90 {np.percentile(per, 90):.3f} max {per.max():.3f} ms")
print("  slowest periods at collect #:", np.argsort(per)[-8:][::-1].tolist(), np.sort(per)[-8:][::-1].round(2).tolist())
