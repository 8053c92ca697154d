#!/usr/bin/env python3
"""x86_gap.py -- canonical vs x86 summation order of LK (pc_context_set_arithmetic) on the benchmark clips at full size
(run ON the GPU box): how many tracked vectors differ at all, by how much.  DESIGN.md section 2; profiles/r03_x86_gap.json."""
import sys, json, numpy as np
sys.path.insert(0, "/root/repo")
from polychase_amd import hip, synth
import torch
ctx = hip.Context(0)
out = {}
for cfg, (w, h, ml) in {"c2": (1920, 1080, 3), "c3": (3840, 2160, 4)}.items():
    clip = synth.NoiseClip(w, h, 300, device="cuda:0")
    fr = {}
    for s in (0, -8, -4, -2, -1, 1, 2, 4, 8):
        f = hip.Frame(ctx, w, h, 10, ml); f.set_rgb(clip.frame_torch(100 + s)); fr[s] = f
    fr[0].detect()
    tg = [fr[s] for s in (-8, -4, -2, -1, 1, 2, 4, 8)]
    opt = hip.flow_options(max_level=ml)
    ctx.set_arithmetic(hip.ARITH_CANONICAL)
    a = hip.lk_track(ctx, fr[0], tg, opt)
    ctx.set_arithmetic(hip.ARITH_LK_X86_ORDER)
    b = hip.lk_track(ctx, fr[0], tg, opt)
    ctx.set_arithmetic(hip.ARITH_CANONICAL)
    m = (a[1] == 1) & (b[1] == 1)
    d = np.abs(a[0] - b[0]).max(axis=2)[m]
    out[cfg] = {"pairs": int(m.size), "tracked_both": int(m.sum()), "status_flips": int((a[1] != b[1]).sum()), "vectors_that_differ": int((d > 0).sum()),
                "above_1e-3_px": int((d > 1e-3).sum()), "max_px": float(d.max())}
    for f in fr.values(): f.close()
print(json.dumps(out))
