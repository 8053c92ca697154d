#!/usr/bin/env python3
"""lk_straggler_sim.py -- what deferring the stragglers of the LK launch could gain (CPU only; the numbers beside the measured
ceiling of tools/lk_tail_ceiling.sh, DESIGN.md section 4 "Divergence").

The oracle records the iterations per (pair, level) of one frame and its 8 targets (pco_set_lk_iter_trace).  The kernel's
wavefront (2 keypoints x 8 targets) issues, per level, the MAXIMUM over its 16 pairs.  With a cap T a wavefront stops a level
after T iterations; the pairs still iterating are stragglers, handed on to wavefronts of 16 stragglers each that first pay a
refill of R iteration-equivalents (the I side of the straggler's keypoint again, its region, its state) and then run the
maximum of the remaining counts.

    python tests/studies/lk_straggler_sim.py [--width 960 --height 540]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--height", type=int, default=540)
    ap.add_argument("--frame", type=int, default=100)
    a = ap.parse_args()
    import oracle   # test infrastructure, used here as a measuring instrument
    from polychase_amd import synth

    skips = (-8, -4, -2, -1, 1, 2, 4, 8)
    clip = synth.NoiseClip(a.width, a.height, 300)
    g = {t: oracle.rgb2gray(clip.frame(t)) for t in [a.frame + s for s in (0,) + skips]}
    kps = oracle.gftt(g[a.frame])
    p0 = oracle.Pyramid(g[a.frame])
    L = oracle.lib()
    L.pco_set_lk_iter_trace.argtypes = [C.c_void_p, C.c_int]
    its = np.zeros((8, len(kps), 4), np.int32)
    for k, s in enumerate(skips):
        buf = np.zeros((len(kps), 4), np.uint8)
        L.pco_set_lk_iter_trace(buf.ctypes.data, 4)
        oracle.lk(p0, oracle.Pyramid(g[a.frame + s]), kps)
        L.pco_set_lk_iter_trace(None, 0)
        its[k] = buf
    T8, N, LV = its.shape
    tile = (kps[:, 1].astype(int) // 64) * 64 + (kps[:, 0].astype(int) // 64)
    I = its[:, np.argsort(tile, kind="stable"), :]
    n2 = N // 2 * 2
    A = I[:, :n2, :].reshape(T8, n2 // 2, 2, LV)          # [target, wave, half, level]
    base = A.max(axis=(0, 2)).sum()                       # iterations the kernel issues
    print(f"{N} keypoints x 8 targets, {n2 // 2} wavefronts; iterations issued {base} ({base / (n2 // 2):.2f} per wavefront; per level "
          f"{A.max(axis=(0, 2)).mean(axis=0).round(2)}); a pair needs {I.mean(axis=(0, 1)).round(2)} per level (in processing order: the LAST entry is level 0, the finest)")
    F = LV - 1                                            # the trace is in processing order: coarsest level first
    h = np.bincount(I[:, :, F].ravel(), minlength=31)
    print("level-0 iteration counts of a pair, cumulative share: " + ", ".join(f"<= {t}: {h[:t + 1].sum() / h.sum():.3f}" for t in (4, 8, 10, 12, 16, 20, 24, 29)))
    for cap in (8, 10, 12, 16, 20, 24):
        capped = np.minimum(A, cap)
        issued = capped.max(axis=(0, 2)).sum()            # every level capped: what tools/lk_tail_ceiling.sh measures
        only0 = A.copy()
        only0[..., F] = np.minimum(A[..., F], cap)
        issued0 = only0.max(axis=(0, 2)).sum()            # the cap at the finest level only
        rest = np.maximum(I[:, :n2, F] - cap, 0).ravel()
        rest = rest[rest > 0]
        n_strag = len(rest)
        line = (f"cap {cap:2d}: all levels capped -> {100 * (1 - issued / base):.1f} % fewer iterations issued (the measured ceiling's model); finest level only -> "
                f"{100 * (1 - issued0 / base):.1f} %; stragglers {n_strag} = {100 * n_strag / (8 * n2):.2f} % of the pairs")
        for refill in (2, 4, 8):
            # stragglers re-packed 16 to a wavefront in arrival order: refill + the maximum of the remaining counts
            m = (n_strag + 15) // 16
            pad = np.concatenate([rest, np.zeros(m * 16 - n_strag, rest.dtype)]).reshape(m, 16)
            extra = (pad.max(axis=1) + refill).sum()
            line += f"; refill {refill}: net {100 * (1 - (issued0 + extra) / base):.1f} %"
        print(line)


if __name__ == "__main__":
    main()
