#!/usr/bin/env python3
"""lk_divergence.py -- how much of the LK kernel's iteration work is SIMT padding, and which regrouping would help (CPU only).

The oracle records the iterations it ran per (keypoint, level) (pco_set_lk_iter_trace) for one C2 frame and its 8 targets;
the script then replays the GPU kernel's wavefront mapping (2 keypoints x 8 targets per wavefront, a level costs the
maximum over its 16 pairs) and alternatives.  The numbers behind DESIGN.md section 4 "What is left".

    python tests/studies/lk_divergence.py [--width 1920 --height 1080 --frame 100]
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
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
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
    T, N, LV = its.shape
    tile = (kps[:, 1].astype(int) // 64) * 64 + (kps[:, 0].astype(int) // 64)
    I = its[:, np.argsort(tile, kind="stable"), :]
    print(f"{N} keypoints x {T} targets; iterations a pair needs: {I.sum() / (T * N):.2f} (per level {I.mean(axis=(0, 1)).round(2)})")
    n2 = N // 2 * 2
    A = I[:, :n2, :].reshape(T, n2 // 2, 2, LV)
    cur = A.max(axis=(0, 2))
    print(f"wavefront = 2 keypoints x 8 targets (the kernel): {cur.sum(axis=1).mean():.2f} iterations issued per wavefront "
          f"(per level {cur.mean(axis=0).round(2)}), efficiency {I[:, :n2].sum() / (cur.sum() * 16):.3f}")
    n16 = N // 16 * 16
    B = I[:, :n16, :].reshape(T, n16 // 16, 16, LV).max(axis=2)
    print(f"wavefront = 16 keypoints x 1 target: efficiency {I[:, :n16].sum() / (B.sum() * 16):.3f}")
    tot = sum(np.sort(I[t, :n16, l]).reshape(-1, 16).max(axis=1).sum() for t in range(T) for l in range(LV))
    print(f"pairs sorted by their true count per level (unattainable bound): efficiency {I[:, :n16].sum() / (tot * 16):.3f}")
    for l in range(LV - 1):
        print(f"correlation of the counts, level {l + 1} vs {l}: {np.corrcoef(I[:, :, l + 1].ravel(), I[:, :, l].ravel())[0, 1]:.3f}")
    print("correlation between two targets of one keypoint, top level:", round(float(np.corrcoef(I[0, :, LV - 1], I[7, :, LV - 1])[0, 1]), 3))
    # idle groups helping: 25 pixel slots per lane with > 8 active pairs, 13 / 7 / 4 with <= 8 / 4 / 2
    W = A.transpose(1, 0, 2, 3).reshape(n2 // 2, 16, LV)
    cost = {"kernel": 0.0, "idle groups help": 0.0}
    for l in range(LV):
        X = W[:, :, l]
        for j in range(1, 31):
            act = (X >= j).sum(axis=1)
            act = act[act > 0]
            cost["kernel"] += (85 + 4 * 25) * len(act)
            slots = np.where(act > 8, 25, np.where(act > 4, 13, np.where(act > 2, 7, 4)))
            cost["idle groups help"] += (85 + 4 * slots).sum()
    print(f"VALU instructions in the iteration loops per wavefront: {cost['kernel'] / len(W):.0f}; with idle groups helping: "
          f"{cost['idle groups help'] / len(W):.0f} ({100 * (1 - cost['idle groups help'] / cost['kernel']):.0f} % fewer)")


if __name__ == "__main__":
    main()
