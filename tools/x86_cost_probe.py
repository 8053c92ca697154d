#!/usr/bin/env python3
"""x86_cost_probe.py -- what the x86 summation order costs on content where its exactness proof FAILS (run ON the GPU box).

The benchmark clips are kind to the proof (99 % of the iterations are decided by it); footage with hard edges is not.  This
times one LK launch (frame 10 into its 8 neighbours) in the canonical mode and in lk_x86 on four kinds of 1920x1080 content and
prints, per kind, the two launch times and the counters of pc_debug_lk_x86_stats: DESIGN.md section 4 "The x86 order"."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def content(kind, w, h, t, rng_seed=7):
    from polychase_amd import synth
    if kind == "c2 texture":
        return None
    rng = np.random.default_rng(rng_seed)
    if kind == "checkerboard":
        return synth.checkerboard_frame(t, w=w, h=h)
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == "binary blocks":          # hard edges at full contrast, drifting
        cells = rng.integers(0, 2, (h // 12 + 4, w // 12 + 4))
        img = 255 * np.kron(cells, np.ones((12, 12)))[t:t + h, 2 * t:2 * t + w]
    else:                                # "edges + texture": blurred step edges of contrast 160 over a fine texture of +-20
        cells = rng.integers(0, 2, (h // 24 + 4, w // 24 + 4))
        big = 160.0 * np.kron(cells, np.ones((24, 24)))[t:t + h + 2, 2 * t:2 * t + w + 2]
        big = (big[:-2, :-2] + big[1:-1, :-2] + big[2:, :-2] + big[:-2, 1:-1] + big[1:-1, 1:-1] + big[2:, 1:-1] + big[:-2, 2:] + big[1:-1, 2:] + big[2:, 2:]) / 9
        tex = rng.integers(-20, 21, (h + 64, w + 64)).astype(np.float64)[t:t + h, 2 * t:2 * t + w]
        img = 40 + big + tex
    g = np.clip(img, 0, 255).astype(np.uint8)
    return np.ascontiguousarray(np.repeat(g[:, :, None], 3, axis=2))


def measure(contents=("c2 texture", "edges + texture", "binary blocks", "checkerboard"), w=1920, h=1080, reps=10, emit=None):
    """one row per content; `sha256` covers every tracked row of the x86 launch (it must not move when the kernel is worked on)"""
    import hashlib
    from polychase_amd import hip, synth
    clip = synth.NoiseClip(w, h, 24)
    rows = []
    for kind in contents:
        rgbs = [clip.frame(t) if kind == "c2 texture" else content(kind, w, h, t) for t in (10, 2, 6, 8, 9, 11, 12, 14, 18)]
        for _once in (0,):
            ctx = hip.Context(0)
            frames = []
            for rgb in rgbs:
                f = hip.Frame(ctx, w, h, 10, 3)
                f.set_rgb(rgb)
                frames.append(f)
            row = {"content": kind}
            for mode, flag in (("canonical", hip.ARITH_CANONICAL), ("lk_x86", hip.ARITH_LK_X86_ORDER)):
                ctx.set_arithmetic(flag)
                frames[0].detect()
                xy, st, err = hip.lk_track(ctx, frames[0], frames[1:], hip.flow_options())
                ctx.enable_timing(["lk"])
                ctx.reset_timing()
                for _ in range(reps):
                    hip.lk_track(ctx, frames[0], frames[1:], hip.flow_options())
                n, ms = ctx.timing()["lk"]
                ctx.enable_timing(False)
                row[mode + "_ms"] = ms / n
                if flag:
                    hsh = hashlib.sha256()
                    m = st == 1
                    for arr in (st, xy[m], err[m]):
                        hsh.update(np.ascontiguousarray(arr).tobytes())
                    row["sha256"] = hsh.hexdigest()[:16]
                    ctx.lk_x86_stats(True)
                    hip.lk_track(ctx, frames[0], frames[1:], hip.flow_options())
                    stt = ctx.lk_x86_stats(False)
                    tot = stt["iterations_proven_exact"] + stt["iterations_x86_order"]
                    row["keypoints"] = frames[0].num_keypoints
                    row["iterations_in_x86_order"] = stt["iterations_x86_order"] / max(1, tot)
                    row["levels_with_ordered_structure_tensor"] = stt["keypoint_levels_x86_order"] / max(1, stt["keypoint_levels"])
            row["x86_over_canonical"] = row["lk_x86_ms"] / row["canonical_ms"]
            rows.append(row)
            if emit:
                emit(row)
            for f in frames:
                f.close()
            ctx.close()
    return rows


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--contents", default="c2 texture,edges + texture,binary blocks,checkerboard")
    a = ap.parse_args()
    measure(tuple(a.contents.split(",")), emit=lambda row: print(json.dumps(row), flush=True))


if __name__ == "__main__":
    main()
