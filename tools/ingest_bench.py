#!/usr/bin/env python3
"""Frame ingestion through polychase_core: host frames as uint8 RGB and as Blender's float32 RGBA,
no database.  `POLYCHASE_COPY_THREADS=1` gives the single-threaded copy for comparison.

    python tools/ingest_bench.py [--config c2] [--frames 120]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "polychase_amd", "core"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--frames", type=int, default=120)
    a = ap.parse_args()
    import numpy as np
    import torch  # noqa: F401
    import polychase_core as core
    from polychase_amd import synth

    w, h, ml = {"c2": (1920, 1080, 3), "c3": (3840, 2160, 4)}[a.config]
    clip = synth.NoiseClip(w, h, 24)
    u8 = [clip.frame(t) for t in range(24)]
    f32 = [np.concatenate([x.astype(np.float32) / 255.0, np.ones((h, w, 1), np.float32)], axis=2) for x in u8[:12]]
    fo = core.OpticalFlowOptions()
    fo.max_level = ml
    vi = core.VideoInfo(w, h, 1, a.frames)
    out = {"config": a.config, "frames": a.frames, "copy_threads": os.environ.get("POLYCHASE_COPY_THREADS", "default")}
    for name, frames in (("uint8_rgb", u8), ("float32_rgba", f32)):
        n = len(frames)

        def acc(fid):
            t = (fid - 1) % (2 * n - 2)
            return frames[t if t < n else 2 * n - 2 - t]
        core.generate_optical_flow_database(core.VideoInfo(w, h, 1, 20), acc, None, "", core.GFTTOptions(), fo)   # warm-up
        t0 = time.perf_counter()
        core.generate_optical_flow_database(vi, acc, None, "", core.GFTTOptions(), fo)
        dt = time.perf_counter() - t0
        out[name] = {"fps": a.frames / dt, "host_GBps": a.frames * frames[0].nbytes / dt / 1e9}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
