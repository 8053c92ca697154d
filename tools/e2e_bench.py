#!/usr/bin/env python3
"""End-to-end rates of GenerateOpticalFlowDatabase through the polychase_core module (what the Blender
addon experiences), next to bench.py's HBM-resident number.  Not the headline metric.

  python tools/e2e_bench.py [--config c2|c3] [--frames 60]

Modes: frames as torch CUDA tensors / host numpy arrays (PCIe upload included), with and without the
SQLite insert."""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--frames", type=int, default=300)
    a = ap.parse_args()
    import torch
    from polychase_amd import synth
    sys.path.insert(0, os.path.join(ROOT, "polychase_amd", "core"))
    import polychase_core as core

    w, h, ml = {"c2": (1920, 1080, 3), "c3": (3840, 2160, 4)}[a.config]
    clip = synth.NoiseClip(w, h, max(a.frames, 30), device="cuda")
    dev = [clip.frame_torch(t) for t in range(a.frames)]
    torch.cuda.synchronize()
    host = [f.cpu().numpy() for f in dev]
    fo = core.OpticalFlowOptions()
    fo.max_level = ml
    vi = core.VideoInfo(w, h, 1, a.frames)
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for name, frames, db in [("device_frames_no_db", dev, ""), ("host_frames_no_db", host, ""),
                                 ("host_frames_sqlite", host, os.path.join(td, "a.db")),
                                 ("device_frames_sqlite", dev, os.path.join(td, "b.db"))]:
            core.generate_optical_flow_database(core.VideoInfo(w, h, 1, 12), lambda f: frames[f - 1], None, "", core.GFTTOptions(), fo)
            c0 = core._async_write_counters()
            t0 = time.perf_counter()
            st = core.generate_optical_flow_database(vi, lambda f: frames[f - 1], None, db, core.GFTTOptions(), fo)
            dt = time.perf_counter() - t0
            c1 = core._async_write_counters()
            out[name] = {"fps": a.frames / dt, "fps_without_setup": a.frames / (dt - st.seconds_setup), "seconds_db": st.seconds_db,
                         "db_bytes": os.path.getsize(db) if db else 0,
                         # the driver thread's stage clock, ms per frame
                         "driver_ms_per_frame": {k: round(1e3 * getattr(st, "seconds_" + k) / a.frames, 4)
                                                 for k in ("accessor", "put", "submit", "collect", "writer_wait")},
                         "setup_ms": round(1e3 * st.seconds_setup, 1),
                         # page writes of the database file: handed to the worker threads / carried out by SQLite's own thread
                         "db_page_writes": {k: c1[k] - c0[k] for k in c1}}
    print(json.dumps({"config": a.config, "frames": a.frames, **out}))


if __name__ == "__main__":
    main()
