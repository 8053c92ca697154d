#!/usr/bin/env python3
"""c5_endtoend.py -- BASELINE config C5 at full size, on a GPU box:

    frames RENDERED from a textured plane under a known camera trajectory
      -> generate_optical_flow_database   (GFTT + pyramidal LK on the GPU, SQLite)
      -> track_sequence                   (ray casting + PnP on the GPU; tracker.cc)
      -> refine_trajectory                (refiner.cc on the GPU)
    timings of every stage, pose error against the ground truth, and -- for the first frames -- against the CPU
    reference of the tracking step (float64 numpy restatement of tracker.cc, oracle/pnp_oracle.py) run on the same
    database.

    python tests/c5_endtoend.py [--width 1920 --height 1080 --frames 300 --oracle-frames 6 --out gpurun_out/c5.json]

Test infrastructure (it uses the oracle as the checker); tests/test_c5_gpu.py runs it at 1920x1080 with 40 frames
inside `pytest -m gpu` and asserts the pose tolerances; the miniature is
tests/test_tracker_gpu.py::test_c5_end_to_end_rendered_plane.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--oracle-frames", type=int, default=6, help="frames solved by the CPU reference as well")
    ap.add_argument("--oracle-stride", type=int, default=0,
                    help="> 0: every stride-th frame of the WHOLE clip is also solved by the CPU reference, from the GPU's poses of "
                         "its source frames and the GPU's pose of the frame before it as the initial guess")
    ap.add_argument("--oracle-workers", type=int, default=1, help="threads over the sampled frames of --oracle-stride")
    ap.add_argument("--refine-iterations", type=int, default=30)
    ap.add_argument("--out", default=None)
    a = ap.parse_args(argv)
    out = run(a.width, a.height, a.frames, a.oracle_frames, a.oracle_stride, a.refine_iterations, a.oracle_workers)
    for fn in os.listdir(td):
        os.remove(os.path.join(td, fn))
    os.rmdir(td)
    return out


if __name__ == "__main__":
    sys.exit(main())
