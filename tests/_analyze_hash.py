#!/usr/bin/env python3
"""Helper of tests/test_env_variants_gpu.py: analyze a small synthetic clip through polychase_core (the product path:
C++ driver over the C ABI) and print the sha256 of the database's logical content.  Run in a subprocess, because the
knobs under test are read from the environment when the library / the context is created."""
import hashlib
import os
import sqlite3
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "polychase_amd", "core"))


def main():
    w, h, n = (int(x) for x in sys.argv[1:4])
    import torch  # noqa: F401  (device discovery like the other GPU tests)
    import polychase_core as core
    from polychase_amd import synth

    clip = synth.NoiseClip(w, h, n)
    frames = [clip.frame(t) for t in range(n)]
    vi = core.VideoInfo(w, h, 1, n)
    fo = core.OpticalFlowOptions()
    fo.max_level = 3
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "clip.db")
        core.generate_optical_flow_database(vi, lambda fid: frames[fid - 1], None, path, core.GFTTOptions(), fo)
        con = sqlite3.connect(path)
        hsh = hashlib.sha256()
        for row in con.execute("select image_id, rows, keypoints from keypoints order by image_id"):
            hsh.update(repr(row[:2]).encode())
            hsh.update(row[2])
        for row in con.execute("select image_id_from, image_id_to, rows, src_keypoints_indices, tgt_keypoints, flow_errors "
                               "from optical_flow order by image_id_from, image_id_to"):
            hsh.update(repr(row[:3]).encode())
            for blob in row[3:]:
                hsh.update(blob)
        con.close()
    print("HASH", hsh.hexdigest())


if __name__ == "__main__":
    main()
