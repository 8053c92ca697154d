#!/usr/bin/env python3
"""fetch_calib.py -- calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on kernels with known bytes (run ON the GPU box).

    python tools/fetch_calib.py build                 (here: hipcc cross-compiles)
    python tools/fetch_calib.py run > profiles/r03_fetch_calibration.json

FETCH_SIZE and WRITE_SIZE are collected in separate `rocprofv3 --pmc` passes (they do not fit one; no other trace domain
is combined with --pmc).  Output: per kernel the known bytes, the counter (KiB) per launch, and counter * 1024 / known.
"""
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "bin", "fetch_calib")


def counters(name):
    d = tempfile.mkdtemp(prefix="calib_", dir="/tmp")
    r = subprocess.run(["rocprofv3", "--pmc", name, "--output-format", "csv", "-d", d, "--", BIN], cwd="/tmp",
                       env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    known = [l for l in r.stdout.splitlines() if l.startswith("{")]
    per = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            per.setdefault(k, []).append(float(row["Counter_Value"]))
    return (json.loads(known[-1]) if known else None), {k: sum(v) / len(v) for k, v in per.items()}, r.stdout[-800:]


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "run"
    if cmd == "build" or not os.path.exists(BIN):
        os.makedirs(os.path.dirname(BIN), exist_ok=True)
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", os.path.join(HERE, "fetch_calib.hip"), "-o", BIN])
    if cmd == "run":
        known, fetch, log1 = counters("FETCH_SIZE")
        _, write, log2 = counters("WRITE_SIZE")
        if not known or not fetch:
            sys.exit("no counters:\n" + log1 + log2)
        out = {"known": known, "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write}
        sr = fetch.get("stream_read16", 0) * 1024
        rw = fetch.get("rows28", 0) * 1024
        sw = write.get("stream_write16", 0) * 1024
        out["factors"] = {
            "stream_read16: FETCH_SIZE*1024 / bytes": sr / known["stream_read16"]["bytes"],
            "rows28: FETCH_SIZE*1024 / bytes asked": rw / known["rows28"]["bytes_asked"],
            "rows28: FETCH_SIZE*1024 / distinct 64-B lines": rw / known["rows28"]["lines64_bytes"],
            "rows28: FETCH_SIZE*1024 / distinct 128-B lines": rw / known["rows28"]["lines128_bytes"],
            "stream_write16: WRITE_SIZE*1024 / bytes": sw / known["stream_write16"]["bytes"],
        }
        print(json.dumps(out, indent=1))
