#!/usr/bin/env python3
"""ta_probe.py -- build (here) / run (GPU box) tools/ta_probe.hip; `counters` runs it under rocprofv3 --pmc and adds the
L1 accesses per load instruction.

    python tools/ta_probe.py build
    python tools/ta_probe.py run      > profiles/r04_ta_probe.jsonl
    python tools/ta_probe.py counters >> profiles/r04_ta_probe.jsonl
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "bin", "ta_probe")

if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "run"
    if cmd == "build" or not os.path.exists(BIN):
        os.makedirs(os.path.dirname(BIN), exist_ok=True)
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", os.path.join(HERE, "ta_probe.hip"), "-o", BIN])
    if cmd == "run":
        sys.exit(subprocess.call([BIN, *sys.argv[2:]]))
    if cmd == "counters":
        d = tempfile.mkdtemp(prefix="ta_probe_", dir="/tmp")
        env = dict(os.environ, TMPDIR="/tmp")
        subprocess.run(["rocprofv3", "--pmc", "TCP_TOTAL_CACHE_ACCESSES_sum", "SQ_INSTS_VMEM_RD", "TA_BUSY_avr", "GRBM_GUI_ACTIVE", "--kernel-trace",
                        "-d", d, "-o", "r", "--output-format", "csv", "--", BIN, *sys.argv[2:]],
                       cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600, check=True)
        f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        best = {}
        for (k, disp), c in acc.items():       # two dispatches per pattern: keep the long one
            if k not in best or c["SQ_INSTS_VMEM_RD"] > best[k]["SQ_INSTS_VMEM_RD"]:
                best[k] = c
        for k, c in best.items():
            n = c["SQ_INSTS_VMEM_RD"]
            print(json.dumps({"kernel": k.split("(")[0], "l1_accesses_per_load_instruction": round(c["TCP_TOTAL_CACHE_ACCESSES_sum"] / n, 1),
                              "ta_busy_fraction": round(c["TA_BUSY_avr"] / (c["GRBM_GUI_ACTIVE"] / 8), 3)}))
