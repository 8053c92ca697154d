#!/usr/bin/env python3
"""valu_issue.py -- build (here) / run (on the GPU box) the VALU issue-rate micro-benchmark tools/valu_issue.hip.

    python tools/valu_issue.py build
    python tools/valu_issue.py run [filter] > profiles/r03_valu_issue.json
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "bin", "valu_issue")

if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "run"
    if cmd == "build" or not os.path.exists(BIN):
        os.makedirs(os.path.dirname(BIN), exist_ok=True)
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", os.path.join(HERE, "valu_issue.hip"), "-o", BIN])
    if cmd == "run":
        sys.exit(subprocess.call([BIN, *sys.argv[2:]]))
