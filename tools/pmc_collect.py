#!/usr/bin/env python3
"""pmc_collect.py -- hardware counters of one kernel, averaged per dispatch (run ON the GPU box).

    python tools/pmc_collect.py [--kernel lk_kernel] [--config c2] [--out gpurun_out/pmc.json] GROUP [GROUP ...]

Each GROUP is a comma-separated list of counters collected in its own rocprofv3 pass
(`rocprofv3 --pmc ... --kernel-trace` is NOT combined with any other trace domain).
"""
import argparse
import csv
import glob
import json
import os
import signal
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="lk_kernel")
    ap.add_argument("--config", default="c2")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "pmc.json"))
    ap.add_argument("--by-kernel", action="store_true", help="per kernel name: counters summed over the run / frames (all kernels)")
    ap.add_argument("groups", nargs="+")
    args = ap.parse_args()
    # One job lane: counter collection serialises dispatches anyway, and without the second lane there is no gate
    # kernel polling for a dispatch the profiler is holding back.
    env = dict(os.environ, TMPDIR="/tmp", POLYCHASE_LK_LANES="1")
    result = {}
    for g in args.groups:
        d = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc", *g.split(","), "--output-format", "csv", "-d", d, "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-c3", "--no-end-to-end", "--config", args.config,
               "--steps", str(args.steps), "--warmup", "4"]
        r = None
        for attempt in range(2):   # a counter pass that hangs (seen once in r03) is killed and tried once more
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                                    start_new_session=True)
            try:
                out, _ = proc.communicate(timeout=180)
                r = subprocess.CompletedProcess(cmd, proc.returncode, out)
                break
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)   # the whole session: rocprofv3 and the bench under it
                proc.communicate()
                print(f"[{g}] rocprofv3 pass timed out (attempt {attempt + 1})", file=sys.stderr)
        if r is None:
            continue
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            print(f"[{g}] no counter file; rocprofv3 said:\n{r.stdout[-1500:]}", file=sys.stderr)
            continue
        sums, disp = {}, set()
        if args.by_kernel:
            per = {}
            for f in files:
                for row in csv.DictReader(open(f)):
                    name = row.get("Kernel_Name", "").split("(")[0][:48]
                    d2 = per.setdefault(name, {"_n": set()})
                    d2["_n"].add((f, row["Dispatch_Id"]))
                    d2[row["Counter_Name"]] = d2.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            for name, d2 in per.items():
                r = result.setdefault(name, {})
                r["dispatches"] = len(d2.pop("_n"))
                r.update(d2)
            continue
        for f in files:
            for row in csv.DictReader(open(f)):
                if args.kernel not in row.get("Kernel_Name", ""):
                    continue
                disp.add((f, row["Dispatch_Id"]))
                sums[row["Counter_Name"]] = sums.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        n = max(1, len(disp))
        for k, v in sums.items():
            result[k] = v / n
        result.setdefault("_dispatches", {})[g] = len(disp)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(result, open(args.out, "w"), indent=1, sort_keys=True)
    print(json.dumps(result, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
