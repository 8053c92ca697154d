#!/usr/bin/env python3
"""soak_parity.py -- the randomized parity sweep of test_random_configs_gpu.py over MANY more seeds (run on a GPU box).

    python tests/soak_parity.py [--first 96] [--count 2000] [--out gpurun_out/soak.json]

Every case compares the HIP path with the CPU oracle bit for bit (the same checks as the pytest cases, which cover
seeds 0-95); the first mismatch is reported with its seed.  Test infrastructure: this is a checker, not a product path.
"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=96)
    ap.add_argument("--count", type=int, default=2000)
    ap.add_argument("--out", default=None)
    ap.add_argument("--jobs", type=int, default=1, help="split the seed range over this many processes (the oracle is the slow side)")
    args = ap.parse_args()
    if args.jobs > 1:
        return fan_out(args)
    import test_arith_modes_gpu as A
    import test_random_configs_gpu as T
    from polychase_amd import hip

    ctx = hip.Context(0)
    t0 = time.time()
    failures = []
    for seed in range(args.first, args.first + args.count):
        try:
            T.test_random_configuration(ctx, seed)
            if seed % 10 == 0:   # every tenth seed also drives a whole random clip through the pipelined analyzer
                T.test_random_clip_through_the_analyzer(ctx, seed)
            if seed % 3 == 0:    # every third: the detector's other branches and the arithmetic modes
                T.test_random_detector_branch_and_arithmetic_mode(ctx, seed)
            if seed % 2 == 0:    # every second: the x86 summation order on hard content, every window 3..31 (both product kernels)
                A.test_x86_order_on_the_two_keypoint_kernel_random(ctx, seed)
        except AssertionError as e:
            import traceback
            tb = traceback.extract_tb(e.__traceback__)[-1]
            where = f"{os.path.basename(tb.filename)}:{tb.lineno}: {tb.line}"
            failures.append({"seed": seed, "where": where[:300], "what": str(e)[:1200]})
            print(f"MISMATCH seed {seed} at {where}: {str(e)[:1200]}", flush=True)
            if len(failures) >= 10:
                break
    ctx.close()
    out = {"first_seed": args.first, "cases": args.count, "mismatches": len(failures), "failures": failures,
           "seconds": round(time.time() - t0, 1)}
    print(json.dumps(out))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(out, open(args.out, "w"), indent=1)
    return 1 if failures else 0


def fan_out(args) -> int:
    import subprocess
    per = -(-args.count // args.jobs)
    procs = []
    for j in range(args.jobs):
        first = args.first + j * per
        n = min(per, args.first + args.count - first)
        if n <= 0:
            break
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--first", str(first), "--count", str(n)],
                                      stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
    t0 = time.time()
    parts = []
    for p in procs:
        out, _ = p.communicate()
        lines = [l for l in out.splitlines() if l.startswith("{")]
        parts.append(json.loads(lines[-1]) if lines else {"first_seed": None, "cases": 0, "mismatches": 1, "failures": [{"what": "worker died: " + out[-300:]}]})
    out = {"first_seed": args.first, "cases": args.count, "jobs": len(procs), "mismatches": sum(p["mismatches"] for p in parts),
           "failures": [f for p in parts for f in p["failures"]], "seconds": round(time.time() - t0, 1),
           "what": "per seed: test_random_configuration (default arithmetic = opencv_x86); every 10th a whole clip through the analyzer; "
                   "every 3rd the detector branches x arithmetic modes; every 2nd the x86 LK order on hard content, windows 3-31"}
    print(json.dumps(out))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(out, open(args.out, "w"), indent=1)
    return 1 if out["mismatches"] else 0


if __name__ == "__main__":
    sys.exit(main())
