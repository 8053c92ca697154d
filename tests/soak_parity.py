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
    args = ap.parse_args()
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
        except AssertionError as e:
            failures.append({"seed": seed, "what": str(e)[:400]})
            print(f"MISMATCH seed {seed}: {str(e)[:400]}", flush=True)
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


if __name__ == "__main__":
    sys.exit(main())
