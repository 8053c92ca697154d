#!/bin/bash
# Run ON the GPU box: HIP / ROCr runtime switches that change launch and wake-up latencies, on the benchmark loop and the
# product call (device frames / host frames / with the SQLite insert).
#   tools/runtime_knobs.sh > gpurun_out/runtime_knobs.jsonl
ROOT=$(cd "$(dirname "$0")/.." && pwd)
one() { # label ; env from the caller
  b=$(timeout 300 python "$ROOT/bench.py" --no-cpu-baseline --no-end-to-end --no-breakdown --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value'],1), round(d['c3']['value'],1))")
  e=$(timeout 300 python "$ROOT/tools/e2e_bench.py" --config c2 --frames 300 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['device_frames_no_db']['fps'],1), round(d['host_frames_no_db']['fps'],1), round(d['host_frames_sqlite']['fps'],1))")
  echo "{\"knob\": \"$1\", \"bench_c2_c3\": \"$b\", \"e2e_dev_host_sqlite\": \"$e\"}"
}
one "defaults"
HIP_FORCE_DEV_KERNARG=1 one "HIP_FORCE_DEV_KERNARG=1"
HIP_FORCE_DEV_KERNARG=0 one "HIP_FORCE_DEV_KERNARG=0"
HSA_ENABLE_INTERRUPT=0 one "HSA_ENABLE_INTERRUPT=0"
ROC_ACTIVE_WAIT_TIMEOUT=200 one "ROC_ACTIVE_WAIT_TIMEOUT=200"
one "defaults again"
