#!/bin/bash
# Run ON the GPU box: the step time for the ways the preparation work can be spread over streams / hardware queues.
#   tools/stream_layouts.sh > gpurun_out/stream_layouts.jsonl
ROOT=$(cd "$(dirname "$0")/.." && pwd)
run() { # label, config ; env comes from the caller
  timeout 300 python "$ROOT/bench.py" --no-cpu-baseline --no-c3 --no-end-to-end --config $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print(json.dumps({'layout': '$1', 'config': '$2', 'fps': round(d['value'],1), 'ms_per_step': round(d['ms_per_step'],4), 'lk_busy_ms': round(d['roofline'].get('busy_ms_per_launch', 0),4), 'kernel_ms_per_frame': {k: round(v,3) for k,v in d['kernel_ms_per_frame'].items()}}))"
}
for c in c2 c3; do
  run "default (one preparation stream, 4 queues)" $c
  POLYCHASE_DETECT_STREAMS=1 run "detection stream, 4 queues" $c
  POLYCHASE_DETECT_STREAMS=1 POLYCHASE_HELPER_PRIO=1 run "detection stream, 4 queues, helpers high" $c
  POLYCHASE_DETECT_STREAMS=1 POLYCHASE_DETECT_CUMASK=1 run "detection stream with a CU mask (own queue), 4 queues" $c
  POLYCHASE_DETECT_STREAMS=1 POLYCHASE_DETECT_CUMASK=1 POLYCHASE_HELPER_PRIO=1 run "detection stream with a CU mask, 4 queues, helpers high" $c
  POLYCHASE_DETECT_STREAMS=2 POLYCHASE_DETECT_CUMASK=1 run "two detection streams with CU masks, 4 queues" $c
  GPU_MAX_HW_QUEUES=8 run "one preparation stream, 8 queues" $c
  GPU_MAX_HW_QUEUES=8 POLYCHASE_DETECT_STREAMS=1 run "detection stream, 8 queues" $c
  GPU_MAX_HW_QUEUES=8 POLYCHASE_DETECT_STREAMS=1 POLYCHASE_HELPER_PRIO=1 run "detection stream, 8 queues, helpers high" $c
  GPU_MAX_HW_QUEUES=8 POLYCHASE_DETECT_STREAMS=2 run "two detection streams, 8 queues" $c
  GPU_MAX_HW_QUEUES=8 POLYCHASE_DETECT_STREAMS=2 POLYCHASE_HELPER_PRIO=1 run "two detection streams, 8 queues, helpers high" $c
done
