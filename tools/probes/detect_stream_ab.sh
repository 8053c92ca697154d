#!/bin/bash
# Run ON the GPU box: the pipeline step with the detection on the preparation stream (default) / on a stream of its own
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/detect_stream_ab.jsonl; : > $OUT
for rep in 1 2 3; do for ds in 0 1; do for hp in auto 0 1; do for cfg in c2 c3; do
  steps=300; [ $cfg = c3 ] && steps=100
  r=$(POLYCHASE_DETECT_STREAMS=$ds POLYCHASE_HELPER_PRIO=$hp GPU_MAX_HW_QUEUES=16 python $ROOT/tools/lane_probe.py --config $cfg --steps $steps --modes full 2>/dev/null | grep '^{' | tail -1)
  echo "{\"detect_streams\": $ds, \"helper_prio\": \"$hp\", \"config\": \"$cfg\", \"rep\": $rep, \"probe\": ${r:-null}}" >> $OUT
done; done; done; done
