#!/bin/bash
# Run ON the GPU box: libraries in polychase_amd/lib/variants (the LK translation unit compiled with other compiler flags)
# against the product library: the isolated launch in both arithmetic modes (tools/lk_bench.py) and the whole pipeline
# (tools/lane_probe.py), alternating, REPS times.   tools/probes/flag_ab.sh <tag> [reps]
TAG=${1:-flags}; REPS=${2:-2}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/${TAG}_ab.jsonl
: > "$OUT"
for rep in $(seq $REPS); do
  for lib in default $ROOT/polychase_amd/lib/variants/libpolychase_hip_*.so; do
    name=default; envs=()
    if [ "$lib" != default ]; then name=$(basename $lib .so); name=${name#libpolychase_hip_}; envs=(POLYCHASE_HIP_LIB=$lib); fi
    for cfg in c2 c3; do
      can=$(env "${envs[@]}" python "$ROOT/tools/lk_bench.py" --config $cfg 2>/dev/null | grep '^{' | tail -1)
      x86=$(env "${envs[@]}" python "$ROOT/tools/lk_bench.py" --config $cfg --arith opencv_x86 2>/dev/null | grep '^{' | tail -1)
      steps=200; [ $cfg = c3 ] && steps=80
      pipe=$(env "${envs[@]}" python "$ROOT/tools/lane_probe.py" --config $cfg --steps $steps --modes full 2>/dev/null | grep '^{' | tail -1)
      echo "{\"variant\": \"$name\", \"rep\": $rep, \"config\": \"$cfg\", \"canonical\": ${can:-null}, \"x86\": ${x86:-null}, \"pipeline\": ${pipe:-null}}" >> "$OUT"
    done
  done
done
