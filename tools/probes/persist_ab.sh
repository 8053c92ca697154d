#!/bin/bash
# Run ON the GPU box: the persistent-wavefront build of the LK kernel (lib/variants/libpolychase_hip_persist.so) against the product
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/persist_ab.jsonl
: > "$OUT"
V=$ROOT/polychase_amd/lib/variants/libpolychase_hip_persist.so
one() {  # name env...
  local name=$1; shift
  for cfg in c2 c3; do
    can=$(env "$@" python "$ROOT/tools/lk_bench.py" --config $cfg 2>/dev/null | grep '^{' | tail -1)
    x86=$(env "$@" python "$ROOT/tools/lk_bench.py" --config $cfg --arith opencv_x86 2>/dev/null | grep '^{' | tail -1)
    steps=200; [ $cfg = c3 ] && steps=80
    pipe=$(env "$@" python "$ROOT/tools/lane_probe.py" --config $cfg --steps $steps --modes lk,full 2>/dev/null | grep '^{' | tr '\n' ',' | sed 's/,$//')
    echo "{\"variant\": \"$name\", \"config\": \"$cfg\", \"canonical\": ${can:-null}, \"x86\": ${x86:-null}, \"pipeline\": [${pipe}]}" >> "$OUT"
  done
}
for rep in 1 2; do
  one default A=1
  one persist3072 POLYCHASE_HIP_LIB=$V
  one persist2560 POLYCHASE_HIP_LIB=$V PC_LK3_PERSIST_SLOTS=2560
  one persist4096 POLYCHASE_HIP_LIB=$V PC_LK3_PERSIST_SLOTS=4096
  one persist1536 POLYCHASE_HIP_LIB=$V PC_LK3_PERSIST_SLOTS=1536
done
