#!/bin/bash
# Run ON the GPU box: the LK kernel variants of polychase_amd/lib/variants (tools/lk_variants/lk_variants.py build ...) and the other
# LK kernels of the library, each measured three ways -- the launch alone (tools/lk_bench.py), the job lanes alone and the
# whole pipeline (tools/lane_probe.py: modes lk, full).  One JSON line per variant -> gpurun_out/<tag>_lk_variants.jsonl
#   tools/lk_ab.sh <tag> [config]
TAG=${1:-r03}
CFG=${2:-c2}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/${TAG}_${CFG}_lk_variants.jsonl
: > "$OUT"
STEPS=200; [ "$CFG" = "c3" ] && STEPS=80
run() {   # name, env assignments...
  local name=$1; shift
  local iso lanes
  iso=$(env "$@" python "$ROOT/tools/lk_bench.py" --config $CFG 2>/dev/null | grep '^{' | tail -1)
  lanes=$(env "$@" python "$ROOT/tools/lane_probe.py" --config $CFG --steps $STEPS --modes lk,full 2>/dev/null | grep '^{' | tr '\n' ',' | sed 's/,$//')
  echo "{\"variant\": \"$name\", \"isolated\": ${iso:-null}, \"pipeline\": [${lanes}]}" >> "$OUT"
}
run default
run "lk (ONE keypoint per wavefront, 8 lanes per pair: max over 8 pairs)" POLYCHASE_LK_VARIANT=1
for v in occ4 occ2 waves2 srows2 nopf nopairs notrim prio1 prio2; do
  lib=$ROOT/polychase_amd/lib/variants/libpolychase_hip_$v.so
  [ -f "$lib" ] && run "$v" POLYCHASE_HIP_LIB=$lib
done
cat "$OUT"
