#!/bin/bash
# Run ON the GPU box: the ceiling of "defer the stragglers" (VERDICT r04 #8).  The LK launch alone with term_max_iters cut from the
# reference's 30 down to T: every pair still iterating after T iterations is simply dropped (WRONG results for those pairs -- a
# measurement, not a mode).  launch(30) - launch(T) is the most that handing those pairs to re-packed straggler wavefronts could
# save, before the cost of the hand-over (state out and in, the I side of the straggler's keypoint again, the second launch).
#   tools/lk_tail_ceiling.sh <tag>   -> gpurun_out/<tag>_lk_tail_ceiling.jsonl
TAG=${1:-r05}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${TAG}_lk_tail_ceiling.jsonl
: > "$OUT"
for c in c2 c3; do
  for a in canonical opencv_x86; do
    for t in 30 24 20 16 12 10 8; do
      timeout 300 python "$ROOT/tools/lk_bench.py" --config $c --arith $a --max-iters $t 2>/dev/null | grep "^{" >> "$OUT"
    done
  done
done
cat "$OUT" | python -c "
import sys, json
rows=[json.loads(l) for l in sys.stdin]
base={}
for r in rows:
    k=(r['config'],r['arith'])
    if r['max_iters']==30: base[k]=r['lk_ms_per_launch']
for r in rows:
    k=(r['config'],r['arith'])
    print(r['config'], r['arith'], 'T', r['max_iters'], 'ms %.4f' % r['lk_ms_per_launch'], 'saved %.1f %%' % (100*(1-r['lk_ms_per_launch']/base[k])), 'tracked', r['tracked_rows'])
"
