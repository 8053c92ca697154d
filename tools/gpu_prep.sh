#!/bin/bash
# Run ON the GPU box: standalone frame-preparation kernels, with rocprofv3 per-kernel stats.  tools/gpu_prep.sh [tag]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-prep}
export TMPDIR=/tmp
cd /tmp
for c in c2 c3; do
  python "$ROOT/tools/prep_bench.py" --config $c 2>/dev/null | tail -1 | tee "$ROOT/gpurun_out/${TAG}_${c}_prep.json"
  rm -rf /tmp/kst && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -- python "$ROOT/tools/prep_bench.py" --config $c > /tmp/kst.log 2>&1
  f=$(find /tmp/kst -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$ROOT/gpurun_out/${TAG}_${c}_prep_kernel_stats.csv" && head -14 "$f" | cut -d, -f1-5
done
