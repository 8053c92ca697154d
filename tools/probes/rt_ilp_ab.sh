#!/bin/bash
# Run ON the GPU box: C5's tracking / refinement half with the product library and with a variant library preloaded
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd /tmp
for rep in 1 2 3; do
  python $ROOT/tests/c5_endtoend.py --frames 300 --oracle-frames 0 --out $ROOT/gpurun_out/rt_default_$rep.json > /tmp/rt.log 2>&1 || tail -3 /tmp/rt.log
  LD_PRELOAD=$ROOT/polychase_amd/lib/variants/libpolychase_hip_rtilp.so python $ROOT/tests/c5_endtoend.py --frames 300 --oracle-frames 0 --out $ROOT/gpurun_out/rt_ilp_$rep.json > /tmp/rt.log 2>&1 || tail -3 /tmp/rt.log
done
