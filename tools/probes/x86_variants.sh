#!/bin/bash
# Run ON the GPU box: tools/x86_cost_probe.py with the product library and every library in lib/variants, alternating, N times
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
N=${1:-2}
for i in $(seq $N); do
  python $ROOT/tools/x86_cost_probe.py > $ROOT/gpurun_out/x86v_product_$i.jsonl 2>/dev/null
  for lib in $ROOT/polychase_amd/lib/variants/libpolychase_hip_*.so; do
    n=$(basename $lib .so); n=${n#libpolychase_hip_}
    POLYCHASE_HIP_LIB=$lib python $ROOT/tools/x86_cost_probe.py > $ROOT/gpurun_out/x86v_${n}_$i.jsonl 2>/dev/null
  done
done
