#!/bin/bash
# Run ON the GPU box: tools/x86_cost_probe.py with the product library and with lib/variants/libpolychase_hip_base.so, alternating
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
N=${1:-2}
for i in $(seq $N); do
  POLYCHASE_HIP_LIB=$ROOT/polychase_amd/lib/variants/libpolychase_hip_base.so python $ROOT/tools/x86_cost_probe.py > $ROOT/gpurun_out/x86ab_base_$i.jsonl 2>/dev/null
  python $ROOT/tools/x86_cost_probe.py > $ROOT/gpurun_out/x86ab_new_$i.jsonl 2>/dev/null
done
