#!/bin/bash
# Run ON the GPU box: isolated LK launch and bench.py frames/s for the default library and every variant under
# polychase_amd/lib/variants/ (tools/lk_variants/lk_variants.py build ...).   tools/gpu_ab.sh [config] [variant ...]
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CFG=${1:-c2}; shift
cd "$ROOT"
python tools/lk_variants/lk_variants.py run --config $CFG --reps 20 "$@" 2>&1 | cut -c1-260
LIBS="default $@"
[ $# -eq 0 ] && LIBS="default $(ls polychase_amd/lib/variants 2>/dev/null | sed 's/libpolychase_hip_//; s/.so//')"
for l in $LIBS; do
  if [ $l = default ]; then unset POLYCHASE_HIP_LIB; else export POLYCHASE_HIP_LIB=$ROOT/polychase_amd/lib/variants/libpolychase_hip_$l.so; fi
  python bench.py --no-cpu-baseline --no-c3 --no-end-to-end --config $CFG 2>/dev/null > gpurun_out/ab_${CFG}_$l.json
  python - "$l" "gpurun_out/ab_${CFG}_$l.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    k = d.get("kernel_ms_per_frame") or {}
    print(f"bench {sys.argv[1]:8s} fps {d['value']:.1f} ms/step {d['ms_per_step']:.3f} lk {d['roofline']['avg_launch_ms']:.3f} busy {d['roofline']['busy_ms_per_launch']:.3f} overlap {d['roofline']['launch_overlap']:.2f} | " + " ".join(f"{a}={b:.3f}" for a, b in k.items()))
except Exception as e:
    print("bench", sys.argv[1], "FAILED", e)
PY
done
