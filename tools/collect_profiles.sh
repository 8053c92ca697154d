#!/bin/bash
# Run ON the GPU box (through gpurun): refresh the measurement artefacts that profiles/ keeps.
#   tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>_*
set -u
TAG=${1:-r01}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for c in c2 c3; do
  python "$ROOT/bench.py" --no-c3 --config $c 2>/dev/null | tail -1 > "$OUT/${TAG}_${c}_bench.json"
done
# per-kernel durations (rocprofv3 --kernel-trace --stats) of the default bench command, and of the 4K configuration
for c in c2 c3; do
  rm -rf /tmp/kstats && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats -- python "$ROOT/bench.py" --no-cpu-baseline --no-c3 --no-end-to-end --no-breakdown --config $c > /tmp/kstats.log 2>&1
  grep "^{" /tmp/kstats.log | tail -1 > "$OUT/${TAG}_${c}_bench_under_rocprofv3.json"
  f=$(find /tmp/kstats -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${TAG}_${c}_rocprofv3_kernel_stats.csv"
done
# HBM traffic of the LK launch: FETCH_SIZE and WRITE_SIZE in separate passes (they do not fit one)
for c in c2 c3; do
  python "$ROOT/tools/pmc_collect.py" --kernel lk3_kernel --config $c --steps 10 --out "$OUT/${TAG}_${c}_lk_hbm_pmc.json" FETCH_SIZE WRITE_SIZE > /dev/null 2>&1
done
python "$ROOT/tools/pmc_collect.py" --kernel lk3_kernel --config c2 --steps 10 --out "$OUT/${TAG}_c2_lk_sq_pmc.json" \
  SQ_WAVES,SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_INSTS_VMEM_RD,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES \
  SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS,SQ_ACTIVE_INST_ANY,SQ_WAIT_INST_ANY,SQ_WAIT_INST_LDS,SQ_WAIT_ANY > /dev/null 2>&1
ls -la "$OUT" | grep "$TAG"
