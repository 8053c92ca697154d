#!/bin/bash
# Run ON the GPU box (through gpurun): refresh the measurement artefacts that profiles/ keeps.
#   tools/collect_profiles.sh <tag> [stage ...]     -> gpurun_out/<tag>_*
# Stages: bench kstats pmc probes e2e micro (default: all, in that order).  Every command runs under its own
# `timeout`: a counter pass of rocprofv3 has been seen to hang for good (r03: the 4K FETCH_SIZE pass sat for 48
# minutes until the call's limit), and one stuck pass must not cost the rest of the collection.
set -u
TAG=${1:-r01}
shift || true
STAGES=${*:-bench kstats pmc probes e2e micro}
want() { case " $STAGES " in *" $1 "*) return 0;; *) return 1;; esac; }
T="timeout -k 10 420"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
if want bench; then
for c in c2 c3; do
  $T python "$ROOT/bench.py" --no-c3 --config $c 2>/dev/null | tail -1 > "$OUT/${TAG}_${c}_bench.json"
done
# the driver's own command line
$T python "$ROOT/bench.py" --steps 20 --warmup 5 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_driver_style.json"
fi
# per-kernel durations (rocprofv3 --kernel-trace --stats) of the default bench command, and of the 4K configuration.
# POLYCHASE_LK_LANES=1: all jobs on one lane, so no two LK launches overlap and a dispatch's duration is the launch's
# own (with two lanes the tail of a launch overlaps the next one: start-to-end times exceed the GPU time a launch
# costs, and under the profiler -- which serialises dispatch hand-over -- the gate between the lanes only adds its
# polling).  The bench line saved beside the CSV is the same profiled run: its roofline.avg_launch_ms is the
# number the CSV's average must agree with.
if want kstats; then
for c in c2 c3; do
  rm -rf /tmp/kstats && POLYCHASE_LK_LANES=1 $T rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats -- python "$ROOT/bench.py" --no-cpu-baseline --no-c3 --no-end-to-end --no-breakdown --config $c > /tmp/kstats.log 2>&1
  grep "^{" /tmp/kstats.log | tail -1 > "$OUT/${TAG}_${c}_bench_under_rocprofv3.json"
  f=$(find /tmp/kstats -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${TAG}_${c}_rocprofv3_kernel_stats.csv"
done
fi
# HBM traffic of the LK launch: FETCH_SIZE and WRITE_SIZE in separate passes (they do not fit one)
if want pmc; then
for c in c2 c3; do
  $T python "$ROOT/tools/pmc_collect.py" --kernel lk3_kernel --config $c --steps 10 --out "$OUT/${TAG}_${c}_lk_hbm_pmc.json" FETCH_SIZE WRITE_SIZE > /dev/null 2>&1
done
for c in c2 c3; do
$T python "$ROOT/tools/pmc_collect.py" --kernel lk3_kernel --config $c --steps 10 --out "$OUT/${TAG}_${c}_lk_sq_pmc.json" \
  SQ_WAVES,SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_INSTS_VMEM_RD,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES \
  SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS,SQ_ACTIVE_INST_ANY,SQ_WAIT_INST_ANY,SQ_WAIT_INST_LDS,SQ_WAIT_ANY > /dev/null 2>&1
done
$T python "$ROOT/tools/pmc_collect.py" --by-kernel --config c2 --steps 10 --out "$OUT/${TAG}_c2_pipeline_by_kernel_pmc.json" SQ_WAVES,SQ_INSTS_VALU,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES > /dev/null 2>&1
fi
# where a step goes: job lanes alone / + pyramid / + detection, helper priority pinned and automatic
if want probes; then
for c in c2 c3; do
  for m in auto 0 1; do POLYCHASE_HELPER_PRIO=$m $T python "$ROOT/tools/lane_probe.py" --config $c --steps 200 2>/dev/null | grep "^{" | sed "s/^{/{\"helper_prio\": \"$m\", /"; done > "$OUT/${TAG}_${c}_lane_probe.jsonl"
  $T python "$ROOT/tools/lk_bench.py" --config $c 2>/dev/null | grep "^{" > "$OUT/${TAG}_${c}_lk_isolated.json"
  $T python "$ROOT/tools/prep_bench.py" --config $c 2>/dev/null | grep "^{" > "$OUT/${TAG}_${c}_prep.json"
done
fi
if want e2e; then
$T python "$ROOT/tools/e2e_bench.py" --config c2 --frames 300 2>/dev/null | grep "^{" > "$OUT/${TAG}_e2e_c2.json"
$T python "$ROOT/tools/e2e_bench.py" --config c3 --frames 100 2>/dev/null | grep "^{" > "$OUT/${TAG}_e2e_c3.json"
$T python "$ROOT/tools/ingest_bench.py" --config c2 --frames 300 2>/dev/null | grep "^{" > "$OUT/${TAG}_ingest.json"
fi
if want micro; then
$T "$ROOT/tools/bin/valu_issue" > "$OUT/${TAG}_valu_issue.json" 2>/dev/null
$T python "$ROOT/tools/fetch_calib.py" run > "$OUT/${TAG}_fetch_calibration.json" 2>/dev/null
# what runs beside the LK launches: a kernel of RCCL's resource shape, a slim one, copies (tools/coresidency.hip)
[ -f "$ROOT/tools/bin/libcoresidency.so" ] && $T python "$ROOT/tools/coresidency_probe.py" --config c2 2>/dev/null | grep "^{" > "$OUT/${TAG}_coresidency_c2.jsonl"
fi
ls -la "$OUT" | grep "$TAG"
