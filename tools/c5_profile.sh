#!/bin/bash
# Run ON the GPU box: the tracking / refinement half of BASELINE's C5 (1920x1080 x 300 rendered frames through
# polychase_core) -- host stage clock, rocprofv3 kernel statistics + kernel / copy timeline of the same command.
#   tools/c5_profile.sh <tag> [frames]   -> gpurun_out/<tag>_c5_*
set -u
TAG=${1:-r05}
FRAMES=${2:-300}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
POLYCHASE_TRACE_STAGES=1 timeout -k 10 600 python "$ROOT/tests/c5_endtoend.py" --frames $FRAMES --oracle-frames 0 \
  --out "$OUT/${TAG}_c5_endtoend.json" > /tmp/c5_plain.log 2> "$OUT/${TAG}_c5_stages.txt"
# the kernel trace; the copy trace is tried first and dropped if the profiler does not survive it
for extra in ""; do   # (--memory-copy-trace: rocprofv3 segfaults at exit with it on this image)
  rm -rf /tmp/c5k
  timeout -k 10 900 rocprofv3 --kernel-trace $extra --stats --output-format csv -d /tmp/c5k -- \
    python "$ROOT/tests/c5_endtoend.py" --frames $FRAMES --oracle-frames 0 --out "$OUT/${TAG}_c5_endtoend_under_rocprofv3.json" > /tmp/c5k.log 2>&1
  f=$(find /tmp/c5k -name "*kernel_stats.csv" 2>/dev/null | head -1)
  [ -n "$f" ] && break
  echo "rocprofv3 --kernel-trace $extra: no output"; grep -v "^{" /tmp/c5k.log | tail -12
done
[ -n "$f" ] && cp "$f" "$OUT/${TAG}_c5_rocprofv3_kernel_stats.csv"
python "$ROOT/tools/c5_timeline.py" /tmp/c5k --out "$OUT/${TAG}_c5_timeline.json" --print-frame > "$OUT/${TAG}_c5_timeline.txt" 2>&1
tail -30 "$OUT/${TAG}_c5_stages.txt"
head -150 "$OUT/${TAG}_c5_timeline.txt"
