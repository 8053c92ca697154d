#!/bin/bash
# Run ON the GPU box: kernel + memory-copy timeline of a short bench run, printed for one LK period.  tools/gpu_timeline.sh [config]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CFG=${1:-c3}
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/ktr && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ktr -- python "$ROOT/bench.py" --no-cpu-baseline --no-c3 --no-end-to-end --no-breakdown --config $CFG --steps 30 > /tmp/ktr.log 2>&1
k=$(find /tmp/ktr -name "*kernel_trace.csv" | head -1)
m=$(find /tmp/ktr -name "*memory_copy_trace.csv" | head -1)
python - "$k" "$m" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q" + r.get("Queue_Id", "?"), r["Kernel_Name"][:60]) for r in rows]
try:
    for r in csv.DictReader(open(sys.argv[2])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
except Exception as e:
    print("no copy trace", e)
ev.sort()
lk = [e for e in ev if "lk3_kernel" in e[3]]
lk = lk[len(lk) * 2 // 3:]
a, b = lk[2], lk[5]
t0 = a[0]
# union of LK intervals / no-LK time in the window
import itertools
win = [e for e in lk if e[0] >= a[0] and e[1] <= lk[-1][1]]
cov, cur_s, cur_e = 0, None, None
for s, e, _, _ in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None: cov += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
cov += cur_e - cur_s
span = win[-1][1] - win[0][0]
print(f"LK coverage {cov/span:.3f} of {span/1e3:.0f} us over {len(win)} launches; period {span/1e3/(len(win)-1):.1f} us")
for s, e, q, n in ev:
    if e >= t0 and s <= b[1]:
        print(f"  {(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f} ({(e-s)/1e3:7.1f})  {q:>5}  {n}")
PY
