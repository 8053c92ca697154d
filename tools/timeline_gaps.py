#!/usr/bin/env python3
"""Gap analysis of a rocprofv3 --kernel-trace CSV: time between consecutive launches of one kernel
and what ran in between.   usage: timeline_gaps.py <kernel_trace.csv> [kernel-substring]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else "lk_kernel"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
lk = [r for r in rows if key in r["Kernel_Name"]]
lk = lk[len(lk) // 4:]           # skip warm-up
gaps, durs, periods = [], [], []
for a, b in zip(lk, lk[1:]):
    gaps.append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
    durs.append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
    periods.append((int(b["Start_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
import statistics as st
print(f"{len(lk)} launches of {key}: duration {st.mean(durs):.1f} us, gap to next {st.mean(gaps):.1f} us (median {st.median(gaps):.1f}), period {st.mean(periods):.1f} us")
a, b = lk[len(lk) // 2], lk[len(lk) // 2 + 1]
t0 = int(a["Start_Timestamp"])
print("one period, times in us relative to the LK start:")
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e >= t0 and s <= int(b["End_Timestamp"]):
        print(f"  {(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f}  q{r.get('Queue_Id', '?'):>3}  {r['Kernel_Name'][:70]}")
