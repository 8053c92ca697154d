#!/usr/bin/env python3
"""Longest kernel launches in a rocprofv3 kernel-trace CSV.  usage: slow_kernels.py trace.csv [N]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows.sort(key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), reverse=True)
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for r in rows[:n]:
    print(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:10.1f} us  at {(int(r['Start_Timestamp']) - t0) / 1e6:9.2f} ms  {r['Kernel_Name'][:80]}")
