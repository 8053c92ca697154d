#!/usr/bin/env python3
"""Dump the kernel summary (calls, total/avg duration in us, %) of a rocprofv3 rocpd .db file as CSV.

usage: tools/rocpd_summary.py gpurun_out/prof/xxx_results.db > profiles/rNN_name_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
    for name, calls, total, avg, pct in rows:
        short = name if len(name) < 160 else name[:157] + "..."
        w.writerow([short, calls, f"{total:.3f}", f"{avg:.3f}", f"{pct:.3f}"])


if __name__ == "__main__":
    main()
