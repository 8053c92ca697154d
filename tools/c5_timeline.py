#!/usr/bin/env python3
"""c5_timeline.py -- the tracking / refinement half of a C5 run out of a rocprofv3 trace:

    rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d DIR -- python tests/c5_endtoend.py ...
    python tools/c5_timeline.py DIR [--out summary.json] [--print-frame]

Splits the kernel trace into the three phases of C5 by kernel name (analysis: lk3 / level / min-eig ...; tracking:
corr_* / pnp_*; refinement: refine_*), and for the tracking phase reports per-kernel calls / total / average duration,
launches and copies per frame, the GPU-busy share of the phase's wall time and the timeline of one frame in the middle.
"""
import argparse
import csv
import glob
import json
import os
import statistics as st

TRACK = ("corr_", "pnp_", "track_")
REFINE = ("refine_",)


def load(d):
    k = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    m = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)
    ev = []
    for r in csv.DictReader(open(k[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "kernel", r["Kernel_Name"].split("(")[0].replace("pc::", "")))
    if m:
        for r in csv.DictReader(open(m[0])):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy",
                       (r.get("Direction", "") + " " + str(r.get("Bytes", r.get("Size", "")))).strip()))
    ev.sort()
    return ev


def phase_summary(ev, prefixes, frame_marker=None):
    ks = [e for e in ev if e[2] == "kernel" and e[3].startswith(prefixes)]
    if not ks:
        return None
    t0, t1 = ks[0][0], max(e[1] for e in ks)
    inside = [e for e in ev if e[0] >= t0 and e[1] <= t1]
    by = {}
    for s, e, kind, name in inside:
        key = name if kind == "kernel" else "copy " + name.split(" ")[0]
        c = by.setdefault(key, [0, 0])
        c[0] += 1
        c[1] += e - s
    # union of busy intervals
    busy, cs, ce = 0, None, None
    for s, e, _, _ in sorted(inside):
        if ce is None or s > ce:
            if ce is not None:
                busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += (ce - cs) if ce is not None else 0
    out = {"wall_ms": (t1 - t0) / 1e6, "gpu_busy_ms": busy / 1e6, "gpu_busy_share": busy / max(1, t1 - t0),
           "by_name": {k: {"calls": v[0], "total_ms": v[1] / 1e6, "avg_us": v[1] / v[0] / 1e3} for k, v in
                       sorted(by.items(), key=lambda kv: -kv[1][1])}}
    if frame_marker:
        marks = [e for e in ks if e[3].startswith(frame_marker)]
        if len(marks) > 4:
            per = [(b[0] - a[0]) / 1e3 for a, b in zip(marks, marks[1:])]
            out["frames"] = len(marks)
            out["period_us_median"] = st.median(per)
            out["period_us_mean"] = st.mean(per)
            out["launches_per_frame"] = sum(v[0] for k, v in by.items() if not k.startswith("copy")) / len(marks)
            out["copies_per_frame"] = sum(v[0] for k, v in by.items() if k.startswith("copy")) / len(marks)
    return out, inside, ks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--out")
    ap.add_argument("--print-frame", action="store_true")
    a = ap.parse_args()
    ev = load(a.dir)
    res = {}
    tr = phase_summary(ev, TRACK, frame_marker="pnp_cost_lm")
    if tr is None:
        tr = phase_summary(ev, TRACK, frame_marker="track_frame")
    if tr:
        res["tracking"] = tr[0]
        if a.print_frame:
            marks = [e for e in tr[2] if e[3].startswith(("pnp_cost_lm", "track_frame"))]
            if len(marks) > 8:
                lo, hi = marks[len(marks) // 2][1], marks[len(marks) // 2 + 1][1]
                print("one tracked frame, us relative to the end of the previous frame's last kernel:")
                for s, e, kind, name in tr[1]:
                    if s >= lo and e <= hi:
                        print(f"  {(s - lo) / 1e3:9.1f} .. {(e - lo) / 1e3:9.1f} ({(e - s) / 1e3:7.1f})  {kind:6s} {name[:70]}")
    rf = phase_summary(ev, REFINE)
    if rf:
        res["refinement"] = rf[0]
    print(json.dumps(res, indent=1))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
