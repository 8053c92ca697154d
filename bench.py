#!/usr/bin/env python3
"""bench.py -- optical-flow frames/s of the MI355X hot path on a synthetic clip.

A "step" = one frame1 of the clip: make the new neighbour frame resident (RGB->gray + LK pyramid),
detect keypoints (GFTT), track them into its 8 neighbours (+-1,2,4,8) with pyramidal LK, filter
status==1 and deliver the records to the host -- i.e. one iteration of the outer loop of
GenerateOpticalFlowDatabase (reference cpp/opticalflow.cc:237-316) without the SQLite insert
(reported separately under "end_to_end").  Frames are resident in HBM before the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c1]

What is timed: after at least 24 untimed frames (every ring slot used, clocks up; --warmup adds to that),
EXACTLY K steps between barrier + synchronize on both sides.  A K-step region shorter than half a second
(K = 20 at 1080p is 11 ms) is measured several times over consecutive stretches of the clip and the MEDIAN
region is reported (`config.timed_regions`), so that a short driver run is not a cold one-off sample.
The headline line is the 1080p configuration C2 (BASELINE.json configs[1]); the same JSON object carries
the 4K configuration C3 under "c3" (the >= 30x target is quoted on 4K) unless --no-c3 -- with N > 1 that block is C4's
per-rank workload and says so -- and, with one GPU, "c5": C5 end to end (analysis -> tracking -> refinement, pose error
against the CPU reference of the tracking step) unless --no-c5.

Arithmetic: the library's default mode, PC_ARITH_OPENCV_X86 (what the OpenCV build the reference links executes; DESIGN.md
section 2), named in config.arith; "arith_modes" carries the same K steps in the canonical mode beside it.

roofline = the contract's HBM block of the dominant kernel (algorithmic bytes of an LK launch / its average duration,
against 8 TB/s); valu_roofline = what actually bounds that kernel (VALU issue), with both ceilings.  Counters
(SQ_INSTS_VALU, FETCH_SIZE, WRITE_SIZE) come from short `rocprofv3 --pmc` passes of tools/lk_bench.py that this script
runs itself after the timed regions ("counters_measured_in_run": true), or, without rocprofv3 / with --no-counters, from
the committed profiles/lk_hbm_traffic.json (false).

N > 1: `python bench.py --gpus N ...` launches N ranks by itself (it re-executes through torch.distributed.run on
127.0.0.1 and fails if the node has fewer than N GPUs); started under torch.distributed.run / torchrun it joins that job
instead (WORLD_SIZE must equal --gpus).  One rank per GPU; every rank analyses its own contiguous range of frame ids
(rank r: ids from 1 + r * RANK_ID_STRIDE, no data-path collective) and the flow records are all-gathered inside the timed
region, in pieces that overlap the analysis: by default each rank pushes its pieces into the peers' receive buffers over xGMI
with the copy engines (distributed.PeerLogStitch: RCCL's kernel allocates 280 registers per lane and gets a CU only by keeping
LK off it -- measured: +12 % on the step); POLYCHASE_BENCH_STITCH=rccl selects the RCCL all-gather, which is also the fallback when the ranks cannot map each
other's buffers.  Control traffic (barrier, region agreement, max-over-ranks time, piece sizes) is RCCL / gloo.
`config.stitch` names the path taken.  scaling = "weak".
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (width, height, max_level, label)
    "c1": (640, 480, 3, "C1 640x480 checkerboard-size noise clip"),
    "c2": (1920, 1080, 3, "C2 1920x1080 300-frame synthetic clip, 3-level pyramidal LK (max_level=3)"),
    "c3": (3840, 2160, 4, "C3 3840x2160 300-frame clip, 4-level LK (max_level=4) + feature detect"),
}
HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
# VALU issue ceiling of the LK kernel's instruction mix, MEASURED (tools/valu_issue.hip, profiles/r03_valu_issue.json):
# v_dot2_i32_i16, v_mad_i32_i16, v_perm_b32 (and every other integer / packed / fp64 / conversion instruction tried) issue
# at 4.2 cycles per wave64 instruction and SIMD with 8 wavefronts per SIMD (4.5 with the kernel's 3), i.e. the 16 lanes per
# clock of the classic CDNA SIMD; only v_fma/mul/add_f32, v_add_u32, v_and_b32, v_ashrrev, v_mov reach the 2.3 cycles
# of the guide's "v_fma_f32 2 cyc (SIMD-32)" row (MI355X_MICROARCH.md, per-instruction constants) -- and only back to
# back with each other: alternating with a dot2 they take a full slot too.
VALU_CYCLES_PER_INST = 4.2
VALU_PEAK_GINST = 256 * 4 * 2.4 / VALU_CYCLES_PER_INST
CLIP_FRAMES = 300
MIN_PREWARM = 24
MIN_REGION_S = 0.5
LK_KERNEL = {"canonical": "lk3_kernel<10, false>", "opencv_x86": "lk3_kernel<10, true>", "lk_x86": "lk3_kernel<10, true>",
             "sobel_fma": "lk3_kernel<10, false>"}
VALU_NOMINAL_CYCLES = 2.0   # the guide's "v_fma_f32 2 cyc (SIMD-32)" row: the ceiling no instruction of this kernel's mix reaches
VALU_NOMINAL_GINST = 256 * 4 * 2.4 / VALU_NOMINAL_CYCLES
C4_FRAMES = 2400            # BASELINE.json configs[3]: 3840x2160, 2400 frames over 8 GPUs
RANK_ID_STRIDE = 1 << 20   # rank r owns frame ids 1 + r * stride ...: disjoint, ordered shards like analyze.py's


# ---- N > 1 safety net ------------------------------------------------------------------------------------------------
# Nothing of the N > 1 path has ever run between two GPUs (no multi-GPU box was leased to this project).  So that the
# first such run cannot end without a line: (i) every rank first times the K steps on the N = 1 code path (no device
# log, no stitch, no data-path collective) and rank 0 keeps a minimal line built from it; (ii) a watchdog thread on
# every rank ends the job when nothing has progressed for POLYCHASE_BENCH_WATCHDOG_S seconds (a hung peer mapping, a
# collective that never returns) or when the launcher asks the rank to terminate because another rank died: rank 0
# prints the best line it has, marked "incomplete", and every rank leaves with exit code 0 if it has one, 3 otherwise.
_WD = {"last": None, "stage": "start", "fallback": None, "rank": 0, "done": False}
_WD_LOCK = __import__("threading").Lock()   # the watchdog thread and the SIGTERM handler may both want to end the rank


def wd_tick(stage=None):
    _WD["last"] = time.monotonic()
    if stage is not None:
        _WD["stage"] = stage


def wd_bail(why):
    """ends this rank now: rank 0 prints the best line it has"""
    if _WD["done"] or not _WD_LOCK.acquire(blocking=False):
        return
    _WD["done"] = True
    if _WD.get("printed"):   # the full line is out: only the shutdown of the process group is stuck
        os._exit(0)
    line = _WD["fallback"]
    print(f"[bench] rank {_WD['rank']}: {why} (stage: {_WD['stage']}); leaving with "
          f"{'the line measured so far' if line is not None else 'no line'}", file=sys.stderr, flush=True)
    if _WD["rank"] == 0 and line is not None:
        line = dict(line)
        line["incomplete"] = f"{why} (stage: {_WD['stage']}): the fields of the full line that were still to be measured are missing"
        print(json.dumps(line), flush=True)
    os._exit(0 if line is not None or _WD["rank"] != 0 else 3)


def wd_start(rank, limit_s):
    """one thread per rank: no progress for limit_s seconds, or SIGTERM from the launcher -> wd_bail"""
    import signal
    import threading

    _WD["rank"] = rank
    wd_tick("start")
    rfd, wfd = os.pipe()
    os.set_blocking(wfd, False)
    os.set_blocking(rfd, False)
    try:
        # the C-level handler writes the signal number into the pipe at once, whatever the main thread is blocked in
        signal.set_wakeup_fd(wfd, warn_on_full_buffer=False)
        signal.signal(signal.SIGTERM, lambda *_: wd_bail("terminated by the launcher (another rank ended)"))
    except (ValueError, OSError):
        pass

    def run():
        import select

        while not _WD["done"]:
            r, _, _ = select.select([rfd], [], [], 2.0)
            if r:
                try:
                    got = os.read(rfd, 64)
                except OSError:
                    got = b""
                if bytes([signal.SIGTERM]) in got:
                    wd_bail("terminated by the launcher (another rank ended)")
            if time.monotonic() - _WD["last"] > limit_s:
                wd_bail(f"no progress for {limit_s:.0f} s")

    threading.Thread(target=run, name="bench-watchdog", daemon=True).start()


def host_block(dev_index=0):
    """what a reader needs to compare this line with one from another box -- the CPU, its sockets / NUMA nodes, the
    node the GPU hangs off, where the library put its own threads (csrc/host/numa_pin.h), and the file system the databases of the
    end-to-end blocks were written to."""
    import platform
    import tempfile
    out = {"python_process_affinity_cpus": len(os.sched_getaffinity(0)), "cpu_count": os.cpu_count(), "kernel": platform.release()}
    try:
        model, sockets = None, set()
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model is None:
                model = line.split(":", 1)[1].strip()
            if line.startswith("physical id"):
                sockets.add(line.split(":", 1)[1].strip())
        out["cpu_model"] = model
        out["sockets"] = len(sockets) or None
    except OSError:
        pass
    try:
        nodes = sorted(d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit())
        out["numa_nodes"] = {d: open(f"/sys/devices/system/node/{d}/cpulist").read().strip() for d in nodes}
    except OSError:
        out["numa_nodes"] = None
    try:
        import json as _json
        sys.path.insert(0, os.path.join(ROOT, "polychase_amd", "core"))
        import polychase_core
        out["thread_placement"] = _json.loads(polychase_core._thread_placement())
    except Exception as e:   # the module is optional for the kernel benchmark
        out["thread_placement"] = {"error": str(e)}
    try:
        tmp = os.path.realpath(tempfile.gettempdir())
        best = ("", "?", "?")
        for line in open("/proc/mounts"):
            dev, mnt, fs = line.split()[:3]
            if (tmp == mnt or tmp.startswith(mnt.rstrip("/") + "/")) and len(mnt) >= len(best[0]):
                best = (mnt, fs, dev)
        out["temp_dir"] = {"path": tmp, "mount": best[0], "file_system": best[1], "device": best[2]}
    except OSError:
        pass
    try:
        out["transparent_hugepage"] = open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip()
    except OSError:
        pass
    return out


def level_pixels(w, h, max_level, win=10):
    s, lw, lh = 0, w, h
    for _ in range(max_level + 1):
        s += lw * lh
        lw, lh = (lw + 1) // 2, (lh + 1) // 2
        if lw <= win or lh <= win:
            break
    return s


def cpu_baseline(fetch_host, f1_candidates, gopt_kw, fopt_kw, target_seconds=12.0):
    """Times the oracle's reference-shaped CPU path (per-pair gray+pyramid rebuild,
    opticalflow.cc:298-302) on this host: a short probe, then a sample sized for ~target_seconds."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle  # test infrastructure: used here ONLY as the reported CPU baseline

    so = os.path.join(ROOT, "oracle", "libpc_oracle_native.so")
    try:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-B", "libpc_oracle_native.so"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:
        so = None
    cores = os.cpu_count() or 1
    pair_threads = min(8, cores)
    feat_threads = max(1, cores // pair_threads)

    def run(f1s):
        lo, hi = f1s[0] - 8, f1s[-1] + 8
        host = [fetch_host(f) for f in range(lo, hi + 1)]
        # timed in the canonical arithmetic: the emulation of the x86 order keeps a second set of accumulators beside the
        # integer ones and would make the port ~25 % slower than it has to be (a real OpenCV pays nothing for its own order)
        native = oracle.lib(so) if so else oracle.lib()
        before = native.pco_get_opencv_emulation()
        native.pco_set_opencv_emulation(oracle.EMU_CANONICAL)
        try:
            t0 = time.perf_counter()
            oracle.analyze_clip(host, first_frame=lo, f1_range=(f1s[0], f1s[-1] + 1), gopt=oracle.gftt_options(**gopt_kw),
                                fopt=oracle.flow_options(**fopt_kw), threads=pair_threads, feature_threads=feat_threads,
                                libpath=so)
            return time.perf_counter() - t0
        finally:
            native.pco_set_opencv_emulation(before)

    probe = run(f1_candidates[:2])
    n = int(max(2, min(len(f1_candidates), round(target_seconds / max(probe / 2, 1e-3)))))
    dt = run(f1_candidates[:n])
    out = {"value": n / dt, "unit": "frames/s", "cores": pair_threads * feat_threads, "kind": "port",
           "sample": f"{n} interior frame1 x 8 pairs of the same clip, oracle/pc_oracle.c (-O3 -march=native), "
                     f"{pair_threads} pair-threads x {feat_threads} feature-threads (all {cores} host cores), "
                     f"per-pair gray+pyramid rebuild as in opticalflow.cc:298-302, canonical arithmetic; {dt:.1f} s"}
    # the reference's own threading: TBB capped at 4 threads over the pairs (opticalflow.cc:271), features serial
    pair_threads, feat_threads = min(4, cores), 1
    dt4 = run(f1_candidates[:1])
    out["reference_threading"] = {"value": 1 / dt4, "unit": "frames/s", "cores": pair_threads,
                                  "sample": f"1 interior frame1 x 8 pairs, 4 pair-threads x 1 feature-thread; {dt4:.1f} s"}
    return out


def end_to_end(cfg, frames_dev, n_frames):
    """GenerateOpticalFlowDatabase through the polychase_core module (what the Blender addon calls) on n_frames frames of
    the clip: frames already on the GPU / as host numpy arrays (PCIe upload inside the call), without and with the SQLite
    insert.  Not part of `value`.  The engine (context + analyzer: 20 resident frame slots) is created by the first call
    of the process and parked for the next one (analysis_driver.cc: EngineCache): its creation is reported separately."""
    sys.path.insert(0, os.path.join(ROOT, "polychase_amd", "core"))
    import polychase_core as core

    w, h, ml, _ = CONFIGS[cfg]
    dev = [frames_dev[i] for i in range(n_frames)]
    host = [f.cpu().numpy() for f in dev]
    fo = core.OpticalFlowOptions()
    fo.max_level = ml
    vi = core.VideoInfo(w, h, 1, n_frames)
    core.release_cached_engine()
    t0 = time.perf_counter()
    st = core.generate_optical_flow_database(core.VideoInfo(w, h, 1, 12), lambda f: dev[f - 1], None, "", core.GFTTOptions(), fo)
    out = {"frames": n_frames, "engine_creation_ms": 1e3 * st.seconds_setup, "first_call_12_frames_ms": 1e3 * (time.perf_counter() - t0),
           "note": "whole calls (clip edges, first-use allocations of the call included) with the process's engine already created; "
                   "engine creation + the cold 12-frame call are the two numbers above"}
    with tempfile.TemporaryDirectory() as td:
        for name, frames, db in [("device_frames_no_db_fps", dev, ""), ("host_frames_over_pcie_no_db_fps", host, ""),
                                 ("host_frames_over_pcie_sqlite_fps", host, os.path.join(td, "a.db"))]:
            t0 = time.perf_counter()
            st = core.generate_optical_flow_database(vi, lambda f: frames[f - 1], None, db, core.GFTTOptions(), fo)
            out[name] = n_frames / (time.perf_counter() - t0)
            if db:
                out["sqlite_bytes_per_frame"] = os.path.getsize(db) / n_frames
                out["sqlite_insert_ms_per_frame"] = 1e3 * st.seconds_db / n_frames
            out[name.replace("_fps", "_driver_ms_per_frame")] = {
                k: round(1e3 * getattr(st, "seconds_" + k) / n_frames, 4) for k in ("accessor", "put", "submit", "collect", "writer_wait")}
    core.release_cached_engine()
    del host
    return out


ARITH_FLAGS = {"canonical": 0, "lk_x86": 1, "sobel_fma": 2, "opencv_x86": 3}
ARITH_NAMES = {v: k for k, v in ARITH_FLAGS.items()}
LAST_ARITH = None   # the arithmetic mode the latest run_config ran in (every rank)


_COUNTERS_BROKEN = False   # a pass hung or failed on this box: no further attempts in this run (worst case: one timeout)


def measure_counters(cfg, arith, timeout_s=45):
    """SQ_INSTS_VALU, FETCH_SIZE and WRITE_SIZE of ONE LK launch of this configuration, measured now on this box: three short
    `rocprofv3 --pmc` passes (counters in their own runs, no other trace domain) of tools/lk_bench.py -- the same kernel on a
    frame of the same clip.  Returns None when rocprofv3 is missing or a pass fails (the caller then falls back to the
    committed profile and says so)."""
    import csv
    import glob
    import shutil
    import signal

    global _COUNTERS_BROKEN
    if _COUNTERS_BROKEN or not shutil.which("rocprofv3"):
        return None
    env = dict(os.environ, TMPDIR="/tmp", POLYCHASE_ARITH=arith)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    got = {}
    for counter in ("SQ_INSTS_VALU", "FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="pcpmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable,
               os.path.join(ROOT, "tools", "lk_bench.py"), "--config", cfg, "--reps", "3", "--arith", arith]
        proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
        try:
            proc.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)
            proc.communicate()
            shutil.rmtree(d, ignore_errors=True)
            _COUNTERS_BROKEN = True
            return None
        total, n = 0.0, 0
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if "lk3_kernel" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                    total += float(row["Counter_Value"])
                    n += 1
        shutil.rmtree(d, ignore_errors=True)
        if n == 0:
            _COUNTERS_BROKEN = True
            return None
        got[counter] = total / n
        got["dispatches_" + counter] = n
    return got


def c5_block():
    """Config C5 (1920x1080, 300 rendered frames): generate_optical_flow_database -> track_sequence -> refine_trajectory
    through polychase_core, pose error of every 29th frame against the CPU reference of the tracking step
    (oracle/pnp_oracle.py: the checker leg, like cpu_baseline).  reference cpp/tracker.cc:133-192, cpp/refiner.cc:716-725.

    Since round 5 the block also says HOW FAST that half is and why: launches per tracked frame, the persistent LM kernel's own
    phase clock, a `roofline` for it (bytes of one residual sweep x sweeps per frame / its duration vs 8 TB/s -- tiny by
    construction: the kernel is a chain of grid barriers, the phases say where the time goes), the refinement's split
    (SQLite read / upload / GPU sweeps / banded Cholesky) and CPU baselines of both beside them (the float64 numpy
    restatements, `kind: "port (numpy float64)"`, cores stated)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import c5_endtoend

    t0 = time.perf_counter()
    r = c5_endtoend.run(width=1920, height=1080, frames=300, oracle_frames=0, oracle_stride=29, refine_iterations=30, oracle_workers=10,
                        refine_oracle_frames=6)
    tr, rf = r["tracking"], r["refinement"]
    s = tr.get("vs_cpu_reference_sampled", {})
    st = tr.get("stages", {})
    frames = max(1, st.get("track/wait for the GPU", [0, 299])[1])
    val = lambda k: st.get(k, [0.0, 0])[0]
    matches, corr, rounds = val("track/matches (count, not ms)"), val("track/correspondences (count, not ms)"), val("track/lm kernel: rounds (count, not ms)")
    lm_ms = val("track/lm kernel: whole launch")
    # one residual sweep reads every match's world point + validity (16 B) and its observation (8 B)
    sweep_bytes = 24.0 * matches / frames
    achieved = sweep_bytes * (rounds / frames) / (lm_ms / frames * 1e-3) / 1e9 if lm_ms > 0 else None
    tracking = {
        "frames_per_s": tr["frames_per_s"], "mean_lm_iterations": tr["mean_lm_iterations"], "keypoints_per_frame": tr["keypoints_per_frame"],
        "matches_per_frame": matches / frames, "correspondences_per_frame": corr / frames, "lm_sweeps_per_frame": rounds / frames,
        "launches_per_frame": 2, "transfers_per_frame": "1 (+ 1 the first time a frame is a source: its keypoints)", "host_waits_per_frame": 1,
        "launches_per_frame_round4": 41.2,
        "gpu_ms_per_frame": {"lm_kernel": lm_ms / frames,
                             "lm_kernel_phases": {k.split(": ", 1)[1]: val(k) / frames for k in st if k.startswith("track/lm kernel: ") and "count" not in k}},
        "host_ms_per_frame": {k[6:]: val(k) / frames for k in st if k.startswith("track/") and "lm kernel" not in k and "count" not in k},
        "roofline": {"bound": "hbm", "kernel": "track_lm_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS if achieved else None,
                     "algorithmic_bytes_per_sweep": sweep_bytes, "sweeps_per_launch": rounds / frames, "avg_launch_ms": lm_ms / frames,
                     "time_base": "the kernel's own 100 MHz clock, whole launch (workgroup 0), summed over the frames",
                     "note": "a latency chain, not a stream: one launch = ~12 rounds of [sweep, publish, grid barrier, 9x9 decision, publish]; "
                             "the matches stay in the L2 between rounds"},
        "cpu_baseline": {"value": (len(s.get("frames", [])) / s["cpu_wall_seconds"]) if s.get("cpu_wall_seconds") else None, "unit": "frames/s",
                         "cores": s.get("cpu_processes"), "kind": "port (numpy float64)",
                         "sample": f"{len(s.get('frames', []))} frames of the clip (every 29th), oracle/pnp_oracle.py: ray casting + PnP LM of tracker.cc:36-131"},
    }
    rs = rf.get("stages", {})
    rv = lambda k: rs.get(k, [0.0, 0])[0] / 1e3
    cpu = rf.get("cpu_reference", {})
    n_res = rs.get("refine/residuals (count, not ms)", [0, 0])[0]

    def sweep_roofline(stage):
        ms, calls = rs.get(stage, [0.0, 0])
        if not calls or not ms:
            return None
        avg = ms / calls
        achieved = 24.0 * n_res / (avg * 1e-3) / 1e9
        return {"launches": calls, "avg_launch_ms": avg, "residuals_per_s": n_res / (avg * 1e-3), "bound": "hbm", "achieved": achieved,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS}

    refinement = {
        "seconds": rf["seconds"], "iterations": rf["iterations"], "cost_before_after": rf["cost"],
        "seconds_split": {"sqlite_read": rv("refine/load segment (SQLite)"), "upload": rv("refine/upload"), "release_host_copy": rv("refine/release host copy"),
                          "gpu_cost_sweeps": rv("refine/cost sweep"), "gpu_normal_equation_sweeps_and_host_assembly": rv("refine/normal equations (sweep + host assembly)"),
                          "banded_cholesky": rv("refine/banded Cholesky")},
        "sweeps": {"cost": rs.get("refine/cost sweep", [0, 0])[1], "normal_equations": rs.get("refine/normal equations (sweep + host assembly)", [0, 0])[1]},
        "residuals": n_res,
        "gpu_kernels": {
            "time_base": "HIP events on the context's stream around every sweep's kernel (pc_refine_problem_timing)",
            "refine_cost_kernel": sweep_roofline("refine/cost sweep: kernel (GPU clock)"),
            "refine_normal_eq_kernel": sweep_roofline("refine/normal equations: kernel (GPU clock)"),
            "algorithmic_bytes_per_residual": 24, "what": "4 B source keypoint index + 8 B tracked position (streamed), 8 B keypoint + 4 B cached "
            "triangle (gathered through the index); the triangle itself is shared by its residuals",
            "note": "both sweeps are bound by instruction issue, not by HBM: per residual a ray / cached-triangle test, two camera transforms "
                    "and a robust loss with ~10 IEEE divisions (cost), plus a 2 x 12 Jacobian and 204 fp64 multiply-adds (normal equations)"},
        "cpu_baseline": {"kind": "port (numpy float64)", "cores": cpu.get("processes"), "unit": "residuals/s",
                         "cost_sweep": cpu.get("residuals", 0) / cpu["cost_sweep_seconds"] if cpu.get("cost_sweep_seconds") else None,
                         "normal_equations": cpu.get("residuals", 0) / cpu["normal_equations_seconds"] if cpu.get("normal_equations_seconds") else None,
                         "sample": f"{cpu.get('frames')} frames in the middle of the clip, {cpu.get('edges')} flows, {cpu.get('residuals')} residuals (oracle/refine_oracle.py)"},
    }
    return {"workload": "C5 1920x1080 300 rendered frames end to end: GFTT + LK -> SQLite -> ray casting + PnP (tracker.cc) -> refiner.cc on the GPU",
            "analysis_with_sqlite_fps": r["analysis"]["fps"], "tracking_fps": tr["frames_per_s"],
            # the same call a second time in the process (the first one pays 6-8 ms of page-locked blocks, streams and device arrays)
            "tracking_fps_second_call": tr.get("second_call", {}).get("frames_per_s"),
            "tracking_mean_lm_iterations": tr["mean_lm_iterations"], "keypoints_per_frame": tr["keypoints_per_frame"],
            "refinement_seconds": rf["seconds"], "refinement_iterations": rf["iterations"],
            "refinement_cost_before_after": rf["cost"],
            "tracking": tracking, "refinement": refinement,
            "pose_error_vs_cpu_reference": {"sampled_frames": len(s.get("frames", [])), "rotation_rad_max": s.get("rotation_rad_max"),
                                            "translation_rel_max": s.get("translation_rel_max"),
                                            "tolerance": "1e-4 rad, 1e-4 |t| (SURVEY 8(d))",
                                            "what": "float64 numpy restatement of tracker.cc (oracle/pnp_oracle.py) on the same database, "
                                                    "each sampled frame from the GPU's poses of its source frames"},
            "pose_error_vs_truth": {"tracking": tr["vs_truth"], "refined": rf["vs_truth"]},
            "seconds_total": round(time.perf_counter() - t0, 2)}


def hard_content_block():
    """What the default (parity) arithmetic costs on content that is NOT the benchmark's: the isolated LK launch
    (one frame1 of 1920x1080 into 8 targets), x86 summation order vs canonical, on blurred step edges over a fine texture and on
    BASELINE's own C1 pattern (the checkerboard) at 1080p, beside the benchmark's texture -- ratio, share of the iterations and
    of the structure tensors that had to be evaluated in the x86 order (pc_debug_lk_x86_stats).  Outside every timed region.
    The headline clip is the KINDEST case; a quote of the headline number should carry these ratios (reference
    cpp/opticalflow.cc:119-125)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import x86_cost_probe
    rows = x86_cost_probe.measure(("c2 texture", "edges + texture", "checkerboard", "binary blocks"), reps=20)
    out = {"what": "isolated LK launch, 1920x1080, 8 targets, lk_x86 vs canonical arithmetic; ms per launch",
           "model": "ratio ~ 1 + 0.06 (the exactness proof, every iteration) + 0.03 x (share of ordered structure tensors) + 0.3 x (share of "
                    "ordered iterations): a rough fit to these rows (DESIGN.md section 4; before the ordered iteration was rewritten in "
                    "round 5: 0.09 / 0.085 / 0.5)"}
    for r in rows:
        out[r["content"]] = {"keypoints": r["keypoints"], "canonical_ms": r["canonical_ms"], "x86_ms": r["lk_x86_ms"],
                             "x86_over_canonical": r["x86_over_canonical"], "iterations_in_x86_order": r["iterations_in_x86_order"],
                             "levels_with_ordered_structure_tensor": r["levels_with_ordered_structure_tensor"], "sha256_of_tracked_rows": r["sha256"]}
    return out


def run_config(cfg, K, W, args, rank, world, dev, with_cpu, with_e2e, arith=None, light=False):
    """One configuration: returns the result object of this rank (rank 0's is printed).  light: the timed regions of the
    headline path only (the other arithmetic mode beside the headline)."""
    import torch
    import torch.distributed as dist

    from polychase_amd import distributed as D
    from polychase_amd import hip, synth
    from polychase_amd.pipeline import ClipAnalyzer

    w, h, max_level, label = CONFIGS[cfg]
    wd_tick(f"{cfg}: set-up")
    clip = synth.NoiseClip(w, h, CLIP_FRAMES, device=str(dev))
    clip_frames = [clip.frame_torch(t) for t in range(CLIP_FRAMES)]
    torch.cuda.synchronize()

    def source(fid):
        # the clip played forwards and backwards over and over: a continuous motion for any number of frame ids
        # (ranks start at different phases of it: fid carries the rank's id offset)
        t = (fid % RANK_ID_STRIDE + (fid // RANK_ID_STRIDE) * 37) % (2 * CLIP_FRAMES - 2)
        return clip_frames[t if t < CLIP_FRAMES else 2 * CLIP_FRAMES - 2 - t]

    ctx = hip.Context(dev.index or 0)
    if arith is not None:
        ctx.set_arithmetic(ARITH_FLAGS[arith])
    global LAST_ARITH
    arith_name = LAST_ARITH = ARITH_NAMES[ctx.arithmetic]
    gopt_kw, fopt_kw = {}, {"max_level": max_level}
    prewarm = max(MIN_PREWARM, W)
    dist_path = world > 1 or args.force_dist_path

    # ---- how many K-step regions: a short probe after the pre-warm gives the step time ----
    first_id = 1 + rank * RANK_ID_STRIDE
    an = ClipAnalyzer(ctx, w, h, first_id, 1 << 30, source, hip.gftt_options(**gopt_kw), hip.flow_options(**fopt_kw), max_jobs=3)

    def barrier():
        torch.cuda.synchronize()
        ctx.synchronize()
        if dist.is_initialized():
            dist.barrier()

    n_kps, n_rows = [], []

    def sink(frame1, kps, detected, flows):
        n_kps.append(len(kps))
        n_rows.append(sum(len(v[0]) for v in flows.values()))

    # the Python driver loop must not stall the GPU pipeline: a generation-2 collection pauses this process for
    # 50-90 ms (the C++ driver of polychase_core has no such pauses)
    gc.collect()
    gc.disable()
    nxt = first_id + 8
    an.run(range(nxt, nxt + prewarm), sink)
    nxt += prewarm
    barrier()
    t0 = time.perf_counter()
    an.run(range(nxt, nxt + 8), sink, copy=False)
    barrier()
    est_step = (time.perf_counter() - t0) / 8
    nxt += 8
    regions = int(min(40, max(1, -(-MIN_REGION_S // max(K * est_step, 1e-6)))))
    if dist.is_initialized():
        tt = torch.tensor([regions], dtype=torch.int64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        regions = int(tt.item())

    # ---- the ways the K steps are run: the N = 1 path (records to the host), and with N > 1 the device log + a stitch ----
    log = None
    stitches = {}      # mode -> stitch object
    unavailable = {}   # mode -> why
    modes = ["n1"]

    def make_stitches():
        nonlocal log, modes
        # device-resident record log: the stitch all-gathers these bytes, no host copy of the payload
        max_kp = int(1.5 * max(n_kps + [kp_hint])) + 4096
        log = torch.empty(D.log_capacity_bytes(K + 2, max_kp), dtype=torch.uint8, device=dev)
        side = None
        if dist.is_initialized():
            try:
                side = dist.new_group(backend="gloo")
            except Exception as e:   # no usable interface for gloo: the sizes go over the main group instead
                print(f"[bench] gloo side group unavailable ({e}); exchanging piece sizes over RCCL", file=sys.stderr)
        # POLYCHASE_BENCH_STITCH=rccl: the RCCL all-gather; default: peer copies over xGMI by the copy engines, falling back
        # to the all-gather when the ranks cannot map each other's buffers (distributed.PeerLogStitch says why)
        prefer = os.environ.get("POLYCHASE_BENCH_STITCH", "peer")
        first = D.make_log_stitch(log, side_group=side, prefer=prefer)
        piece_frames = max(1, K // 8)

        def prepare(st):
            st.warm_up()
            st.reserve(K // piece_frames + piece_frames + 1, D.log_capacity_bytes(piece_frames + 1, max_kp))
            return st

        first_mode = "peer" if isinstance(first, D.PeerLogStitch) else "rccl"
        stitches[first_mode] = prepare(first)
        modes = [first_mode]
        if not light:
            # A/B inside ONE job: the other stitch over the same log, and the N = 1 path on all ranks at once
            if first_mode == "peer":
                stitches["rccl"] = prepare(D.ChunkedLogStitch(log, side_group=side))
                modes.append("rccl")
            elif prefer == "peer":
                unavailable["peer"] = ("the ranks could not map each other's buffers (HIP IPC); the job fell back to the RCCL all-gather"
                                       if dist.is_initialized() else "one rank without a process group: nothing to push")
            modes.append("n1")

    def run_regions(mode):
        """`regions` timed K-step regions in one mode -> per-region statistics"""
        nonlocal nxt
        stitch = stitches.get(mode)
        res = {"region_s": [], "lk_avg": [], "lk_busy": [], "lk_launches": 0, "rank_dt": [], "finish_ms": [], "log_bytes": []}
        for region in range(regions):
            wd_tick(f"{cfg} {arith_name}: mode {mode}, region {region + 1} of {regions}")
            timed = range(nxt, nxt + K)
            nxt += K
            if stitch is not None:
                an.an.set_device_log(log)   # resets the log (synchronises: outside the timed region)
                stitch.reset()              # ... and the pieces gathered in the previous region
            barrier()
            ctx.enable_timing(["lk"])   # HIP events around the dominant kernel only (2 records per step)
            ctx.reset_timing()
            finish_ms = 0.0
            t0 = time.perf_counter()
            if stitch is None:
                an.run(timed, sink, copy=False)
            else:
                # the same loop as ClipAnalyzer.run, plus: whenever the last frame1 of a piece has been collected (its log
                # bytes are complete), all-gather that piece -- the transfer runs beside the LK launches of the
                # following frames; only the last piece is exposed (the one collective of the path)
                piece = max(1, K // 8)
                log_end = {}
                piece_start = 0

                def collect_one():
                    nonlocal piece_start
                    r = an.an.collect(False)
                    sink(*r)
                    done = r[0] - timed.start + 1
                    # the last frames of the region go one by one: what is still on the wire when the last job has been
                    # collected is one frame, not a piece
                    if done % piece == 0 or done > K - piece:
                        stitch.gather(piece_start, log_end[r[0]])
                        piece_start = log_end[r[0]]

                for f in timed:
                    if an.an.pending == an.max_jobs:
                        collect_one()
                    an.submit(f)
                    log_end[f] = an.an.device_log_used
                while an.an.pending:
                    collect_one()
                tf = time.perf_counter()
                stitch.finish()             # peer copies: this rank's pushes landed, sizes exchanged, the ranks have met
                torch.cuda.synchronize()    # the all-gather's payload is stream-ordered: this wait ends it
                finish_ms = (time.perf_counter() - tf) * 1e3
                res["log_bytes"].append(int(an.an.device_log_used))
            barrier()
            dt = time.perf_counter() - t0
            mine = [dt, finish_ms]
            if dist.is_initialized():
                allr = [torch.zeros(2, dtype=torch.float64, device=dev) for _ in range(world)]
                dist.all_gather(allr, torch.tensor(mine, dtype=torch.float64, device=dev))
                allr = [[float(x) for x in t.tolist()] for t in allr]
            else:
                allr = [mine]
            dt = max(a[0] for a in allr)          # the MAX over ranks is the region's time
            res["region_s"].append(dt)
            res["rank_dt"].append([a[0] for a in allr])
            res["finish_ms"].append(max(a[1] for a in allr))
            timing = ctx.timing()
            lk_n, lk_ms = timing["lk"]
            res["lk_launches"] += lk_n
            res["lk_avg"].append(lk_ms / max(1, lk_n))
            res["lk_busy"].append(ctx.busy_ms("lk") / max(1, lk_n))
            ctx.enable_timing(False)
            if stitch is not None:
                an.an.set_device_log(None)
            if stitch is not None and (region == 0 or region == regions - 1):
                # outside the timed region: every rank's shard must parse and hold exactly K records in frame order.  (First and
                # last region of a mode only: the check downloads every rank's records -- 3.5 GB per 4K region at 8 ranks.)
                for r, (buf, used) in enumerate(stitch.rank_logs()):
                    recs = D.parse_device_log(buf, used)
                    assert len(recs) == K and [x[0] for x in recs] == list(range(recs[0][0], recs[0][0] + K)), "stitched log is not K consecutive frames"
                    # the shards are disjoint and in rank order: rank r's ids are this rank's ids shifted by (r - rank) strides
                    assert recs[0][0] == timed.start + (r - rank) * RANK_ID_STRIDE, "stitched shards are not the ranks' disjoint id ranges"
                    if r == rank:
                        assert [x[0] for x in recs] == list(timed)
                        assert [len(x[1]) for x in recs] == n_kps[-K:] and [sum(len(v[0]) for v in x[2].values()) for x in recs] == n_rows[-K:]
        order = np.argsort(res["region_s"])
        res["mid"] = int(order[len(order) // 2])      # the median region
        return res

    kp_hint = max(n_kps)
    n_kps.clear()
    n_rows.clear()
    results = {}
    if dist_path:
        if not light:
            # first the N = 1 code path on every rank (no log, no stitch): the line the watchdog falls back to
            wd_tick(f"{cfg} {arith_name}: the N = 1 code path on all ranks")
            results["n1"] = run_regions("n1")
            if rank == 0 and _WD["fallback"] is None:
                r1 = results["n1"]
                dt1 = r1["region_s"][r1["mid"]]
                _WD["fallback"] = {
                    "metric": "optical-flow frames/sec", "value": world * K / dt1, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
                    "ms_per_step": dt1 / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                    "dtype": "u8/int32 fixed-point + f32 2x2 solve", "data": "synthetic",
                    "config": {"workload": label, "width": w, "height": h, "max_level": max_level, "window": 10, "pairs_per_frame": 8,
                               "frames_per_gpu": K, "arith": arith_name, "parallelism": f"frame-shard x{world}" if world > 1 else "single GPU",
                               "stitch": "none: the analysis-only rate (every rank's records delivered to its host, no all-gather)",
                               "timed_regions": len(r1["region_s"])}}
        wd_tick(f"{cfg} {arith_name}: creating the device log and the stitches")
        if os.environ.get("POLYCHASE_BENCH_TEST_HANG") == "stitch":   # tests/test_bench_gpu.py: the watchdog's own test
            time.sleep(1e6)
        make_stitches()
    for m in modes:
        if m not in results:
            results[m] = run_regions(m)
    wd_tick(f"{cfg} {arith_name}: after the timed regions")
    gc.enable()
    stitch_names = {m: st.name for m, st in stitches.items()}
    for st in stitches.values():
        st.close()
    stitches.clear()
    log = None
    an.close()
    head = results[modes[0]]
    mid = head["mid"]
    dt = head["region_s"][mid]
    fps = world * K / dt

    def mode_summary(m):
        r = results[m]
        i = r["mid"]
        d = {"value": world * K / r["region_s"][i], "unit": "frames/s", "ms_per_step": r["region_s"][i] / K * 1e3,
             "per_rank_ms_per_step_min_max": [min(r["rank_dt"][i]) / K * 1e3, max(r["rank_dt"][i]) / K * 1e3],
             "timed_regions": len(r["region_s"]),
             "region_ms_min_median_max": [min(r["region_s"]) * 1e3, r["region_s"][i] * 1e3, max(r["region_s"]) * 1e3]}
        if m != "n1":
            b = r["log_bytes"][i]
            d.update({"stitch": stitch_names[m],
                      # after the last job of the region has been collected: waiting for the pieces still on the wire
                      "exposed_stitch_ms_per_region": r["finish_ms"][i],
                      "record_bytes_per_rank_per_region": b,
                      # xGMI is point to point: each of the world - 1 peers receives one copy of this rank's records
                      "per_peer_link_GBs": b / r["region_s"][i] / 1e9,
                      "received_GBs_per_gpu": (world - 1) * b / r["region_s"][i] / 1e9})
        else:
            d["what"] = ("the N = 1 code path (records delivered to the host, no device log, no stitch) on all ranks at once: "
                         "the analysis-only rate, comparable with the one-GPU benchmark line")
        return d

    # per-class kernel breakdown from a short extra pass over the same frames (not part of `value`)
    breakdown = None
    if not args.no_breakdown and not light:
        n_extra = min(K, 20)
        an = ClipAnalyzer(ctx, w, h, first_id, 1 << 30, source, hip.gftt_options(**gopt_kw), hip.flow_options(**fopt_kw), max_jobs=3)
        an.run(range(first_id + 8, first_id + 8 + 20), None)
        ctx.synchronize()
        ctx.enable_timing(True)
        ctx.reset_timing()
        an.run(range(first_id + 28, first_id + 28 + n_extra), None)
        breakdown = {k: v[1] / n_extra for k, v in ctx.timing().items()}
        ctx.enable_timing(False)
        an.close()

    # The dominant kernel's OWN duration (the roofline's time base): a short extra pass over the same frames
    # with every job on one lane (POLYCHASE_LK_LANES=1, read when the analyzer is created) -- no two LK launches overlap, so
    # the HIP events around a launch on its lane's stream bracket that launch alone.  This is the recipe of the committed
    # profiles/r0N_c{2,3}_rocprofv3_kernel_stats.csv (tools/collect_profiles.sh), whose average must agree.  Not part of `value`.
    lk_own_ms = None
    if not light and rank == 0:
        prev = os.environ.get("POLYCHASE_LK_LANES")
        os.environ["POLYCHASE_LK_LANES"] = "1"
        try:
            n_own = 40
            an = ClipAnalyzer(ctx, w, h, first_id, 1 << 30, source, hip.gftt_options(**gopt_kw), hip.flow_options(**fopt_kw), max_jobs=3)
            an.run(range(first_id + 8, first_id + 8 + 20), None)
            ctx.synchronize()
            ctx.enable_timing(True)
            ctx.reset_timing()
            an.run(range(first_id + 28, first_id + 28 + n_own), None)
            t = ctx.timing().get("lk")
            if t and t[0] > 0:
                lk_own_ms = t[1] / t[0]
            ctx.enable_timing(False)
            an.close()
        finally:
            if prev is None:
                os.environ.pop("POLYCHASE_LK_LANES", None)
            else:
                os.environ["POLYCHASE_LK_LANES"] = prev

    out = None
    if rank == 0:
        P = w * h
        S = level_pixels(w, h, max_level)
        lk_avg_ms, lk_busy_ms = head["lk_avg"][mid], head["lk_busy"][mid]
        lk_bytes = (5 + 8) * S  # LK I-side 5S (image S + derivs 4S) + J-side S per target, K_f = 8
        frame_bytes = 14 * P + (12 + 8) * S
        base_ms = lk_own_ms if lk_own_ms else lk_busy_ms
        achieved = lk_bytes / (base_ms * 1e-3) / 1e9 if base_ms > 0 else 0.0
        kernel = LK_KERNEL.get(arith_name, "lk3_kernel<10>")
        # counters of the LK launch: measured now by short rocprofv3 --pmc passes (one GPU, not under a profiler already),
        # else from the committed profile of a builder-run pass, labelled as such
        counters = None
        if not light and world == 1 and not args.no_counters and not args.force_dist_path:
            try:
                counters = measure_counters(cfg, arith_name)
            except Exception as e:
                print(f"[bench] counter passes failed: {e}", file=sys.stderr)
        traffic = valu = src = calib = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "lk_hbm_traffic.json")))
            calib = tj.get("fetch_size_calibration")
            if counters is None:
                blk = tj.get(arith_name, tj).get(cfg) or tj[cfg]
                traffic, valu, src = blk["traffic_bytes"], blk.get("valu_insts"), tj.get("source")
        except Exception:
            pass
        if counters is not None:
            # FETCH_SIZE tallies every 128-B line the L2 fetches as 64 B (tools/fetch_calib.hip, profiles/r03_fetch_calibration.json)
            traffic = (2.0 * counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024.0
            valu = counters["SQ_INSTS_VALU"]
            src = ("this run: rocprofv3 --pmc SQ_INSTS_VALU / FETCH_SIZE / WRITE_SIZE, one pass each, of tools/lk_bench.py "
                   f"--config {cfg} --arith {arith_name} on this box (the same kernel on frame 100 of the same clip), per launch; "
                   "traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 B")
        roofline = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "counters_measured_in_run": counters is not None,
                    "traffic_source": src or "none",
                    "traffic_calibration": calib,
                    "traffic_over_algorithmic": traffic / lk_bytes if traffic else None,
                    "algorithmic_bytes_per_launch": lk_bytes, "launches": head["lk_launches"],
                    # ONE time base: `achieved` = algorithmic bytes / the kernel's own average duration, measured in this run
                    # with every job on one lane (no overlapping launches) -- what a rocprofv3 --kernel-trace --stats CSV of the
                    # one-lane run reports as the kernel's average.  The other two clocks of the timed regions stay as fields of
                    # their own: the start-to-end time of the OVERLAPPING launches of the two-lane pipeline, and the GPU-busy
                    # time per launch there (the union of the launches' intervals / launches).
                    "kernel_avg_duration_ms": base_ms,
                    "time_base": ("the kernel's own duration: HIP events around each launch on its stream, all jobs on one lane "
                                  "(POLYCHASE_LK_LANES=1), 40 launches over the same frames" if lk_own_ms else
                                  "GPU-busy time per launch of the two-lane pipeline (the one-lane pass did not run)"),
                    "avg_launch_ms": lk_avg_ms,
                    "launch_overlap": lk_avg_ms / lk_busy_ms if lk_busy_ms > 0 else None,
                    "busy_ms_per_launch": lk_busy_ms,
                    "note": "a gather that is VALU-issue bound: see valu_roofline for the limiter"}
        valu_roofline = None
        if valu and lk_busy_ms > 0:
            v_ach = valu / (lk_busy_ms * 1e-3) / 1e9
            valu_roofline = {"bound": "valu", "kernel": kernel, "achieved": v_ach, "unit": "G wave-instructions/s",
                             "peak_measured_for_the_mix": VALU_PEAK_GINST, "frac_of_measured_peak": v_ach / VALU_PEAK_GINST,
                             "peak_nominal": VALU_NOMINAL_GINST, "frac_of_nominal_peak": v_ach / VALU_NOMINAL_GINST,
                             "valu_wave_instructions_per_launch": valu, "counters_measured_in_run": counters is not None,
                             "valu_source": src,
                             "peak_source": f"measured: tools/valu_issue.hip, profiles/r03_valu_issue.json: {VALU_CYCLES_PER_INST} cycles per wave64 "
                                            "v_dot2_i32_i16 / v_mad_i32_i16 / v_perm_b32 and SIMD at 8 wavefronts per SIMD (4.5 at the kernel's 3); "
                                            f"nominal: the guide's {VALU_NOMINAL_CYCLES:.0f}-cycle VALU rate (MI355X_MICROARCH.md); 1024 SIMDs x 2.4 GHz",
                             "time_base": "GPU-busy time of the launches (busy_ms_per_launch), measured in this run with HIP events on the lanes' streams"}
        workload = label
        cfg_extra = {}
        if world > 1 and cfg == "c3":
            fpg = C4_FRAMES // 8
            workload = (f"C4 3840x2160 2400-frame clip frame-sharded across {world} x MI355X (the per-rank workload of C4: contiguous frame1 "
                        f"ranges, {fpg} frames per GPU at 8 GPUs + an 8-frame halo per side, stitch = all-gather of the flow records), "
                        f"timed in steady-state regions of {K} frame1 per rank")
            cfg_extra = {"c4_clip_frames": C4_FRAMES, "c4_frames_per_gpu_at_8": fpg, "halo_frames_per_side": 8,
                         "halo_overhead": "16 extra frames per shard are converted and pyramided, not detected or tracked: "
                                          f"16 / {fpg} of the pyramid kernel's time per shard (kernel_ms_per_frame.pyramid), not in the steady-state step"}
        out = {
            "metric": "optical-flow frames/sec", "value": fps, "unit": "frames/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
            # the dominant kernel's GPU-busy time per step (one LK launch per step; the union of the overlapping launches' intervals
            # of the two-lane pipeline / launches): <= ms_per_step.  (roofline.kernel_avg_duration_ms -- one launch ALONE, from the
            # one-lane pass -- may exceed ms_per_step by the overlap: the launches of consecutive steps share the GPU in their tails.)
            "gpu_busy_ms_per_step": lk_busy_ms,
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8/int32 fixed-point + f32 2x2 solve",
            "data": "synthetic",
            "config": {"workload": workload, "width": w, "height": h, "max_level": max_level,
                       "window": 10, "pairs_per_frame": 8, "frames_per_gpu": K, "arith": arith_name,
                       "mean_keypoints": float(np.mean(n_kps)), "mean_flow_rows": float(np.mean(n_rows)),
                       "parallelism": f"frame-shard x{world}" if world > 1 else "single GPU",
                       "untimed_prewarm_steps": prewarm + 8, "timed_regions": len(head["region_s"]),
                       "region_ms_min_median_max": [min(head["region_s"]) * 1e3, dt * 1e3, max(head["region_s"]) * 1e3], **cfg_extra},
            "roofline": roofline,
            "valu_roofline": valu_roofline,
            "path_roofline": {"algorithmic_bytes_per_frame": frame_bytes,
                              "achieved_GBs": frame_bytes * (K / dt) / 1e9,
                              "frac": frame_bytes * (K / dt) / 1e9 / HBM_PEAK_GBS},
            "kernel_ms_per_frame": breakdown,
        }
        if dist.is_initialized():
            out["collectives_backend"] = dist.get_backend()   # "nccl" = RCCL; "gloo" only under the SHARE_GPU testing aid
            out["world_size"] = dist.get_world_size()
        if world > 1 or modes[0] != "n1":
            # every N > 1 figure is the ANALYSIS-ONLY rate -- detection + LK + the stitch of the records, no SQLite
            # insert.  The product call with the insert is bound by its single writer (one connection, one pwrite thread: DESIGN.md
            # section 5) whatever N is: product_rate_with_insert below is that rate, measured on rank 0 of this job.
            out["claim"] = ("analysis-only (no insert): frames analysed and their records stitched per second over all ranks; the "
                            "end-to-end product rate with the SQLite insert does not scale with N (one writer) -- see product_rate_with_insert")
        if modes[0] != "n1":
            out["config"]["stitch"] = stitch_names[modes[0]]
            if not light:
                ab = {m: mode_summary(m) for m in modes if m != "n1"}
                ab.update({m: {"unavailable": why} for m, why in unavailable.items()})
                out["stitch_ab"] = ab
                out["analysis_only"] = mode_summary("n1")
                out["per_rank_ms_per_step_min_max"] = ab[modes[0]]["per_rank_ms_per_step_min_max"]
        if with_cpu:
            f1s = [first_id + 8 + i for i in range(K)]
            out["cpu_baseline"] = cpu_baseline(lambda f: source(f).cpu().numpy(), f1s, gopt_kw, fopt_kw,
                                               target_seconds=args.cpu_seconds)
            out["speedup_vs_cpu_baseline"] = fps / out["cpu_baseline"]["value"]
        if with_e2e:
            try:
                out["end_to_end"] = end_to_end(cfg, clip_frames, 300 if cfg != "c3" else 100)
            except Exception as e:   # the module is optional for the kernel benchmark
                out["end_to_end"] = {"error": str(e)}
        elif world > 1 and not light and not args.no_end_to_end:
            try:   # the writer-bound rate the N > 1 figure must be read beside (rank 0 alone; the other ranks wait at the next barrier)
                e2e = end_to_end(cfg, clip_frames, 100 if cfg != "c3" else 50)
                out["product_rate_with_insert"] = {"value": e2e.get("host_frames_over_pcie_sqlite_fps"), "unit": "frames/s", "n_gpus_that_matter": 1,
                                                   "what": "generate_optical_flow_database with the SQLite insert, one GPU + one writer: the ceiling "
                                                           "of the multi-GPU product call (polychase_amd/analyze.py) whatever the number of GPUs",
                                                   "frames": e2e.get("frames"), "sqlite_insert_ms_per_frame": e2e.get("sqlite_insert_ms_per_frame")}
            except Exception as e:
                out["product_rate_with_insert"] = {"error": str(e)}
    ctx.close()
    del clip_frames, clip
    torch.cuda.empty_cache()
    return out


def free_port() -> int:
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: run N ranks of this command through torch.distributed.run (one
    process per GPU, rendezvous on 127.0.0.1) and return its exit code.  The sharded loop being launched is the
    reference's frame loop, cpp/opticalflow.cc:209-321, on N disjoint frame ranges."""
    import torch

    have = torch.cuda.device_count()
    if os.environ.get("POLYCHASE_BENCH_SHARE_GPU") != "1" and have < n:
        print(f"bench.py: --gpus {n} needs {n} GPUs, this node has {have} "
              "(POLYCHASE_BENCH_SHARE_GPU=1 puts all ranks on GPU 0 over gloo: a testing aid, not a measurement)", file=sys.stderr)
        return 2
    port = free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--no-c3", action="store_true", help="skip the nested 4K configuration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true",
                    help="skip the extra per-class timing pass (profilers: only warm-up + timed launches remain)")
    ap.add_argument("--force-dist-path", action="store_true",
                    help="exercise the device-log + stitch code path with a single rank (testing)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the CPU-baseline sample")
    ap.add_argument("--no-counters", action="store_true", help="no rocprofv3 --pmc passes after the timed regions (committed profile instead)")
    ap.add_argument("--no-arith-modes", action="store_true", help="skip the second run of the K steps in the other arithmetic mode")
    ap.add_argument("--no-c5", action="store_true", help="skip the C5 end-to-end block (analysis -> tracking -> refinement)")
    ap.add_argument("--arith", default=None, choices=sorted(ARITH_FLAGS), help="arithmetic mode of the headline (default: the library's)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))     # N ranks of this very command; rank 0 of them prints the JSON line
    # Before the process first touches HIP: every stream of the engine gets a hardware queue of its own (the runtime's default
    # is four for the whole process; the library raises it to sixteen when it is loaded first -- here torch initialises HIP
    # before that -- csrc/hip/api.hip: pc_runtime_defaults).  N > 1 adds the stitch's push streams and RCCL's stream.  Idle
    # queues cost nothing (profiles/r03_stream_layouts.jsonl, "one preparation stream, 8 queues").
    # Several processes on ONE GPU (the SHARE_GPU testing aid) keep the runtime's four: a GPU whose hardware queue slots are
    # oversubscribed by several processes time-slices them (two ranks with 16 queues each on one GPU: 10 x slower).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "4" if os.environ.get("POLYCHASE_BENCH_SHARE_GPU") == "1" else ("8" if args.gpus > 1 else "16"))
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(1, args.gpus):
        sys.exit(f"bench.py: --gpus {args.gpus} but the job has WORLD_SIZE={world}: refusing to report a mislabelled number")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # the container hostname may not resolve
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # testing aid (tests/test_bench_gpu.py): several ranks on ONE GPU over gloo -- the N > 1 control flow of this script
    # without an N-GPU node.  RCCL wants one GPU per rank, so the real thing stays the default.
    share_gpu = os.environ.get("POLYCHASE_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # testing aid (tests/test_rccl_gpu.py): --force-dist-path with POLYCHASE_BENCH_RCCL_WORLD1=1 builds a ONE-rank RCCL
    # communicator, and every collective of the N > 1 path (barrier, all_reduce, all_gather, all_gather_into_tensor of the
    # log pieces) goes through RCCL on the one GPU a test box has -- the calls, dtypes and buffer shapes of --gpus N
    rccl_world1 = world == 1 and args.force_dist_path and os.environ.get("POLYCHASE_BENCH_RCCL_WORLD1") == "1"
    if world > 1 or os.environ.get("POLYCHASE_BENCH_TEST_HANG"):
        wd_start(rank, float(os.environ.get("POLYCHASE_BENCH_WATCHDOG_S", "300")))
    if world > 1:
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    elif rccl_world1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        port = int(os.environ.get("MASTER_PORT", "0")) or free_port()
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)

    K, W = args.steps, args.warmup
    single = world == 1 and not args.force_dist_path
    under_profiler = any(k.startswith("ROCPROF") for k in os.environ)   # a run under rocprofv3 starts no counter passes of its own
    if under_profiler:
        args.no_counters = args.no_arith_modes = True

    def with_arith_modes(cfg, steps, **kw):
        """the configuration in the headline's arithmetic mode, then the same K steps in the other one (light: regions only)"""
        o = run_config(cfg, steps, W, args, rank, world, dev, arith=args.arith, **kw)
        if not args.no_arith_modes:
            head_mode = LAST_ARITH
            other = "canonical" if head_mode != "canonical" else "opencv_x86"
            alt = run_config(cfg, steps, W, args, rank, world, dev, with_cpu=False, with_e2e=False, arith=other, light=True)
            if rank == 0:
                o["arith_modes"] = {head_mode: {"value": o["value"], "ms_per_step": o["ms_per_step"], "lk_avg_launch_ms": o["roofline"]["avg_launch_ms"]},
                                    other: {"value": alt["value"], "ms_per_step": alt["ms_per_step"], "lk_avg_launch_ms": alt["roofline"]["avg_launch_ms"]},
                                    "unit": "frames/s", "default": "opencv_x86",
                                    "what": "opencv_x86 = the execution of the OpenCV build the reference links (FMA in the AVX2 Sobel column "
                                            "filter, LK sums in the SSE lane order); canonical = no FMA, LK sums exact in integers (DESIGN.md section 2)"}
        return o

    out = with_arith_modes(args.config, K, with_cpu=single and not args.no_cpu_baseline, with_e2e=single and not args.no_end_to_end)
    if args.config == "c2" and not args.no_c3:
        # the 4K configuration rides along with the same K steps per region; its CPU sample is shorter.  With N > 1 this is
        # C4's per-rank workload (config.workload says so)
        args.cpu_seconds = min(args.cpu_seconds, 8.0)
        c3 = with_arith_modes("c3", K, with_cpu=single and not args.no_cpu_baseline, with_e2e=single and not args.no_end_to_end)
        if rank == 0:
            out["c3"] = c3
    if single and rank == 0 and args.config == "c2" and not args.no_arith_modes and not under_profiler:
        try:
            out["hard_content"] = hard_content_block()
        except Exception as e:
            out["hard_content"] = {"error": f"{type(e).__name__}: {e}"}
    if single and rank == 0 and args.config == "c2" and not args.no_c5 and not under_profiler:
        try:
            out["c5"] = c5_block()
        except Exception as e:
            out["c5"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        try:
            out["host"] = host_block()
        except Exception as e:
            out["host"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(out), flush=True)
    _WD["printed"] = True
    wd_tick("shutting the process group down")
    if dist.is_initialized():
        dist.destroy_process_group()
    _WD["done"] = True


if __name__ == "__main__":
    main()
