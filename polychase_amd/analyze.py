"""Multi-GPU "Analyze Video": GenerateOpticalFlowDatabase (reference cpp/opticalflow.cc:209-321) with the frame1 loop
sharded over the GPUs of one node, one process per GPU (BASELINE.json config C4).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \\
        -m polychase_amd.analyze --synthetic c3 --frames 2400 --database /tmp/clip.db

Every rank runs the SAME host code as the single-GPU path -- polychase_core's C++ driver over the C ABI -- on its
contiguous range of frame1 ids (polychase_core.generate_optical_flow_records: the frames up to 8 outside the range are
ingested as tracking targets only, no detection) and appends the records to a log in its GPU's memory.  There is no
collective on the data path.  The one exchange is the stitch: an all-gather of the logs over RCCL (torch.distributed,
backend "nccl"), after which rank 0 stores them through polychase_core.write_optical_flow_records in frame order -- the
statements of the single-process run in the same order, so the SQLite file does not depend on the number of ranks.

`analyze()` is the library entry point (frames from any accessor, like generate_optical_flow_database); the command
line reads a synthetic clip (--synthetic c1|c2|c3) or an .npy stack of RGB frames.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

from . import distributed as D

_HERE = os.path.dirname(os.path.abspath(__file__))


def _core():
    sys.path.insert(0, os.path.join(_HERE, "core"))
    import polychase_core   # the C++ module: there is no other implementation of the path

    return polychase_core


def log_capacity(n_frames: int, width: int, height: int, n_targets: int = 8, keypoints_per_frame: int | None = None) -> int:
    """Bytes of device log for `n_frames` frame1 records.  Default estimate: one keypoint per 40 pixels (the 5-px minimum
    distance of the detector yields ~1 per 51 px on a fully textured frame), which `analyze` doubles once if a shard
    turns out denser."""
    kp = keypoints_per_frame or (width * height // 40 + 4096)
    return D.log_capacity_bytes(n_frames, kp, n_targets)


def analyze(width: int, height: int, first_frame: int, num_frames: int, frame_accessor, database_path: str,
            detector_options=None, flow_options=None, callback=None, group=None, device=None):
    """Analyze frames first_frame .. first_frame + num_frames - 1 with all ranks of `group` (default: the world; one rank
    when torch.distributed is not initialised) and store the flow database at `database_path` (written by rank 0).
    frame_accessor(frame_id) -> H x W x 3 uint8 (numpy, or a torch tensor on this rank's GPU), or float32 H x W x 3|4.
    Returns a dict of per-phase seconds and counts (on every rank)."""
    import torch
    import torch.distributed as dist

    core = _core()
    have_pg = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if have_pg else 1
    rank = dist.get_rank(group) if have_pg else 0
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    os.environ["POLYCHASE_DEVICE"] = str(dev.index or 0)     # the C++ driver's device (analysis_driver.cc)
    begin, end = D.shard_range(first_frame, num_frames, world, rank)
    vi = core.VideoInfo(width, height, first_frame, num_frames)
    gopt = detector_options or core.GFTTOptions()
    fopt = flow_options or core.OpticalFlowOptions()
    t0 = time.perf_counter()
    cap = log_capacity(end - begin + 1, width, height)
    used, stats = 0, None
    for attempt in range(3):
        log = torch.empty(cap, dtype=torch.uint8, device=dev)
        try:
            used, stats = core.generate_optical_flow_records(vi, frame_accessor, callback, begin, end, log.data_ptr(), log.numel(), gopt, fopt)
            break
        except RuntimeError as e:
            if "device log full" not in str(e) or attempt == 2:
                raise
            del log
            cap *= 2      # a denser clip than the estimate: once more with twice the room
    t1 = time.perf_counter()
    # ---- the stitch: the one collective of the path ----
    if world > 1:
        gathered, sizes = D.all_gather_device_log(log, used, group)
        torch.cuda.synchronize(dev)
    else:
        gathered, sizes = log[:used][None], [used]
    t2 = time.perf_counter()
    written = None
    if rank == 0:
        rows_kp = rows_flow = 0
        for r in range(world):      # ranks own increasing frame ranges: rank order is frame order
            host = gathered[r, :sizes[r]].cpu().numpy()
            st = core.write_optical_flow_records(database_path, host, int(sizes[r]))
            rows_kp += st.keypoint_rows_written
            rows_flow += st.flow_rows_written
        written = {"keypoint_rows": rows_kp, "flow_rows": rows_flow}
    if world > 1:
        dist.barrier(group)
    t3 = time.perf_counter()
    return {"rank": rank, "world": world, "shard": [begin, end], "frames": end - begin, "log_bytes": int(used),
            "seconds_analysis": t1 - t0, "seconds_stitch": t2 - t1, "seconds_database": t3 - t2, "written": written,
            "frames_processed": stats.frames_processed if stats else 0}


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--synthetic", choices=["c1", "c2", "c3"], help="synthetic clip of that configuration (BASELINE.md section 4)")
    ap.add_argument("--npy", help=".npy file of RGB frames, shape (N, H, W, 3) uint8")
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--first-frame", type=int, default=1)
    ap.add_argument("--max-level", type=int, default=None)
    ap.add_argument("--database", required=True)
    args = ap.parse_args(argv)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # testing aid (tests/test_sharded_gpu.py): all ranks on GPU 0 over gloo -- the launcher end to end without an N-GPU node
    share_gpu = os.environ.get("POLYCHASE_ANALYZE_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share_gpu:
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    core = _core()
    fopt = core.OpticalFlowOptions()
    if args.npy:
        stack = np.load(args.npy, mmap_mode="r")
        n, h, w, _ = stack.shape
        n = min(n, args.frames)
        accessor = lambda fid: np.ascontiguousarray(stack[fid - args.first_frame])
    else:
        from . import synth
        w, h, ml = {"c1": (640, 480, 3), "c2": (1920, 1080, 3), "c3": (3840, 2160, 4)}[args.synthetic or "c2"]
        fopt.max_level = ml
        n = args.frames
        clip = synth.NoiseClip(w, h, min(n, 500), device=str(dev))
        period = 2 * clip.n - 2

        def accessor(fid):     # the clip played forwards and backwards: any number of frames, continuous motion
            t = (fid - args.first_frame) % period
            return clip.frame_torch(t if t < clip.n else period - t)
    if args.max_level is not None:
        fopt.max_level = args.max_level
    if int(os.environ.get("RANK", "0")) == 0 and os.path.exists(args.database):
        os.remove(args.database)
    if world > 1:
        dist.barrier()
    out = analyze(w, h, args.first_frame, n, accessor, args.database, core.GFTTOptions(), fopt, device=dev)
    import json
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
