"""Multi-GPU "Analyze Video": GenerateOpticalFlowDatabase (reference cpp/opticalflow.cc:209-321) with the frame1 loop
sharded over the GPUs of one node, one process per GPU (BASELINE.json config C4).

Every rank runs the SAME host code as the single-GPU path -- polychase_core's C++ driver over the C ABI -- on its
contiguous range of frame1 ids (the frames up to 8 outside the range are ingested as tracking targets only, no
detection).  There is no collective on the data path.  Rank 0, which owns the SQLite file, stores its shard as it goes;
the other ranks append their records to a log in GPU memory that is handed over in pieces of a few frames while the
analysis continues: the pieces travel to rank 0 over RCCL (ncclSend / ncclRecv, device to device over xGMI) in frame order
under credit flow control and are stored as they arrive -- the statements of the single-process run in the same order, so
the SQLite file does not depend on the number of ranks, and no rank ever holds more than a few pieces.  (Only rank 0 reads
the records, so this is a gather; bench.py times the all-gather variant of the same pieces, distributed.ChunkedLogStitch.)

ONE implementation of that protocol: csrc/host/multi_gpu.cc (GenerateOpticalFlowDatabaseMultiGpu), reached here through
polychase_core.generate_optical_flow_database_multi_gpu -- this module is its launcher and its frame source, nothing more
(rounds 3-5 carried a Python twin of the protocol over torch.distributed: removed in round 6).  The ranks find each other
through MASTER_ADDR / MASTER_PORT (+ 17: the port itself belongs to the launcher's rendezvous) or POLYCHASE_MULTI_GPU_PORT.

EVERY N > 1 FIGURE OF THIS PATH IS BOUND BY ITS SINGLE WRITER once the analysis outruns one SQLite connection (4K: ~370-460
frames/s on one writer against ~670 analysed per GPU): what scales with N is the analysis-only rate (bench.py --gpus N).

    python -m polychase_amd.analyze --gpus 8 --synthetic c3 --frames 2400 --database /tmp/clip.db

launches the eight ranks itself (the same through torch.distributed.run / torchrun works too).

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
    distance of the detector yields ~1 per 51 px on a fully textured frame: 162 k at 3840 x 2160, i.e. 22 MB per frame
    with its 8 flows).  A piece that fills up early simply ends early (analysis_driver.cc), so the estimate only sets
    how many frames a piece usually holds."""
    kp = keypoints_per_frame or (width * height // 40 + 4096)
    return D.log_capacity_bytes(n_frames, kp, n_targets)


def analyze(width: int, height: int, first_frame: int, num_frames: int, frame_accessor, database_path: str,
            detector_options=None, flow_options=None, callback=None, group=None, device=None, piece_frames: int = 16,
            keypoints_per_frame: int | None = None):
    """Analyze frames first_frame .. first_frame + num_frames - 1 with all ranks of `group` (default: the world; one rank
    when torch.distributed is not initialised) and store the flow database at `database_path` (written by rank 0).
    frame_accessor(frame_id) -> H x W x 3 uint8 (numpy, or a torch tensor on this rank's GPU), or float32 H x W x 3|4.

    Rank 0 runs its own shard straight into the database (the single-GPU path: inserts overlap the analysis); every other
    rank analyses its shard into a two-part device log and hands the log over in pieces of `piece_frames` frames, which
    travel to rank 0 over RCCL in frame order (csrc/host/multi_gpu.cc) and are stored as they arrive.  GPU and
    host memory are bounded by the pieces in flight, whatever the length of the clip.  An existing database must hold
    the same analysis (write_optical_flow_records refuses other keypoints).
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
    out = {"rank": rank, "world": world, "shard": [begin, end], "frames": end - begin}
    if world == 1:
        st = core.generate_optical_flow_shard(vi, frame_accessor, callback, database_path, begin, end, detector_options=gopt,
                                              flow_options=fopt)["stats"]
        t1 = time.perf_counter()
        out.update(seconds_analysis=t1 - t0, seconds_stitch=0.0, seconds_database=st.seconds_db, seconds_total=t1 - t0, log_bytes=0,
                   frames_processed=st.frames_processed, cancelled=False,
                   written={"keypoint_rows": st.keypoint_rows_written, "flow_rows": st.flow_rows_written})
        return out

    # N ranks: the C++ protocol (csrc/host/multi_gpu.cc) -- rank 0 stores its own shard while it receives the first pieces of rank 1
    share_gpu = os.environ.get("POLYCHASE_ANALYZE_SHARE_GPU") == "1"
    port = int(os.environ.get("POLYCHASE_MULTI_GPU_PORT", "0")) or int(os.environ.get("MASTER_PORT", "29594")) + 17
    res = core.generate_optical_flow_database_multi_gpu(
        vi, frame_accessor, callback, database_path, world, rank, master_addr=os.environ.get("MASTER_ADDR", "127.0.0.1"), master_port=port,
        device=dev.index or 0, piece_frames=piece_frames,
        transport="tcp" if share_gpu else "rccl",     # (RCCL refuses two ranks on one device: the one-GPU testing aid moves the pieces over TCP)
        keypoints_per_frame=keypoints_per_frame or 0, detector_options=gopt, flow_options=fopt)
    st = res["stats"]
    out.update(seconds_analysis=res["seconds_analysis"], seconds_stitch=res["seconds_blocked"], seconds_database=st.seconds_db,
               log_bytes=int(res["bytes_moved"]), frames_processed=st.frames_processed, pieces=int(res["pieces"]), cancelled=bool(res["cancelled"]),
               written={"keypoint_rows": st.keypoint_rows_written, "flow_rows": st.flow_rows_written} if rank == 0 else None)
    out["seconds_total"] = time.perf_counter() - t0
    return out


def _self_launch(n: int, argv) -> int:
    """`python -m polychase_amd.analyze --gpus N ...`: N ranks of this command through torch.distributed.run, one per GPU."""
    import socket
    import subprocess

    import torch

    have = torch.cuda.device_count()
    if os.environ.get("POLYCHASE_ANALYZE_SHARE_GPU") != "1" and have < n:
        print(f"polychase_amd.analyze: --gpus {n} needs {n} GPUs, this node has {have}", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    # the ranks keep the caller's working directory (relative --database / --npy paths mean what the user typed); the
    # package is found through PYTHONPATH instead of a change of directory
    parent = os.path.dirname(_HERE)
    env["PYTHONPATH"] = parent + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "polychase_amd.analyze", *argv]
    return subprocess.call(cmd, env=env)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--synthetic", choices=["c1", "c2", "c3"], help="synthetic clip of that configuration (BASELINE.md section 4)")
    ap.add_argument("--npy", help=".npy file of RGB frames, shape (N, H, W, 3) uint8")
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--first-frame", type=int, default=1)
    ap.add_argument("--max-level", type=int, default=None)
    ap.add_argument("--database", required=True)
    ap.add_argument("--piece-frames", type=int, default=16, help="frames per piece of the record log handed to rank 0")
    ap.add_argument("--gpus", type=int, default=0, help="launch this many ranks (one per GPU) of this command")
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return _self_launch(args.gpus, sys.argv[1:] if argv is None else list(argv))
    if args.gpus > 1 and int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit(f"polychase_amd.analyze: --gpus {args.gpus} but the job has WORLD_SIZE={os.environ['WORLD_SIZE']}")

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
    out = analyze(w, h, args.first_frame, n, accessor, args.database, core.GFTTOptions(), fopt, device=dev,
                  piece_frames=args.piece_frames)
    import json
    sys.stdout.write(json.dumps(out) + "\n")     # ONE write per rank: the ranks share the launcher's stdout
    sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
