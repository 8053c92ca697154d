"""GPU, two or more devices: the first contact of this repository with RCCL on separate GPUs (one process per GPU,
torch.distributed backend "nccl" over xGMI) -- skipped on a one-GPU box, where tests/test_bench_gpu.py and
tests/test_sharded_gpu.py run the same entry points with all ranks on GPU 0 over gloo.

    bench.py --gpus 2          the benchmark's N > 1 path: per-rank shards, pieces all-gathered inside the timed region
    polychase_amd.analyze      the product: rank 0 stores, rank 1 streams its record pieces with send / recv in frame order
(reference loop being sharded: cpp/opticalflow.cc:209-321)"""
import json
import os
import sqlite3
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gpus():
    import torch
    return torch.cuda.device_count()


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "POLYCHASE_BENCH_SHARE_GPU",
                                                             "POLYCHASE_ANALYZE_SHARE_GPU")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


def test_bench_collectives_over_a_one_rank_rccl_communicator():
    """Runs on ANY GPU box: bench.py's N > 1 code path (device log, pieces all-gathered inside the timed region, region
    agreement, max-over-ranks timing) with a ONE-rank RCCL communicator -- init_process_group("nccl", device_id=...),
    barrier, all_reduce of int64 / float64, all_gather of the piece sizes, all_gather_into_tensor of uint8 log pieces all
    execute in RCCL; the stitched log must parse into exactly the frames analysed (asserted inside bench.py)."""
    env = _env()
    env["POLYCHASE_BENCH_RCCL_WORLD1"] = "1"
    env["POLYCHASE_BENCH_STITCH"] = "rccl"      # the all-gather itself (the default stitch pushes with the copy engines)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "16", "--warmup", "4", "--config", "c1", "--no-c3",
                        "--no-breakdown", "--force-dist-path"], text=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900,
                       env=env, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-4000:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0
    assert out.get("collectives_backend") == "nccl", out.get("collectives_backend")


@pytest.mark.parametrize("stitch", ["peer", "rccl"])
def test_bench_two_ranks_over_rccl(stitch):
    if _gpus() < 2:
        pytest.skip("one GPU: RCCL needs a device per rank")
    env = _env()
    env["POLYCHASE_BENCH_STITCH"] = stitch
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "16", "--warmup", "4", "--config", "c1",
                        "--no-c3", "--no-breakdown"], text=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200, env=env, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-4000:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["parallelism"] == "frame-shard x2"
    assert out["config"]["stitch"].startswith("xgmi peer copies" if stitch == "peer" else "rccl all_gather")


def test_analyze_two_ranks_over_rccl(tmp_path):
    if _gpus() < 2:
        pytest.skip("one GPU: RCCL needs a device per rank")

    def run(gpus, db):
        cmd = [sys.executable, "-m", "polychase_amd.analyze", "--synthetic", "c1", "--frames", "41", "--piece-frames", "4", "--database", db]
        if gpus > 1:
            cmd += ["--gpus", str(gpus)]
        r = subprocess.run(cmd, text=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200, env=_env(), cwd=ROOT)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])

    def dump(path):
        con = sqlite3.connect(path)
        rows = (list(con.execute("select rowid, image_id, rows, keypoints from keypoints order by rowid")),
                list(con.execute("select rowid, image_id_from, image_id_to, rows, src_keypoints_indices, tgt_keypoints, flow_errors "
                                 "from optical_flow order by rowid")))
        con.close()
        return rows

    one, two = str(tmp_path / "one.db"), str(tmp_path / "two.db")
    run(1, one)
    run(2, two)
    a = dump(one)
    assert len(a[0]) == 41 and dump(two) == a
