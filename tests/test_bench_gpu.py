"""GPU: bench.py's multi-rank code path on one GPU (`--force-dist-path`: device-resident record log + the chunked stitch,
world size 1), over several timed regions -- the path the driver runs with --gpus N > 1, where every region reuses the log
and the stitch (a stale piece list made rank_logs() return the records of all regions; found in round 2)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_forced_dist_path_runs_several_regions():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist-path", "--no-cpu-baseline", "--no-c3",
                        "--no-end-to-end", "--no-breakdown", "--config", "c1", "--steps", "16", "--warmup", "4"],
                       text=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, r.stderr[-3000:]
    out = json.loads(lines[-1])
    assert out["steps"] == 16 and out["n_gpus"] == 1 and out["value"] > 0
    assert out["config"]["timed_regions"] > 1, "the short C1 regions must repeat (the case that reuses the stitch)"


@pytest.mark.parametrize("stitch", ["peer", "rccl", "peer-broken"])
def test_two_ranks_on_one_gpu_over_gloo(stitch):
    """bench.py exactly as the driver launches it for --gpus 2 (torch.distributed.run, two processes), except that both
    ranks sit on GPU 0 and the process group is gloo (POLYCHASE_BENCH_SHARE_GPU=1): the region count agreed by all-reduce,
    the stitch with two shards, the per-rank record checks and the max-over-ranks timing all run.  "peer" (the default): each
    rank maps the other's receive buffer through HIP IPC and pushes its pieces with device-to-device copies
    (distributed.PeerLogStitch; here both buffers live on GPU 0); "rccl": the chunked all-gather (gloo here);
    "peer-broken": rank 1 cannot export its buffer -- BOTH ranks must fall back to the all-gather, nobody waits."""
    env = dict(os.environ, POLYCHASE_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1", POLYCHASE_BENCH_STITCH=stitch.split("-")[0],
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    if stitch == "peer-broken":
        env["POLYCHASE_TEST_BREAK_PEER_EXPORT"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "16",
                        "--warmup", "4", "--config", "c1", "--no-c3", "--no-breakdown"],
                       text=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 16 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["stitch"].startswith("xgmi peer copies" if stitch == "peer" else "rccl all_gather"), (out["config"]["stitch"], r.stderr[-2000:])


def test_gpus_flag_launches_the_ranks_itself():
    """`python3 bench.py --gpus 2 --steps 16 --warmup 4` -- the form the driver uses, no torchrun around it: bench.py
    re-executes itself through torch.distributed.run with two ranks (on this one-GPU box both on GPU 0 over gloo, the
    SHARE_GPU testing aid) and exactly one JSON line with n_gpus == 2 comes out (round 2 parsed --gpus and ignored it)."""
    env = dict(os.environ, POLYCHASE_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "16", "--warmup", "4",
                        "--config", "c1", "--no-c3", "--no-breakdown"],
                       text=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 16 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["parallelism"] == "frame-shard x2"


def test_peer_stitch_selftest_three_ranks_with_logs_of_different_sizes():
    """distributed.PeerLogStitch alone, three processes on GPU 0: every rank reads back every rank's bytes, twice (the slot
    size must be agreed: each rank's log has its own length -- the first version used the local length and wrote to the
    wrong offsets of the peers' buffers)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", GPU_MAX_HW_QUEUES="4")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                        "--master-port", "29547", os.path.join(ROOT, "tools", "peer_stitch_selftest.py")],
                       text=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])


def test_four_ranks_push_on_three_streams():
    """The stitch of more than four ranks deals the peers onto three copy streams; here four ranks on GPU 0 are told to
    (POLYCHASE_PEER_PUSH_STREAMS=3): every rank must still end up with every rank's records (asserted inside bench.py)."""
    env = dict(os.environ, POLYCHASE_BENCH_SHARE_GPU="1", POLYCHASE_PEER_PUSH_STREAMS="3", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "16", "--warmup", "4",
                        "--config", "c1", "--no-c3", "--no-breakdown"],
                       text=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["config"]["stitch"].startswith("xgmi peer copies")


def test_gpus_flag_fails_loudly_without_enough_gpus():
    """Without the testing aid a one-GPU box must refuse --gpus 2 instead of measuring one GPU and labelling it two."""
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "POLYCHASE_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"],
                       text=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "needs 2 GPUs" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]
