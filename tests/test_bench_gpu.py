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


def test_two_ranks_on_one_gpu_over_gloo():
    """bench.py exactly as the driver launches it for --gpus 2 (torch.distributed.run, two processes), except that both
    ranks sit on GPU 0 and the process group is gloo (POLYCHASE_BENCH_SHARE_GPU=1): the region count agreed by all-reduce,
    the chunked stitch with two shards, the per-rank record checks and the max-over-ranks timing all run."""
    env = dict(os.environ, POLYCHASE_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "16",
                        "--warmup", "4", "--config", "c1", "--no-c3", "--no-breakdown"],
                       text=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 16 and out["scaling"] == "weak" and out["value"] > 0
