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
