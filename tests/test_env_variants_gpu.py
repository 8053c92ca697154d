"""GPU: every execution-strategy knob of the library must leave the flow database unchanged.

The knobs select kernels (LK variants 1/2/3, fused / unfused pyramid), the detection path (device-count bucket sort / the
synchronous slow path with the rocPRIM sort) and the stream layout of the analyzer (gate between the job lanes, detection
on its own stream(s), the helper kernels' issue priority, the copy stream of host frames, one job lane, the parked engine).  They are read when the library or a context is created, so each variant runs in its own process
(tests/_analyze_hash.py: polychase_core.generate_optical_flow_database on a 26-frame clip -> sha256 of all rows)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

VARIANTS = [
    {"POLYCHASE_LK_GATE": "0"},
    {"POLYCHASE_DETECT_STREAMS": "1"},
    {"POLYCHASE_DETECT_STREAMS": "2", "GPU_MAX_HW_QUEUES": "8"},
    {"POLYCHASE_GFTT_SLOW_PATH": "1"},
    {"POLYCHASE_LK_VARIANT": "1"},
    {"POLYCHASE_PYRAMID_VARIANT": "1"},
    {"POLYCHASE_MINEIG_VARIANT": "1"},
    {"POLYCHASE_GFTT_GENERAL": "1"},
    {"POLYCHASE_HELPER_PRIO": "0"},
    {"POLYCHASE_HELPER_PRIO": "1"},
    {"POLYCHASE_COPY_STREAM": "0"},
    {"POLYCHASE_LK_LANES": "1"},
    {"POLYCHASE_ENGINE_CACHE": "0"},
]


def _hash(extra_env, size=(416, 304, 26)):
    env = dict(os.environ)
    for k in ("POLYCHASE_LK_GATE", "POLYCHASE_DETECT_STREAMS", "POLYCHASE_GFTT_SLOW_PATH", "POLYCHASE_LK_VARIANT",
              "POLYCHASE_PYRAMID_VARIANT", "POLYCHASE_MINEIG_VARIANT", "POLYCHASE_GFTT_GENERAL", "GPU_MAX_HW_QUEUES", "POLYCHASE_HELPER_PRIO", "POLYCHASE_COPY_STREAM", "POLYCHASE_LK_LANES",
              "POLYCHASE_ENGINE_CACHE", "POLYCHASE_ARITH"):
        env.pop(k, None)
    env.update(extra_env)
    r = subprocess.run([sys.executable, os.path.join(HERE, "_analyze_hash.py"), *map(str, size)], env=env, text=True,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("HASH ")]
    assert r.returncode == 0 and lines, f"variant {extra_env} failed:\n{r.stdout[-2000:]}\n{r.stderr[-3000:]}"
    return lines[-1].split()[1]


@pytest.fixture(scope="module")
def reference_hash():
    return _hash({})


# (Round 6 tried the variants' processes four at a time to save suite time: 307 s instead of 48 -- processes that share the GPU
# time-slice whole contexts, and the analyzer's stream gates spin meanwhile.  One after the other.)
@pytest.mark.parametrize("variant", VARIANTS, ids=lambda v: ",".join(f"{k}={x}" for k, x in v.items()))
def test_knob_does_not_change_the_database(reference_hash, variant):
    assert _hash(variant) == reference_hash


def test_the_three_min_eig_kernels_agree_in_the_row_fma_mode():
    """POLYCHASE_ARITH=opencv_x86_rows (PC_ARITH_SOBEL_ROW_FMA on top of the default: the second hypothesis about the AVX2 build):
    the fused kernel, the LDS-tiled one and the general pair give one database -- and an unknown mode name is refused."""
    base = _hash({"POLYCHASE_ARITH": "opencv_x86_rows"})
    assert _hash({"POLYCHASE_ARITH": "opencv_x86_rows", "POLYCHASE_MINEIG_VARIANT": "1"}) == base
    assert _hash({"POLYCHASE_ARITH": "opencv_x86_rows", "POLYCHASE_GFTT_GENERAL": "1"}) == base
    with pytest.raises(AssertionError):
        _hash({"POLYCHASE_ARITH": "no_such_mode"})
