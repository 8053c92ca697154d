"""CPU: the C-ABI library loads and exports every symbol include/polychase_hip.h declares."""
import ctypes
import os
import re

import pytest

from polychase_amd import build, hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "polychase_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pc_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(hip.SYMBOLS)


def test_library_exports_all_symbols():
    path = build.hip_library_path()
    if not os.path.exists(path):
        build.build_hip()
    lib = ctypes.CDLL(path)
    for s in _declared_symbols():
        assert hasattr(lib, s), s


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hip.PolychaseHipError):
        hip.Context(0)
    assert b"polychase_hip" in hip.load().pc_version()


def test_default_options_match_reference():
    g = hip.gftt_options()
    assert (g.quality_level, g.min_distance, g.block_size, g.gradient_size, g.max_corners, g.use_harris,
            g.harris_k, g.grid_rows, g.grid_cols) == (0.01, 5.0, 3, 3, 0, 0, 0.04, 4, 4)
    f = hip.flow_options()
    assert (f.window_size, f.max_level, f.term_max_iters, f.term_epsilon, f.min_eigen_threshold) == (10, 3, 30, 0.01, 1e-4)


@pytest.mark.parametrize("preset", [None, "5"])
def test_runtime_init_raises_the_hardware_queue_default_and_dlopen_does_not(preset):
    """include/polychase_hip.h: pc_runtime_init -- sets GPU_MAX_HW_QUEUES=16 in the process environment (the HIP runtime
    reads it at its first call: a stream per hardware queue), leaves a user's value alone, reports that
    the ROCm runtime was not up yet, and is idempotent.  LOADING the library must change nothing (round 3 did it in a
    constructor: a side effect of dlopen on every HIP user of the process).  Checked through libc's getenv in a fresh
    process (os.environ is a snapshot taken at interpreter start)."""
    import subprocess
    import sys

    path = build.hip_library_path()
    if not os.path.exists(path):
        build.build_hip()
    code = ("import ctypes, sys\n"
            "libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p\n"
            "before = libc.getenv(b'GPU_MAX_HW_QUEUES')\n"
            f"L = ctypes.CDLL({path!r})\n"
            "loaded = libc.getenv(b'GPU_MAX_HW_QUEUES')\n"
            "up, q = ctypes.c_int(-1), ctypes.c_int(-1)\n"
            "assert L.pc_runtime_init(ctypes.byref(up), ctypes.byref(q)) == 0\n"
            "after = libc.getenv(b'GPU_MAX_HW_QUEUES')\n"
            "assert L.pc_runtime_init(None, None) == 0\n"
            "print(before, loaded, after, up.value, q.value)\n")
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    if preset is not None:
        env["GPU_MAX_HW_QUEUES"] = preset
    r = subprocess.run([sys.executable, "-c", code], env=env, text=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    before, loaded, after, up, q = r.stdout.split()
    assert up == "0", "no HIP call has been made in that process"
    if preset is None:
        assert before == loaded == "None" and after == "b'16'" and q == "16", r.stdout
    else:
        assert before == loaded == after == f"b'{preset}'" and q == preset, r.stdout
