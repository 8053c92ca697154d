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
