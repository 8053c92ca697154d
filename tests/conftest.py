import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu() -> bool:
    try:
        import torch
        if torch.cuda.is_available():
            return True
    except Exception:
        pass
    return os.path.exists("/dev/kfd") and bool(os.environ.get("POLYCHASE_ASSUME_GPU"))


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a box without a GPU skips the gpu-marked tests instead of failing them.  An explicit
    `-m gpu` run is left alone: there a missing device (or a missing HIP library) must FAIL loudly."""
    if "gpu" in (config.getoption("-m") or ""):
        return
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device here (the hot path has no CPU fallback); run with -m gpu on an MI355X")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The CPU oracle is test infrastructure: build it on demand (gcc only, ~2 s)."""
    so = os.path.join(ROOT, "oracle", "libpc_oracle.so")
    src = os.path.join(ROOT, "oracle", "pc_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libpc_oracle.so"])
    yield
