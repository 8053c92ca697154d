"""In-tree build of the native libraries (hipcc cross-compiles gfx950 without a GPU).

  polychase_amd/lib/libpolychase_hip.so   -- HIP kernels + C ABI (include/polychase_hip.h)

Run:  python -m polychase_amd.build [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB_DIR = os.path.join(PKG, "lib")
OBJ_DIR = os.path.join(PKG, "lib", "obj")
HIP_DIR = os.path.join(PKG, "csrc", "hip")

HIP_SOURCES = ["kernels_image.hip", "kernels_gftt.hip", "kernels_lk.hip", "api.hip"]
# -ffp-contract=off: the float stages must match the oracle bit-for-bit (no FMA fusion).
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall",
             "-Wno-unused-function"]


def _newer(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _run(cmd: list[str]) -> None:
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout))


def hip_library_path() -> str:
    return os.path.join(LIB_DIR, "libpolychase_hip.so")


def build_hip(force: bool = False) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = [os.path.join(HIP_DIR, h) for h in os.listdir(HIP_DIR) if h.endswith(".hpp")]
    headers.append(os.path.join(ROOT, "include", "polychase_hip.h"))
    out = hip_library_path()
    srcs = [os.path.join(HIP_DIR, s) for s in HIP_SOURCES]
    if not force and _newer(out, srcs + headers):
        return out
    objs = [os.path.join(OBJ_DIR, os.path.splitext(s)[0] + ".o") for s in HIP_SOURCES]

    def compile_one(pair):
        src, obj = pair
        if force or not _newer(obj, [src] + headers):
            _run(["hipcc", *HIP_FLAGS, "-c", src, "-o", obj])

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(compile_one, zip(srcs, objs)))
    _run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs])
    return out


def build_all(force: bool = False) -> None:
    build_hip(force)


if __name__ == "__main__":
    build_all("--force" in sys.argv)
    print(hip_library_path())
