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

HIP_SOURCES = ["kernels_image.hip", "kernels_pyramid.hip", "kernels_gftt.hip", "kernels_lk.hip", "kernels_lk3.hip", "kernels_lk4a.hip", "kernels_lk4b.hip", "kernels_lk4c.hip", "kernels_tracker.hip", "kernels_refiner.hip", "kernels_bvh.hip", "api.hip", "api_analyzer.hip", "api_tracker.hip", "api_comm.hip"]
# -ffp-contract=off: the float stages must match the oracle bit-for-bit (no FMA fusion).
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall",
             "-Wno-unused-function"]
# Per-source additions.  The LK kernel with LLVM's max-ILP scheduling strategy: the same instructions (VALU / LDS counts and the
# 136-VGPR allocation unchanged, 9 % fewer s_nop / s_waitcnt), same bits, 1-1.5 % off the step at C2 and C3 in alternating
# runs against the default strategy, max-memory-clause, no post-RA scheduler, -O2 and the AMDGPU pressure trackers
# (tools/probes/flag_ab.sh, profiles/r05_lk_compiler_flags.jsonl).
HIP_SOURCE_FLAGS = {"kernels_lk3.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}


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
    # POLYCHASE_HIP_LIB: an alternative build of the same C ABI (tools/lk_variants/: kernel experiments on a patched tree)
    return os.environ.get("POLYCHASE_HIP_LIB") or os.path.join(LIB_DIR, "libpolychase_hip.so")


def build_hip(force: bool = False) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = [os.path.join(HIP_DIR, h) for h in os.listdir(HIP_DIR) if h.endswith(".hpp")]
    headers.append(os.path.join(ROOT, "include", "polychase_hip.h"))
    out = os.path.join(LIB_DIR, "libpolychase_hip.so")
    srcs = [os.path.join(HIP_DIR, s) for s in HIP_SOURCES]
    if not force and _newer(out, srcs + headers):
        return out
    objs = [os.path.join(OBJ_DIR, os.path.splitext(s)[0] + ".o") for s in HIP_SOURCES]

    def compile_one(pair):
        src, obj = pair
        if force or not _newer(obj, [src] + headers):
            _run(["hipcc", *HIP_FLAGS, *HIP_SOURCE_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj])

    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(compile_one, zip(srcs, objs)))
    _run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs])
    return out


HOST_DIR = os.path.join(PKG, "csrc", "host")
PYBIND_DIR = os.path.join(PKG, "csrc", "pybind")
CORE_DIR = os.path.join(PKG, "core")


def core_module_path() -> str:
    import sysconfig

    return os.path.join(CORE_DIR, "polychase_core" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_core(force: bool = False) -> str:
    """polychase_amd/core/polychase_core*.so: the reference's pybind11 surface over the C ABI."""
    import sysconfig

    import pybind11

    os.makedirs(CORE_DIR, exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = sorted([os.path.join(HOST_DIR, f) for f in os.listdir(HOST_DIR) if f.endswith(".cc")] +
                  [os.path.join(PYBIND_DIR, f) for f in os.listdir(PYBIND_DIR) if f.endswith(".cc")])
    headers = [os.path.join(d, f) for d in (HOST_DIR, PYBIND_DIR) for f in os.listdir(d) if f.endswith(".h")]
    headers.append(os.path.join(ROOT, "include", "polychase_hip.h"))
    out = core_module_path()
    if not force and _newer(out, srcs + headers + [hip_library_path()]):
        return out
    inc = ["-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"], "-I/opt/conda/include"]
    flags = ["-std=c++17", "-O2", "-fPIC", "-fvisibility=hidden", "-Wall", "-pthread"]
    if os.path.exists(os.path.join(PYBIND_DIR, "tracker_bindings.cc")):
        flags.append("-DPC_WITH_TRACKER")
    objs = [os.path.join(OBJ_DIR, "core_" + os.path.splitext(os.path.basename(s))[0] + ".o") for s in srcs]

    def compile_one(pair):
        src, obj = pair
        if force or not _newer(obj, [src] + headers):
            _run(["g++", *flags, *inc, "-c", src, "-o", obj])

    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(compile_one, zip(srcs, objs)))
    # $ORIGIN/../lib: libpolychase_hip.so travels in-tree; libsqlite3.so.0 is the system one
    _run(["g++", "-shared", "-o", out, *objs, "-L" + LIB_DIR, "-lpolychase_hip", "-l:libsqlite3.so.0", "-l:libz.so.1",
          "-L/usr/lib/x86_64-linux-gnu", "-Wl,-rpath,$ORIGIN/../lib", "-pthread"])
    return out


def multi_gpu_tool_path() -> str:
    return os.path.join(LIB_DIR, "polychase_multi_gpu")


def build_multi_gpu_tool(force: bool = False) -> str:
    """polychase_amd/lib/polychase_multi_gpu: tools/multi_gpu/multi_gpu_analyze.cc + the host driver objects of build_core --
    the multi-GPU analysis (csrc/host/multi_gpu.cc) as a plain C++ program: no Python, no torch."""
    src = os.path.join(ROOT, "tools", "multi_gpu", "multi_gpu_analyze.cc")
    out = multi_gpu_tool_path()
    need = ["analysis_driver", "flow_database", "async_write_vfs", "frame_pool", "debug_images", "gpu_context", "multi_gpu", "numa_pin"]
    objs = [os.path.join(OBJ_DIR, f"core_{n}.o") for n in need]
    if not force and _newer(out, [src, hip_library_path()] + objs):
        return out
    _run(["g++", "-std=c++17", "-O2", "-Wall", "-pthread", src, *objs, "-o", out, "-L" + LIB_DIR, "-lpolychase_hip", "-l:libsqlite3.so.0",
          "-l:libz.so.1", "-L/usr/lib/x86_64-linux-gnu", "-Wl,-rpath,$ORIGIN"])
    return out


def build_all(force: bool = False) -> None:
    build_hip(force)
    build_core(force)
    build_multi_gpu_tool(force)


if __name__ == "__main__":
    build_all("--force" in sys.argv)
    print(hip_library_path())
