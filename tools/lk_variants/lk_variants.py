#!/usr/bin/env python3
"""lk_variants.py -- build (here, no GPU needed) or run (on the GPU box) kernel variants of the library.

    python tools/lk_variants/lk_variants.py build  tag:-DFLAG=1,-DOTHER=2 [tag2:...]     -> polychase_amd/lib/variants/
    python tools/lk_variants/lk_variants.py run [--config c2] [--reps 20] [tag ...]      -> one JSON line per variant

Experiment harness only: the product always loads polychase_amd/lib/libpolychase_hip.so.
"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "polychase_amd", "lib", "variants")


def build_hip_variant(tag, extra_flags):
    """lib/variants/libpolychase_hip_<tag>.so: the library's sources compiled with extra -D flags (on a tree patched with
    r04_experiments.patch: the product kernel has no experiment switches left)"""
    import subprocess
    from concurrent.futures import ThreadPoolExecutor

    from polychase_amd import build as b
    vdir, odir = os.path.join(b.LIB_DIR, "variants"), os.path.join(b.LIB_DIR, "obj_" + tag)
    os.makedirs(vdir, exist_ok=True)
    os.makedirs(odir, exist_ok=True)
    out = os.path.join(vdir, f"libpolychase_hip_{tag}.so")
    srcs = [os.path.join(b.HIP_DIR, s) for s in b.HIP_SOURCES]
    objs = [os.path.join(odir, os.path.splitext(s)[0] + ".o") for s in b.HIP_SOURCES]
    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(lambda so: subprocess.check_call(["hipcc", *b.HIP_FLAGS, *b.HIP_SOURCE_FLAGS.get(os.path.basename(so[0]), []), *extra_flags, "-c", so[0], "-o", so[1]]), zip(srcs, objs)))
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs])
    return out


def main():
    cmd = sys.argv[1]
    if cmd == "build":
        for spec in sys.argv[2:]:
            tag, _, flags = spec.partition(":")
            print(build_hip_variant(tag, [f for f in flags.split(",") if f]))
    elif cmd == "run":
        args, tags = [], []
        it = iter(sys.argv[2:])
        for a in it:
            if a.startswith("--"):
                args += [a, next(it)]
            else:
                tags.append(a)
        libs = [("default", None)] + [(os.path.basename(p)[len("libpolychase_hip_"):-3], p)
                                      for p in sorted(glob.glob(os.path.join(VDIR, "libpolychase_hip_*.so")))]
        for tag, path in libs:
            if tags and tag not in tags:
                continue
            env = dict(os.environ)
            if path:
                env["POLYCHASE_HIP_LIB"] = path
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lk_bench.py"), *args], env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print(tag, line[-1] if line else f"FAILED rc={r.returncode}: {r.stderr[-800:]}", flush=True)


if __name__ == "__main__":
    main()
