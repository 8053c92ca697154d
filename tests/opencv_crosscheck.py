#!/usr/bin/env python3
"""opencv_crosscheck.py -- pins oracle/pc_oracle.c against a REAL OpenCV, where one is installed.

The reference calls OpenCV 4.x for all pixel arithmetic; this image has no
cv2, so the oracle is a restatement of those algorithms and the parity claim of this repository is
"unpinned at the OpenCV boundary".  On any machine with `opencv-python` (CPU is enough: the oracle is
plain C) this script runs the same calls the reference makes and compares them with the oracle:

    cv2.cvtColor(COLOR_RGB2GRAY)             opticalflow.cc:259        bit-exact expected
    cv2.cornerMinEigenVal(gray, 3, 3)        gftt.cc:35                bit-exact expected (float order fixed, no FMA)
    cv2.cornerHarris(gray, 3, 3, 0.04)       gftt.cc:31-33             bit-exact expected in one of the two modes
    cv2.buildOpticalFlowPyramid              opticalflow.cc:184        bit-exact expected (images + Scharr planes)
    cv2.calcOpticalFlowPyrLK                 opticalflow.cc:119-125    status equal; positions within 1e-3 px of the
                                                                       canonical oracle (the bound
                                                                       tests/test_oracle_emulation_cpu.py measures between
                                                                       the canonical and the emulated x86 order), and
                                                                       bit-exact against the oracle run in the emulated
                                                                       order of the build at hand (PCO_EMU_LK_SIMD on x86,
                                                                       PCO_EMU_SOBEL_FMA where the AVX2 filters dispatch)

    python tests/opencv_crosscheck.py [--width 640 --height 360] [--write-golden [PATH]]

--write-golden is the PIN KIT (tests/opencv_golden.py): it writes tests/golden/opencv_<version>_<isa>.npz -- inputs and THIS
library's outputs for two small cases -- which the CPU and GPU test suites then hold oracle and HIP path to.

Exit code 0: everything within tolerance; 1: a mismatch (printed); 2: cv2 is not importable.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=360)
    ap.add_argument("--write-golden", nargs="?", const="", default=None, metavar="PATH",
                    help="also write the pin kit's golden file (default path: tests/golden/opencv_<version>_<isa>.npz): inputs + this "
                         "OpenCV's outputs, consumed by tests/test_opencv_golden_{cpu,gpu}.py (tests/opencv_golden.py)")
    args = ap.parse_args()
    try:
        import cv2
    except Exception as e:  # pragma: no cover - depends on the machine
        print(f"cv2 is not importable here ({e}); nothing checked.  The oracle stays 'parity unpinned'.")
        return 2
    import oracle
    from polychase_amd import synth

    if args.write_golden is not None:
        import opencv_golden as og
        backend = og.Cv2Backend()
        path = args.write_golden or os.path.join(og.GOLDEN_DIR, f"opencv_{backend.tag}.npz")
        arith = og.write(path, backend)
        print(f"wrote {path} ({os.path.getsize(path) / 1e6:.1f} MB): {backend.source}; the oracle reproduces it bit for bit in mode "
              f"'{arith}'.  Commit it: tests/test_opencv_golden_cpu.py and tests/test_opencv_golden_gpu.py pin oracle and HIP path to it.")

    # the comparisons below name their execution explicitly (the oracle's DEFAULT is the x86 one since round 4)
    oracle.lib().pco_set_opencv_emulation(oracle.EMU_CANONICAL)
    w, h = args.width, args.height
    clip = synth.NoiseClip(w, h, 30)
    rgb0, rgb1 = clip.frame(10), clip.frame(12)
    bad = 0

    def report(name, ok, detail=""):
        nonlocal bad
        print(f"{'ok  ' if ok else 'FAIL'} {name} {detail}")
        bad += 0 if ok else 1

    # A.1 gray
    g0, g1 = oracle.rgb2gray(rgb0), oracle.rgb2gray(rgb1)
    c0 = cv2.cvtColor(rgb0, cv2.COLOR_RGB2GRAY)
    report("cvtColor RGB2GRAY", np.array_equal(g0, c0), f"max |diff| {np.abs(g0.astype(int) - c0.astype(int)).max()}")

    # A.2 min-eigenvalue map
    e_o = oracle.min_eigen_val(g0, 3, 3)
    e_c = cv2.cornerMinEigenVal(c0, 3, ksize=3)
    same = np.array_equal(e_o.view(np.uint32), e_c.view(np.uint32))
    rel = np.abs(e_o - e_c).max() / max(float(np.abs(e_c).max()), 1e-30)
    with oracle.emulation(oracle.EMU_SOBEL_FMA):
        e_f = oracle.min_eigen_val(g0, 3, 3)
    same_f = np.array_equal(e_f.view(np.uint32), e_c.view(np.uint32))
    # the same bound as tests/test_oracle_emulation_cpu.py::test_gftt_gap_to_avx2_sobel
    report("cornerMinEigenVal", same or same_f or rel < 1e-6,
           f"bit-exact vs canonical={same}, vs AVX2-FMA emulation={same_f}, max |diff| / max = {rel:.2e}")
    k_o = oracle.gftt(g0)
    # the detector's unused branch (gftt.cc:31-33): calcHarris' scalar expression everywhere (canonical) / its float vector
    # loop over the first w / 4 * 4 columns (the x86 execution, under EMU_SOBEL_FMA).  g0 is cropped to a width that is not a
    # multiple of 4 so that both loops are exercised
    gh, ch = np.ascontiguousarray(g0[:, :g0.shape[1] - 3]), np.ascontiguousarray(c0[:, :c0.shape[1] - 3])
    h_c = cv2.cornerHarris(ch, 3, 3, 0.04)
    h_same = {}
    for name, flags in (("canonical", 0), ("sobel_fma", oracle.EMU_SOBEL_FMA)):
        with oracle.emulation(flags):
            h_same[name] = bool(np.array_equal(oracle.corner_harris(gh, 3, 3, 0.04).view(np.uint32), h_c.view(np.uint32)))
    h_rel = np.abs(oracle.corner_harris(gh, 3, 3, 0.04) - h_c).max() / max(float(np.abs(h_c).max()), 1e-30)
    report("cornerHarris", any(h_same.values()) or h_rel < 1e-5, f"bit-exact vs canonical={h_same['canonical']}, vs x86 emulation={h_same['sobel_fma']}, "
                                                                    f"max |diff| / max = {h_rel:.2e}")

    # A.3 pyramid with derivatives
    win, max_level = 10, 3
    p_o = oracle.Pyramid(g0, win, max_level)
    n_lv, pyr = cv2.buildOpticalFlowPyramid(c0, (win, win), max_level)   # withDerivatives=True, REFLECT_101 / CONSTANT
    report("pyramid level count", p_o.num_levels == n_lv + 1, f"oracle {p_o.num_levels}, OpenCV maxLevel {n_lv}")
    for l in range(min(p_o.num_levels, n_lv + 1)):
        img_c, der_c = pyr[2 * l], pyr[2 * l + 1]
        report(f"pyramid image level {l}", np.array_equal(p_o.image(l, padded=False), img_c))
        d_o = p_o.deriv(l, padded=False)            # (H, W, 2) int16: dx, dy
        report(f"Scharr plane level {l}", np.array_equal(d_o, der_c.reshape(d_o.shape)))

    # A.4 LK
    kps = oracle.gftt(g0)
    p1_o = oracle.Pyramid(g1, win, max_level)
    xy_o, st_o, err_o = oracle.lk(p_o, p1_o, kps)
    crit = (cv2.TERM_CRITERIA_COUNT + cv2.TERM_CRITERIA_EPS, 30, 0.01)
    xy_c, st_c, err_c = cv2.calcOpticalFlowPyrLK(c0, cv2.cvtColor(rgb1, cv2.COLOR_RGB2GRAY), kps.reshape(-1, 1, 2), None,
                                                 winSize=(win, win), maxLevel=max_level, criteria=crit, flags=0,
                                                 minEigThreshold=1e-4)
    xy_c, st_c, err_c = xy_c.reshape(-1, 2), st_c.reshape(-1), err_c.reshape(-1)
    report("LK status", np.array_equal(st_o, st_c), f"{int((st_o != st_c).sum())} of {len(st_o)} differ")
    m = (st_o == 1) & (st_c == 1)
    d = np.abs(xy_o[m] - xy_c[m]).max() if m.any() else 0.0
    exact = np.array_equal(xy_o[m].view(np.uint32), xy_c[m].view(np.uint32))
    report("LK positions", d <= 1e-3, f"bit-exact={exact} max |diff| {d:.2e} px over {int(m.sum())} tracks")
    de = np.abs(err_o[m] - err_c[m]).max() if m.any() else 0.0
    report("LK error", de <= 1e-4 * max(1.0, float(np.abs(err_c[m]).max()) if m.any() else 1.0), f"max |diff| {de:.2e}")
    with oracle.emulation(oracle.EMU_LK_SIMD):
        xy_e, st_e, err_e = oracle.lk(p_o, p1_o, kps)
    me = (st_e == 1) & (st_c == 1)
    exact_e = np.array_equal(st_e, st_c) and np.array_equal(xy_e[me].view(np.uint32), xy_c[me].view(np.uint32))
    d_e = np.abs(xy_e[me] - xy_c[me]).max() if me.any() else 0.0
    report("LK positions, oracle in the x86 SIMD order", d_e <= 1e-3, f"bit-exact={exact_e} max |diff| {d_e:.2e} px "
           "(bit-exact expected on an SSE2/SSE3-baseline x86 build of OpenCV)")
    # the sharp-edged C1 clip is where the two orders differ at all (tests/test_oracle_emulation_cpu.py)
    cb = synth.checkerboard_clip(12)
    gb0, gb1 = oracle.rgb2gray(cb[10]), oracle.rgb2gray(cb[11])
    kb = oracle.gftt(gb0)
    xb_c, sb_c, _ = cv2.calcOpticalFlowPyrLK(gb0, gb1, kb.reshape(-1, 1, 2), None, winSize=(win, win), maxLevel=max_level,
                                             criteria=crit, flags=0, minEigThreshold=1e-4)
    xb_c, sb_c = xb_c.reshape(-1, 2), sb_c.reshape(-1)
    xb_o, sb_o, _ = oracle.lk(oracle.Pyramid(gb0, win, max_level), oracle.Pyramid(gb1, win, max_level), kb)
    mb = (sb_o == 1) & (sb_c == 1)
    db = np.abs(xb_o[mb] - xb_c[mb]).max(axis=1) if mb.any() else np.zeros(1)
    report("LK on the checkerboard (C1), canonical oracle", np.array_equal(sb_o, sb_c) and db.max() <= 5e-3 and (db > 1e-3).mean() <= 0.02,
           f"status equal={np.array_equal(sb_o, sb_c)}, max |diff| {db.max():.2e} px, {(db > 1e-3).sum()} of {len(db)} above 1e-3 px")
    # ---- which arithmetic mode of the GPU library reproduces THIS OpenCV build bit for bit (pc_context_set_arithmetic,
    # POLYCHASE_ARITH): keypoint order at 1920 x 1080 (where the FMA of the AVX2 Sobel swaps neighbours of the list) and the
    # LK vectors on the checkerboard (where the fp32 lane sums of the SSE path round)
    big = oracle.rgb2gray(synth.NoiseClip(1920, 1080, 4).frame(2))
    e_big = cv2.cornerMinEigenVal(big, 3, ksize=3)
    verdict = {}
    for name, flags in (("canonical", 0), ("sobel_fma", oracle.EMU_SOBEL_FMA), ("sobel_fma_rows", oracle.EMU_SOBEL_FMA | oracle.EMU_SOBEL_ROW_FMA)):
        with oracle.emulation(flags):
            verdict[name] = bool(np.array_equal(oracle.min_eigen_val(big, 3, 3).view(np.uint32), e_big.view(np.uint32)))
    with oracle.emulation(oracle.EMU_LK_SIMD):
        xb_e, sb_e, _ = oracle.lk(oracle.Pyramid(gb0, win, max_level), oracle.Pyramid(gb1, win, max_level), kb)
    lk_modes = {"canonical": bool(np.array_equal(sb_o, sb_c) and np.array_equal(xb_o[mb].view(np.uint32), xb_c[mb].view(np.uint32))),
                "lk_x86": bool(np.array_equal(sb_e, sb_c) and np.array_equal(xb_e[sb_c == 1].view(np.uint32), xb_c[sb_c == 1].view(np.uint32)))}
    gftt_mode = next((m for m in ("sobel_fma", "sobel_fma_rows", "canonical") if verdict[m]), None)
    lk_mode = "lk_x86" if lk_modes["lk_x86"] else ("canonical" if lk_modes["canonical"] else None)
    print(f"min-eig map at 1920x1080 bit-exact: canonical={verdict['canonical']} sobel_fma={verdict['sobel_fma']} sobel_fma_rows={verdict['sobel_fma_rows']};  "
          f"LK on the checkerboard bit-exact: canonical={lk_modes['canonical']} lk_x86={lk_modes['lk_x86']}")
    if gftt_mode and lk_mode:
        arith = {("canonical", "canonical"): "canonical", ("sobel_fma", "canonical"): "sobel_fma",
                 ("canonical", "lk_x86"): "lk_x86", ("sobel_fma", "lk_x86"): "opencv_x86",
                 ("sobel_fma_rows", "canonical"): "sobel_fma_rows", ("sobel_fma_rows", "lk_x86"): "opencv_x86_rows"}[(gftt_mode, lk_mode)]
        print(f"==> this OpenCV build is reproduced bit for bit by POLYCHASE_ARITH={arith} (DESIGN.md section 2)")
    else:
        report("an arithmetic mode that reproduces this build", False, "neither mode is bit-exact: the restatement needs another look")
    print("all within tolerance: the oracle is pinned on this machine" if bad == 0 else f"{bad} mismatch(es)")
    return 0 if bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
