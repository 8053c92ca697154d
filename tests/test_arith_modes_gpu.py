"""GPU: the arithmetic modes of the C ABI (pc_context_set_arithmetic) against the oracle's emulation of an x86 OpenCV build
(oracle/pc_oracle.c: pco_set_opencv_emulation).

Where OpenCV's result depends on how the host executes it there are two executions on record: the canonical one (no FMA,
LK sums exact in integers) and the x86 one (fp32 lane sums of LKTrackerInvoker's CV_SIMD128 path, fused multiply-add in
the AVX2 column filter of Sobel).  Each GPU mode must be BIT-EXACT with the oracle in the same mode -- keypoints in value
and ORDER (the order is the keypoint index the database stores, cpp/feature_detection/gftt.cc:7-12, :98), LK vectors,
status and error (cv::calcOpticalFlowPyrLK at cpp/opticalflow.cc:119-125) -- on content where the two executions differ
(C1's step edges) and on the benchmark's content."""
import numpy as np
import pytest

import oracle
from polychase_amd import hip, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = hip.Context(0)
    assert c.arithmetic == hip.ARITH_OPENCV_X86, "the library's default is the x86 execution"
    yield c
    c.close()


def _frames(ctx, rgbs, win=10, max_level=3):
    out = []
    for rgb in rgbs:
        h, w = rgb.shape[:2]
        f = hip.Frame(ctx, w, h, win, max_level)
        f.set_rgb(rgb)
        out.append(f)
    return out


def _lk_both(ctx, rgbs, flags, emu, win=10, max_level=3):
    """GPU in mode `flags` vs oracle under emulation `emu`; returns (gpu xy, oracle xy of the canonical order, status)."""
    grays = [oracle.rgb2gray(r) for r in rgbs]
    ctx.set_arithmetic(flags)
    fr = _frames(ctx, rgbs, win, max_level)
    fr[0].detect()
    kps = fr[0].keypoints()
    opt = hip.flow_options(window_size=win, max_level=max_level)
    xy, st, err = hip.lk_track(ctx, fr[0], fr[1:], opt)
    oopt = oracle.flow_options(window_size=win, max_level=max_level)
    p0 = oracle.Pyramid(grays[0], win, max_level)
    with oracle.emulation(emu):
        okps = oracle.gftt(grays[0])
        assert np.array_equal(kps, okps), "keypoints (value and order) differ from the oracle in this mode"
        res = [oracle.lk(p0, oracle.Pyramid(g, win, max_level), kps, oopt) for g in grays[1:]]
    for k, (oxy, ost, oerr) in enumerate(res):
        assert np.array_equal(st[k], ost), f"target {k}: {(st[k] != ost).sum()} status flips"
        m = ost == 1
        assert np.array_equal(xy[k][m].view(np.uint32), oxy[m].view(np.uint32)), f"target {k}: LK vectors are not bit-exact in mode {flags}"
        assert np.array_equal(err[k][m].view(np.uint32), oerr[m].view(np.uint32))
    for f in fr:
        f.close()
    return xy, st, kps


def test_x86_lk_order_is_bit_exact_on_step_edges_and_differs_from_canonical(ctx):
    """C1's checkerboard: window sums exceed 2^24, the fp32 lane sums round -- the content where the modes part."""
    frames = synth.checkerboard_clip(20)
    rgbs = [frames[10]] + [frames[t] for t in (2, 6, 8, 9, 11, 12, 14, 18)]
    x86, st_x, kps_x = _lk_both(ctx, rgbs, hip.ARITH_LK_X86_ORDER, oracle.EMU_LK_SIMD)
    can, st_c, kps_c = _lk_both(ctx, rgbs, hip.ARITH_CANONICAL, 0)
    assert np.array_equal(kps_x, kps_c) and np.array_equal(st_x, st_c)
    m = st_c == 1
    d = np.abs(x86 - can)[m]
    assert d.max() > 0, "the x86 order changed nothing: the mode is not exercised"
    assert d.max() < 5e-3          # the two executions of OpenCV stay within 2e-3 px of each other here


@pytest.mark.parametrize("win,max_level,n_targets", [(10, 3, 8), (7, 2, 2), (13, 3, 3), (16, 1, 1), (4, 2, 2), (8, 3, 5), (5, 2, 3), (6, 3, 8),
                                                      (9, 2, 4), (11, 3, 7), (3, 1, 2), (12, 2, 1), (14, 2, 3), (15, 3, 8), (17, 2, 2), (20, 1, 3),
                                                      (21, 3, 8), (24, 2, 2), (27, 1, 1), (29, 2, 3), (31, 2, 4)])
def test_x86_lk_order_every_window_geometry(ctx, win, max_level, n_targets):
    """(win / 8) * 8 vector columns + scalar rest: 0 + 3 ... 0 + 7, 8 + 0 ... 8 + 5, 16 + 0 columns; windows 4-11 run on the
    two-keypoint kernel (kernels_lk3.hip, X86 = true), 3 and 12-31 on the eight-lanes-per-target kernel (lk4_kernel.hpp: up to
    three 8-column blocks of vector lanes + up to 7 scalar columns)"""
    frames = synth.checkerboard_clip(16, w=320, h=240)
    _lk_both(ctx, [frames[6]] + [frames[6 + k + 1] for k in range(n_targets)], hip.ARITH_LK_X86_ORDER, oracle.EMU_LK_SIMD, win, max_level)
    clip = synth.NoiseClip(320, 240, 12)
    _lk_both(ctx, [clip.frame(4)] + [clip.frame(5 + k % 6) for k in range(n_targets)], hip.ARITH_LK_X86_ORDER, oracle.EMU_LK_SIMD, win, max_level)


def test_sobel_fma_mode_min_eig_map_and_keypoint_order_at_1080p(ctx):
    """The AVX2 Sobel fuses one multiply-add: at 1920x1080 that swaps neighbours in the (value, address) order, i.e.
    keypoint INDICES -- the GPU mode must reproduce the emulated order exactly, map included."""
    clip = synth.NoiseClip(1920, 1080, 4)
    rgb = clip.frame(2)
    gray = oracle.rgb2gray(rgb)
    got = {}
    # (the third: the second hypothesis about the real build -- Dy's row smoothing fused as well, PC_ARITH_SOBEL_ROW_FMA)
    for flags, emu in ((hip.ARITH_SOBEL_FMA, oracle.EMU_SOBEL_FMA), (hip.ARITH_CANONICAL, 0),
                       (hip.ARITH_SOBEL_FMA | hip.ARITH_SOBEL_ROW_FMA, oracle.EMU_SOBEL_FMA | oracle.EMU_SOBEL_ROW_FMA)):
        ctx.set_arithmetic(flags)
        f = hip.Frame(ctx, 1920, 1080)
        f.set_rgb(rgb)
        f.detect()
        with oracle.emulation(emu):
            assert np.array_equal(f.min_eig().view(np.uint32), oracle.min_eigen_val(gray).view(np.uint32))
            okps = oracle.gftt(gray)
        got[flags] = f.keypoints()
        assert np.array_equal(got[flags], okps), f"mode {flags}: keypoint list differs from the oracle's"
        f.close()
    a, b = got[hip.ARITH_SOBEL_FMA], got[hip.ARITH_CANONICAL]
    assert len(a) == len(b) and set(map(tuple, a.astype(int))) == set(map(tuple, b.astype(int)))   # the same corners ...
    assert (a != b).any(), "... in another order at this size: otherwise the mode would not matter"


def test_opencv_x86_mode_through_the_analyzer(ctx):
    """both flags together, through pc_analyzer (the path the database comes from): records == oracle.analyze in the
    same emulation"""
    from polychase_amd.pipeline import ClipAnalyzer
    frames = synth.checkerboard_clip(12, w=320, h=240)
    ctx.set_arithmetic(hip.ARITH_OPENCV_X86)
    an = ClipAnalyzer(ctx, 320, 240, 1, 12, lambda fid: frames[fid - 1])
    got = {}
    an.run(range(1, 13), lambda f1, k, det, flows: got.__setitem__(f1, (k.copy(), {t: [a.copy() for a in v] for t, v in flows.items()})))
    an.close()
    grays = [oracle.rgb2gray(f) for f in frames]
    with oracle.emulation(oracle.EMU_LK_SIMD | oracle.EMU_SOBEL_FMA):
        for f1 in (1, 6, 12):
            okps = oracle.gftt(grays[f1 - 1])
            assert np.array_equal(got[f1][0], okps)
            p0 = oracle.Pyramid(grays[f1 - 1])
            for f2, (idx, xy, err) in got[f1][1].items():
                oxy, ost, oerr = oracle.lk(p0, oracle.Pyramid(grays[f2 - 1]), okps)
                keep = np.nonzero(ost == 1)[0].astype(np.uint32)
                assert np.array_equal(idx, keep) and np.array_equal(xy, oxy[keep]) and np.array_equal(err, oerr[keep])


def _hard(rng, w, h, kind):
    """content for the x86 order: white noise (huge window sums), binary blocks (step edges at full contrast), smooth"""
    if kind == "noise":
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if kind == "binary":
        cell = int(rng.integers(3, 12))
        small = rng.integers(0, 2, (h // cell + 2, w // cell + 2, 1), dtype=np.uint8) * 255
        return np.ascontiguousarray(np.repeat(np.kron(small, np.ones((cell, cell, 1), np.uint8))[:h, :w], 3, axis=2))
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    img = 128 + 50 * np.sin(x / rng.uniform(4, 20)) * np.cos(y / rng.uniform(4, 20)) + rng.uniform(-4, 4, (h, w))
    return np.repeat(np.clip(img, 0, 255).astype(np.uint8)[:, :, None], 3, axis=2)


@pytest.mark.parametrize("seed", range(106))
def test_x86_order_on_the_two_keypoint_kernel_random(ctx, seed):
    """Windows 4-11 in PC_ARITH_LK_X86_ORDER run the two-keypoint kernel, 3 and 12-31 the eight-lanes-per-target kernel (seeds from
    48 on: every window 3..31 in turn): the canonical integer data path plus a proof that the fp32 sums of the x86 order are exact,
    and the x86 order itself where the proof fails.  Random content of three kinds (proof always fails / fails at edges / always
    holds), every window, 1-8 targets, odd keypoint counts -- bit for bit against the oracle's emulation, and the diagnostics
    counters must add up."""
    rng = np.random.default_rng(77000 + seed)
    win = 4 + seed % 8 if seed < 48 else 3 + (seed - 48) % 29
    kind = ["noise", "binary", "smooth"][(seed // 8) % 3] if seed < 48 else ["noise", "binary", "smooth"][seed % 3]
    w, h = int(rng.integers(60, 260)), int(rng.integers(60, 200))
    max_level = int(rng.integers(0, 4))
    n_targets = int(rng.integers(1, 9))
    src = _hard(rng, w, h, kind)
    tgts = []
    moved = False
    for _ in range(n_targets):
        dx, dy = rng.integers(-4, 5, 2)
        moved = moved or bool(dx or dy)
        t = np.roll(src, (int(dy), int(dx)), axis=(0, 1)).astype(np.int16) + rng.integers(-5, 6, src.shape, dtype=np.int16)
        tgts.append(np.clip(t, 0, 255).astype(np.uint8))
    fk = dict(window_size=win, max_level=max_level, term_max_iters=int(rng.choice([3, 30])))
    case = f"seed {seed}: {w}x{h} {kind} win {win} L {max_level} targets {n_targets} {fk}"
    before = ctx.arithmetic
    ctx.set_arithmetic(hip.ARITH_LK_X86_ORDER)
    try:
        f1 = hip.Frame(ctx, w, h, win, max_level)
        f1.set_rgb(src)
        g1 = oracle.rgb2gray(src)
        f1.detect(hip.gftt_options(min_distance=float(rng.choice([2.0, 5.0]))))
        kps = f1.keypoints()
        if len(kps) == 0:
            kps = np.floor(rng.uniform([0, 0], [w, h], (9, 2))).astype(np.float32)
            f1.set_keypoints(kps)
        frames = []
        for t in tgts:
            f = hip.Frame(ctx, w, h, win, max_level)
            f.set_rgb(t)
            frames.append(f)
        ctx.lk_x86_stats(True)
        xy, st, err = hip.lk_track(ctx, f1, frames, hip.flow_options(**fk))
        stats = ctx.lk_x86_stats(False)
        p1 = oracle.Pyramid(g1, win, max_level)
        with oracle.emulation(oracle.EMU_LK_SIMD):
            for k, t in enumerate(tgts):
                oxy, ost, oerr = oracle.lk(p1, oracle.Pyramid(oracle.rgb2gray(t), win, max_level), kps, oracle.flow_options(**fk))
                assert np.array_equal(st[k], ost), f"{case}: status of target {k}: {(st[k] != ost).sum()} differ"
                m = ost == 1
                assert np.array_equal(xy[k][m].view(np.uint32), oxy[m].view(np.uint32)), f"{case}: positions of target {k} {stats}"
                assert np.array_equal(err[k][m].view(np.uint32), oerr[m].view(np.uint32)), f"{case}: errors of target {k}"
        assert stats["keypoint_levels"] == len(kps) * (min(max_level, f1.num_levels - 1) + 1), (case, stats)
        if kind == "noise" and win >= 8:      # window sums of white noise exceed 2^24 from 8 x 8 on: both ordered evaluations ran
            assert stats["keypoint_levels_x86_order"] > 0, (case, stats)
            # (a target that is the source plus +-5 grey levels, not moved, differs too little for the proof to fail:
            # soak seeds 1056340, 1060684, 1068510)
            assert stats["iterations_x86_order"] > 0 or not moved, (case, stats)
        for f in [f1] + frames:
            f.close()
    finally:
        ctx.set_arithmetic(before)


def test_x86_order_is_decided_by_the_proof_on_benchmark_content(ctx):
    """On the benchmark's band-limited texture the exactness proof holds nearly everywhere: the x86 mode runs at the
    canonical kernel's speed plus the proof.  (C1's step edges: the other extreme, see the counters there.)"""
    clip = synth.NoiseClip(960, 540, 12)
    rgbs = [clip.frame(4)] + [clip.frame(t) for t in (3, 5, 6, 8)]
    before = ctx.arithmetic
    ctx.set_arithmetic(hip.ARITH_LK_X86_ORDER)
    try:
        fr = _frames(ctx, rgbs)
        fr[0].detect()
        ctx.lk_x86_stats(True)
        hip.lk_track(ctx, fr[0], fr[1:], hip.flow_options())
        stats = ctx.lk_x86_stats(False)
        for f in fr:
            f.close()
    finally:
        ctx.set_arithmetic(before)
    total = stats["iterations_proven_exact"] + stats["iterations_x86_order"]
    print("x86 stats on C2 content:", stats)
    assert total > 0 and stats["iterations_x86_order"] <= 0.2 * total, stats


def test_hard_edged_clip_through_the_analyzer_in_the_default_mode():
    """Blurred step edges over a fine texture -- the content class on which a third of the LK iterations and nine tenths of the
    structure tensors take the ORDERED x86 evaluation (tools/x86_cost_probe.py) -- as a whole clip through pc_analyzer in the
    library's default arithmetic: every record equals the oracle's in its default emulation."""
    from polychase_amd.pipeline import ClipAnalyzer
    rng = np.random.default_rng(11)
    w, h, n = 384, 256, 14
    cells = rng.integers(0, 2, (h // 24 + 6, w // 24 + 6))
    tex_all = rng.integers(-20, 21, (h + 64, w + 64)).astype(np.float64)
    frames = []
    for t in range(n):
        big = 160.0 * np.kron(cells, np.ones((24, 24)))[t:t + h + 2, 2 * t:2 * t + w + 2]
        big = sum(big[dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)) / 9
        g = np.clip(40 + big + tex_all[t:t + h, 2 * t:2 * t + w], 0, 255).astype(np.uint8)
        frames.append(np.ascontiguousarray(np.repeat(g[:, :, None], 3, axis=2)))
    ctx = hip.Context(0)
    assert ctx.arithmetic == hip.ARITH_OPENCV_X86
    ctx.lk_x86_stats(True)
    an = ClipAnalyzer(ctx, w, h, 1, n, lambda fid: frames[fid - 1])
    got = {}
    an.run(range(1, n + 1), lambda f1, k, det, flows: got.__setitem__(f1, (k.copy(), {t: [a.copy() for a in v] for t, v in flows.items()})))
    an.close()
    stats = ctx.lk_x86_stats(False)
    ctx.close()
    assert stats["iterations_x86_order"] > 0.05 * stats["iterations_proven_exact"] and stats["keypoint_levels_x86_order"] > 0, stats
    kps_o, flows_o = oracle.analyze_clip(frames, first_frame=1, threads=4)
    assert sorted(got) == sorted(kps_o)
    for f1 in kps_o:
        assert np.array_equal(got[f1][0], kps_o[f1]), f"keypoints of frame {f1}"
        for f2, rec in got[f1][1].items():
            for a, b in zip(rec, flows_o[(f1, f2)]):
                assert a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"flow {f1} -> {f2}"
