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
    yield c
    c.set_arithmetic(hip.ARITH_CANONICAL)
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


@pytest.mark.parametrize("win,max_level,n_targets", [(10, 3, 8), (7, 2, 2), (13, 3, 3), (16, 1, 1), (4, 2, 2), (8, 3, 5)])
def test_x86_lk_order_every_window_geometry(ctx, win, max_level, n_targets):
    """(win / 8) * 8 vector columns + scalar rest: 0 + 4, 0 + 7, 8 + 0, 8 + 2, 8 + 5, 16 + 0 columns"""
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
    for flags, emu in ((hip.ARITH_SOBEL_FMA, oracle.EMU_SOBEL_FMA), (hip.ARITH_CANONICAL, 0)):
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
    ctx.set_arithmetic(hip.ARITH_CANONICAL)
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
