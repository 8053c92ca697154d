"""GPU: randomized parity sweep -- image content, image size, every GFTT and LK option, target count --
HIP path against the CPU oracle, bit for bit.  Seeds are fixed, so a failure names a reproducible case."""
import numpy as np
import pytest

import oracle
from polychase_amd import hip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = hip.Context(0)
    yield c
    c.close()


def _image(rng, w, h, kind):
    if kind == "noise":                                   # white noise: corners everywhere, many equal responses
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if kind == "blocks":                                  # piecewise constant: ties, plateaus, exact zeros of the gradient
        cell = int(rng.integers(3, 17))
        small = rng.integers(0, 256, (h // cell + 2, w // cell + 2, 3), dtype=np.uint8)
        return np.ascontiguousarray(np.kron(small, np.ones((cell, cell, 1), np.uint8))[:h, :w])
    if kind == "smooth":                                  # low-contrast gradients + a few blobs
        y, x = np.mgrid[0:h, 0:w].astype(np.float32)
        img = 128 + 60 * np.sin(x / rng.uniform(5, 40)) * np.cos(y / rng.uniform(5, 40))
        for _ in range(12):
            cx, cy, r = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(2, 9)
            img += rng.uniform(-80, 80) * np.exp(-((x - cx) ** 2 + (y - cy) ** 2) / (2 * r * r))
        return np.repeat(np.clip(img, 0, 255).astype(np.uint8)[:, :, None], 3, axis=2)
    raise ValueError(kind)


def _shifted(rng, img, max_shift):
    """a target frame: the source rolled by a few pixels plus noise (LK must converge, diverge or leave the image)"""
    dx, dy = rng.integers(-max_shift, max_shift + 1, 2)
    out = np.roll(img, (int(dy), int(dx)), axis=(0, 1)).astype(np.int16)
    out += rng.integers(-6, 7, out.shape, dtype=np.int16)
    return np.clip(out, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("seed", range(96))
def test_random_configuration(ctx, seed):
    rng = np.random.default_rng(1000 + seed)
    w, h = int(rng.integers(24, 260)), int(rng.integers(24, 200))
    kind = ["noise", "blocks", "smooth"][seed % 3]
    win = int(rng.integers(3, 32))
    max_level = int(rng.integers(0, 6))
    gk = dict(quality_level=float(rng.choice([0.001, 0.01, 0.05, 0.3])), min_distance=float(rng.choice([0.0, 1.0, 2.5, 5.0, 11.0, 70.0])),
              max_corners=int(rng.choice([0, 0, 7, 300])), grid_rows=int(rng.integers(1, 7)), grid_cols=int(rng.integers(1, 7)))
    fk = dict(window_size=win, max_level=max_level, term_max_iters=int(rng.choice([1, 3, 30, 60])),
              term_epsilon=float(rng.choice([0.0, 0.001, 0.01, 0.3])), min_eigen_threshold=float(rng.choice([1e-6, 1e-4, 1e-2])))
    n_targets = int(rng.integers(1, 9))
    src = _image(rng, w, h, kind)
    tgts = [_shifted(rng, src, int(rng.integers(0, 7))) for _ in range(n_targets)]
    case = f"seed {seed}: {w}x{h} {kind} win {win} L {max_level} gftt {gk} lk {fk} targets {n_targets}"

    f1 = hip.Frame(ctx, w, h, win, max_level)
    f1.set_rgb(src)
    g1 = oracle.rgb2gray(src)
    assert np.array_equal(f1.gray(), g1), case
    f1.detect(hip.gftt_options(**gk))
    kps_o, _, ncand_o = oracle.gftt(g1, oracle.gftt_options(**gk), want_eig=True)
    assert np.array_equal(f1.min_eig().view(np.uint32), oracle.min_eigen_val(g1).view(np.uint32)), case
    assert f1.num_candidates == ncand_o, case
    assert np.array_equal(f1.keypoints(), kps_o), case

    p1 = oracle.Pyramid(g1, win, max_level)
    assert f1.num_levels == p1.num_levels, case
    for l in range(p1.num_levels):
        assert np.array_equal(f1.level(l), p1.image(l)) and np.array_equal(f1.deriv(l), p1.deriv(l)), case
    frames = []
    for t in tgts:
        f = hip.Frame(ctx, w, h, win, max_level)
        f.set_rgb(t)
        frames.append(f)
    if len(kps_o) == 0:   # also track a few arbitrary points, some of them at the border
        kps_o = np.floor(rng.uniform([0, 0], [w, h], (9, 2))).astype(np.float32)
        f1.set_keypoints(kps_o)
    xy, st, err = hip.lk_track(ctx, f1, frames, hip.flow_options(**fk))
    oopt = oracle.flow_options(**fk)
    for k, t in enumerate(tgts):
        oxy, ost, oerr = oracle.lk(p1, oracle.Pyramid(oracle.rgb2gray(t), win, max_level), kps_o, oopt)
        assert np.array_equal(st[k], ost), f"{case}: status of target {k}: {(st[k] != ost).sum()} differ"
        m = ost == 1
        assert np.array_equal(xy[k][m].view(np.uint32), oxy[m].view(np.uint32)), f"{case}: positions of target {k}"
        assert np.array_equal(err[k][m].view(np.uint32), oerr[m].view(np.uint32)), f"{case}: errors of target {k}"
    for f in [f1] + frames:
        f.close()


@pytest.mark.parametrize("seed", range(16))
def test_random_clip_through_the_analyzer(ctx, seed):
    """Whole clips through the pipelined engine (three streams, ring of resident frames, pre-ordering) with random
    length, first frame id, options and job depth: every record equals the reference-shaped CPU path."""
    from polychase_amd.pipeline import ClipAnalyzer
    rng = np.random.default_rng(5000 + seed)
    w, h = int(rng.integers(48, 200)), int(rng.integers(48, 160))
    n, first = int(rng.integers(1, 27)), int(rng.integers(-3, 40))
    kind = ["noise", "blocks", "smooth"][seed % 3]
    win, max_level = int(rng.integers(3, 26)), int(rng.integers(0, 4))
    gk = dict(quality_level=float(rng.choice([0.01, 0.05])), min_distance=float(rng.choice([0.0, 3.0, 5.0])),
              max_corners=int(rng.choice([0, 40])), grid_rows=int(rng.integers(1, 5)), grid_cols=int(rng.integers(1, 5)))
    fk = dict(window_size=win, max_level=max_level, term_max_iters=int(rng.choice([5, 30])))
    base = _image(rng, w + 64, h + 64, kind)
    frames = []
    for t in range(n):                                  # a drifting crop of one texture, plus noise
        ox, oy = 32 + int(round(1.3 * t)) % 30, 32 - int(round(0.7 * t)) % 30
        fr = base[oy:oy + h, ox:ox + w].astype(np.int16) + rng.integers(-3, 4, (h, w, 3), dtype=np.int16)
        frames.append(np.ascontiguousarray(np.clip(fr, 0, 255).astype(np.uint8)))
    kps_o, flows_o = oracle.analyze_clip(frames, first_frame=first, gopt=oracle.gftt_options(**gk), fopt=oracle.flow_options(**fk),
                                         threads=2)
    got_kps, got_flows = {}, {}

    def sink(f1, kps, detected, flows):
        got_kps[f1] = kps
        for f2, rec in flows.items():
            got_flows[(f1, f2)] = rec

    an = ClipAnalyzer(ctx, w, h, first, n, lambda fid: frames[fid - first], hip.gftt_options(**gk), hip.flow_options(**fk),
                      max_jobs=int(rng.integers(1, 5)))
    an.run(range(first, first + n), sink)
    an.close()
    case = f"seed {seed}: {w}x{h} {kind} n {n} first {first} win {win} L {max_level} {gk} {fk}"
    assert sorted(got_kps) == sorted(kps_o), case
    for f in kps_o:
        assert np.array_equal(got_kps[f], kps_o[f]), f"{case}: keypoints of frame {f}"
    assert sorted(got_flows) == sorted(flows_o), case
    for key in flows_o:
        for a, b in zip(got_flows[key], flows_o[key]):
            assert a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"{case}: flow {key}"


@pytest.mark.parametrize("seed", list(range(48)) + [561, 3129])   # the last two: found by the soak (SOBEL_FMA box-sum order at a border pixel)
def test_random_detector_branch_and_arithmetic_mode(ctx, seed):
    """The dimensions round 3 added: cornerHarris / any block_size (gftt.cc:31-36) and the arithmetic modes of the context
    (pc_context_set_arithmetic: x86 LK summation order, FMA in the Sobel column filter) -- response map, keypoints in value
    and order, LK positions / status / error against the oracle under the matching emulation, bit for bit."""
    rng = np.random.default_rng(9000 + seed)
    w, h = int(rng.integers(40, 300)), int(rng.integers(40, 220))
    kind = ["noise", "blocks", "smooth"][seed % 3]
    win, max_level = int(rng.integers(3, 32)), int(rng.integers(0, 4))
    block = int(rng.choice([1, 2, 3, 3, 4, 5, 7, 13, 34]))
    harris = bool(rng.random() < 0.35)
    arith = int(rng.choice([hip.ARITH_CANONICAL, hip.ARITH_LK_X86_ORDER, hip.ARITH_SOBEL_FMA, hip.ARITH_OPENCV_X86,
                            hip.ARITH_SOBEL_FMA | hip.ARITH_SOBEL_ROW_FMA, hip.ARITH_OPENCV_X86 | hip.ARITH_SOBEL_ROW_FMA, hip.ARITH_SOBEL_ROW_FMA]))
    gk = dict(quality_level=float(rng.choice([0.01, 0.05, 0.3])), min_distance=float(rng.choice([0.0, 2.5, 5.0])), block_size=block,
              use_harris=int(harris), harris_k=float(rng.choice([0.04, 0.06, 0.15])), grid_rows=int(rng.integers(1, 5)),
              grid_cols=int(rng.integers(1, 5)))
    # round 5: the aperture of the derivative (gftt.cc:31-36: Sobel 3 / 5 / 7, Scharr = -1), drawn from a generator of its own so
    # that the other dimensions of a seed stay what they were
    ksize = int(np.random.default_rng(77000 + seed).choice([3, 3, 5, 7, -1]))
    gk["gradient_size"] = ksize
    fk = dict(window_size=win, max_level=max_level, term_max_iters=int(rng.choice([3, 30])))
    n_targets = int(rng.integers(1, 5))
    src = _image(rng, w, h, kind)
    if seed % 4 == 0:                          # hard edges at full contrast: where the fp32 lane sums of the x86 order round
        src = np.where(src > 127, 255, 0).astype(np.uint8)
    tgts = [_shifted(rng, src, int(rng.integers(0, 5))) for _ in range(n_targets)]
    case = f"seed {seed}: {w}x{h} {kind} win {win} L {max_level} gftt {gk} lk {fk} targets {n_targets} arith {arith}"
    emu = (oracle.EMU_LK_SIMD if arith & hip.ARITH_LK_X86_ORDER else 0) | (oracle.EMU_SOBEL_FMA if arith & hip.ARITH_SOBEL_FMA else 0) | \
          (oracle.EMU_SOBEL_ROW_FMA if arith & hip.ARITH_SOBEL_ROW_FMA else 0)
    before = ctx.arithmetic
    ctx.set_arithmetic(arith)
    try:
        f1 = hip.Frame(ctx, w, h, win, max_level)
        f1.set_rgb(src)
        g1 = oracle.rgb2gray(src)
        f1.detect(hip.gftt_options(**gk))
        with oracle.emulation(emu):
            want = oracle.corner_harris(g1, block, ksize, gk["harris_k"]) if harris else oracle.min_eigen_val(g1, block, ksize)
            kps_o = oracle.gftt(g1, oracle.gftt_options(**gk))
        assert np.array_equal(f1.min_eig().view(np.uint32), want.view(np.uint32)), case
        assert np.array_equal(f1.keypoints(), kps_o), case
        frames = []
        for t in tgts:
            f = hip.Frame(ctx, w, h, win, max_level)
            f.set_rgb(t)
            frames.append(f)
        if len(kps_o) == 0:
            kps_o = np.floor(rng.uniform([0, 0], [w, h], (9, 2))).astype(np.float32)
            f1.set_keypoints(kps_o)
        xy, st, err = hip.lk_track(ctx, f1, frames, hip.flow_options(**fk))
        p1 = oracle.Pyramid(g1, win, max_level)
        with oracle.emulation(emu):
            for k, t in enumerate(tgts):
                oxy, ost, oerr = oracle.lk(p1, oracle.Pyramid(oracle.rgb2gray(t), win, max_level), kps_o, oracle.flow_options(**fk))
                assert np.array_equal(st[k], ost), f"{case}: status of target {k}: {(st[k] != ost).sum()} differ"
                m = ost == 1
                assert np.array_equal(xy[k][m].view(np.uint32), oxy[m].view(np.uint32)), f"{case}: positions of target {k}"
                assert np.array_equal(err[k][m].view(np.uint32), oerr[m].view(np.uint32)), f"{case}: errors of target {k}"
        for f in [f1] + frames:
            f.close()
    finally:
        ctx.set_arithmetic(before)
