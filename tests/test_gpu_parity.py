"""GPU parity: every HIP stage, called through the C ABI, against the CPU oracle on the same inputs.

Integer stages (gray, pyramid, Scharr) and feature indices/counts: bit-exact.
Float stages: the min-eig map and the LK outputs are ALSO required to be bit-exact (both sides avoid
FMA contraction and accumulate LK sums exactly); the stated tolerance of the north star (1e-3 px) is
asserted separately so that a future relaxation cannot silently exceed it.
"""
import os
import numpy as np
import pytest

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from polychase_amd import hip, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = hip.Context(0)
    yield c
    c.close()


def _noise_frames(w, h, ts, n=30):
    clip = synth.NoiseClip(w, h, n)
    return clip, [clip.frame(t) for t in ts]


@pytest.mark.parametrize("w,h", [(640, 480), (641, 479), (37, 29), (1920, 1080)])
def test_gray_and_pyramid_bit_exact(ctx, w, h):
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    f = hip.Frame(ctx, w, h, 10, 4)
    f.set_rgb(rgb)
    g = oracle.rgb2gray(rgb)
    assert np.array_equal(f.gray(), g)
    p = oracle.Pyramid(g, 10, 4)
    assert f.num_levels == p.num_levels
    for l in range(p.num_levels):
        assert f.level_size(l) == p.level_size(l)
        assert np.array_equal(f.level(l), p.image(l)), f"image level {l}"
        assert np.array_equal(f.deriv(l), p.deriv(l)), f"deriv level {l}"
    f.close()


def test_device_resident_rgb(ctx):
    import torch
    rng = np.random.default_rng(2)
    rgb = rng.integers(0, 256, (360, 640, 3), dtype=np.uint8)
    f = hip.Frame(ctx, 640, 360)
    t = torch.from_numpy(rgb).cuda()
    f.set_rgb(t)
    assert np.array_equal(f.gray(), oracle.rgb2gray(rgb))
    f.close()


def _blender_floats(rgb_u8, channels, rng):
    """float32 pixels whose addon-side conversion `(x * 255).astype(np.uint8)` (analysis.py:232) gives rgb_u8 back"""
    x = (rgb_u8.astype(np.float32) + rng.uniform(0.05, 0.95, rgb_u8.shape).astype(np.float32)) / np.float32(255.0)
    if channels == 4:
        x = np.concatenate([x, rng.uniform(0, 1, rgb_u8.shape[:2] + (1,)).astype(np.float32)], axis=2)
    return np.ascontiguousarray(x)


@pytest.mark.parametrize("channels,on_device", [(3, False), (4, False), (4, True)])
def test_float_frames_convert_like_the_addon(ctx, channels, on_device):
    """Frame ingestion: Blender's float pixels go to the GPU as they are; the addon's numpy pass
    `(image_data * 255).astype(np.uint8)` (+ dropping alpha) happens in the RGB->gray kernel."""
    import torch
    rng = np.random.default_rng(7)
    w, h = 333, 211
    x = _blender_floats(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), channels, rng)
    x[0, 0, :3] = [1.0, 0.0, 0.999999]            # exactly white / black / just below white
    x[0, 1, :3] = [1.5, -0.1, 2.0]                # out of range: numpy keeps the low byte of the int32
    want_u8 = (x[:, :, :3] * 255).astype(np.uint8)
    assert want_u8[0, 0].tolist() == [255, 0, 254] and want_u8[0, 1].tolist() == [126, 231, 254]
    f = hip.Frame(ctx, w, h, 10, 3)
    f.set_rgb(torch.from_numpy(x).cuda() if on_device else x)
    assert np.array_equal(f.gray(), oracle.rgb2gray(want_u8))
    p = oracle.Pyramid(oracle.rgb2gray(want_u8), 10, 3)
    for l in range(p.num_levels):
        assert np.array_equal(f.level(l), p.image(l)) and np.array_equal(f.deriv(l), p.deriv(l))
    with pytest.raises(hip.PolychaseHipError):
        f.set_rgb(np.zeros((h, w, 2), np.float32))
    f.close()


@pytest.mark.parametrize("w,h,kind", [(640, 480, "checker"), (640, 360, "noise"), (333, 211, "noise"),
                                      (1920, 1080, "noise")])
def test_gftt_bit_exact(ctx, w, h, kind):
    if kind == "checker":
        rgb = synth.checkerboard_frame(3, w, h)
    else:
        _, (rgb,) = _noise_frames(w, h, [7])
    g = oracle.rgb2gray(rgb)
    f = hip.Frame(ctx, w, h)
    f.set_rgb(rgb)
    f.detect()
    eig = oracle.min_eigen_val(g)
    assert np.array_equal(f.min_eig().view(np.uint32), eig.view(np.uint32)), "min-eig map must be bit-exact"
    xy, _, ncand = oracle.gftt(g, want_eig=True)
    assert f.num_candidates == ncand
    assert f.num_keypoints == len(xy)
    assert np.array_equal(f.keypoints(), xy), "keypoints must match in value AND order"
    f.close()


def test_gftt_options(ctx):
    _, (rgb,) = _noise_frames(320, 240, [3])
    g = oracle.rgb2gray(rgb)
    f = hip.Frame(ctx, 320, 240)
    f.set_rgb(rgb)
    for kw in [dict(max_corners=50), dict(min_distance=0.0), dict(min_distance=9.5, quality_level=0.05),
               dict(grid_rows=1, grid_cols=1), dict(grid_rows=7, grid_cols=9)]:
        f.detect(hip.gftt_options(**kw))
        xy = oracle.gftt(g, oracle.gftt_options(**kw))
        assert np.array_equal(f.keypoints(), xy), kw
    for bad in (0, 1, 2, 4, 9, -2):                         # not an aperture cornerEigenValsVecs has
        with pytest.raises(hip.PolychaseHipError):
            f.detect(hip.gftt_options(gradient_size=bad))
    with pytest.raises(hip.PolychaseHipError):
        f.detect(hip.gftt_options(quality_level=0.0))
    f.close()


@pytest.mark.parametrize("kw", [dict(use_harris=1), dict(use_harris=1, harris_k=0.15), dict(block_size=5), dict(block_size=2),
                                dict(block_size=1), dict(block_size=7, use_harris=1), dict(block_size=4, min_distance=3.0, grid_rows=2),
                                dict(gradient_size=5), dict(gradient_size=7), dict(gradient_size=-1), dict(gradient_size=5, use_harris=1),
                                dict(gradient_size=7, block_size=2), dict(gradient_size=-1, block_size=5, use_harris=1, harris_k=0.08),
                                dict(block_size=12), dict(block_size=33), dict(block_size=40, use_harris=1, gradient_size=5)],
                         ids=lambda kw: ",".join(f"{k}={v}" for k, v in kw.items()))
def test_harris_and_other_block_sizes(ctx, kw):
    """The detector's other branches (reference cpp/feature_detection/gftt.cc:31-36: cornerHarris, any block_size, and since
    round 5 every gradient_size the bound GFTTOptions can carry -- Sobel 5 / 7, Scharr = -1; block sizes from 12 on take the
    two-pass box filter, any size runs), which the addon never selects:
    response map and keypoints -- value and order -- bit-exact against the oracle, on sizes that do not divide the kernels'
    tiles (odd, so that every border reflection of the 7-tap aperture is exercised); also through the analyzer."""
    for (w, h) in ((333, 211), (640, 360)):
        _, (rgb,) = _noise_frames(w, h, [3])
        g = oracle.rgb2gray(rgb)
        f = hip.Frame(ctx, w, h)
        f.set_rgb(rgb)
        f.detect(hip.gftt_options(**kw))
        bs, k, ks = kw.get("block_size", 3), kw.get("harris_k", 0.04), kw.get("gradient_size", 3)
        want = oracle.corner_harris(g, bs, ks, k) if kw.get("use_harris") else oracle.min_eigen_val(g, bs, ks)
        assert np.array_equal(f.min_eig().view(np.uint32), want.view(np.uint32)), "response map"
        xy = oracle.gftt(g, oracle.gftt_options(**kw))
        assert len(xy) > 10 and np.array_equal(f.keypoints(), xy), "keypoints must match in value AND order"
        f.close()
    from polychase_amd.pipeline import ClipAnalyzer
    clip = synth.NoiseClip(320, 240, 6)
    frames = {i + 1: clip.frame(i) for i in range(6)}
    an = ClipAnalyzer(ctx, 320, 240, 1, 6, lambda fid: frames[fid], hip.gftt_options(**kw))
    got = {}
    an.run(range(1, 7), lambda f1, kp, det, flows: got.__setitem__(f1, kp.copy()))
    an.close()
    for fid in (1, 4):
        assert np.array_equal(got[fid], oracle.gftt(oracle.rgb2gray(frames[fid]), oracle.gftt_options(**kw)))


@pytest.mark.parametrize("kw", [dict(min_distance=65.0), dict(min_distance=100.5, quality_level=0.001), dict(min_distance=300.0), dict(min_distance=2000.0),
                                dict(min_distance=64.0), dict(min_distance=80.0, max_corners=7), dict(grid_rows=20, grid_cols=30),
                                dict(grid_rows=32, grid_cols=32, min_distance=70.0)],
                         ids=lambda kw: ",".join(f"{k}={v}" for k, v in kw.items()))
def test_large_min_distance_and_fine_grids(ctx, kw):
    """GFTTOptions.min_distance is a free double in the reference (cpp/feature_detection/gftt.h:5-21): above 64 px the HIP path
    runs the reference's own greedy loop against a grid of accepted corners on one wavefront (kernels_gftt.hip:
    suppress_large_radius_kernel) instead of the table-driven parallel suppression -- keypoints in value AND order; a radius larger
    than the frame leaves one corner.  Grids up to 1024 cells."""
    for (w, h) in ((640, 360), (333, 211)):
        _, (rgb,) = _noise_frames(w, h, [3])
        g = oracle.rgb2gray(rgb)
        f = hip.Frame(ctx, w, h)
        f.set_rgb(rgb)
        f.detect(hip.gftt_options(**kw))
        xy = oracle.gftt(g, oracle.gftt_options(**kw))
        assert len(xy) >= 1 and np.array_equal(f.keypoints(), xy), "keypoints must match in value AND order"
        f.close()
    from polychase_amd.pipeline import ClipAnalyzer
    clip = synth.NoiseClip(320, 240, 5)
    frames = {i + 1: clip.frame(i) for i in range(5)}
    an = ClipAnalyzer(ctx, 320, 240, 1, 5, lambda fid: frames[fid], hip.gftt_options(**kw))
    got = {}
    an.run(range(1, 6), lambda f1, kp, det, flows: got.__setitem__(f1, kp.copy()))
    an.close()
    for fid in (1, 4):
        assert np.array_equal(got[fid], oracle.gftt(oracle.rgb2gray(frames[fid]), oracle.gftt_options(**kw)))


def test_flat_image_has_no_keypoints(ctx):
    f = hip.Frame(ctx, 128, 96)
    f.set_gray(np.full((96, 128), 77, np.uint8))
    f.detect()
    assert f.num_keypoints == 0
    tgt = hip.Frame(ctx, 128, 96)
    tgt.set_gray(np.full((96, 128), 77, np.uint8))
    res = hip.lk_track_filtered(ctx, f, [tgt])
    assert len(res) == 1 and len(res[0][0]) == 0
    f.close()
    tgt.close()


def _lk_case(ctx, w, h, t0, ts, max_level=3, win=10, n=30):
    clip, frames = _noise_frames(w, h, [t0] + ts, n)
    grays = [oracle.rgb2gray(fr) for fr in frames]
    f1 = hip.Frame(ctx, w, h, win, max_level)
    f1.set_rgb(frames[0])
    f1.detect()
    tg = []
    for fr in frames[1:]:
        f = hip.Frame(ctx, w, h, win, max_level)
        f.set_rgb(fr)
        tg.append(f)
    opt = hip.flow_options(window_size=win, max_level=max_level)
    xy, st, err = hip.lk_track(ctx, f1, tg, opt)
    kps = f1.keypoints()
    p1 = oracle.Pyramid(grays[0], win, max_level)
    oopt = oracle.flow_options(window_size=win, max_level=max_level)
    for k, g in enumerate(grays[1:]):
        oxy, ost, oerr = oracle.lk(p1, oracle.Pyramid(g, win, max_level), kps, oopt)
        assert np.array_equal(st[k], ost), f"status mismatch target {k}: {(st[k] != ost).sum()}"
        m = ost == 1
        assert np.abs(xy[k][m] - oxy[m]).max() <= 1e-3          # north-star tolerance
        assert np.array_equal(xy[k][m].view(np.uint32), oxy[m].view(np.uint32)), "LK must be bit-exact"
        assert np.array_equal(err[k][m].view(np.uint32), oerr[m].view(np.uint32))
    # filtered path == raw path filtered on the host (opticalflow.cc:130-147)
    flt = hip.lk_track_filtered(ctx, f1, tg, opt)
    for k in range(len(tg)):
        idx = np.nonzero(st[k] == 1)[0].astype(np.uint32)
        assert np.array_equal(flt[k][0], idx)
        assert np.array_equal(flt[k][1], xy[k][idx])
        assert np.array_equal(flt[k][2], err[k][idx])
    # analytic ground truth: median error small
    gt = clip.flow(kps, t0, ts[0])
    m = st[0] == 1
    assert np.median(np.abs(xy[0][m] - gt[m])) < 0.1
    for f in [f1] + tg:
        f.close()


def test_lk_bit_exact_small(ctx):
    _lk_case(ctx, 640, 360, 10, [11, 12, 14, 18, 9, 8, 6, 2])


def test_lk_bit_exact_odd_size_and_windows(ctx):
    _lk_case(ctx, 333, 211, 5, [6, 4], max_level=2, win=7)
    _lk_case(ctx, 333, 211, 5, [7], max_level=4, win=13)


@pytest.mark.parametrize("win,n_targets", [(3, 2), (4, 1), (5, 3), (6, 8), (8, 5), (9, 7), (11, 4), (12, 2), (13, 8), (14, 1), (15, 5), (16, 3),
                                           (17, 2), (18, 8), (19, 1), (20, 3), (21, 8), (22, 2), (23, 4), (24, 1), (25, 6), (26, 2), (27, 1),
                                           (28, 3), (29, 2), (30, 1), (31, 8)])
def test_lk_every_window_size_and_target_count(ctx, win, n_targets):
    """Windows 4..11 run the two-keypoints-per-wavefront kernel (chains + remainder columns differ per size), 3 and 12..31
    (PC_MAX_WINDOW; OpenCV's own default is 21) the one-keypoint, eight-lanes-per-target kernel (lk4_kernel.hpp: one to four
    column chains per lane, entries in registers up to 16 px and in LDS above); target counts below 8 leave groups idle."""
    ts = [11, 12, 14, 18, 9, 8, 6, 2][:n_targets]
    _lk_case(ctx, 320, 200, 10, ts, max_level=2, win=win)


def test_lk_odd_keypoint_count_and_tiny_sets(ctx):
    """An odd number of keypoints leaves the last wavefront's second half without a keypoint; 1 and 2 keypoints."""
    w, h = 320, 200
    clip, frames = _noise_frames(w, h, [4, 6, 7])
    grays = [oracle.rgb2gray(fr) for fr in frames]
    fr = [hip.Frame(ctx, w, h) for _ in frames]
    for f, im in zip(fr, frames):
        f.set_rgb(im)
    rng = np.random.default_rng(4)
    for n in (1, 2, 3, 77, 1001):
        pts = np.floor(rng.uniform([12, 12], [w - 12, h - 12], (n, 2))).astype(np.float32)
        fr[0].set_keypoints(pts)
        xy, st, err = hip.lk_track(ctx, fr[0], fr[1:])
        for k in range(2):
            oxy, ost, oerr = oracle.lk(oracle.Pyramid(grays[0]), oracle.Pyramid(grays[1 + k]), pts)
            assert np.array_equal(st[k], ost)
            m = ost == 1
            assert np.array_equal(xy[k][m].view(np.uint32), oxy[m].view(np.uint32))
            assert np.array_equal(err[k][m].view(np.uint32), oerr[m].view(np.uint32))
    for f in fr:
        f.close()


def test_lk_bit_exact_1080p(ctx):
    _lk_case(ctx, 1920, 1080, 15, [16, 23], n=30)


@pytest.mark.parametrize("win", [17, 21, 31])
def test_lk_large_windows_odd_size_and_1080p(ctx, win):
    """OpticalFlowOptions.window_size above 16 (cpp/opticalflow.h:27-33 accepts any; OpenCV's default is 21): bit-exact at an odd
    frame size (levels end early: the next level would be <= the window) and at 1920x1080"""
    _lk_case(ctx, 333, 211, 5, [6, 4, 9], max_level=3, win=win)
    _lk_case(ctx, 1920, 1080, 15, [16, 11], n=30, win=win)


def test_lk_border_features(ctx):
    """Keypoints within a window of the border exercise the REFLECT_101 / zero paddings and the
    out-of-range status paths."""
    w, h = 320, 200
    clip, frames = _noise_frames(w, h, [4, 12])
    grays = [oracle.rgb2gray(fr) for fr in frames]
    pts = np.array([[0, 0], [1, 1], [w - 1, h - 1], [w - 1, 0], [0, h - 1], [3, 100], [w - 2, 50], [160, 1],
                    [160, h - 1], [5, 5], [w - 6, h - 6]], np.float32)
    f1 = hip.Frame(ctx, w, h)
    f1.set_rgb(frames[0])
    f1.set_keypoints(pts)
    f2 = hip.Frame(ctx, w, h)
    f2.set_rgb(frames[1])
    xy, st, err = hip.lk_track(ctx, f1, [f2])
    oxy, ost, oerr = oracle.lk(oracle.Pyramid(grays[0]), oracle.Pyramid(grays[1]), pts)
    assert np.array_equal(st[0], ost)
    m = ost == 1
    assert np.array_equal(xy[0][m].view(np.uint32), oxy[m].view(np.uint32))
    assert np.array_equal(err[0][m].view(np.uint32), oerr[m].view(np.uint32))
    f1.close()
    f2.close()


def test_errors(ctx):
    with pytest.raises(hip.PolychaseHipError):
        hip.Frame(ctx, 64, 64, window_size=2)
    hip.Frame(ctx, 64, 64, window_size=17).close()            # windows up to PC_MAX_WINDOW = 31 run
    hip.Frame(ctx, 64, 64, window_size=31).close()
    with pytest.raises(hip.PolychaseHipError):
        hip.Frame(ctx, 64, 64, window_size=32)
    f = hip.Frame(ctx, 64, 64)
    g = hip.Frame(ctx, 64, 64)
    f.set_gray(np.zeros((64, 64), np.uint8))
    g.set_gray(np.zeros((64, 64), np.uint8))
    with pytest.raises(hip.PolychaseHipError):  # no keypoints yet
        hip.lk_track(ctx, f, [g])
    f.close()
    g.close()


def _compare_clip(kps_o, flows_o, got_kps, got_flows):
    assert sorted(got_kps) == sorted(kps_o)
    for f in kps_o:
        assert np.array_equal(got_kps[f], kps_o[f]), f"keypoints of frame {f}"
    assert sorted(got_flows) == sorted(flows_o)
    for key in flows_o:
        for a, b in zip(got_flows[key], flows_o[key]):
            assert a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"flow {key}"


def test_analyzer_whole_clip_matches_reference_shaped_cpu_path(ctx):
    """C1-like plumbing case: a whole short clip through the pipelined analyzer equals the
    reference-shaped CPU path (per-pair rebuild) record for record, byte for byte."""
    from polychase_amd.pipeline import ClipAnalyzer
    w, h, n, first = 320, 240, 20, 1
    clip = synth.NoiseClip(w, h, n)
    frames = [clip.frame(t) for t in range(n)]
    kps_o, flows_o = oracle.analyze_clip(frames, first_frame=first, threads=4)
    got_kps, got_flows = {}, {}

    def sink(f1, kps, detected, flows):
        assert detected
        got_kps[f1] = kps
        for f2, rec in flows.items():
            got_flows[(f1, f2)] = rec

    an = ClipAnalyzer(ctx, w, h, first, n, lambda fid: frames[fid - first])
    an.run(range(first, first + n), sink)
    an.close()
    assert len(got_flows) == 8 * n - 30
    _compare_clip(kps_o, flows_o, got_kps, got_flows)


def test_analyzer_checkerboard_c1(ctx):
    """BASELINE config C1: 640x480x30 translating checkerboard."""
    from polychase_amd.pipeline import ClipAnalyzer
    frames = synth.checkerboard_clip(30)
    kps_o, flows_o = oracle.analyze_clip(frames, first_frame=1, threads=4)
    got_kps, got_flows = {}, {}

    def sink(f1, kps, detected, flows):
        got_kps[f1] = kps
        for f2, rec in flows.items():
            got_flows[(f1, f2)] = rec

    an = ClipAnalyzer(ctx, 640, 480, 1, 30, lambda fid: frames[fid - 1])
    an.run(range(1, 31), sink)
    an.close()
    _compare_clip(kps_o, flows_o, got_kps, got_flows)
    # analytic truth: +1 frame => (+1.25, +0.75) px
    idx, xy, _ = got_flows[(10, 11)]
    d = xy - got_kps[10][idx]
    assert np.abs(np.median(d, axis=0) - [1.25, 0.75]).max() < 0.1


def test_analyzer_degenerate_clips(ctx):
    """Ragged inputs through the pipelined engine: a one-frame clip (no targets at all), a two-frame clip, flat frames
    (no keypoints, so empty flow records) between textured ones."""
    from polychase_amd.pipeline import ClipAnalyzer
    w, h = 192, 128
    clip, tex = _noise_frames(w, h, [3, 4, 5, 6], 12)
    flat = np.full((h, w, 3), 90, np.uint8)
    for frames in ([tex[0]], [tex[0], tex[1]], [tex[0], flat, tex[1], flat, flat, tex[2], tex[3]]):
        n = len(frames)
        kps_o, flows_o = oracle.analyze_clip(frames, first_frame=1, threads=2)
        got_kps, got_flows = {}, {}

        def sink(f1, kps, detected, flows):
            got_kps[f1] = kps
            for f2, rec in flows.items():
                got_flows[(f1, f2)] = rec

        an = ClipAnalyzer(ctx, w, h, 1, n, lambda fid: frames[fid - 1])
        an.run(range(1, n + 1), sink)
        an.close()
        assert sorted(got_kps) == list(range(1, n + 1))
        _compare_clip(kps_o, flows_o, got_kps, got_flows)
        if n == 7:
            assert len(got_kps[2]) == 0 and len(got_kps[4]) == 0 and len(got_kps[1]) > 0
            assert all(len(got_flows[(2, f2)][0]) == 0 for f2 in (1, 3, 4, 6))


def test_analyzer_resume_with_supplied_keypoints(ctx):
    from polychase_amd.pipeline import ClipAnalyzer
    w, h, n = 320, 240, 12
    clip = synth.NoiseClip(w, h, n)
    frames = [clip.frame(t) for t in range(n)]
    an = ClipAnalyzer(ctx, w, h, 1, n, lambda fid: frames[fid - 1])
    out = {}
    an.run([3], lambda f1, kps, det, flows: out.update(a=(kps, det, flows)))
    kps = out["a"][0]
    # second pass over the same frame with keypoints "read from the database": not re-detected,
    # only the requested pair
    an.an.set_keypoints(3, kps[::-1].copy())
    an.an.submit(3, [4])
    f1, kps2, det2, flows2 = an.an.collect()
    assert not det2 and np.array_equal(kps2, kps[::-1])
    idx_a, xy_a, err_a = out["a"][2][4]
    idx_b, xy_b, err_b = flows2[4]
    # same keypoints in reversed order -> same tracks, reversed
    m = len(kps) - 1 - idx_b[::-1]
    assert np.array_equal(m.astype(np.uint32), idx_a)
    assert np.array_equal(xy_b[::-1], xy_a)
    an.close()


def test_frame_sharding_is_result_invariant(ctx):
    """SURVEY 8(e) determinism: records do not depend on the number of ranks.  Two shards (each with
    its 8-frame halo) processed independently == one pass over the whole clip, byte for byte."""
    from polychase_amd import distributed as D
    from polychase_amd.pipeline import ClipAnalyzer
    w, h, n, first = 320, 240, 26, 3
    clip = synth.NoiseClip(w, h, n)
    frames = {first + t: clip.frame(t) for t in range(n)}

    def run(f1_begin, f1_end):
        out = []
        an = ClipAnalyzer(ctx, w, h, first, n, lambda fid: frames[fid])
        an.run(range(f1_begin, f1_end), lambda f1, kps, det, flows: out.append((f1, kps, flows)))
        an.close()
        return out

    whole = run(first, first + n)
    parts = []
    for r in range(2):
        b, e = D.shard_range(first, n, 2, r)
        parts += run(b, e)
    hw, pw = D.pack_records(whole)
    hp, pp = D.pack_records(parts)
    assert np.array_equal(hw, hp) and np.array_equal(pw, pp)


def test_device_log_matches_host_records(ctx):
    """The device-resident record log (multi-GPU stitch payload) parses to exactly the host records."""
    import torch
    from polychase_amd import distributed as D
    from polychase_amd.pipeline import ClipAnalyzer
    w, h, n = 320, 240, 14
    clip = synth.NoiseClip(w, h, n)
    frames = [clip.frame(t) for t in range(n)]
    an = ClipAnalyzer(ctx, w, h, 1, n, lambda fid: frames[fid - 1])
    log = torch.empty(D.log_capacity_bytes(n, 4000), dtype=torch.uint8, device="cuda")
    an.an.set_device_log(log)
    host = []
    an.run(range(1, n + 1), lambda f1, kps, det, flows: host.append((f1, kps, flows)))
    ctx.synchronize()
    used = an.an.device_log_used
    recs = D.parse_device_log(log.cpu().numpy(), used)
    an.close()
    ha, pa = D.pack_records(host)
    hb, pb = D.pack_records(recs)
    assert np.array_equal(ha, hb) and np.array_equal(pa, pb)
    # a log that is too small fails loudly, not silently
    an = ClipAnalyzer(ctx, w, h, 1, n, lambda fid: frames[fid - 1])
    an.an.set_device_log(torch.empty(4096, dtype=torch.uint8, device="cuda"))
    with pytest.raises(hip.PolychaseHipError, match="device log full"):
        an.run(range(1, 4), None)
    an.close()


@pytest.mark.parametrize("name", ["oracle_small.npz", "oracle_x86_small.npz"])
def test_hip_path_against_committed_golden_vectors(ctx, name):
    """tests/golden/*.npz: frozen inputs and expected outputs of every stage, in the canonical execution and in the x86
    execution (key "emulation" = the arithmetic flags of pc_context_set_arithmetic)."""
    import os
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name))
    before = ctx.arithmetic
    ctx.set_arithmetic(int(G["emulation"]) if "emulation" in G else hip.ARITH_CANONICAL)
    try:
        _golden_case(ctx, G)
    finally:
        ctx.set_arithmetic(before)


def _golden_case(ctx, G):
    frames = G["frames"]
    h, w, _ = frames[0].shape
    fr = []
    for f in frames:
        x = hip.Frame(ctx, w, h, 10, 2)
        x.set_rgb(np.ascontiguousarray(f))
        fr.append(x)
    assert np.array_equal(fr[0].gray(), G["gray0"])
    assert fr[0].num_levels == int(G["num_levels"])
    for l in range(fr[0].num_levels):
        assert np.array_equal(fr[0].level(l), G[f"level{l}"]) and np.array_equal(fr[0].deriv(l), G[f"deriv{l}"])
    fr[0].detect()
    assert np.array_equal(fr[0].min_eig().view(np.uint32), G["min_eig0"].view(np.uint32))
    assert fr[0].num_candidates == int(G["n_candidates0"])
    assert np.array_equal(fr[0].keypoints(), G["keypoints0"])
    xy, st, err = hip.lk_track(ctx, fr[0], fr[1:], hip.flow_options(max_level=2))
    for k in (1, 2):
        assert np.array_equal(st[k - 1], G[f"lk_status_{k}"])
        m = st[k - 1] == 1
        assert np.array_equal(xy[k - 1][m].view(np.uint32), G[f"lk_xy_{k}"][m].view(np.uint32))
        assert np.array_equal(err[k - 1][m].view(np.uint32), G[f"lk_err_{k}"][m].view(np.uint32))
    for x in fr:
        x.close()


def test_detection_slow_path_and_launch_hint(ctx):
    """Detection keeps every count on the device and sizes its launches from the previous frames' candidate counts; a
    frame beyond those bounds is redone on the slow path (counts on the host, rocPRIM sort).  (a) The forced slow path
    (POLYCHASE_GFTT_SLOW_PATH, read once per process: a subprocess) gives the keypoints of the fast path; (b) a clip
    whose textured area -- and candidate count -- jumps by more than 5x in the middle comes out of the pipelined analyzer with the
    keypoints of the stand-alone detection of every frame."""
    import subprocess
    import sys
    from polychase_amd.pipeline import ClipAnalyzer
    w, h = 480, 360
    clip = synth.NoiseClip(w, h, 24)
    frames = []
    for t in range(24):
        f = clip.frame(t)
        if t < 12:      # first half: texture only in one corner, the rest flat -> a fraction of the candidates
            g = np.full_like(f, 128)
            g[:96, :96] = f[:96, :96]
            f = g
        frames.append(np.ascontiguousarray(f))
    want = {}
    fr = hip.Frame(ctx, w, h)
    for t in range(24):
        fr.set_rgb(frames[t])
        fr.detect()
        want[t + 1] = fr.keypoints()
        assert np.array_equal(want[t + 1], oracle.gftt(oracle.rgb2gray(frames[t])))
    fr.close()
    assert len(want[20]) > 5 * len(want[5])
    got = {}
    an = ClipAnalyzer(ctx, w, h, 1, 24, lambda fid: frames[fid - 1])
    an.run(range(1, 25), lambda f1, k, det, flows: got.__setitem__(f1, k.copy()))
    an.close()
    for f1 in range(1, 25):
        assert np.array_equal(got[f1], want[f1]), f1
    # (a) the slow path in a fresh process
    np.save("/tmp/_pc_slow_in.npy", frames[18])
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from polychase_amd import hip; c = hip.Context(0); "
            "f = hip.Frame(c, %d, %d); f.set_rgb(np.load('/tmp/_pc_slow_in.npy')); f.detect(); np.save('/tmp/_pc_slow_out.npy', f.keypoints())"
            % (ROOT, w, h))
    subprocess.check_call([sys.executable, "-c", code], env=dict(os.environ, POLYCHASE_GFTT_SLOW_PATH="1"))
    assert np.array_equal(np.load("/tmp/_pc_slow_out.npy"), want[19])
