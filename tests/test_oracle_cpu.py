"""CPU: pins the C restatement (oracle/pc_oracle.c) against INDEPENDENT restatements of the same
published OpenCV algorithms (scipy.ndimage / numpy, written separately) and against analytic ground
truth.  There is no executable OpenCV in the image and the reference holds no golden vectors for
this path, so this is what "pinning" can be here."""
import numpy as np
import pytest
from scipy import ndimage

import oracle
from polychase_amd import synth


@pytest.fixture(scope="module")
def gray():
    rng = np.random.default_rng(5)
    g = ndimage.gaussian_filter(rng.integers(0, 256, (123, 157)).astype(np.float64), 1.5)
    return np.clip((g - g.min()) / (g.max() - g.min()) * 255, 0, 255).astype(np.uint8)


def test_rgb2gray_formula():
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, (37, 41, 3), dtype=np.uint8)
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    assert np.array_equal(oracle.rgb2gray(rgb), ((r * 9798 + g * 19235 + b * 3735 + 16384) >> 15).astype(np.uint8))
    # known values: pure white stays 255, pure primaries follow the 15-bit coefficients
    px = np.array([[[255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255]]], np.uint8)
    assert oracle.rgb2gray(px).tolist() == [[255, 76, 150, 29]]


def test_pyrdown_and_padding_vs_scipy(gray):
    p = oracle.Pyramid(gray, 10, 3)
    k = np.array([1, 4, 6, 4, 1], np.int64)
    cur = gray.astype(np.int64)
    for l in range(1, p.num_levels):
        full = ndimage.correlate1d(ndimage.correlate1d(cur, k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
        down = ((full + 128) >> 8)[::2, ::2]
        assert p.level_size(l) == (down.shape[1], down.shape[0])
        assert np.array_equal(p.image(l, padded=False), down.astype(np.uint8))
        cur = down
        assert np.array_equal(p.image(l), np.pad(down.astype(np.uint8), 10, mode="reflect"))   # REFLECT_101
    assert np.array_equal(p.image(0), np.pad(gray, 10, mode="reflect"))


def test_pyramid_stops_when_next_level_too_small():
    g = np.zeros((45, 200), np.uint8)
    p = oracle.Pyramid(g, 10, 5)     # 45 -> 23 -> 12 -> 6: level 3 would be <= 10 high
    assert p.num_levels == 3 and p.level_size(2) == (50, 12)


def test_scharr_vs_scipy(gray):
    p = oracle.Pyramid(gray, 10, 0)
    g = gray.astype(np.int64)
    sm, df = np.array([3, 10, 3]), np.array([-1, 0, 1])
    dx = ndimage.correlate1d(ndimage.correlate1d(g, sm, axis=0, mode="mirror"), df, axis=1, mode="mirror")
    dy = ndimage.correlate1d(ndimage.correlate1d(g, df, axis=0, mode="mirror"), sm, axis=1, mode="mirror")
    d = p.deriv(0, padded=False)
    assert np.array_equal(d[..., 0], dx) and np.array_equal(d[..., 1], dy)
    full = p.deriv(0)
    assert not full[:10].any() and not full[:, :10].any() and not full[-10:].any() and not full[:, -10:].any()


def test_min_eigen_val_vs_float64(gray):
    g = gray.astype(np.float64)
    s = 1.0 / (4 * 3 * 255)
    dx = ndimage.correlate1d(ndimage.correlate1d(g, [-1, 0, 1], axis=1, mode="mirror"), np.array([1, 2, 1]) * s, axis=0, mode="mirror")
    dy = ndimage.correlate1d(ndimage.correlate1d(g, np.array([1, 2, 1]) * s, axis=1, mode="mirror"), [-1, 0, 1], axis=0, mode="mirror")
    box = lambda a: ndimage.uniform_filter(a, 3, mode="mirror") * 9
    a, b, c = box(dx * dx) * 0.5, box(dx * dy), box(dy * dy) * 0.5
    ref = (a + c) - np.sqrt((a - c) ** 2 + b * b)
    got = oracle.min_eigen_val(gray)
    assert np.abs(got - ref).max() < 2e-7 * max(1.0, np.abs(ref).max()) + 1e-9


@pytest.mark.parametrize("ksize,block", [(5, 3), (7, 3), (-1, 3), (5, 2), (7, 5), (-1, 4)])
def test_min_eigen_val_other_apertures_vs_float64(gray, ksize, block):
    """gradient_size 5 / 7 (Sobel, binomial taps from getSobelKernels' recurrence -- rebuilt here by convolution, not copied
    from the C tables) and -1 (Scharr 3 / 10 / 3), cornerEigenValsVecs' scale 1 / (2^(ksize-1) * block * 255) (x 1/2 for
    Scharr), any block size with anchor block / 2: the C restatement against float64 scipy, both arithmetic hypotheses."""
    g = gray.astype(np.float64)
    if ksize > 0:
        sm = np.array([1.0])
        for _ in range(ksize - 1):
            sm = np.convolve(sm, [1, 1])
        dv = np.array([1.0])
        for _ in range(ksize - 2):
            dv = np.convolve(dv, [1, 1])
        dv = np.convolve(dv, [1, -1])[::-1]            # correlation taps: minus on the left
        s = 1.0 / (2 ** (ksize - 1) * block * 255)
    else:
        sm, dv, s = np.array([3.0, 10.0, 3.0]), np.array([-1.0, 0.0, 1.0]), 1.0 / (4 * block * 255) / 2
    assert dv[0] < 0 and dv[-1] > 0 and abs(dv.sum()) < 1e-12
    dx = ndimage.correlate1d(ndimage.correlate1d(g, dv, axis=1, mode="mirror"), sm * s, axis=0, mode="mirror")
    dy = ndimage.correlate1d(ndimage.correlate1d(g, sm * s, axis=1, mode="mirror"), dv, axis=0, mode="mirror")
    # unnormalised box sum, window rows y - block//2 .. y - block//2 + block - 1 (anchor = block / 2), REFLECT_101 on the products
    def box(a):
        a0 = block // 2
        p = np.pad(a, ((a0, block - 1 - a0), (a0, block - 1 - a0)), mode="reflect")
        out = np.zeros_like(a)
        for j in range(block):
            for i in range(block):
                out += p[j:j + a.shape[0], i:i + a.shape[1]]
        return out
    a, b, c = box(dx * dx) * 0.5, box(dx * dy), box(dy * dy) * 0.5
    ref = (a + c) - np.sqrt((a - c) ** 2 + b * b)
    for emu in (oracle.EMU_CANONICAL, oracle.EMU_OPENCV_X86, oracle.EMU_OPENCV_X86 | oracle.EMU_SOBEL_ROW_FMA):
        with oracle.emulation(emu):
            got = oracle.min_eigen_val(gray, block, ksize)
        assert np.abs(got - ref).max() < 1e-6 * max(1.0, np.abs(ref).max()) + 1e-9, (emu, np.abs(got - ref).max(), np.abs(ref).max())
    # Harris takes the same covariance sums
    ref_h = (4 * a * c - b * b) - 0.04 * (2 * a + 2 * c) ** 2
    with oracle.emulation(oracle.EMU_CANONICAL):
        got_h = oracle.corner_harris(gray, block, ksize, 0.04)
    assert np.abs(got_h - ref_h).max() < 1e-5 * max(1e-12, np.abs(ref_h).max())
    assert lib_refuses(gray)


def lib_refuses(gray):
    """apertures OpenCV's getSobelKernels would also reject (even, > 7 in this restatement) are an error, not a guess"""
    import ctypes as C
    out = np.empty(gray.shape, np.float32)
    g = np.ascontiguousarray(gray)
    return all(oracle.lib().pco_min_eigen_val(g.ctypes.data, g.shape[1], g.shape[0], 3, k, out.ctypes.data) != 0 for k in (0, 1, 2, 4, 6, 9, -2))


def test_gftt_invariants_and_grid(gray):
    xy, eig, ncand = oracle.gftt(gray, want_eig=True)
    assert len(xy) > 20 and ncand >= len(xy)
    assert np.array_equal(xy, np.floor(xy))                       # integer coordinates (gftt.cc:157)
    h, w = gray.shape
    assert xy[:, 0].min() >= 1 and xy[:, 0].max() <= w - 2 and xy[:, 1].min() >= 1 and xy[:, 1].max() <= h - 2
    d = xy[:, None, :] - xy[None, :, :]
    d2 = (d ** 2).sum(-1) + np.eye(len(xy)) * 1e9
    assert d2.min() >= 25                                         # min_distance 5
    vals = eig[xy[:, 1].astype(int), xy[:, 0].astype(int)]
    assert (np.diff(vals) <= 0).all()                             # acceptance order = value descending
    # every 4x4 grid cell with texture contributes features (the point of the per-cell threshold)
    bh, bw = -(-h // 4), -(-w // 4)
    cells = {(int(y) // bh, int(x) // bw) for x, y in xy}
    assert len(cells) == 16
    # max_corners is a prefix of the unlimited result
    few = oracle.gftt(gray, oracle.gftt_options(max_corners=17))
    assert np.array_equal(few, xy[:17])


def test_gftt_ties_break_by_descending_address():
    """Two identical isolated blobs -> identical responses; the one with the larger linear address
    (lower in the image) comes first (gftt.cc:7-12)."""
    g = np.full((60, 60), 50, np.uint8)
    for cy, cx in ((15, 15), (44, 44)):
        g[cy - 2:cy + 3, cx - 2:cx + 3] = 200
    xy = oracle.gftt(g, oracle.gftt_options(grid_rows=1, grid_cols=1, quality_level=0.5))
    assert len(xy) >= 2
    top = xy[:8]
    assert top[0][1] > 30       # first accepted corner belongs to the lower blob


def _lk_float64(I, J, pts, win=10, iters=30, eps=0.01):
    """Independent single-level LK (float64, bilinear via map_coordinates, Scharr/32 gradients)."""
    I, J = I.astype(np.float64), J.astype(np.float64)
    sm, df = np.array([3, 10, 3]) / 32.0, np.array([-1, 0, 1])
    Ix = ndimage.correlate1d(ndimage.correlate1d(I, sm, axis=0, mode="mirror"), df, axis=1, mode="mirror")
    Iy = ndimage.correlate1d(ndimage.correlate1d(I, df, axis=0, mode="mirror"), sm, axis=1, mode="mirror")
    out = []
    yy, xx = np.mgrid[0:win, 0:win]
    for px, py in pts:
        x0, y0 = px - (win - 1) / 2, py - (win - 1) / 2
        samp = lambda A, x, y: ndimage.map_coordinates(A, [yy + y, xx + x], order=1, mode="mirror")
        i, gx, gy = samp(I, x0, y0), samp(Ix, x0, y0), samp(Iy, x0, y0)
        A = np.array([[(gx * gx).sum(), (gx * gy).sum()], [(gx * gy).sum(), (gy * gy).sum()]])
        q = np.array([x0, y0])
        for _ in range(iters):
            d = samp(J, q[0], q[1]) - i
            b = np.array([(d * gx).sum(), (d * gy).sum()])
            delta = -np.linalg.solve(A, b)
            q += delta
            if delta @ delta <= eps * eps:
                break
        out.append(q + (win - 1) / 2)
    return np.array(out)


def test_lk_vs_independent_float64_and_analytic_truth():
    clip = synth.NoiseClip(320, 240, 12)
    g0, g1 = oracle.rgb2gray(clip.frame(5)), oracle.rgb2gray(clip.frame(6))
    pts = oracle.gftt(g0)
    pts = pts[(pts[:, 0] > 30) & (pts[:, 0] < 290) & (pts[:, 1] > 30) & (pts[:, 1] < 210)][:150]
    out, st, err = oracle.lk(oracle.Pyramid(g0, 10, 0), oracle.Pyramid(g1, 10, 0), pts, oracle.flow_options(max_level=0))
    ref = _lk_float64(g0, g1, pts)
    ok = st == 1
    assert ok.mean() > 0.95
    assert np.median(np.abs(out[ok] - ref[ok])) < 0.01            # fixed-point vs float64: ~1/100 px
    truth = clip.flow(pts, 5, 6)
    assert np.median(np.abs(out[ok] - truth[ok])) < 0.12   # resampled-texture bias; both LKs agree to 0.01
    # pyramidal: 8-frame skip (7 px of motion) needs the coarse levels
    g8 = oracle.rgb2gray(clip.frame(11))
    out3, st3, _ = oracle.lk(oracle.Pyramid(g0), oracle.Pyramid(g8), pts)
    truth8 = clip.flow(pts, 5, 11)
    ok3 = st3 == 1
    assert ok3.mean() > 0.9 and np.median(np.abs(out3[ok3] - truth8[ok3])) < 0.2


def test_lk_status_and_err_semantics():
    g = oracle.rgb2gray(synth.NoiseClip(200, 150, 4).frame(1))
    p = oracle.Pyramid(g)
    pts = np.array([[100, 75], [-30, 75], [100, 400], [3, 3]], np.float32)
    out, st, err = oracle.lk(p, p, pts)
    assert st[0] == 1 and np.allclose(out[0], pts[0], atol=1e-3) and err[0] == 0.0   # same image: zero flow, zero L1 error
    assert st[1] == 0 and st[2] == 0 and err[1] == 0 and err[2] == 0                 # window outside the image at level 0
    flat = oracle.Pyramid(np.full((150, 200), 9, np.uint8))
    _, stf, _ = oracle.lk(flat, flat, pts[:1])
    assert stf[0] == 0                                                                # minEig below threshold


def test_reference_shaped_clip_driver_counts():
    frames = synth.checkerboard_clip(18, w=160, h=120)
    kps, flows = oracle.analyze_clip(frames, first_frame=1, threads=4)
    assert sorted(kps) == list(range(1, 19))
    assert len(flows) == 8 * 18 - 30
    for (a, b), (idx, xy, err) in flows.items():
        assert b - a in (-8, -4, -2, -1, 1, 2, 4, 8)
        assert (np.diff(idx.astype(np.int64)) > 0).all() and len(idx) == len(xy) == len(err)
    # threads do not change results
    kps1, flows1 = oracle.analyze_clip(frames, first_frame=1, threads=1, feature_threads=1)
    assert all(np.array_equal(kps[f], kps1[f]) for f in kps)
    assert all(all(np.array_equal(x, y) for x, y in zip(flows[k], flows1[k])) for k in flows)
