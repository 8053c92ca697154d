"""GPU, BASELINE.json full sizes (C3: 3840x2160, max_level=4): the oracle is too slow to run whole
frames in a unit test, so parity at full size is checked (a) bit-exactly on a random SUBSET of
keypoints against the oracle's LK, (b) through size-independent properties of the domain."""
import numpy as np
import pytest

import oracle
from polychase_amd import hip, synth

pytestmark = pytest.mark.gpu
W, H, ML = 3840, 2160, 4


@pytest.fixture(scope="module")
def setup():
    ctx = hip.Context(0)
    clip = synth.NoiseClip(W, H, 30, device="cuda")
    t0, t1 = 12, 16
    f0, f1 = clip.frame_torch(t0), clip.frame_torch(t1)
    a, b = hip.Frame(ctx, W, H, 10, ML), hip.Frame(ctx, W, H, 10, ML)
    a.set_rgb(f0)
    b.set_rgb(f1)
    a.detect()
    yield ctx, clip, (t0, t1), (f0, f1), (a, b)
    a.close()
    b.close()
    ctx.close()


def test_4k_pyramid_bit_exact(setup):
    ctx, clip, _, (f0, _), (a, _) = setup
    g = oracle.rgb2gray(f0.cpu().numpy())
    assert np.array_equal(a.gray(), g)
    p = oracle.Pyramid(g, 10, ML)
    assert a.num_levels == p.num_levels == 5
    for l in range(5):
        assert np.array_equal(a.level(l), p.image(l)) and np.array_equal(a.deriv(l), p.deriv(l))


def test_4k_keypoint_properties(setup):
    ctx, clip, _, _, (a, _) = setup
    kps = a.keypoints()
    n = len(kps)
    assert n > 100_000 and a.num_candidates >= n
    assert np.array_equal(kps, np.floor(kps))
    assert kps[:, 0].min() >= 1 and kps[:, 0].max() <= W - 2 and kps[:, 1].min() >= 1 and kps[:, 1].max() <= H - 2
    # acceptance order = response descending (ties by address descending)
    eig = a.min_eig()
    v = eig[kps[:, 1].astype(int), kps[:, 0].astype(int)]
    assert (np.diff(v) <= 0).all()
    lin = kps[:, 1].astype(np.int64) * W + kps[:, 0].astype(np.int64)
    ties = np.diff(v) == 0
    assert (np.diff(lin)[ties] < 0).all()
    # minimum distance >= 5 px: no two keypoints within the same or adjacent 5-px cells closer than 5
    cell = (kps // 5).astype(np.int64)
    order = np.lexsort((cell[:, 0], cell[:, 1]))
    from scipy.spatial import cKDTree
    d, _ = cKDTree(kps).query(kps, k=2)
    assert d[:, 1].min() >= 5.0
    # maximality: every rejected candidate must have an accepted keypoint within 5 px -- checked on the
    # candidates the oracle's suppression would see in a crop (bit-exact GFTT on a 512x512 region of a
    # DIFFERENT size is not comparable because the per-cell thresholds depend on the full frame),
    # so instead: idempotence -- detecting again gives the identical list
    a.detect()
    assert np.array_equal(a.keypoints(), kps)


def test_4k_lk_subset_bit_exact_and_truth(setup):
    ctx, clip, (t0, t1), (f0, f1), (a, b) = setup
    kps = a.keypoints()
    xy, st, err = hip.lk_track(ctx, a, [b, a], hip.flow_options(max_level=ML))
    # target == source: zero flow, all tracked, zero error (idempotence of the identity pair)
    ok_self = st[1] == 1
    assert ok_self.mean() > 0.999
    assert np.abs(xy[1][ok_self] - kps[ok_self]).max() < 2e-3 and err[1][ok_self].max() == 0.0
    # oracle on a random subset (the oracle needs the full pyramids, built once: ~1 s)
    rng = np.random.default_rng(0)
    sel = np.sort(rng.choice(len(kps), 3000, replace=False))
    p0 = oracle.Pyramid(oracle.rgb2gray(f0.cpu().numpy()), 10, ML)
    p1 = oracle.Pyramid(oracle.rgb2gray(f1.cpu().numpy()), 10, ML)
    oxy, ost, oerr = oracle.lk(p0, p1, kps[sel], oracle.flow_options(max_level=ML))
    assert np.array_equal(st[0][sel], ost)
    m = ost == 1
    assert np.array_equal(xy[0][sel][m].view(np.uint32), oxy[m].view(np.uint32))
    assert np.array_equal(err[0][sel][m].view(np.uint32), oerr[m].view(np.uint32))
    # analytic ground truth over ALL keypoints
    truth = clip.flow(kps, t0, t1)
    good = st[0] == 1
    assert good.mean() > 0.97 and np.median(np.abs(xy[0][good] - truth[good])) < 0.15


def test_4k_filtered_equals_raw(setup):
    ctx, clip, _, _, (a, b) = setup
    xy, st, err = hip.lk_track(ctx, a, [b], hip.flow_options(max_level=ML))
    (idx, fxy, ferr), = hip.lk_track_filtered(ctx, a, [b], hip.flow_options(max_level=ML))
    keep = np.nonzero(st[0] == 1)[0].astype(np.uint32)
    assert np.array_equal(idx, keep) and np.array_equal(fxy, xy[0][keep]) and np.array_equal(ferr, err[0][keep])
