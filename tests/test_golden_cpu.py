"""CPU: the oracle against the committed golden vectors (tests/golden/oracle_small.npz, made by
tests/golden/make_oracle_fixtures.py).  A change in oracle/pc_oracle.c that alters any output bit
fails here before it can silently move the parity target."""
import os

import numpy as np

import oracle

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_small.npz"))


def test_oracle_reproduces_golden_vectors():
    frames = G["frames"]
    grays = [oracle.rgb2gray(f) for f in frames]
    assert np.array_equal(grays[0], G["gray0"])
    assert np.array_equal(oracle.min_eigen_val(grays[0]).view(np.uint32), G["min_eig0"].view(np.uint32))
    kps, _, ncand = oracle.gftt(grays[0], want_eig=True)
    assert ncand == int(G["n_candidates0"]) and np.array_equal(kps, G["keypoints0"])
    p = [oracle.Pyramid(g, 10, 2) for g in grays]
    assert p[0].num_levels == int(G["num_levels"])
    for l in range(p[0].num_levels):
        assert np.array_equal(p[0].image(l), G[f"level{l}"]) and np.array_equal(p[0].deriv(l), G[f"deriv{l}"])
    for k in (1, 2):
        xy, st, err = oracle.lk(p[0], p[k], kps, oracle.flow_options(max_level=2))
        assert np.array_equal(st, G[f"lk_status_{k}"])
        assert np.array_equal(xy.view(np.uint32), G[f"lk_xy_{k}"].view(np.uint32))
        assert np.array_equal(err.view(np.uint32), G[f"lk_err_{k}"].view(np.uint32))
