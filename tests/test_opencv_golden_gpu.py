"""GPU: the HIP path against golden vectors made by a REAL OpenCV (tests/golden/opencv_*.npz; tests/opencv_golden.py is the kit,
`python tests/opencv_crosscheck.py --write-golden` writes a file wherever cv2 exists).  The context runs in the arithmetic mode
the file names -- the execution that reproduces that OpenCV build bit for bit -- and every stage the reference takes from OpenCV
(cpp/opticalflow.cc:119-125, :184-186, :259; cpp/feature_detection/gftt.cc:35) must then be bit-exact: gray, min-eig map,
keypoints in value and order, every pyramid plane, LK positions / status / error.  Without such a file the first test only says
what is missing; the second one runs the same consumer on a stand-in made by the oracle in a temporary directory."""
import os

import numpy as np
import pytest

import opencv_golden as og
import oracle
from polychase_amd import hip

pytestmark = pytest.mark.gpu


def hip_outputs(ctx, G, name):
    frames = G[f"{name}_frames"]
    h, w = frames[0].shape[:2]
    fr = []
    for f in frames:
        x = hip.Frame(ctx, w, h, og.WIN, og.MAX_LEVEL)
        x.set_rgb(np.ascontiguousarray(f))
        fr.append(x)
    res = {"gray": np.stack([x.gray() for x in fr])}
    fr[0].detect()
    res["min_eig"] = fr[0].min_eig()
    res["keypoints"] = fr[0].keypoints()
    for ks in og.OTHER_APERTURES:
        fr[0].detect(hip.gftt_options(gradient_size=ks))
        res[f"min_eig_k{ks}"] = fr[0].min_eig()
    win = og.WIN
    for l in range(fr[0].num_levels):
        res[f"level{l}"] = fr[0].level(l)[win:-win, win:-win]
        res[f"deriv{l}"] = fr[0].deriv(l)[win:-win, win:-win]
    fr[0].set_keypoints(np.ascontiguousarray(G[f"{name}_keypoints"]))     # LK from the file's keypoints, whatever was detected
    xy, st, err = hip.lk_track(ctx, fr[0], fr[1:], hip.flow_options(window_size=og.WIN, max_level=og.MAX_LEVEL))
    for k in range(1, len(frames)):
        res[f"lk_xy_{k}"], res[f"lk_status_{k}"], res[f"lk_err_{k}"] = xy[k - 1], st[k - 1], err[k - 1]
    for x in fr:
        x.close()
    return res


def check(G, what):
    arith = str(G["arith"])
    ctx = hip.Context(0)
    try:
        ctx.set_arithmetic(og.ARITH_FLAGS[arith] if arith != "none" else hip.ARITH_OPENCV_X86)
        for name in ("c1", "c2"):
            bad = og.compare(G, name, hip_outputs(ctx, G, name), exact_float=arith != "none")
            assert not bad, f"{what} ({G['source']}, mode {arith}): " + "; ".join(bad)
    finally:
        ctx.close()


@pytest.mark.parametrize("path", og.golden_files() or [None])
def test_hip_path_against_real_opencv_vectors(path):
    if path is None:
        pytest.skip("no tests/golden/opencv_*.npz: parity is unpinned at the OpenCV boundary until someone runs "
                    "`python tests/opencv_crosscheck.py --write-golden` where cv2 exists and commits the file")
    G = np.load(path)
    assert str(G["source"]).startswith("cv2 ")
    check(G, os.path.basename(path))


def test_pin_kit_consumer_on_a_stand_in(tmp_path):
    path = str(tmp_path / "opencv_selftest.npz")
    assert og.write(path, og.OracleBackend()) == "opencv_x86"
    check(np.load(path), "stand-in")
