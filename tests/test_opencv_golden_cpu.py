"""CPU: the oracle against golden vectors made by a REAL OpenCV (tests/golden/opencv_*.npz, written by
`python tests/opencv_crosscheck.py --write-golden` on a machine that has cv2: tests/opencv_golden.py).  No such file can be made
in this image (no cv2): the first test then only says so, and the second one runs the whole kit on a stand-in file made by the
oracle itself in a temporary directory, so that the writer, the reference-side keypoint selection and the consumers are exercised
code, not dead code."""
import os

import numpy as np
import pytest

import opencv_golden as og
import oracle


@pytest.mark.parametrize("path", og.golden_files() or [None])
def test_oracle_against_real_opencv_vectors(path):
    if path is None:
        pytest.skip("no tests/golden/opencv_*.npz: parity is unpinned at the OpenCV boundary until someone runs "
                    "`python tests/opencv_crosscheck.py --write-golden` where cv2 exists and commits the file")
    G = np.load(path)
    assert str(G["source"]).startswith("cv2 "), "only files made by a real OpenCV belong in tests/golden"
    arith = str(G["arith"])
    for name in ("c1", "c2"):
        if arith != "none":
            bad = og.compare(G, name, og.oracle_outputs(G, name, og.ARITH_FLAGS[arith]), exact_float=True)
        else:   # no execution of the restatement is bit-exact against this build: north_star's tolerances still hold
            bad = og.compare(G, name, og.oracle_outputs(G, name, oracle.EMU_OPENCV_X86), exact_float=False)
        assert not bad, f"{os.path.basename(path)} ({G['source']}, mode {arith}): " + "; ".join(bad)


def test_pin_kit_end_to_end_on_a_stand_in(tmp_path):
    path = str(tmp_path / "opencv_selftest.npz")
    arith = og.write(path, og.OracleBackend())
    assert arith == "opencv_x86", "the stand-in is the oracle's default execution"
    G = np.load(path)
    assert str(G["source"]) == "oracle-selftest"
    # the numpy restatement of the reference's selection (gftt.cc:38-164) on the library's response map must give the
    # keypoints the oracle's C restatement gives, in value and order -- two independent restatements of the same code
    for name in ("c1", "c2"):
        with oracle.emulation(oracle.EMU_OPENCV_X86):
            want = oracle.gftt(oracle.rgb2gray(np.ascontiguousarray(G[f"{name}_frames"][0])))
        assert len(want) > 50 and np.array_equal(G[f"{name}_keypoints"], want), name
        assert not og.compare(G, name, og.oracle_outputs(G, name, oracle.EMU_OPENCV_X86), exact_float=True)
    # ... and the canonical execution must NOT reproduce the step-edge case bit for bit (else the file's mode means nothing)
    assert og.compare(G, "c1", og.oracle_outputs(G, "c1", oracle.EMU_CANONICAL), exact_float=True)
    assert not og.compare(G, "c1", og.oracle_outputs(G, "c1", oracle.EMU_CANONICAL), exact_float=False)
