"""GPU: BASELINE config C5 at its real frame size -- 1920x1080 frames RENDERED from a textured plane under a known camera
path -> generate_optical_flow_database (GFTT + pyramidal LK, SQLite) -> track_sequence (ray casting + PnP,
cpp/tracker.cc:36-131) -> refine_trajectory (cpp/refiner.cc) -- with 40 frames instead of 300 so that it fits the
test run.  Pose error against the ground truth and, for the first frames, against the CPU reference of the tracking
step (oracle/pnp_oracle.py, float64) on the same database: rotation <= 1e-4 rad, translation <= 1e-4 * |t|
(the tolerance BASELINE.json's north_star states for PnP poses).  tests/c5_endtoend.py is the runner (300 frames
stand-alone: profiles/r01_c5_endtoend.json)."""
import json
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_c5_1080p_analysis_tracking_refinement(tmp_path):
    sys.path.insert(0, HERE)
    import c5_endtoend
    out_path = str(tmp_path / "c5.json")
    assert c5_endtoend.main(["--width", "1920", "--height", "1080", "--frames", "40", "--oracle-frames", "3",
                             "--refine-iterations", "15", "--out", out_path]) == 0
    r = json.load(open(out_path))
    print(json.dumps(r, indent=1))
    tr = r["tracking"]
    assert min(tr["keypoints_per_frame"]) > 20_000                      # full-size frames: tens of thousands of keypoints
    assert tr["min_inlier_ratio"] >= 0.8       # the 40-frame path moves 7.5x faster than the 300-frame one: some tracks leave the plane
    assert tr["vs_truth"]["rotation_rad_max"] <= 1e-3 and tr["vs_truth"]["translation_max"] <= 6e-3   # dead reckoning over 40 frames
    ref = tr["vs_cpu_reference"]
    assert ref["frames"] == 3
    assert ref["rotation_rad_max"] <= 1e-4 and ref["translation_rel_max"] <= 1e-4
    rf = r["refinement"]
    assert rf["cost"][1] <= rf["cost"][0]
    assert rf["vs_truth"]["rotation_rad_max"] <= 1e-3 and rf["vs_truth"]["translation_max"] <= 6e-3


def test_c5_1080p_300_frames_poses_sampled_against_the_cpu_reference(tmp_path):
    """BASELINE config C5 with all 300 frames: analysis, tracking and refinement of the whole clip; every 10th frame's pose
    against the float64 CPU reference of the tracking step (cpp/tracker.cc:36-131 restated in oracle/pnp_oracle.py) solved
    from the same database, the GPU's poses of its source frames and the same initial guess."""
    sys.path.insert(0, HERE)
    import c5_endtoend
    out_path = str(tmp_path / "c5_300.json")
    assert c5_endtoend.main(["--width", "1920", "--height", "1080", "--frames", "300", "--oracle-frames", "0", "--oracle-stride", "10", "--oracle-workers", "8",
                             "--refine-iterations", "10", "--out", out_path]) == 0
    r = json.load(open(out_path))
    print(json.dumps(r, indent=1))
    tr = r["tracking"]
    s = tr["vs_cpu_reference_sampled"]
    assert s["frames"] == list(range(11, 301, 10))          # 29 frames spread over the clip
    assert s["rotation_rad_max"] <= 1e-4 and s["translation_rel_max"] <= 1e-4
    assert tr["min_inlier_ratio"] >= 0.9
    assert tr["vs_truth"]["rotation_rad_max"] <= 1e-3 and tr["vs_truth"]["translation_max"] <= 6e-3
    rf = r["refinement"]
    assert rf["cost"][1] <= rf["cost"][0]
