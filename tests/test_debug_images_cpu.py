"""CPU: the debug dump of analysed frames (write_images; reference cpp/opticalflow.cc:80-96 SaveImageForDebugging):
%06d.png holds the frame, keypoints_%06d.png the frame with an 11-px cross on every keypoint in the colours a copy of
cv::theRNG() produces (multiply-with-carry, coefficient 4164903690, state 0xffffffff -- restated, no cv2 here)."""
import os
import struct
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "polychase_amd", "core"))


def read_png(path):
    """minimal decoder for what the writer emits: 8-bit RGB, filter type 0 on every row"""
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    o, idat, w, h = 8, b"", 0, 0
    while o < len(b):
        n, typ = struct.unpack(">I4s", b[o:o + 8])
        data = b[o + 8:o + 8 + n]
        assert struct.unpack(">I", b[o + 8 + n:o + 12 + n])[0] == zlib.crc32(typ + data) & 0xFFFFFFFF
        if typ == b"IHDR":
            w, h, depth, ctype, comp, filt, inter = struct.unpack(">IIBBBBB", data)
            assert (depth, ctype, comp, filt, inter) == (8, 2, 0, 0, 0)
        elif typ == b"IDAT":
            idat += data
        o += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 3)


def cv_rng_colours(n):
    state, out = 0xFFFFFFFF, []
    for _ in range(n):
        c = []
        for _ in range(3):
            state = ((state & 0xFFFFFFFF) * 4164903690 + (state >> 32)) & 0xFFFFFFFFFFFFFFFF
            c.append((state & 0xFFFFFFFF) % 256)
        out.append((c[2], c[1], c[0]))      # drawn as (B, G, R) on the BGR image -> (R, G, B) in the file
    return out


def test_png_pair_of_one_frame(tmp_path):
    import polychase_core as core

    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    kps = np.array([[10, 12], [0, 0], [52, 36], [30, 3], [2, 20]], np.float32)     # corners: the crosses are clipped
    core._save_image_for_debugging(img, 42, str(tmp_path), kps)
    plain = read_png(tmp_path / "000042.png")
    marked = read_png(tmp_path / "keypoints_000042.png")
    assert np.array_equal(plain, img)
    want = img.copy()
    for (x, y), c in zip(kps.astype(int), cv_rng_colours(len(kps))):
        for d in range(-5, 6):
            if 0 <= x + d < 53:
                want[y, x + d] = c
        for d in range(-5, 6):
            if 0 <= y + d < 37:
                want[y + d, x] = c
    assert np.array_equal(marked, want)
    assert (marked != img).any(axis=2).sum() <= 21 * len(kps)
