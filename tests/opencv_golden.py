"""opencv_golden.py -- the PIN KIT: golden vectors made by a REAL OpenCV, and what consumes them.

The reference calls OpenCV 4.x for all pixel arithmetic; this image has no cv2, so parity is "unpinned at the
OpenCV boundary".  On any machine with `opencv-python` ONE command closes that gap:

    python tests/opencv_crosscheck.py --write-golden            # -> tests/golden/opencv_<version>_<isa>.npz; commit it

The file holds inputs AND the real library's outputs of the calls the reference makes (cpp/opticalflow.cc:119-125, :184-186,
:259; cpp/feature_detection/gftt.cc:35-162) for two small cases -- the step-edge checkerboard of C1 and a crop of C2's texture:
gray, min-eig map, the keypoints the reference's own selection code makes of that map (restated below in numpy: thresholds,
3 x 3 dilate, sort, greedy suppression -- no float arithmetic of its own), every pyramid plane, and LK positions / status / error
into three targets.  `tests/test_opencv_golden_cpu.py` then checks the oracle and `tests/test_opencv_golden_gpu.py` the HIP path
against it, in the arithmetic mode the file names (the mode that reproduces that OpenCV build bit for bit) -- from then on the
parity of this repository is pinned to an executed OpenCV, not to its restatement.

make(backend) builds the dict; the backend is real cv2 (Cv2Backend) -- or, ONLY for the self-test of this kit in an image
without cv2, the oracle (OracleBackend: source = "oracle-selftest", refused by the consumers outside tmp directories)."""
from __future__ import annotations

import glob
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN_DIR = os.path.join(HERE, "golden")
WIN, MAX_LEVEL = 10, 3
ARITH_FLAGS = {"canonical": 0, "lk_x86": 1, "sobel_fma": 2, "opencv_x86": 3, "sobel_fma_rows": 6, "opencv_x86_rows": 7}   # = POLYCHASE_ARITH values


def cases():
    """name -> list of RGB frames: frame 0 is tracked into the others (skips -1, +1, +8 of the reference's pair list)"""
    import sys
    sys.path.insert(0, os.path.dirname(HERE))
    from polychase_amd import synth
    cb = synth.checkerboard_clip(20, w=320, h=240)
    clip = synth.NoiseClip(480, 270, 32)
    return {"c1": [cb[10], cb[9], cb[11], cb[18]], "c2": [clip.frame(20), clip.frame(19), clip.frame(21), clip.frame(28)]}


def reference_keypoints(eig: np.ndarray, quality_level=0.01, min_distance=5.0, grid_rows=4, grid_cols=4, max_corners=0) -> np.ndarray:
    """The selection of cpp/feature_detection/gftt.cc:38-164 on a given response map, in numpy (comparisons and maxima only: no
    arithmetic that could differ from the C++): per-cell threshold `eig > (float)(maxVal * quality)` else 0 (:61-65), 3 x 3 dilate
    with -inf outside (:70), candidates strictly inside the image with val != 0 && val == dilated (:76-86), sort by (value desc,
    address desc) (:7-12, :98), greedy minimum-distance suppression over a grid of cells of cvRound(min_distance) px (:100-164)."""
    h, w = eig.shape
    thr = eig.copy()
    ch, cw = -(-h // grid_rows), -(-w // grid_cols)
    for r in range(grid_rows):
        for c in range(grid_cols):
            cell = thr[r * ch:(r + 1) * ch, c * cw:(c + 1) * cw]
            if cell.size == 0:
                continue
            t = np.float32(np.float64(cell.max()) * quality_level)
            cell[~(cell > t)] = 0.0
    pad = np.full((h + 2, w + 2), -np.inf, np.float32)
    pad[1:-1, 1:-1] = thr
    dil = np.max(np.stack([pad[dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)]), axis=0)
    cand = (thr != 0) & (thr == dil)
    cand[0, :] = cand[-1, :] = False
    cand[:, 0] = cand[:, -1] = False
    ys, xs = np.nonzero(cand)
    vals = thr[ys, xs]
    order = np.lexsort((-(ys * w + xs), -vals.astype(np.float64)))
    ys, xs = ys[order], xs[order]
    if min_distance < 1:
        pts = np.stack([xs, ys], 1).astype(np.float32)
        return pts[:max_corners] if max_corners > 0 else pts
    cell = int(round(min_distance))
    gw, gh = (w + cell - 1) // cell, (h + cell - 1) // cell
    grid = [[] for _ in range(gw * gh)]
    md2 = np.float64(min_distance) * np.float64(min_distance)
    out = []
    for x, y in zip(xs.tolist(), ys.tolist()):
        xc, yc = x // cell, y // cell
        good = True
        for yy in range(max(0, yc - 1), min(gh - 1, yc + 1) + 1):
            for xx in range(max(0, xc - 1), min(gw - 1, xc + 1) + 1):
                for (px, py) in grid[yy * gw + xx]:
                    dx, dy = np.float32(x) - np.float32(px), np.float32(y) - np.float32(py)
                    if np.float64(np.float32(dx * dx + dy * dy)) < md2:
                        good = False
                        break
                if not good:
                    break
            if not good:
                break
        if good:
            grid[yc * gw + xc].append((x, y))
            out.append((x, y))
            if max_corners > 0 and len(out) == max_corners:
                break
    return np.array(out, np.float32).reshape(-1, 2)


OTHER_APERTURES = (5, 7, -1)     # Sobel 5 / 7, Scharr


class Cv2Backend:
    def __init__(self):
        import cv2
        self.cv2 = cv2
        isa = "x86_64" if "x86" in os.uname().machine else os.uname().machine
        self.source = f"cv2 {cv2.__version__}"
        self.tag = f"{cv2.__version__}_{isa}"
        info = cv2.getBuildInformation()
        self.build = "\n".join(l for l in info.splitlines() if any(k in l for k in ("CPU/HW", "Baseline", "Dispatched", "requested", "Version control", "Timestamp")))

    def gray(self, rgb):
        return self.cv2.cvtColor(rgb, self.cv2.COLOR_RGB2GRAY)

    def min_eig(self, gray, ksize=3):
        return self.cv2.cornerMinEigenVal(gray, 3, ksize=ksize)      # ksize -1: Scharr (cornerEigenValsVecs, aperture_size < 0)

    def pyramid(self, gray):
        n, pyr = self.cv2.buildOpticalFlowPyramid(gray, (WIN, WIN), MAX_LEVEL)
        out = []
        for l in range(n + 1):
            img, der = pyr[2 * l], pyr[2 * l + 1]
            out.append((np.ascontiguousarray(img), np.ascontiguousarray(der).reshape(img.shape[0], img.shape[1], 2)))
        return out

    def lk(self, g0, g1, pts):
        cv2 = self.cv2
        crit = (cv2.TERM_CRITERIA_COUNT + cv2.TERM_CRITERIA_EPS, 30, 0.01)
        xy, st, err = cv2.calcOpticalFlowPyrLK(g0, g1, pts.reshape(-1, 1, 2), None, winSize=(WIN, WIN), maxLevel=MAX_LEVEL,
                                               criteria=crit, flags=0, minEigThreshold=1e-4)
        return xy.reshape(-1, 2), st.reshape(-1), err.reshape(-1)


class OracleBackend:
    """self-test of the kit only: the oracle in its default (x86) execution stands in for the library"""
    source, tag, build = "oracle-selftest", "selftest", "oracle/pc_oracle.c"

    def __init__(self):
        import oracle
        self.o = oracle

    def gray(self, rgb):
        return self.o.rgb2gray(rgb)

    def min_eig(self, gray, ksize=3):
        return self.o.min_eigen_val(gray, 3, ksize)

    def pyramid(self, gray):
        p = self.o.Pyramid(gray, WIN, MAX_LEVEL)
        return [(p.image(l, padded=False), p.deriv(l, padded=False)) for l in range(p.num_levels)]

    def lk(self, g0, g1, pts):
        return self.o.lk(self.o.Pyramid(g0, WIN, MAX_LEVEL), self.o.Pyramid(g1, WIN, MAX_LEVEL), pts)


def make(backend) -> dict:
    out = {"source": np.array(backend.source), "build": np.array(backend.build), "win": np.int32(WIN), "max_level": np.int32(MAX_LEVEL)}
    for name, frames in cases().items():
        grays = [backend.gray(f) for f in frames]
        eig = backend.min_eig(grays[0])
        kps = reference_keypoints(eig)
        out[f"{name}_frames"] = np.stack(frames)
        out[f"{name}_gray"] = np.stack(grays)
        out[f"{name}_min_eig"] = eig
        for ks in OTHER_APERTURES:       # GFTTOptions.gradient_size the addon never sets (gftt.cc:31-36): the response map only
            out[f"{name}_min_eig_k{ks}"] = backend.min_eig(grays[0], ks)
        out[f"{name}_keypoints"] = kps
        for l, (img, der) in enumerate(backend.pyramid(grays[0])):
            out[f"{name}_level{l}"] = img
            out[f"{name}_deriv{l}"] = der
        for k in range(1, len(frames)):
            xy, st, err = backend.lk(grays[0], grays[k], kps)
            out[f"{name}_lk_xy_{k}"], out[f"{name}_lk_status_{k}"], out[f"{name}_lk_err_{k}"] = xy, st.astype(np.uint8), err
    return out


def oracle_outputs(G, name, emu):
    """what the oracle computes for case `name` of a golden dict under emulation flags `emu` -> dict with the same keys"""
    import oracle
    frames = G[f"{name}_frames"]
    res = {}
    with oracle.emulation(emu):
        grays = [oracle.rgb2gray(np.ascontiguousarray(f)) for f in frames]
        res["gray"] = np.stack(grays)
        res["min_eig"] = oracle.min_eigen_val(grays[0], 3, 3)
        for ks in OTHER_APERTURES:
            res[f"min_eig_k{ks}"] = oracle.min_eigen_val(grays[0], 3, ks)
        res["keypoints"] = oracle.gftt(grays[0])
        p0 = oracle.Pyramid(grays[0], WIN, MAX_LEVEL)
        for l in range(p0.num_levels):
            res[f"level{l}"], res[f"deriv{l}"] = p0.image(l, padded=False), p0.deriv(l, padded=False)
        kps = np.ascontiguousarray(G[f"{name}_keypoints"])      # the LK inputs are the file's keypoints, whatever the oracle detects
        for k in range(1, len(frames)):
            res[f"lk_xy_{k}"], res[f"lk_status_{k}"], res[f"lk_err_{k}"] = oracle.lk(p0, oracle.Pyramid(grays[k], WIN, MAX_LEVEL), kps)
    return res


def compare(G, name, got, exact_float: bool):
    """-> list of mismatch descriptions.  Integer stages always bit-exact; float stages bit-exact (exact_float) or within
    north_star's tolerance: LK positions 1e-3 px, error 2e-2 gray levels, status equal; min-eig 1e-6 of its maximum."""
    bad = []
    n_targets = len(G[f"{name}_frames"]) - 1
    for key in ["gray"] + [k for k in got if k.startswith("level") or k.startswith("deriv")]:
        if not np.array_equal(got[key], G[f"{name}_{key}"]):
            bad.append(f"{name}/{key}: integer stage differs")
    e, ge = got["min_eig"], G[f"{name}_min_eig"]
    if exact_float:
        if not np.array_equal(e.view(np.uint32), ge.view(np.uint32)):
            bad.append(f"{name}/min_eig: {(e.view(np.uint32) != ge.view(np.uint32)).sum()} pixels differ in bits")
        if not np.array_equal(got["keypoints"], G[f"{name}_keypoints"]):
            bad.append(f"{name}/keypoints: value or order differs ({len(got['keypoints'])} vs {len(G[f'{name}_keypoints'])})")
    else:
        rel = float(np.abs(e - ge).max() / max(float(np.abs(ge).max()), 1e-30))
        if rel > 1e-6:
            bad.append(f"{name}/min_eig: max |diff| / max = {rel:.2e}")
        a, b = set(map(tuple, got["keypoints"].astype(int))), set(map(tuple, G[f"{name}_keypoints"].astype(int)))
        if len(a ^ b) > max(2, len(b) // 1000):
            bad.append(f"{name}/keypoints: {len(a ^ b)} corners not common")
    for ks in OTHER_APERTURES:
        key = f"min_eig_k{ks}"
        if f"{name}_{key}" not in G or key not in got:      # files written before round 5 do not hold them
            continue
        e2, g2 = got[key], G[f"{name}_{key}"]
        if exact_float:
            if not np.array_equal(e2.view(np.uint32), g2.view(np.uint32)):
                bad.append(f"{name}/{key}: {(e2.view(np.uint32) != g2.view(np.uint32)).sum()} pixels differ in bits")
        elif float(np.abs(e2 - g2).max() / max(float(np.abs(g2).max()), 1e-30)) > 1e-6:
            bad.append(f"{name}/{key}: max |diff| / max = {float(np.abs(e2 - g2).max() / max(float(np.abs(g2).max()), 1e-30)):.2e}")
    for k in range(1, n_targets + 1):
        st, gst = got[f"lk_status_{k}"], G[f"{name}_lk_status_{k}"]
        if not np.array_equal(st, gst):
            bad.append(f"{name}/lk_status_{k}: {(st != gst).sum()} flips")
            continue
        m = gst == 1
        xy, gxy = got[f"lk_xy_{k}"][m], G[f"{name}_lk_xy_{k}"][m]
        er, ger = got[f"lk_err_{k}"][m], G[f"{name}_lk_err_{k}"][m]
        if exact_float:
            if not np.array_equal(xy.view(np.uint32), gxy.view(np.uint32)):
                bad.append(f"{name}/lk_xy_{k}: {(xy.view(np.uint32) != gxy.view(np.uint32)).any(axis=1).sum()} vectors differ in bits, max {np.abs(xy - gxy).max():.2e} px")
            if not np.array_equal(er.view(np.uint32), ger.view(np.uint32)):
                bad.append(f"{name}/lk_err_{k}: differs in bits")
        else:
            if m.any() and np.abs(xy - gxy).max() > 1e-3:
                bad.append(f"{name}/lk_xy_{k}: max |diff| {np.abs(xy - gxy).max():.2e} px > 1e-3")
            # the L1 patch error (gray levels per pixel) moves with the position: on step edges 5e-4 px are worth 1e-2
            if m.any() and np.abs(er - ger).max() > 2e-2:
                bad.append(f"{name}/lk_err_{k}: max |diff| {np.abs(er - ger).max():.2e}")
    return bad


def matching_arith(G) -> str | None:
    """the arithmetic mode in which the ORACLE reproduces the file bit for bit (tried: the default opencv_x86 first)"""
    for mode in ("opencv_x86", "opencv_x86_rows", "lk_x86", "sobel_fma", "sobel_fma_rows", "canonical"):
        if all(not compare(G, name, oracle_outputs(G, name, ARITH_FLAGS[mode]), True) for name in ("c1", "c2")):
            return mode
    return None


def golden_files(directory=GOLDEN_DIR):
    return sorted(glob.glob(os.path.join(directory, "opencv_*.npz")))


def write(path: str, backend) -> str:
    G = make(backend)
    arith = matching_arith(G)
    G["arith"] = np.array(arith or "none")
    np.savez_compressed(path, **G)
    return arith or "none"
