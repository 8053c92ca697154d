"""Synthetic clips for tests and bench.

Two generators, both with analytic ground-truth flow:

* ``checkerboard_clip``  -- C1: 640x480x30 anti-aliased checkerboard, pure translation.
* ``NoiseClip``          -- C2..C5: band-limited noise canvas (xorshift64*, Gaussian sigma=2,
                            stretched to [16,240]) seen through a similarity motion
                            (translation (0.9, 0.4) px/frame + rotation 0.02 deg/frame about the
                            frame centre, time origin at the middle of the clip so the 256-px canvas
                            margin is never left).

Frames are RGB uint8, H x W x 3, C-contiguous: the frame format of the reference's accessor
(cpp/opticalflow.cc:189-202).  Frame ids start at 1 (Blender convention).

torch is used only to resample the canvas (CPU here, GPU in bench.py); no model code.
"""
from __future__ import annotations

import math

import numpy as np

_SEED = 0x9E3779B97F4A7C15
_MASK = (1 << 64) - 1


# ----------------------------------------------------------------------------------------------
# C1: checkerboard
# ----------------------------------------------------------------------------------------------
def _stripe_cover(x0: np.ndarray, square: float) -> np.ndarray:
    """Fraction of [x0, x0+1] lying in odd stripes of width `square` (exact area coverage)."""
    period = 2.0 * square

    def F(u):  # integral_0^u of s(v), s = floor(v/square) mod 2
        k = np.floor(u / period)
        r = u - k * period
        return k * square + np.maximum(0.0, r - square)

    return F(x0 + 1.0) - F(x0)


def checkerboard_frame(t: int, w: int = 640, h: int = 480, square: int = 32, lo: int = 64,
                       hi: int = 192, vx: float = 1.25, vy: float = 0.75) -> np.ndarray:
    """Frame `t` (0-based) of the translating checkerboard; content moves by (+vx, +vy) px/frame."""
    # frame pixel [x, x+1] shows board coordinates [x - vx t, x + 1 - vx t]; offset keeps u > 0
    off = 1024.0
    ax = _stripe_cover(np.arange(w, dtype=np.float64) - vx * t + off, float(square))
    ay = _stripe_cover(np.arange(h, dtype=np.float64) - vy * t + off, float(square))
    cover = ax[None, :] * (1.0 - ay[:, None]) + (1.0 - ax[None, :]) * ay[:, None]
    g = np.rint(lo + (hi - lo) * cover).astype(np.uint8)
    return np.ascontiguousarray(np.repeat(g[:, :, None], 3, axis=2))


def checkerboard_clip(n: int = 30, **kw) -> list[np.ndarray]:
    return [checkerboard_frame(t, **kw) for t in range(n)]


# ----------------------------------------------------------------------------------------------
# C2..C5: band-limited noise under a similarity motion
# ----------------------------------------------------------------------------------------------
def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def xorshift_noise(h: int, w: int, seed: int) -> np.ndarray:
    """uint8 noise, one xorshift64* stream per row (row streams seeded by splitmix64(seed + row))."""
    with np.errstate(over="ignore"):
        s = _splitmix64(np.uint64(seed & _MASK) + np.arange(h, dtype=np.uint64))
        s[s == 0] = np.uint64(0x1234567)
        out = np.empty((h, w), dtype=np.uint8)
        mult = np.uint64(0x2545F4914F6CDD1D)
        for x in range(w):
            s ^= s >> np.uint64(12)
            s ^= s << np.uint64(25)
            s ^= s >> np.uint64(27)
            out[:, x] = ((s * mult) >> np.uint64(56)).astype(np.uint8)
    return out


def _gauss_blur(img: np.ndarray, sigma: float) -> np.ndarray:
    from scipy.ndimage import correlate1d

    r = int(math.ceil(4 * sigma))
    k = np.exp(-0.5 * (np.arange(-r, r + 1) / sigma) ** 2)
    k /= k.sum()
    a = correlate1d(img.astype(np.float64), k, axis=1, mode="mirror")
    return correlate1d(a, k, axis=0, mode="mirror")


def noise_canvas(w: int, h: int, margin: int = 256, sigma: float = 2.0, seed: int = _SEED) -> np.ndarray:
    """float32 canvas (3, h+2m, w+2m) in [16, 240]."""
    ch = []
    for c in range(3):
        n = xorshift_noise(h + 2 * margin, w + 2 * margin, seed + c).astype(np.float32)
        b = _gauss_blur(n, sigma)
        lo, hi = float(b.min()), float(b.max())
        ch.append((16.0 + (b - lo) * (224.0 / (hi - lo))).astype(np.float32))
    return np.stack(ch, axis=0)


class NoiseClip:
    """Band-limited noise clip. ``frame(t)`` returns H x W x 3 uint8 (numpy or torch on `device`)."""

    def __init__(self, w: int, h: int, n_frames: int, device: str = "cpu", margin: int = 256,
                 vx: float = 0.9, vy: float = 0.4, rot_deg: float = 0.02, seed: int = _SEED):
        import torch

        self.w, self.h, self.n, self.margin = w, h, n_frames, margin
        self.vx, self.vy, self.rot = vx, vy, math.radians(rot_deg)
        self.device = torch.device(device)
        self.canvas = torch.from_numpy(noise_canvas(w, h, margin, seed=seed)).to(self.device)[None]
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32, device=self.device),
                                torch.arange(w, dtype=torch.float32, device=self.device), indexing="ij")
        self._xs, self._ys = xs, ys

    # pose of frame t (0-based): canvas point u = C + R(theta_t) (p - c) + d_t
    def _pose(self, t: float):
        tc = t - 0.5 * (self.n - 1)
        return self.rot * tc, self.vx * tc, self.vy * tc

    def frame_torch(self, t: int):
        import torch
        import torch.nn.functional as F

        th, dx, dy = self._pose(t)
        cx, cy = 0.5 * (self.w - 1), 0.5 * (self.h - 1)
        c, s = math.cos(th), math.sin(th)
        X, Y = self._xs - cx, self._ys - cy
        u = c * X - s * Y + dx + cx + self.margin
        v = s * X + c * Y + dy + cy + self.margin
        H, W = self.canvas.shape[-2:]
        grid = torch.stack([u * (2.0 / (W - 1)) - 1.0, v * (2.0 / (H - 1)) - 1.0], dim=-1)[None]
        img = F.grid_sample(self.canvas, grid, mode="bicubic", padding_mode="reflection", align_corners=True)
        img = img[0].clamp_(0.0, 255.0).round_().to(torch.uint8)  # 3,H,W
        return img.permute(1, 2, 0).contiguous()

    def frame(self, t: int) -> np.ndarray:
        return self.frame_torch(t).cpu().numpy()

    def flow(self, pts: np.ndarray, t0: int, t1: int) -> np.ndarray:
        """Analytic position in frame t1 of scene points seen at `pts` (N,2) in frame t0."""
        th0, dx0, dy0 = self._pose(t0)
        th1, dx1, dy1 = self._pose(t1)
        cx, cy = 0.5 * (self.w - 1), 0.5 * (self.h - 1)
        p = np.asarray(pts, dtype=np.float64) - [cx, cy]
        c0, s0 = math.cos(th0), math.sin(th0)
        ux = c0 * p[:, 0] - s0 * p[:, 1] + dx0 - dx1
        uy = s0 * p[:, 0] + c0 * p[:, 1] + dy0 - dy1
        c1, s1 = math.cos(-th1), math.sin(-th1)
        return np.stack([c1 * ux - s1 * uy + cx, s1 * ux + c1 * uy + cy], axis=1)
