"""ctypes binding of oracle/libpc_oracle.so (the CPU restatement). TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "libpc_oracle.so")


class GfttOptions(C.Structure):
    _fields_ = [("quality_level", C.c_double), ("min_distance", C.c_double), ("block_size", C.c_int),
                ("gradient_size", C.c_int), ("max_corners", C.c_int), ("use_harris", C.c_int),
                ("harris_k", C.c_double), ("grid_rows", C.c_int), ("grid_cols", C.c_int)]


class FlowOptions(C.Structure):
    _fields_ = [("window_size", C.c_int), ("max_level", C.c_int), ("term_max_iters", C.c_int),
                ("term_epsilon", C.c_double), ("min_eigen_threshold", C.c_double)]


_RECORD_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int32, C.c_int32, C.c_int,
                         C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.POINTER(C.c_float))

_lib = None


def lib(path: str | None = None):
    global _lib
    if _lib is not None and path is None:
        return _lib
    so = path or _SO
    if path is None and not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle"), "libpc_oracle.so"])
    L = C.CDLL(so)
    u8p, f32p, i16p = C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_int16)
    L.pco_rgb2gray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.pco_min_eigen_val.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.pco_min_eigen_val.restype = C.c_int
    L.pco_gftt_default_options.argtypes = [C.POINTER(GfttOptions)]
    L.pco_flow_default_options.argtypes = [C.POINTER(FlowOptions)]
    L.pco_gftt.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(GfttOptions), C.c_void_p, C.c_int,
                           C.c_void_p, C.POINTER(C.c_int)]
    L.pco_gftt.restype = C.c_int
    L.pco_pyramid_build.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.pco_pyramid_build.restype = C.c_void_p
    L.pco_pyramid_free.argtypes = [C.c_void_p]
    L.pco_pyramid_num_levels.argtypes = [C.c_void_p]
    L.pco_pyramid_num_levels.restype = C.c_int
    L.pco_pyramid_win.argtypes = [C.c_void_p]
    L.pco_pyramid_win.restype = C.c_int
    L.pco_pyramid_level_size.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.pco_pyramid_image.argtypes = [C.c_void_p, C.c_int]
    L.pco_pyramid_image.restype = u8p
    L.pco_pyramid_deriv.argtypes = [C.c_void_p, C.c_int]
    L.pco_pyramid_deriv.restype = i16p
    L.pco_lk.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double,
                         C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pco_analyze_clip.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int32, C.c_int32,
                                   C.c_int32, C.POINTER(GfttOptions), C.POINTER(FlowOptions), C.c_int,
                                   C.c_int, _RECORD_CB, C.c_void_p]
    L.pco_analyze_clip.restype = C.c_int
    L.pco_set_opencv_emulation.argtypes = [C.c_int]
    L.pco_get_opencv_emulation.restype = C.c_int
    if path is None:
        _lib = L
    return L


EMU_CANONICAL, EMU_LK_SIMD, EMU_SOBEL_FMA, EMU_OPENCV_X86 = 0, 1, 2, 3   # the default is EMU_OPENCV_X86 (pc_oracle.c)
EMU_SOBEL_ROW_FMA = 4   # hypothesis flag on top: Dy's 8u -> 32f row smoothing as a fused chain (pc_oracle.h)


class emulation:
    """with oracle.emulation(flags): the oracle runs in that execution order of OpenCV (pc_oracle.c; EMU_CANONICAL = 0,
    default EMU_OPENCV_X86)."""

    def __init__(self, flags: int):
        self.flags = flags

    def __enter__(self):
        self.prev = lib().pco_get_opencv_emulation()
        lib().pco_set_opencv_emulation(self.flags)
        return self

    def __exit__(self, *exc):
        lib().pco_set_opencv_emulation(self.prev)
        return False


def gftt_options(**kw) -> GfttOptions:
    o = GfttOptions()
    lib().pco_gftt_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def flow_options(**kw) -> FlowOptions:
    o = FlowOptions()
    lib().pco_flow_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def rgb2gray(rgb: np.ndarray) -> np.ndarray:
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    out = np.empty((h, w), np.uint8)
    lib().pco_rgb2gray(rgb.ctypes.data, w, h, out.ctypes.data)
    return out


def corner_harris(gray: np.ndarray, block_size=3, ksize=3, k=0.04) -> np.ndarray:
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    out = np.empty((h, w), np.float32)
    lib().pco_corner_harris.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
    rc = lib().pco_corner_harris(gray.ctypes.data, w, h, block_size, ksize, k, out.ctypes.data)
    assert rc == 0
    return out


def min_eigen_val(gray: np.ndarray, block_size=3, ksize=3) -> np.ndarray:
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    out = np.empty((h, w), np.float32)
    rc = lib().pco_min_eigen_val(gray.ctypes.data, w, h, block_size, ksize, out.ctypes.data)
    assert rc == 0
    return out


def gftt(gray: np.ndarray, opt: GfttOptions | None = None, want_eig=False):
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    opt = opt or gftt_options()
    cap = h * w // 4 + 16
    xy = np.empty((cap, 2), np.float32)
    eig = np.empty((h, w), np.float32) if want_eig else None
    ncand = C.c_int(0)
    n = lib().pco_gftt(gray.ctypes.data, w, h, C.byref(opt), xy.ctypes.data, cap,
                       eig.ctypes.data if want_eig else None, C.byref(ncand))
    if n < 0:
        raise RuntimeError(f"pco_gftt failed: {n}")
    if want_eig:
        return xy[:n].copy(), eig, ncand.value
    return xy[:n].copy()


class Pyramid:
    def __init__(self, gray: np.ndarray, win=10, max_level=3):
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        h, w = gray.shape
        self._p = lib().pco_pyramid_build(gray.ctypes.data, w, h, win, max_level)
        assert self._p
        self.win = win
        self.num_levels = lib().pco_pyramid_num_levels(self._p)

    def level_size(self, l):
        w, h = C.c_int(), C.c_int()
        lib().pco_pyramid_level_size(self._p, l, C.byref(w), C.byref(h))
        return w.value, h.value

    def image(self, l, padded=True) -> np.ndarray:
        w, h = self.level_size(l)
        pw, ph = w + 2 * self.win, h + 2 * self.win
        a = np.ctypeslib.as_array(lib().pco_pyramid_image(self._p, l), shape=(ph, pw)).copy()
        return a if padded else a[self.win:-self.win, self.win:-self.win]

    def deriv(self, l, padded=True) -> np.ndarray:
        w, h = self.level_size(l)
        pw, ph = w + 2 * self.win, h + 2 * self.win
        a = np.ctypeslib.as_array(lib().pco_pyramid_deriv(self._p, l), shape=(ph, pw, 2)).copy()
        return a if padded else a[self.win:-self.win, self.win:-self.win]

    def __del__(self):
        if getattr(self, "_p", None):
            lib().pco_pyramid_free(self._p)
            self._p = None


def lk(prev: Pyramid, nxt: Pyramid, pts: np.ndarray, opt: FlowOptions | None = None):
    opt = opt or flow_options()
    pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 2)
    n = len(pts)
    out = np.zeros((n, 2), np.float32)
    st = np.zeros(n, np.uint8)
    err = np.zeros(n, np.float32)
    lib().pco_lk(prev._p, nxt._p, pts.ctypes.data, n, opt.max_level, opt.term_max_iters, opt.term_epsilon,
                 opt.min_eigen_threshold, out.ctypes.data, st.ctypes.data, err.ctypes.data)
    return out, st, err


def analyze_clip(frames, first_frame=1, f1_range=None, gopt=None, fopt=None, threads=1, feature_threads=1,
                 libpath=None):
    """Run the reference-shaped CPU path. Returns ({frame: xy}, {(from,to): (idx, xy, err)})."""
    L = lib(libpath)
    frames = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames]
    h, w, _ = frames[0].shape
    n = len(frames)
    ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
    gopt = gopt or gftt_options()
    fopt = fopt or flow_options()
    kps, flows = {}, {}

    def cb(user, kind, f_from, f_to, rows, idx, xy, err):
        if kind == 0:
            kps[f_from] = np.ctypeslib.as_array(xy, shape=(rows, 2)).copy() if rows else np.zeros((0, 2), np.float32)
        else:
            if rows:
                flows[(f_from, f_to)] = (np.ctypeslib.as_array(idx, shape=(rows,)).copy(),
                                         np.ctypeslib.as_array(xy, shape=(rows, 2)).copy(),
                                         np.ctypeslib.as_array(err, shape=(rows,)).copy())
            else:
                flows[(f_from, f_to)] = (np.zeros(0, np.uint32), np.zeros((0, 2), np.float32), np.zeros(0, np.float32))

    cbf = _RECORD_CB(cb)
    b, e = f1_range if f1_range else (first_frame, first_frame + n)
    rc = L.pco_analyze_clip(ptrs, n, w, h, first_frame, b, e, C.byref(gopt), C.byref(fopt), threads,
                            feature_threads, cbf, None)
    assert rc == 0
    return kps, flows
