"""ctypes binding of the C ABI in include/polychase_hip.h (polychase_amd/lib/libpolychase_hip.so).

Thin and literal: one Python method per C entry point.  There is NO fallback: importing works
without a GPU (so symbol checks can run on CPU), but `Context()` raises when no gfx950 device is
usable, and `load()` raises when the library has not been built.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build

PC_MAX_TARGETS = 8
KERNEL_CLASSES = ["gray", "pyramid", "min_eig", "nms", "sort", "suppress", "lk", "compact"]

# every symbol include/polychase_hip.h declares (tests check they are all exported)
SYMBOLS = [
    "pc_gftt_default_options", "pc_flow_default_options", "pc_last_error", "pc_version",
    "pc_context_create", "pc_context_destroy", "pc_context_synchronize", "pc_context_stream",
    "pc_runtime_init", "pc_context_set_arithmetic", "pc_context_get_arithmetic", "pc_context_download",
    "pc_context_enable_timing", "pc_context_get_timing", "pc_context_get_busy_time", "pc_context_reset_timing",
    "pc_debug_lk_profile", "pc_debug_lk_x86_stats", "pc_debug_llt9",
    "pc_frame_create", "pc_frame_destroy", "pc_frame_set_rgb", "pc_frame_set_rgb_f32", "pc_frame_set_gray",
    "pc_host_buffer_alloc", "pc_host_buffer_free",
    "pc_frame_num_levels", "pc_frame_level_size", "pc_frame_download_gray", "pc_frame_download_level",
    "pc_frame_download_deriv", "pc_frame_detect", "pc_frame_download_min_eig", "pc_frame_num_candidates",
    "pc_frame_num_keypoints", "pc_frame_download_keypoints", "pc_frame_set_keypoints",
    "pc_lk_track", "pc_lk_track_filtered",
    "pc_analyzer_create", "pc_analyzer_destroy", "pc_analyzer_reset", "pc_analyzer_put_frame", "pc_analyzer_put_frame_f32",
    "pc_analyzer_has_frame", "pc_analyzer_frame_ingested",
    "pc_analyzer_set_keypoints", "pc_analyzer_submit", "pc_analyzer_pending", "pc_analyzer_collect",
    "pc_analyzer_set_device_log", "pc_analyzer_device_log_used", "pc_analyzer_redirect_device_log",
    "pc_analyzer_set_host_records",
    "pc_peer_buffer_alloc", "pc_peer_buffer_free", "pc_peer_buffer_export", "pc_peer_buffer_open", "pc_peer_buffer_close",
    "pc_peer_copy_async", "pc_peer_buffer_download",
    "pc_comm_unique_id", "pc_comm_create", "pc_comm_destroy", "pc_comm_world_size", "pc_comm_rank", "pc_comm_all_gather_log",
    "pc_comm_send", "pc_comm_recv",
    "pc_mesh_create", "pc_mesh_set_mask", "pc_mesh_destroy", "pc_raycast_pixels", "pc_raycast_pixels_sweep",
    "pc_corr_set_create", "pc_corr_set_destroy", "pc_corr_set_clear", "pc_corr_set_recycle", "pc_corr_set_append", "pc_corr_set_size",
    "pc_corr_set_download", "pc_pnp_problem_from_set",
    "pc_pnp_problem_create", "pc_pnp_problem_destroy", "pc_pnp_normal_equations", "pc_pnp_normal_equations_cost",
    "pc_pnp_solve", "pc_pnp_total_cost", "pc_track_solve_frame", "pc_track_frame_upload", "pc_track_frame_launch", "pc_track_frame_launch_chained", "pc_corr_set_reserve", "pc_context_pci_bus_id", "pc_track_frame_finish",
    "pc_track_download_points",
    "pc_refine_problem_create", "pc_refine_problem_create_parts", "pc_refine_problem_destroy", "pc_refine_total_cost", "pc_refine_normal_equations", "pc_refine_problem_timing",
]


ARITH_CANONICAL, ARITH_LK_X86_ORDER, ARITH_SOBEL_FMA, ARITH_OPENCV_X86 = 0, 1, 2, 3
ARITH_SOBEL_ROW_FMA = 4   # on top of the others (include/polychase_hip.h: PC_ARITH_SOBEL_ROW_FMA)


class GfttOptions(C.Structure):
    """GFTTOptions (reference cpp/feature_detection/gftt.h:5-21)."""
    _fields_ = [("quality_level", C.c_double), ("min_distance", C.c_double), ("block_size", C.c_int),
                ("gradient_size", C.c_int), ("max_corners", C.c_int), ("use_harris", C.c_int),
                ("harris_k", C.c_double), ("grid_rows", C.c_int), ("grid_cols", C.c_int)]


class FlowOptions(C.Structure):
    """OpticalFlowOptions (reference cpp/opticalflow.h:27-33)."""
    _fields_ = [("window_size", C.c_int), ("max_level", C.c_int), ("term_max_iters", C.c_int),
                ("term_epsilon", C.c_double), ("min_eigen_threshold", C.c_double)]


class FrameResult(C.Structure):
    """pc_frame_result (include/polychase_hip.h)."""
    _fields_ = [("frame1", C.c_int32), ("n_keypoints", C.c_int32), ("keypoints_detected", C.c_int32),
                ("keypoints_xy", C.POINTER(C.c_float)), ("n_targets", C.c_int32),
                ("targets", C.c_int32 * 8), ("row_offset", C.c_int64 * 9),
                ("src_indices", C.POINTER(C.c_uint32)), ("tgt_xy", C.POINTER(C.c_float)),
                ("flow_err", C.POINTER(C.c_float))]


class PolychaseHipError(RuntimeError):
    pass


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    path = _build.hip_library_path()
    if not os.path.exists(path):
        raise PolychaseHipError(
            f"{path} is missing: run `python -m polychase_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback for the hot path.")
    # One HIP runtime per process: torch wheels bundle their own libamdhip64.so (SONAME
    # libamdhip64.so.7, same as /opt/rocm's).  Loading torch FIRST makes our DT_NEEDED entry resolve
    # to the copy torch already mapped; the other order maps two runtimes and the second one finds
    # "no HIP GPUs".  Without torch the library uses /opt/rocm/lib via its RUNPATH.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the C ABI
        pass
    L = C.CDLL(path)
    vp, ip = C.c_void_p, C.POINTER(C.c_int)
    L.pc_runtime_init.argtypes = [ip, ip]
    L.pc_runtime_init(None, None)   # GPU_MAX_HW_QUEUES before this library's first HIP call (include/polychase_hip.h)
    L.pc_last_error.restype = C.c_char_p
    L.pc_version.restype = C.c_char_p
    L.pc_gftt_default_options.argtypes = [C.POINTER(GfttOptions)]
    L.pc_flow_default_options.argtypes = [C.POINTER(FlowOptions)]
    L.pc_context_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.pc_context_destroy.argtypes = [vp]
    L.pc_context_destroy.restype = None
    L.pc_context_synchronize.argtypes = [vp]
    L.pc_context_stream.argtypes = [vp]
    L.pc_context_download.argtypes = [vp, vp, vp, C.c_size_t]
    L.pc_context_set_arithmetic.argtypes = [vp, C.c_int]
    L.pc_context_get_arithmetic.argtypes = [vp]
    L.pc_context_stream.restype = vp
    L.pc_context_enable_timing.argtypes = [vp, C.c_int]
    L.pc_context_get_timing.argtypes = [vp, C.c_int, ip, C.POINTER(C.c_double)]
    L.pc_context_get_busy_time.argtypes = [vp, C.c_int, C.POINTER(C.c_double)]
    L.pc_context_reset_timing.argtypes = [vp]
    L.pc_debug_lk_profile.argtypes = [vp, C.POINTER(C.c_ulonglong)]
    L.pc_debug_lk_x86_stats.argtypes = [vp, C.c_int, C.POINTER(C.c_ulonglong)]
    L.pc_debug_llt9.argtypes = [vp, vp, vp, vp, vp, ip]
    L.pc_frame_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.pc_frame_destroy.argtypes = [vp]
    L.pc_frame_destroy.restype = None
    L.pc_frame_set_rgb.argtypes = [vp, vp, vp, C.c_size_t, C.c_int]
    L.pc_frame_set_gray.argtypes = [vp, vp, vp, C.c_size_t, C.c_int]
    L.pc_frame_set_rgb_f32.argtypes = [vp, vp, vp, C.c_size_t, C.c_int, C.c_int]
    L.pc_frame_num_levels.argtypes = [vp]
    L.pc_frame_level_size.argtypes = [vp, C.c_int, ip, ip]
    L.pc_frame_download_gray.argtypes = [vp, vp, vp]
    L.pc_frame_download_level.argtypes = [vp, vp, C.c_int, vp]
    L.pc_frame_download_deriv.argtypes = [vp, vp, C.c_int, vp]
    L.pc_frame_detect.argtypes = [vp, vp, C.POINTER(GfttOptions)]
    L.pc_frame_download_min_eig.argtypes = [vp, vp, vp]
    L.pc_frame_num_candidates.argtypes = [vp, vp, ip]
    L.pc_frame_num_keypoints.argtypes = [vp, vp, ip]
    L.pc_frame_download_keypoints.argtypes = [vp, vp, vp, C.c_int]
    L.pc_frame_set_keypoints.argtypes = [vp, vp, vp, C.c_int]
    L.pc_lk_track.argtypes = [vp, vp, C.POINTER(vp), C.c_int, C.POINTER(FlowOptions), vp, vp, vp]
    L.pc_lk_track_filtered.argtypes = [vp, vp, C.POINTER(vp), C.c_int, C.POINTER(FlowOptions), vp, vp, vp, vp]
    L.pc_analyzer_create.argtypes = [vp, C.c_int, C.c_int, C.POINTER(GfttOptions), C.POINTER(FlowOptions), C.c_int,
                                     C.c_int, C.POINTER(vp)]
    L.pc_analyzer_destroy.argtypes = [vp]
    L.pc_analyzer_destroy.restype = None
    L.pc_analyzer_put_frame.argtypes = [vp, C.c_int32, vp, C.c_size_t, C.c_int, C.c_int]
    L.pc_analyzer_put_frame_f32.argtypes = [vp, C.c_int32, vp, C.c_size_t, C.c_int, C.c_int, C.c_int]
    L.pc_analyzer_has_frame.argtypes = [vp, C.c_int32]
    L.pc_analyzer_frame_ingested.argtypes = [vp, C.c_int32]
    L.pc_analyzer_set_keypoints.argtypes = [vp, C.c_int32, vp, C.c_int]
    L.pc_analyzer_submit.argtypes = [vp, C.c_int32, C.POINTER(C.c_int32), C.c_int]
    L.pc_analyzer_pending.argtypes = [vp]
    L.pc_analyzer_collect.argtypes = [vp, C.POINTER(FrameResult)]
    L.pc_analyzer_set_device_log.argtypes = [vp, vp, C.c_size_t]
    L.pc_analyzer_device_log_used.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.pc_analyzer_reset.argtypes = [vp]
    L.pc_analyzer_redirect_device_log.argtypes = [vp, vp, C.c_size_t]
    L.pc_analyzer_set_host_records.argtypes = [vp, C.c_int]
    L.pc_peer_buffer_alloc.argtypes = [C.c_int, C.c_size_t, C.POINTER(vp)]
    L.pc_peer_buffer_free.argtypes = [C.c_int, vp]
    L.pc_peer_buffer_export.argtypes = [C.c_int, vp, C.c_char_p]
    L.pc_peer_buffer_open.argtypes = [C.c_int, C.c_char_p, C.POINTER(vp)]
    L.pc_peer_buffer_close.argtypes = [C.c_int, vp]
    L.pc_peer_copy_async.argtypes = [C.c_int, vp, vp, C.c_size_t, vp]
    L.pc_peer_buffer_download.argtypes = [C.c_int, vp, vp, C.c_size_t]
    L.pc_comm_unique_id.argtypes = [vp]
    L.pc_comm_create.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.pc_comm_destroy.argtypes = [vp]
    L.pc_comm_destroy.restype = None
    L.pc_comm_world_size.argtypes = [vp]
    L.pc_comm_rank.argtypes = [vp]
    L.pc_comm_all_gather_log.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.pc_comm_send.argtypes = [vp, vp, C.c_uint64, C.c_int]
    L.pc_comm_recv.argtypes = [vp, vp, C.c_uint64, C.c_int]
    _lib = L
    return L


def _check(rc: int):
    if rc != 0:
        raise PolychaseHipError(f"polychase_hip error {rc}: {load().pc_last_error().decode()}")


def gftt_options(**kw) -> GfttOptions:
    o = GfttOptions()
    load().pc_gftt_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def flow_options(**kw) -> FlowOptions:
    o = FlowOptions()
    load().pc_flow_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class Context:
    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        _check(load().pc_context_create(device, C.byref(self._h)))

    def close(self):
        if self._h:
            load().pc_context_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_arithmetic(self, flags: int):
        """ARITH_CANONICAL | ARITH_LK_X86_ORDER | ARITH_SOBEL_FMA | ARITH_SOBEL_ROW_FMA (include/polychase_hip.h: pc_context_set_arithmetic)."""
        _check(load().pc_context_set_arithmetic(self._h, int(flags)))

    @property
    def arithmetic(self) -> int:
        return load().pc_context_get_arithmetic(self._h)

    def synchronize(self):
        _check(load().pc_context_synchronize(self._h))

    @property
    def stream(self) -> int:
        return load().pc_context_stream(self._h)

    def enable_timing(self, classes=True):
        """classes: True = all, False = none, or an iterable of KERNEL_CLASSES names."""
        if classes is True:
            mask = 0xFF
        elif not classes:
            mask = 0
        else:
            mask = sum(1 << KERNEL_CLASSES.index(c) for c in classes)
        _check(load().pc_context_enable_timing(self._h, mask))

    def reset_timing(self):
        _check(load().pc_context_reset_timing(self._h))

    def timing(self) -> dict:
        out = {}
        for k, name in enumerate(KERNEL_CLASSES):
            n, ms = C.c_int(), C.c_double()
            _check(load().pc_context_get_timing(self._h, k, C.byref(n), C.byref(ms)))
            out[name] = (n.value, ms.value)
        return out

    def lk_profile(self) -> list:
        """per-phase cycle sums of the LK kernel since the last call (zeros unless the library was built with -DPC_LK_PROFILE)"""
        out = (C.c_ulonglong * 16)()
        _check(load().pc_debug_lk_profile(self._h, out))
        return list(out)

    def lk_x86_stats(self, enable: bool = True) -> dict:
        """counters of the x86 summation order on the two-keypoint LK kernel since counting was enabled
        (include/polychase_hip.h: pc_debug_lk_x86_stats); enable=True restarts counting, False stops it"""
        out = (C.c_ulonglong * 4)()
        _check(load().pc_debug_lk_x86_stats(self._h, 1 if enable else 0, out))
        return {"iterations_proven_exact": out[0], "iterations_x86_order": out[1], "keypoint_levels": out[2],
                "keypoint_levels_x86_order": out[3]}

    def llt9(self, a: np.ndarray, b: np.ndarray):
        """the device solver's 9x9 float32 Cholesky + solve -> (L, x, positive_definite)"""
        a = np.ascontiguousarray(a, np.float32).reshape(9, 9)
        b = np.ascontiguousarray(b, np.float32).reshape(9)
        l, x, ok = np.zeros((9, 9), np.float32), np.zeros(9, np.float32), C.c_int()
        _check(load().pc_debug_llt9(self._h, a.ctypes.data, b.ctypes.data, l.ctypes.data, x.ctypes.data, C.byref(ok)))
        return l, x, bool(ok.value)

    def busy_ms(self, kernel_class: str) -> float:
        """wall time during which at least one launch of the class was executing (launches may overlap)"""
        ms = C.c_double()
        _check(load().pc_context_get_busy_time(self._h, KERNEL_CLASSES.index(kernel_class), C.byref(ms)))
        return ms.value


def _ptr(a):
    """numpy array (host) or torch tensor (host or device) -> (address, is_device, row_pitch_bytes)."""
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data, 0, a.strides[0]
    # torch tensor
    assert a.is_contiguous()
    if a.is_cuda:
        # the library enqueues on its own HIP stream: the tensor must be complete before it is read
        import torch
        torch.cuda.current_stream(a.device).synchronize()
    return a.data_ptr(), 1 if a.is_cuda else 0, a.stride(0) * a.element_size()


def _is_float32(a) -> bool:
    return str(a.dtype) in ("float32", "torch.float32")


class Frame:
    """Gray image + LK pyramid + keypoints of one video frame, resident in HBM."""

    def __init__(self, ctx: Context, width: int, height: int, window_size: int = 10, max_level: int = 3):
        self.ctx, self.w, self.h, self.win = ctx, width, height, window_size
        self._h = C.c_void_p()
        _check(load().pc_frame_create(ctx._h, width, height, window_size, max_level, C.byref(self._h)))

    def close(self):
        if self._h and self.ctx._h:
            load().pc_frame_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_rgb(self, rgb):
        """uint8 (H, W, 3), or float32 (H, W, 3|4) as Blender hands frames out (converted on the GPU)."""
        p, dev, pitch = _ptr(rgb)
        if _is_float32(rgb):
            assert tuple(rgb.shape[:2]) == (self.h, self.w) and len(rgb.shape) == 3, rgb.shape
            _check(load().pc_frame_set_rgb_f32(self.ctx._h, self._h, p, pitch, int(rgb.shape[2]), dev))
        else:
            assert tuple(rgb.shape) == (self.h, self.w, 3), rgb.shape
            _check(load().pc_frame_set_rgb(self.ctx._h, self._h, p, pitch, dev))
        self._keep = rgb  # device sources must outlive the async kernels

    def set_gray(self, gray):
        assert tuple(gray.shape) == (self.h, self.w), gray.shape
        p, dev, pitch = _ptr(gray)
        _check(load().pc_frame_set_gray(self.ctx._h, self._h, p, pitch, dev))
        self._keep = gray

    @property
    def num_levels(self) -> int:
        return load().pc_frame_num_levels(self._h)

    def level_size(self, level: int):
        w, h = C.c_int(), C.c_int()
        _check(load().pc_frame_level_size(self._h, level, C.byref(w), C.byref(h)))
        return w.value, h.value

    def gray(self) -> np.ndarray:
        out = np.empty((self.h, self.w), np.uint8)
        _check(load().pc_frame_download_gray(self.ctx._h, self._h, out.ctypes.data))
        return out

    def level(self, l: int) -> np.ndarray:
        w, h = self.level_size(l)
        out = np.empty((h + 2 * self.win, w + 2 * self.win), np.uint8)
        _check(load().pc_frame_download_level(self.ctx._h, self._h, l, out.ctypes.data))
        return out

    def deriv(self, l: int) -> np.ndarray:
        w, h = self.level_size(l)
        out = np.empty((h + 2 * self.win, w + 2 * self.win, 2), np.int16)
        _check(load().pc_frame_download_deriv(self.ctx._h, self._h, l, out.ctypes.data))
        return out

    def detect(self, opt: GfttOptions | None = None):
        opt = opt or gftt_options()
        _check(load().pc_frame_detect(self.ctx._h, self._h, C.byref(opt)))

    def min_eig(self) -> np.ndarray:
        out = np.empty((self.h, self.w), np.float32)
        _check(load().pc_frame_download_min_eig(self.ctx._h, self._h, out.ctypes.data))
        return out

    @property
    def num_candidates(self) -> int:
        n = C.c_int()
        _check(load().pc_frame_num_candidates(self.ctx._h, self._h, C.byref(n)))
        return n.value

    @property
    def num_keypoints(self) -> int:
        n = C.c_int()
        _check(load().pc_frame_num_keypoints(self.ctx._h, self._h, C.byref(n)))
        return n.value

    def keypoints(self) -> np.ndarray:
        n = self.num_keypoints
        out = np.empty((n, 2), np.float32)
        _check(load().pc_frame_download_keypoints(self.ctx._h, self._h, out.ctypes.data, n))
        return out

    def set_keypoints(self, xy: np.ndarray):
        xy = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
        _check(load().pc_frame_set_keypoints(self.ctx._h, self._h, xy.ctypes.data, len(xy)))


def lk_track(ctx: Context, frame1: Frame, targets: list[Frame], opt: FlowOptions | None = None):
    """Raw LK outputs: next_xy [T,N,2], status [T,N], err [T,N]."""
    opt = opt or flow_options()
    n, t = frame1.num_keypoints, len(targets)
    xy = np.zeros((t, n, 2), np.float32)
    st = np.zeros((t, n), np.uint8)
    err = np.zeros((t, n), np.float32)
    arr = (C.c_void_p * t)(*[f._h for f in targets])
    _check(load().pc_lk_track(ctx._h, frame1._h, arr, t, C.byref(opt), xy.ctypes.data, st.ctypes.data,
                              err.ctypes.data))
    return xy, st, err


def lk_track_filtered(ctx: Context, frame1: Frame, targets: list[Frame], opt: FlowOptions | None = None):
    """status==1 rows per target: list of (src_indices u32 [M], tgt_xy f32 [M,2], err f32 [M])."""
    opt = opt or flow_options()
    n, t = frame1.num_keypoints, len(targets)
    rows = max(1, n * t)
    idx = np.zeros(rows, np.uint32)
    xy = np.zeros((rows, 2), np.float32)
    err = np.zeros(rows, np.float32)
    off = np.zeros(t + 1, np.int64)
    arr = (C.c_void_p * t)(*[f._h for f in targets])
    _check(load().pc_lk_track_filtered(ctx._h, frame1._h, arr, t, C.byref(opt), idx.ctypes.data, xy.ctypes.data,
                                       err.ctypes.data, off.ctypes.data))
    out = []
    for k in range(t):
        a, b = int(off[k]), int(off[k + 1])
        out.append((idx[a:b].copy(), xy[a:b].copy(), err[a:b].copy()))
    return out


class Analyzer:
    """pc_analyzer: pipelined per-clip engine (ring of resident frames + asynchronous frame1 jobs)."""

    def __init__(self, ctx: Context, width: int, height: int, gftt: GfttOptions | None = None,
                 flow: FlowOptions | None = None, ring_frames: int = 17, max_jobs: int = 3):
        self.ctx, self.w, self.h = ctx, width, height
        self.gftt = gftt or gftt_options()
        self.flow = flow or flow_options()
        self._h = C.c_void_p()
        self._keep = {}
        _check(load().pc_analyzer_create(ctx._h, width, height, C.byref(self.gftt), C.byref(self.flow), ring_frames,
                                         max_jobs, C.byref(self._h)))
        self.ring = ring_frames

    def close(self):
        if self._h and self.ctx._h:
            load().pc_analyzer_destroy(self._h)
        self._h = C.c_void_p()
        self._keep = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        """No resident frame, no job, no log; allocations kept (pc_analyzer_reset)."""
        _check(load().pc_analyzer_reset(self._h))
        self._keep = {}

    def put_frame(self, frame_id: int, rgb, will_detect: bool = True):
        p, dev, pitch = _ptr(rgb)
        if _is_float32(rgb):
            assert tuple(rgb.shape[:2]) == (self.h, self.w) and len(rgb.shape) == 3, rgb.shape
            _check(load().pc_analyzer_put_frame_f32(self._h, frame_id, p, pitch, int(rgb.shape[2]), dev,
                                                    1 if will_detect else 0))
        else:
            assert tuple(rgb.shape) == (self.h, self.w, 3), rgb.shape
            _check(load().pc_analyzer_put_frame(self._h, frame_id, p, pitch, dev, 1 if will_detect else 0))
        if dev:
            self._keep[frame_id % self.ring] = rgb  # device sources must outlive the async kernels

    def has_frame(self, frame_id: int) -> bool:
        return bool(load().pc_analyzer_has_frame(self._h, frame_id))

    def set_keypoints(self, frame_id: int, xy: np.ndarray):
        xy = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
        _check(load().pc_analyzer_set_keypoints(self._h, frame_id, xy.ctypes.data, len(xy)))

    def submit(self, frame1: int, targets):
        t = list(targets)
        arr = (C.c_int32 * max(1, len(t)))(*t)
        _check(load().pc_analyzer_submit(self._h, frame1, arr, len(t)))

    @property
    def pending(self) -> int:
        return load().pc_analyzer_pending(self._h)

    def set_device_log(self, tensor):
        """tensor: torch uint8 CUDA tensor (or None) that receives the device-resident record log."""
        if tensor is None:
            _check(load().pc_analyzer_set_device_log(self._h, None, 0))
            self._log = None
            return
        assert tensor.is_cuda and tensor.is_contiguous() and tensor.element_size() == 1
        _check(load().pc_analyzer_set_device_log(self._h, tensor.data_ptr(), tensor.numel()))
        self._log = tensor

    def redirect_device_log(self, tensor):
        """The following jobs append to `tensor` from offset 0; nothing is waited for (pc_analyzer_redirect_device_log)."""
        assert tensor.is_cuda and tensor.is_contiguous() and tensor.element_size() == 1
        _check(load().pc_analyzer_redirect_device_log(self._h, tensor.data_ptr(), tensor.numel()))
        self._log = tensor

    def set_host_records(self, enabled: bool):
        _check(load().pc_analyzer_set_host_records(self._h, 1 if enabled else 0))

    @property
    def device_log_used(self) -> int:
        n = C.c_size_t()
        _check(load().pc_analyzer_device_log_used(self._h, C.byref(n)))
        return n.value

    def collect_raw(self) -> FrameResult:
        r = FrameResult()
        _check(load().pc_analyzer_collect(self._h, C.byref(r)))
        return r

    def collect(self, copy: bool = True):
        """-> (frame1, keypoints [N,2], detected, {frame2: (src_idx, tgt_xy, err)})"""
        r = self.collect_raw()
        n = r.n_keypoints
        if n and not r.keypoints_xy:     # set_host_records(False): the records left through the device log only
            return r.frame1, None, bool(r.keypoints_detected), {int(r.targets[t]): None for t in range(r.n_targets)}
        kps = np.ctypeslib.as_array(r.keypoints_xy, shape=(n, 2)) if n else np.zeros((0, 2), np.float32)
        flows = {}
        for t in range(r.n_targets):
            a, b = int(r.row_offset[t]), int(r.row_offset[t + 1])
            if b > a:
                idx = np.ctypeslib.as_array(r.src_indices, shape=(b,))[a:b]
                xy = np.ctypeslib.as_array(r.tgt_xy, shape=(b, 2))[a:b]
                err = np.ctypeslib.as_array(r.flow_err, shape=(b,))[a:b]
            else:
                idx, xy, err = np.zeros(0, np.uint32), np.zeros((0, 2), np.float32), np.zeros(0, np.float32)
            flows[int(r.targets[t])] = (idx.copy(), xy.copy(), err.copy()) if copy else (idx, xy, err)
        return r.frame1, (kps.copy() if copy else kps), bool(r.keypoints_detected), flows
