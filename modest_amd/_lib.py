"""ctypes binding of libmodest_hip.so (the C ABI declared in include/modest_hip.h).

There is no CPU fallback: if the shared library is missing, or no HIP device
is visible when a context is requested, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("MODEST_HIP_LIB", _PKG / "lib" / "libmodest_hip.so"))

c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_u8p = C.POINTER(C.c_uint8)
VP = C.c_void_p

# name -> (restype, argtypes); mirrors include/modest_hip.h one to one.
SIGNATURES = {
    "modest_version": (C.c_int, []),
    "modest_last_error": (C.c_char_p, []),
    "modest_device_count": (C.c_int, []),
    "modest_ctx_create": (C.c_int, [C.c_int, C.POINTER(VP)]),
    "modest_ctx_destroy": (C.c_int, [VP]),
    "modest_ctx_reserve_arena": (C.c_int, [VP, C.c_uint64]),
    "modest_warmup": (C.c_int, [VP]),
    "modest_ctx_profile_begin": (C.c_int, [VP, C.c_int]),
    "modest_ctx_profile_collect": (C.c_int, [VP, VP, C.c_int, VP]),
    "modest_transform_points": (C.c_int, [VP, VP, C.c_int64, C.c_int, VP, C.c_int, VP, VP, VP]),
    "modest_pp_count": (C.c_int, [VP, VP, C.c_int, VP, VP, C.c_int, C.c_double, VP, VP]),
    "modest_pp_entropy": (C.c_int, [VP, VP, C.c_int, C.c_int, VP, VP]),
    "modest_pp_score": (C.c_int, [VP, VP, C.c_int, VP, VP, C.c_int, C.c_double, VP, VP, VP]),
    "modest_frame_table_tiles": (C.c_int, []),
    "modest_frame_sort": (C.c_int, [VP, VP, C.c_int, VP, VP]),
    "modest_frame_sort_async": (C.c_int, [VP, VP, C.c_int, VP, VP, VP]),
    "modest_pp_score_frames": (C.c_int, [VP, VP, VP, VP, C.c_int, C.c_int, VP, C.c_double, VP, VP, VP]),
    "modest_pp_score_frames_batch": (C.c_int, [VP, C.c_int, VP, VP, VP, VP, C.c_int, C.c_double, VP, VP, VP]),
    "modest_pp_block_limits": (C.c_int, [VP, VP, VP]),
    "modest_pp_score_block": (C.c_int, [VP, VP, C.c_int, VP, C.c_int, C.c_int, C.c_double, C.c_double, VP]),
    "modest_pp_score_block_mixed": (C.c_int, [VP, VP, C.c_int, VP, C.c_int, VP, C.c_double, C.c_double, VP]),
    "modest_plane_candidates": (C.c_int, [VP, VP, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                          C.c_float, C.c_float, VP, VP, VP, VP]),
    "modest_mad_threshold": (C.c_int, [VP, VP, C.c_int, VP, VP]),
    "modest_mad_threshold_batch": (C.c_int, [VP, VP, VP, C.c_int, VP, VP]),
    "modest_ransac_score_trials": (C.c_int, [VP, VP, C.c_int, VP, C.c_int, C.c_float, VP, VP, VP, VP, VP]),
    "modest_ransac_trials": (C.c_int, [VP, VP, C.c_int, VP, C.c_int, VP, VP, VP, VP, VP, VP, VP]),
    "modest_ransac_refit": (C.c_int, [VP, VP, C.c_int, VP, C.c_float, VP, VP, VP]),
    "modest_plane_range_mask": (C.c_int, [VP, VP, C.c_int, C.c_int, VP, C.c_double, VP, VP, VP, VP, VP,
                                          VP, VP]),
    "modest_cluster_dbscan": (C.c_int, [VP, VP, VP, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                        VP, VP, VP, VP]),
    "modest_cluster_dbscan_ex": (C.c_int, [VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                           C.c_int, VP, VP, VP, VP]),
    "modest_project_velo_to_rect": (C.c_int, [VP, VP, C.c_int, C.c_int, VP, VP, VP, VP]),
    "modest_ransac_plane": (C.c_int, [VP, VP, C.c_int, C.c_float, VP, VP, C.c_int, C.c_double, C.c_int, VP, VP, VP, VP, VP,
                                      VP, VP]),
    "modest_mt19937_triplets": (C.c_int, [VP, VP, C.c_uint32, C.c_int, VP]),
    "modest_plane_prepare": (C.c_int, [VP, VP, C.c_int, C.c_int, VP, VP, VP, VP, VP, VP]),
    "modest_mask_stage": (C.c_int, [VP, VP, C.c_int, C.c_int, VP, VP, VP, VP, VP, VP, VP, VP, VP]),
    "modest_mask_cluster": (C.c_int, [VP, VP, C.c_int, C.c_int, VP, VP, C.c_double, VP, VP, C.c_int, C.c_int, C.c_int,
                                      C.c_double, C.c_double, C.c_int, VP, VP, VP, VP]),
    "modest_mask_stage_batch": (C.c_int, [VP, C.c_int, VP, VP]),
    "modest_cluster_stats": (C.c_int, [VP, VP, C.c_int, C.c_int, VP, VP, C.c_int, VP, C.c_double, VP, VP]),
    "modest_boxes_pp_stats": (C.c_int, [VP, VP, C.c_int, VP, VP, C.c_int, C.c_double, VP, VP]),
    "modest_fit_boxes_closeness": (C.c_int, [VP, VP, VP, C.c_int, VP, C.c_int, C.c_double, VP, VP, VP]),
    "modest_fit_boxes_closeness_host": (C.c_int, [VP, VP, VP, C.c_int, VP, C.c_int, C.c_double, VP, VP, VP, VP]),
    "modest_fit_boxes_variance": (C.c_int, [VP, VP, VP, C.c_int, VP, C.c_int, VP, VP, VP]),
    "modest_fit_boxes_pca": (C.c_int, [VP, VP, VP, C.c_int, VP, VP]),
    "modest_lowest_point": (C.c_int, [VP, VP, C.c_int, VP, C.c_int, VP, VP]),
    "modest_boxes_overlap_bev": (C.c_int, [VP, C.c_int, VP, C.c_int, VP, VP]),
    "modest_boxes_iou_bev": (C.c_int, [VP, C.c_int, VP, C.c_int, VP, VP]),
    "modest_nms_bev": (C.c_int, [VP, VP, C.c_int, C.c_float, VP, VP, VP]),
    "modest_nms_normal": (C.c_int, [VP, VP, C.c_int, C.c_float, VP, VP, VP]),
    "modest_boxes_iou_bev_host": (C.c_int, [VP, VP, C.c_int, VP, C.c_int, VP, VP]),
    "modest_scan_boxes_batch": (C.c_int, [VP, C.c_int, VP, VP]),
    "modest_seed_chain": (C.c_int, [VP, C.c_int, VP, VP, C.c_int, C.c_int, VP]),
    "modest_scan_boxes": (C.c_int, [VP, VP, VP, C.c_int, C.c_int, VP, C.c_int, VP, VP, VP, VP, VP]),
    "modest_objs_iou_batch": (C.c_int, [VP, VP, VP, C.c_int, VP, VP]),
    "modest_objs_iou": (C.c_int, [VP, VP, C.c_int, VP, VP]),
    "modest_pp_block_tables": (C.c_int, [VP, C.c_int, VP, VP, VP, VP, VP, C.c_int, VP, C.c_int32, VP, VP, VP, VP, VP]),
    "modest_host_read_files": (C.c_int64, [VP, C.c_int, VP, C.c_uint64, VP, C.c_int]),
    "modest_label_lines": (C.c_int, [VP, VP, C.c_int, VP, VP, VP, VP, VP, VP, C.c_int32, VP]),
}

_lib = None
_lock = threading.Lock()


class ModestHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """dlopen the library and attach prototypes (raises if it is not built)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not LIB_PATH.exists():
            raise ModestHipError(
                f"{LIB_PATH} is missing: build it with `python -m modest_amd.build` "
                "(there is no CPU fallback for the MODEST hot path)")
        # Device pointers cross this boundary from PyTorch-ROCm, so both must sit on
        # ONE HIP runtime instance: import torch first, then libamdhip64.so.7 is
        # already mapped when our DT_NEEDED entry is resolved.
        import torch  # noqa: F401
        lib = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().modest_last_error()
        raise ModestHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


class Context:
    """One scratch arena per process x GPU x stream (modest_ctx)."""

    def __init__(self, device: int = 0):
        lib = load()
        self._h = VP()
        check(lib.modest_ctx_create(int(device), C.byref(self._h)), "modest_ctx_create")
        self.device = int(device)

    @property
    def handle(self):
        return self._h

    def host_counter(self):
        """A pinned (device-visible) int32 word of this context: the compaction kernels write their
        totals straight into it, the caller reads it after a stream sync -- no copy kernel."""
        import torch
        if getattr(self, "_counter", None) is None:
            self._counter = torch.zeros((16,), dtype=torch.int32).pin_memory()
        return self._counter

    def warmup(self) -> None:
        """load the library's device code now instead of at the first launch of every translation unit"""
        check(load().modest_warmup(self._h), "modest_warmup")

    def reserve_arena(self, nbytes: int) -> None:
        """take the scratch arena in one driver call (it grows on demand otherwise: synchronise + free + allocate)"""
        check(load().modest_ctx_reserve_arena(self._h, int(nbytes)), "modest_ctx_reserve_arena")

    def profile_begin(self, capacity: int = 4096) -> None:
        check(load().modest_ctx_profile_begin(self._h, int(capacity)), "modest_ctx_profile_begin")

    def profile_collect(self, capacity: int = 4096):
        import numpy as np
        ms = np.zeros(capacity, dtype=np.float32)
        n = C.c_int(0)
        check(load().modest_ctx_profile_collect(self._h, ms.ctypes.data, capacity, C.byref(n)),
              "modest_ctx_profile_collect")
        return ms[: n.value]

    def close(self):
        if self._h:
            load().modest_ctx_destroy(self._h)
            self._h = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device: int = 0) -> Context:
    """One context (scratch arena) per device AND per host thread: calls on a context are
    stream ordered, and every host thread drives its own HIP stream."""
    key = (device, threading.get_ident())
    ctx = _default_ctx.get(key)
    if ctx is None:
        ctx = _default_ctx[key] = Context(device)
    return ctx
