"""Thin Python wrappers over the C ABI (include/modest_hip.h).

PyTorch-ROCm tensors are used only as device-memory handles and for the
current HIP stream; every computation happens inside libmodest_hip.so.  All
functions raise if the library or a GPU is missing (no CPU fallback).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import Context, check, default_context, load


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise ValueError(f"{name} must be a device tensor (PyTorch-ROCm 'cuda')")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


def _ctx(ctx: Optional[Context], t: torch.Tensor) -> Context:
    return ctx if ctx is not None else default_context(t.device.index or 0)


def _np_ptr(a: np.ndarray) -> int:
    return a.ctypes.data


# --------------------------------------------------------------------------- PP score
def pp_count(live_xyz: torch.Tensor, hist_xyz: torch.Tensor, trav_offsets: Sequence[int],
             radius: float = 0.3, ctx: Optional[Context] = None) -> torch.Tensor:
    """count_neighbors (pre_compute_pp_score.py:54-60): (N,T) int32 on device."""
    lib = load()
    _dev(live_xyz, torch.float32, "live_xyz")
    _dev(hist_xyz, torch.float32, "hist_xyz")
    off = np.ascontiguousarray(np.asarray(trav_offsets, dtype=np.int64))
    T = off.shape[0] - 1
    n = live_xyz.shape[0]
    assert live_xyz.ndim == 2 and live_xyz.shape[1] == 3
    assert hist_xyz.ndim == 2 and hist_xyz.shape[1] == 3 and off[-1] <= hist_xyz.shape[0]
    counts = torch.empty((n, T), dtype=torch.int32, device=live_xyz.device)
    c = _ctx(ctx, live_xyz)
    check(lib.modest_pp_count(c.handle, live_xyz.data_ptr(), n, hist_xyz.data_ptr(), _np_ptr(off), T,
                              float(radius), counts.data_ptr(), _stream()), "modest_pp_count")
    return counts


def pp_entropy(counts: torch.Tensor, ctx: Optional[Context] = None) -> torch.Tensor:
    """compute_ephe_score (pre_compute_pp_score.py:68-75) + float32 cast (:195-196)."""
    lib = load()
    _dev(counts, torch.int32, "counts")
    n, T = counts.shape
    H = torch.empty((n,), dtype=torch.float32, device=counts.device)
    c = _ctx(ctx, counts)
    check(lib.modest_pp_entropy(c.handle, counts.data_ptr(), n, T, H.data_ptr(), _stream()),
          "modest_pp_entropy")
    return H


def pp_score(live_xyz: torch.Tensor, hist_xyz: torch.Tensor, trav_offsets: Sequence[int],
             radius: float = 0.3, ctx: Optional[Context] = None, return_counts: bool = False,
             out: Optional[torch.Tensor] = None):
    """Fused count + entropy: (N,) float32 PP score on device."""
    lib = load()
    _dev(live_xyz, torch.float32, "live_xyz")
    _dev(hist_xyz, torch.float32, "hist_xyz")
    off = np.ascontiguousarray(np.asarray(trav_offsets, dtype=np.int64))
    T = off.shape[0] - 1
    n = live_xyz.shape[0]
    assert hist_xyz.ndim == 2 and hist_xyz.shape[1] == 3 and off[-1] <= hist_xyz.shape[0]
    H = out if out is not None else torch.empty((n,), dtype=torch.float32, device=live_xyz.device)
    counts = torch.empty((n, T), dtype=torch.int32, device=live_xyz.device) if return_counts else None
    c = _ctx(ctx, live_xyz)
    check(lib.modest_pp_score(c.handle, live_xyz.data_ptr(), n, hist_xyz.data_ptr(), _np_ptr(off), T,
                              float(radius), counts.data_ptr() if counts is not None else None,
                              H.data_ptr(), _stream()), "modest_pp_score")
    return (H, counts) if return_counts else H
