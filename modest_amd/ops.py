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


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """handle of PyTorch's current HIP stream (the raw getter skips the Stream object: ~8 us per call)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
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


# --------------------------------------------------------------------------- transform
def transform_points(pts: torch.Tensor, T: np.ndarray, remove_center: bool = False,
                     ctx: Optional[Context] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """transform_points (utils/pointcloud_utils.py:11-19) of an (n,3|4) float32 frame,
    optionally preceded by remove_center (pre_compute_pp_score.py:48-52)."""
    lib = load()
    _dev(pts, torch.float32, "pts")
    assert pts.ndim == 2 and pts.shape[1] in (3, 4)
    n = pts.shape[0]
    T16 = np.ascontiguousarray(np.asarray(T, dtype=np.float32).reshape(4, 4))
    if out is None:
        out = torch.empty((n, 3), dtype=torch.float32, device=pts.device)
    else:
        _dev(out, torch.float32, "out")
        assert out.shape == (n, 3)
    c = _ctx(ctx, pts)
    if not remove_center:
        check(lib.modest_transform_points(c.handle, pts.data_ptr(), n, pts.shape[1], _np_ptr(T16), 0,
                                          out.data_ptr(), None, _stream()), "modest_transform_points")
        return out
    n_out = torch.zeros((1,), dtype=torch.int64, device=pts.device)
    check(lib.modest_transform_points(c.handle, pts.data_ptr(), n, pts.shape[1], _np_ptr(T16), 1,
                                      out.data_ptr(), n_out.data_ptr(), _stream()), "modest_transform_points")
    return out[: int(n_out.item())]


def project_velo_to_rect(pts: torch.Tensor, V2C: np.ndarray, R0: np.ndarray, ctx: Optional[Context] = None) -> torch.Tensor:
    """Calibration.project_velo_to_rect (kitti_util.py:327-329) of the scan rows: (n,3) float64 on the
    device, bit-identical to numpy's two dgemm products."""
    lib = load()
    _dev(pts, torch.float32, "pts")
    v = np.ascontiguousarray(V2C, dtype=np.float64).reshape(12)
    r = np.ascontiguousarray(R0, dtype=np.float64).reshape(9)
    out = torch.empty((pts.shape[0], 3), dtype=torch.float64, device=pts.device)
    c = _ctx(ctx, pts)
    check(lib.modest_project_velo_to_rect(c.handle, pts.data_ptr(), pts.shape[0], pts.shape[1], _np_ptr(v), _np_ptr(r),
                                          out.data_ptr(), _stream()), "modest_project_velo_to_rect")
    return out


# --------------------------------------------------------------------------- plane / RANSAC
def plane_candidates(pts: torch.Tensor, max_hs: float, ptc_range, ctx: Optional[Context] = None):
    """Candidate mask of estimate_plane (pointcloud_utils.py:45-49), compacted in input order.
    Returns (cand_xyz (m,3) f32, cand_idx (m,) i32)."""
    lib = load()
    _dev(pts, torch.float32, "pts")
    n = pts.shape[0]
    cand = torch.empty((n, 3), dtype=torch.float32, device=pts.device)
    idx = torch.empty((n,), dtype=torch.int32, device=pts.device)
    c = _ctx(ctx, pts)
    cnt = c.host_counter()   # pinned host word, always written by the call (n == 0: a memset)
    (xlo, xhi), (ylo, yhi) = ptc_range
    check(lib.modest_plane_candidates(c.handle, pts.data_ptr(), n, pts.shape[1], float(max_hs), float(xlo),
                                      float(xhi), float(ylo), float(yhi), cand.data_ptr(), idx.data_ptr(),
                                      cnt.data_ptr(), _stream()), "modest_plane_candidates")
    torch.cuda.current_stream().synchronize()
    m = int(cnt[0])
    return cand[:m], idx[:m]


def plane_prepare(pts: torch.Tensor, specs, ctx: Optional[Context] = None):
    """Candidates and MAD thresholds of two estimate_plane calls on one scan ((max_hs, ptc_range) each):
    one pass over the rows, one stream sync.  Returns [(cand (m,3) f32 device, thr np.float32 | None)] x 2."""
    lib = load()
    _dev(pts, torch.float32, "pts")
    assert len(specs) == 2
    n = pts.shape[0]
    flat = []
    for max_hs, ((xlo, xhi), (ylo, yhi)) in specs:
        flat += [max_hs, xlo, xhi, ylo, yhi]
    spec = np.ascontiguousarray(flat, dtype=np.float32)
    cands = [torch.empty((n, 3), dtype=torch.float32, device=pts.device) for _ in range(2)]
    cnt, mad = np.zeros(2, dtype=np.int32), np.zeros(2, dtype=np.float32)
    c = _ctx(ctx, pts)
    check(lib.modest_plane_prepare(c.handle, pts.data_ptr(), n, pts.shape[1], _np_ptr(spec), cands[0].data_ptr(),
                                   cands[1].data_ptr(), _np_ptr(cnt), _np_ptr(mad), _stream()), "modest_plane_prepare")
    return [(cands[k][: int(cnt[k])], np.float32(mad[k]) if cnt[k] >= 1 else None) for k in range(2)]


def mad_threshold(cand: torch.Tensor, ctx: Optional[Context] = None) -> np.float32:
    lib = load()
    _dev(cand, torch.float32, "cand")
    out = C.c_float(0)
    c = _ctx(ctx, cand)
    check(lib.modest_mad_threshold(c.handle, cand.data_ptr(), cand.shape[0], C.byref(out), _stream()),
          "modest_mad_threshold")
    return np.float32(out.value)


def mad_threshold_batch(cands: Sequence[torch.Tensor], ctx: Optional[Context] = None) -> np.ndarray:
    """MAD thresholds of up to four candidate sets from one launch (one workgroup per set)."""
    lib = load()
    for t in cands:
        _dev(t, torch.float32, "cand")
    k = len(cands)
    ptrs = (C.c_void_p * k)(*[t.data_ptr() for t in cands])
    ns = np.ascontiguousarray([t.shape[0] for t in cands], dtype=np.int32)
    out = np.zeros(k, dtype=np.float32)
    c = _ctx(ctx, cands[0])
    check(lib.modest_mad_threshold_batch(c.handle, ptrs, _np_ptr(ns), k, _np_ptr(out), _stream()),
          "modest_mad_threshold_batch")
    return out


def ransac_score_trials(cand: torch.Tensor, models: np.ndarray, thr: float, ctx: Optional[Context] = None):
    """Inlier counts (+ SSE / sum z / sum z^2 over the inliers, float64) of K trial planes."""
    lib = load()
    _dev(cand, torch.float32, "cand")
    models = np.ascontiguousarray(models, dtype=np.float32).reshape(-1, 3)
    K = models.shape[0]
    n_in = np.zeros(K, dtype=np.int32)
    sse, sy, syy = (np.zeros(K, dtype=np.float64) for _ in range(3))
    c = _ctx(ctx, cand)
    check(lib.modest_ransac_score_trials(c.handle, cand.data_ptr(), cand.shape[0], _np_ptr(models), K,
                                         float(np.float32(thr)), _np_ptr(n_in), _np_ptr(sse), _np_ptr(sy),
                                         _np_ptr(syy), _stream()), "modest_ransac_score_trials")
    return n_in, sse, sy, syy


def ransac_trials(cand: torch.Tensor, triplets: np.ndarray, thr: Optional[float] = None,
                  ctx: Optional[Context] = None):
    """MAD threshold (when thr is None) + exact-fit planes of the triplets + their scores, one
    device round trip.  Returns (thr, models (K,3) f32, n_inliers, sse, sy, syy)."""
    lib = load()
    _dev(cand, torch.float32, "cand")
    trip = np.ascontiguousarray(triplets, dtype=np.int32).reshape(-1, 3)
    K = trip.shape[0]
    thr_io = C.c_float(-1.0 if thr is None else float(np.float32(thr)))
    models = np.zeros((K, 3), dtype=np.float32)
    n_in = np.zeros(K, dtype=np.int32)
    sse, sy, syy = (np.zeros(K, dtype=np.float64) for _ in range(3))
    c = _ctx(ctx, cand)
    check(lib.modest_ransac_trials(c.handle, cand.data_ptr(), cand.shape[0], _np_ptr(trip), K, C.byref(thr_io),
                                   _np_ptr(models), _np_ptr(n_in), _np_ptr(sse), _np_ptr(sy), _np_ptr(syy),
                                   _stream()), "modest_ransac_trials")
    return np.float32(thr_io.value), models, n_in, sse, sy, syy


def ransac_plane_native(cand: torch.Tensor, rs: np.random.RandomState, thr: float, max_trials: int = 100,
                        stop_probability: float = 0.99, batch: int = 48, ctx: Optional[Context] = None):
    """The whole RANSAC fit behind one library call (include/modest_hip.h: modest_ransac_plane): `rs`
    (legacy MT19937 RandomState) is advanced in place by the executed trials.
    Returns (status, model64 (3,), best_model (3,) f32, triplets (n_trials,3), n_trials, n_inliers)."""
    lib = load()
    _dev(cand, torch.float32, "cand")
    st = rs.get_state()
    assert st[0] == "MT19937"
    key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
    pos = C.c_int32(int(st[2]))
    model64, best = np.zeros(3, dtype=np.float64), np.zeros(3, dtype=np.float32)
    trip = np.zeros((max_trials, 3), dtype=np.int32)
    n_trials, n_in, status = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    c = _ctx(ctx, cand)
    check(lib.modest_ransac_plane(c.handle, cand.data_ptr(), cand.shape[0], float(np.float32(thr)), _np_ptr(key),
                                  C.byref(pos), int(max_trials), float(stop_probability), int(batch), _np_ptr(model64),
                                  _np_ptr(best), _np_ptr(trip), C.byref(n_trials), C.byref(n_in), C.byref(status),
                                  _stream()), "modest_ransac_plane")
    rs.set_state((st[0], key, int(pos.value), st[3], st[4]))
    return int(status.value), model64, best, trip[: n_trials.value].astype(np.int64), int(n_trials.value), int(n_in.value)


def mt19937_triplets(rs: np.random.RandomState, n_population: int, n_trials: int) -> np.ndarray:
    """n_trials draws of sklearn's sample_without_replacement(n_population > 300, 3) from `rs`, made by
    the library's generator (host only); `rs` is advanced in place."""
    lib = load()
    st = rs.get_state()
    key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
    pos = C.c_int32(int(st[2]))
    out = np.zeros((n_trials, 3), dtype=np.int32)
    check(lib.modest_mt19937_triplets(_np_ptr(key), C.byref(pos), int(n_population), int(n_trials), _np_ptr(out)),
          "modest_mt19937_triplets")
    rs.set_state((st[0], key, int(pos.value), st[3], st[4]))
    return out


def ransac_refit(cand: torch.Tensor, model: np.ndarray, thr: float, ctx: Optional[Context] = None):
    lib = load()
    _dev(cand, torch.float32, "cand")
    model = np.ascontiguousarray(model, dtype=np.float32).reshape(3)
    out = np.zeros(3, dtype=np.float64)
    n_in = C.c_int32(0)
    c = _ctx(ctx, cand)
    check(lib.modest_ransac_refit(c.handle, cand.data_ptr(), cand.shape[0], _np_ptr(model),
                                  float(np.float32(thr)), _np_ptr(out), C.byref(n_in), _stream()),
          "modest_ransac_refit")
    return out, int(n_in.value)


def plane_range_mask(pts: torch.Tensor, plane: np.ndarray, offset: float, only_range, limit_range,
                     ctx: Optional[Context] = None):
    """above_plane (pointcloud_utils.py:68-74) AND the limit_range mask (generate_mask.py:61-65).
    Returns (mask (n,) bool device, kept_xyz (m,3) f32, kept_idx (m,) i32)."""
    lib = load()
    _dev(pts, torch.float32, "pts")
    n = pts.shape[0]
    plane = np.ascontiguousarray(plane, dtype=np.float64).reshape(4)
    lim = np.ascontiguousarray(np.asarray(limit_range, dtype=np.float64).reshape(4))
    onl = None if only_range is None else np.ascontiguousarray(np.asarray(only_range, dtype=np.float64).reshape(4))
    mask = torch.empty((n,), dtype=torch.uint8, device=pts.device)
    kept = torch.empty((n, 3), dtype=torch.float32, device=pts.device)
    idx = torch.empty((n,), dtype=torch.int32, device=pts.device)
    c = _ctx(ctx, pts)
    cnt = c.host_counter()   # pinned host word, always written by the call
    check(lib.modest_plane_range_mask(c.handle, pts.data_ptr(), n, pts.shape[1], _np_ptr(plane), float(offset),
                                      None if onl is None else _np_ptr(onl), _np_ptr(lim), mask.data_ptr(),
                                      kept.data_ptr(), idx.data_ptr(), cnt.data_ptr(), _stream()),
          "modest_plane_range_mask")
    torch.cuda.current_stream().synchronize()
    m = int(cnt[0])
    return mask.view(torch.bool), kept[:m], idx[:m]   # 0/1 bytes reinterpreted, no kernel


# --------------------------------------------------------------------------- clustering
GRAPH_TYPES = {"radius_mutual_knn": 0, "radius": 1, "knn": 2, "sym_knn": 3, "mutual_knn": 4}
AFFINITY_TYPES = {"l1": 0, "exp": 1, "3d_l2_distance": 2}


def cluster_dbscan(xyz: torch.Tensor, pp: torch.Tensor, n_neighbors: int = 70, radius: float = 2.0,
                   eps: float = 0.1, min_samples: int = 10, return_kth: bool = False,
                   ctx: Optional[Context] = None, neighbor_type: str = "radius_mutual_knn",
                   affinity_type: str = "l1", intensity: Optional[torch.Tensor] = None):
    """Affinity graph (precompute_affinity_matrix, clustering_utils.py:7-60) + DBSCAN(precomputed)
    labels, (n,) int32 on device.  neighbor_type radius_mutual_knn | radius; affinity_type l1 | exp |
    3d_l2_distance (the latter needs the scan's intensity column, see include/modest_hip.h)."""
    lib = load()
    _dev(xyz, torch.float32, "xyz")
    _dev(pp, torch.float32, "pp")
    if neighbor_type not in GRAPH_TYPES:
        raise NotImplementedError(neighbor_type)
    if affinity_type not in AFFINITY_TYPES:
        raise NotImplementedError(affinity_type)
    n = xyz.shape[0]
    assert pp.shape[0] == n
    if affinity_type == "3d_l2_distance":
        if intensity is None:
            raise ValueError("3d_l2_distance needs the intensity column of the scan rows")
        _dev(intensity, torch.float32, "intensity")
        assert intensity.shape[0] == n
    labels = torch.empty((n,), dtype=torch.int32, device=xyz.device)
    kth = torch.empty((n,), dtype=torch.float64, device=xyz.device) if return_kth else None
    ncl = C.c_int32(0)
    c = _ctx(ctx, xyz)
    check(lib.modest_cluster_dbscan_ex(c.handle, xyz.data_ptr(), pp.data_ptr(),
                                       intensity.data_ptr() if intensity is not None else None, n,
                                       GRAPH_TYPES[neighbor_type], AFFINITY_TYPES[affinity_type], int(n_neighbors),
                                       float(radius), float(eps), int(min_samples), labels.data_ptr(),
                                       kth.data_ptr() if kth is not None else None, C.byref(ncl), _stream()),
          "modest_cluster_dbscan_ex")
    return (labels, int(ncl.value), kth) if return_kth else (labels, int(ncl.value))


def mask_cluster(pts: torch.Tensor, pp: torch.Tensor, plane: np.ndarray, offset: float, only_range, limit_range,
                 n_neighbors: int = 70, radius: float = 2.0, eps: float = 0.1, min_samples: int = 10,
                 neighbor_type: str = "radius_mutual_knn", affinity_type: str = "l1",
                 ctx: Optional[Context] = None):
    """plane_range_mask + cluster_dbscan + ``labels[ptc_mask] = ...`` (generate_mask.py:57-88) in one
    call.  Returns (labels (n,) int32 device, -1 = masked out or noise; number of kept rows)."""
    lib = load()
    _dev(pts, torch.float32, "pts")
    _dev(pp, torch.float32, "pp")
    if neighbor_type not in GRAPH_TYPES:
        raise NotImplementedError(neighbor_type)
    if affinity_type not in AFFINITY_TYPES:
        raise NotImplementedError(affinity_type)
    n = pts.shape[0]
    assert pp.shape[0] == n
    plane = np.ascontiguousarray(plane, dtype=np.float64).reshape(4)
    lim = np.ascontiguousarray(np.asarray(limit_range, dtype=np.float64).reshape(4))
    onl = None if only_range is None else np.ascontiguousarray(np.asarray(only_range, dtype=np.float64).reshape(4))
    labels = torch.empty((n,), dtype=torch.int32, device=pts.device)
    n_kept, ncl = C.c_int32(0), C.c_int32(0)
    c = _ctx(ctx, pts)
    rc = lib.modest_mask_cluster(c.handle, pts.data_ptr(), n, pts.shape[1], pp.data_ptr(), _np_ptr(plane), float(offset),
                                 None if onl is None else _np_ptr(onl), _np_ptr(lim), GRAPH_TYPES[neighbor_type],
                                 AFFINITY_TYPES[affinity_type], int(n_neighbors), float(radius), float(eps),
                                 int(min_samples), labels.data_ptr(), C.byref(n_kept), C.byref(ncl), _stream())
    if rc and neighbor_type != "radius" and 0 < n_kept.value <= n_neighbors:
        raise ValueError(f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {n_neighbors + 1}, "
                         f"n_samples_fit = {n_kept.value}, n_samples = {n_kept.value}")
    check(rc, "modest_mask_cluster")
    return labels, int(n_kept.value)


class MaskParams(C.Structure):
    """modest_mask_params (include/modest_hip.h)"""
    _fields_ = [("max_hs1", C.c_float), ("range1", C.c_float * 4), ("max_hs2", C.c_float), ("range2", C.c_float * 4),
                ("offset", C.c_double), ("use_only_range", C.c_int32), ("only_range", C.c_double * 4),
                ("limit_range", C.c_double * 4), ("neighbor_type", C.c_int32), ("affinity_type", C.c_int32),
                ("k_neighbors", C.c_int32), ("min_samples", C.c_int32), ("radius", C.c_double), ("eps", C.c_double),
                ("min_points", C.c_int32), ("max_min_height", C.c_double), ("min_max_height", C.c_double),
                ("quantile", C.c_double), ("min_percentile_pp_score", C.c_float), ("max_trials", C.c_int32),
                ("batch", C.c_int32), ("stop_probability", C.c_double)]


STAGE_STATUS = {1: "small candidate set", 2: "no consensus set", 3: "degenerate consensus set", 4: "too few kept rows",
                5: "RANSAC trial bound on a rounding boundary (the host loop decides)"}


def mask_stage(pts: torch.Tensor, pp: torch.Tensor, params: MaskParams, rs: np.random.RandomState,
               ctx: Optional[Context] = None):
    """generate_mask_scan up to ``labels_filtered`` behind one library call (modest_mask_stage).
    Returns None when the library hands the scan back to the host statement (rare inputs: see
    STAGE_STATUS; ``rs`` is untouched then), else (labels_filtered (n,) int64, plane1, plane2, info):
    ``rs`` has been advanced by the executed RANSAC trials of both fits."""
    lib = load()
    _dev(pts, torch.float32, "pts")
    _dev(pp, torch.float32, "pp")
    n = pts.shape[0]
    assert pp.shape[0] == n and n >= 1
    st = rs.get_state()
    assert st[0] == "MT19937"
    key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
    pos = C.c_int32(int(st[2]))
    plane1, plane2 = np.zeros(4, dtype=np.float64), np.zeros(4, dtype=np.float64)
    labels = np.empty(n, dtype=np.int64)
    info = np.zeros(8, dtype=np.int32)
    c = _ctx(ctx, pts)
    check(lib.modest_mask_stage(c.handle, pts.data_ptr(), n, pts.shape[1], pp.data_ptr(), C.byref(params), _np_ptr(key),
                                C.byref(pos), _np_ptr(plane1), _np_ptr(plane2), _np_ptr(labels), _np_ptr(info),
                                _stream()), "modest_mask_stage")
    if info[3] != 0:
        return None
    rs.set_state((st[0], key, int(pos.value), st[3], st[4]))
    return labels, plane1, plane2, info


class MaskStageScan(C.Structure):
    """modest_mask_stage_scan (include/modest_hip.h)"""
    _fields_ = [("ctx", C.c_void_p), ("pts_dev", C.c_void_p), ("n", C.c_int32), ("stride", C.c_int32), ("pp_dev", C.c_void_p),
                ("mt_key624", C.c_void_p), ("mt_pos", C.c_void_p), ("plane1_out", C.c_void_p), ("plane2_out", C.c_void_p),
                ("labels_out", C.c_void_p), ("info_out", C.c_void_p), ("members_out", C.c_void_p), ("n_members_out", C.c_void_p)]


_CHAIN_CTXS = {}


def chain_contexts(n: int, device: int = 0):
    """`n` library contexts of the calling thread for the scans of a chain (every scan of a chain works in its own
    context: scratch, pinned words, persistent counters)"""
    import threading
    key = (threading.get_ident(), int(device))
    have = _CHAIN_CTXS.setdefault(key, [])
    while len(have) < n:
        have.append(Context(int(device)))
    return have[:n]


def mask_stage_batch(items, params: MaskParams, ctxs=None):
    """modest_mask_stage_batch: generate_mask_scan up to ``labels_filtered`` for a CHAIN of scans -- the ground fits per
    scan, the mask / graph / DBSCAN block and the cluster statistics as one launch per kernel for the whole chain.
    items: [(pts_dev (n,3|4) f32, pp_dev (n,) f32, RandomState)]; returns a list with, per scan, what mask_stage
    returns (None: the library hands that scan back to the host statement, its generator untouched)."""
    lib = load()
    B = len(items)
    if B == 0:
        return []
    if ctxs is None:
        ctxs = chain_contexts(B, items[0][0].device.index or 0)
    arr = (MaskStageScan * B)()
    keep = []
    for i, (pts, pp, rs) in enumerate(items):
        _dev(pts, torch.float32, "pts")
        _dev(pp, torch.float32, "pp")
        n = pts.shape[0]
        assert pp.shape[0] == n and n >= 1
        st = rs.get_state()
        assert st[0] == "MT19937"
        key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
        pos = np.array([int(st[2])], dtype=np.int32)
        plane1, plane2 = np.zeros(4, dtype=np.float64), np.zeros(4, dtype=np.float64)
        labels = np.empty(n, dtype=np.int64)
        info = np.zeros(8, dtype=np.int32)
        members, n_mem = np.empty(n, dtype=np.int32), np.full(1, -1, dtype=np.int32)
        keep.append((st, key, pos, plane1, plane2, labels, info, members, n_mem))
        a = arr[i]
        a.ctx = C.cast(ctxs[i].handle, C.c_void_p).value
        a.pts_dev, a.n, a.stride, a.pp_dev = pts.data_ptr(), n, pts.shape[1], pp.data_ptr()
        a.mt_key624, a.mt_pos = _np_ptr(key), _np_ptr(pos)
        a.plane1_out, a.plane2_out, a.labels_out, a.info_out = _np_ptr(plane1), _np_ptr(plane2), _np_ptr(labels), _np_ptr(info)
        a.members_out, a.n_members_out = _np_ptr(members), _np_ptr(n_mem)
    check(lib.modest_mask_stage_batch(C.byref(arr), B, C.byref(params), _stream()), "modest_mask_stage_batch")
    out = []
    for (pts, pp, rs), (st, key, pos, plane1, plane2, labels, info, members, n_mem) in zip(items, keep):
        if info[3] != 0:
            out.append(None)
            continue
        rs.set_state((st[0], key, int(pos[0]), st[3], st[4]))
        # (a fifth entry: the indices of the points with a label > 0, ascending -- None when the library did not list them)
        out.append((labels, plane1, plane2, info, members[:int(n_mem[0])] if n_mem[0] >= 0 else None))
    return out


class BoxesParams(C.Structure):
    """modest_boxes_params (include/modest_hip.h)"""
    _fields_ = [("V2C", C.c_double * 12), ("R0", C.c_double * 9), ("angles", C.c_void_p), ("cossin", C.c_void_p),
                ("cossin90", C.c_void_p), ("n_angles", C.c_int32), ("d0", C.c_double), ("min_volume", C.c_double),
                ("max_volume", C.c_double)]


class LabelsParams(C.Structure):
    """modest_labels_params (include/modest_hip.h)"""
    _fields_ = [("P", C.c_double * 12), ("nms_enable", C.c_int32), ("nms_threshold", C.c_float), ("fov_only", C.c_int32),
                ("image_h", C.c_double), ("image_w", C.c_double)]


def scan_boxes(pts_dev: torch.Tensor, pts_host: np.ndarray, labels_filtered: np.ndarray, n_lab: int, V2C, R0,
               angles: np.ndarray, cossin: np.ndarray, cossin90: np.ndarray, d0: float, min_volume: float,
               max_volume: float, ctx: Optional[Context] = None):
    """The box tail of generate_mask.py:88-103 behind one library call (modest_scan_boxes): members, rect
    points, closeness fit, get_obj, volume gate, relabelling.  Returns None when the library hands the scan
    back to the host statement, else (final labels (n,) int64, objs (n_lab,8) float64 rows
    {t0, t1, t2, l, w, h, ry, volume}, keep (n_lab,) bool)."""
    lib = load()
    _dev(pts_dev, torch.float32, "pts")
    assert pts_host.dtype == np.float32 and pts_host.flags.c_contiguous and pts_host.shape == tuple(pts_dev.shape)
    n = pts_host.shape[0]
    P = BoxesParams()
    P.V2C[:] = [float(x) for x in np.asarray(V2C, dtype=np.float64).reshape(12)]
    P.R0[:] = [float(x) for x in np.asarray(R0, dtype=np.float64).reshape(9)]
    assert angles.dtype == np.float64 and cossin.dtype == np.float64 and cossin90.dtype == np.float64
    assert cossin.flags.c_contiguous and cossin90.flags.c_contiguous and angles.flags.c_contiguous
    P.angles, P.cossin, P.cossin90, P.n_angles = _np_ptr(angles), _np_ptr(cossin), _np_ptr(cossin90), angles.shape[0]
    P.d0, P.min_volume, P.max_volume = float(d0), float(min_volume), float(max_volume)
    labels = np.ascontiguousarray(labels_filtered, dtype=np.int64).copy()
    objs = np.zeros((max(n_lab, 1), 8), dtype=np.float64)
    keep = np.zeros(max(n_lab, 1), dtype=np.int32)
    info = np.zeros(2, dtype=np.int32)
    c = _ctx(ctx, pts_dev)
    check(lib.modest_scan_boxes(c.handle, pts_dev.data_ptr(), _np_ptr(pts_host), n, pts_host.shape[1], _np_ptr(labels),
                                int(n_lab), C.byref(P), _np_ptr(objs), _np_ptr(keep), _np_ptr(info), _stream()),
          "modest_scan_boxes")
    if info[1] != 0:
        return None
    return labels, objs[:n_lab], keep[:n_lab].astype(bool)


class BoxesScan(C.Structure):
    """modest_boxes_scan (include/modest_hip.h)"""
    _fields_ = [("ctx", C.c_void_p), ("pts_dev", C.c_void_p), ("pts_host", C.c_void_p), ("n", C.c_int32), ("stride", C.c_int32),
                ("labels_inout", C.c_void_p), ("n_lab", C.c_int32), ("objs_out", C.c_void_p), ("keep_out", C.c_void_p),
                ("info_out", C.c_void_p), ("members", C.c_void_p), ("n_members", C.c_int32)]


def scan_boxes_batch(items, V2C, R0, angles: np.ndarray, cossin: np.ndarray, cossin90: np.ndarray, d0: float,
                     min_volume: float, max_volume: float, ctxs=None):
    """modest_scan_boxes_batch: the box tail of a CHAIN of scans (one calibration) -- host phases per scan, one
    closeness launch over all clusters, one lowest-point launch over all boxes.  items: [(pts_dev, pts_host,
    labels_filtered, n_lab[, members])] -- members: the ascending indices of the points with a label > 0 as mask_stage_batch
    lists them (the host passes then touch those only); returns per scan what scan_boxes returns (None: host statement)."""
    lib = load()
    B = len(items)
    if B == 0:
        return []
    if ctxs is None:
        ctxs = chain_contexts(B, items[0][0].device.index or 0)
    P = BoxesParams()
    P.V2C[:] = [float(x) for x in np.asarray(V2C, dtype=np.float64).reshape(12)]
    P.R0[:] = [float(x) for x in np.asarray(R0, dtype=np.float64).reshape(9)]
    assert angles.dtype == np.float64 and cossin.dtype == np.float64 and cossin90.dtype == np.float64
    assert cossin.flags.c_contiguous and cossin90.flags.c_contiguous and angles.flags.c_contiguous
    P.angles, P.cossin, P.cossin90, P.n_angles = _np_ptr(angles), _np_ptr(cossin), _np_ptr(cossin90), angles.shape[0]
    P.d0, P.min_volume, P.max_volume = float(d0), float(min_volume), float(max_volume)
    arr = (BoxesScan * B)()
    keepalive = []
    for i, item in enumerate(items):
        pts_dev, pts_host, labels_filtered, n_lab = item[:4]
        members = item[4] if len(item) > 4 else None
        _dev(pts_dev, torch.float32, "pts")
        assert pts_host.dtype == np.float32 and pts_host.flags.c_contiguous and pts_host.shape == tuple(pts_dev.shape)
        labels = np.ascontiguousarray(labels_filtered, dtype=np.int64).copy()
        objs = np.zeros((max(n_lab, 1), 8), dtype=np.float64)
        keep = np.zeros(max(n_lab, 1), dtype=np.int32)
        info = np.zeros(2, dtype=np.int32)
        if members is not None:
            members = np.ascontiguousarray(members, dtype=np.int32)
        keepalive.append((labels, objs, keep, info, int(n_lab), members))
        a = arr[i]
        a.ctx = C.cast(ctxs[i].handle, C.c_void_p).value
        a.pts_dev, a.pts_host, a.n, a.stride = pts_dev.data_ptr(), _np_ptr(pts_host), pts_host.shape[0], pts_host.shape[1]
        a.labels_inout, a.n_lab = _np_ptr(labels), int(n_lab)
        a.objs_out, a.keep_out, a.info_out = _np_ptr(objs), _np_ptr(keep), _np_ptr(info)
        a.members, a.n_members = (_np_ptr(members), int(members.shape[0])) if members is not None else (None, 0)
    check(lib.modest_scan_boxes_batch(C.byref(arr), B, C.byref(P), _stream()), "modest_scan_boxes_batch")
    return [None if info[1] != 0 else (labels, objs[:n_lab], keep[:n_lab].astype(bool))
            for labels, objs, keep, info, n_lab, _m in keepalive]


class SeedScan(C.Structure):
    """modest_seed_scan (include/modest_hip.h)"""
    _fields_ = [("ctx", C.c_void_p), ("pts_dev", C.c_void_p), ("pts_host", C.c_void_p), ("n", C.c_int32), ("stride", C.c_int32),
                ("pp_dev", C.c_void_p), ("mt_key624", C.c_void_p), ("mt_pos", C.c_void_p), ("plane1_out", C.c_void_p),
                ("plane2_out", C.c_void_p), ("labels_out", C.c_void_p), ("members_scratch", C.c_void_p), ("objs_out", C.c_void_p),
                ("iou_out", C.c_void_p), ("info_out", C.c_void_p)]


def _boxes_params(V2C, R0, angles, cossin, cossin90, d0, min_volume, max_volume) -> BoxesParams:
    P = BoxesParams()
    P.V2C[:] = [float(x) for x in np.asarray(V2C, dtype=np.float64).reshape(12)]
    P.R0[:] = [float(x) for x in np.asarray(R0, dtype=np.float64).reshape(9)]
    assert angles.dtype == np.float64 and cossin.dtype == np.float64 and cossin90.dtype == np.float64
    assert cossin.flags.c_contiguous and cossin90.flags.c_contiguous and angles.flags.c_contiguous
    P.angles, P.cossin, P.cossin90, P.n_angles = _np_ptr(angles), _np_ptr(cossin), _np_ptr(cossin90), angles.shape[0]
    P.d0, P.min_volume, P.max_volume = float(d0), float(min_volume), float(max_volume)
    return P


SEED_MAX_BOXES = 96   # clusters per scan the one-call chain has room for (a Lyft-shape scan has 10-40; more: the scan takes the separate calls)


def seed_chain(items, mparams: MaskParams, V2C, R0, angles: np.ndarray, cossin: np.ndarray, cossin90: np.ndarray, d0: float,
               min_volume: float, max_volume: float, nms_enable: bool, ctxs=None):
    """modest_seed_chain: stages 2 + 3 of a CHAIN of scans (one calibration) behind ONE library call -- mask stage, box tail and the
    IoU matrices of the kept boxes.  items: [(pts_dev (n,3|4) f32, pts_host (same, numpy), pp_dev (n,) f32, RandomState)].
    Returns per scan (status, labels (n,) int64, rows (k,8) f64, iou (k,k) f32 | None, plane1, info): status 0 = all of it is final;
    1 = the mask stage handed the scan back (its generator is untouched: run the host statement); 2 / 3 = labels are
    `labels_filtered` and the generator is advanced, the box tail is the caller's (scan_boxes / host statement)."""
    lib = load()
    B = len(items)
    if B == 0:
        return []
    if ctxs is None:
        ctxs = chain_contexts(B, items[0][0].device.index or 0)
    P = _boxes_params(V2C, R0, angles, cossin, cossin90, d0, min_volume, max_volume)
    K = SEED_MAX_BOXES
    ns = [int(it[0].shape[0]) for it in items]
    offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
    labels_all = np.empty(int(offs[-1]), dtype=np.int64)       # one allocation per kind for the whole chain
    members_all = np.empty(int(offs[-1]), dtype=np.int32)
    keys_all = np.empty((B, 624), dtype=np.uint32)
    pos_all = np.zeros(B, dtype=np.int32)
    planes_all = np.zeros((B, 2, 4), dtype=np.float64)
    rows_all = np.zeros((B, K, 8), dtype=np.float64)
    iou_all = np.zeros((B, K * K), dtype=np.float32) if nms_enable else None
    info_all = np.zeros((B, 12), dtype=np.int32)
    arr = (SeedScan * B)()
    states = []
    for i, (pts, pts_host, pp, rs) in enumerate(items):
        _dev(pts, torch.float32, "pts")
        _dev(pp, torch.float32, "pp")
        n = ns[i]
        assert pp.shape[0] == n and n >= 1 and pts_host.dtype == np.float32 and pts_host.flags.c_contiguous and pts_host.shape == tuple(pts.shape)
        st = rs.get_state()
        assert st[0] == "MT19937"
        keys_all[i] = st[1]
        pos_all[i] = int(st[2])
        states.append(st)
        a = arr[i]
        a.ctx = C.cast(ctxs[i].handle, C.c_void_p).value
        a.pts_dev, a.pts_host, a.n, a.stride, a.pp_dev = pts.data_ptr(), _np_ptr(pts_host), n, pts.shape[1], pp.data_ptr()
        a.mt_key624, a.mt_pos = keys_all[i].ctypes.data, pos_all[i:].ctypes.data
        a.plane1_out, a.plane2_out = planes_all[i, 0].ctypes.data, planes_all[i, 1].ctypes.data
        a.labels_out, a.members_scratch = labels_all[offs[i]:].ctypes.data, members_all[offs[i]:].ctypes.data
        a.objs_out = rows_all[i].ctypes.data
        a.iou_out = iou_all[i].ctypes.data if nms_enable else None
        a.info_out = info_all[i].ctypes.data
    check(lib.modest_seed_chain(C.byref(arr), B, C.byref(mparams), C.byref(P), K, int(bool(nms_enable)), _stream()), "modest_seed_chain")
    out = []
    for i, (pts, pts_host, pp, rs) in enumerate(items):
        info = info_all[i]
        status = int(info[10])
        if status != 1:
            st = states[i]
            rs.set_state((st[0], keys_all[i], int(pos_all[i]), st[3], st[4]))
        k = int(info[9])
        out.append((status, labels_all[offs[i]:offs[i + 1]], rows_all[i, :k], (iou_all[i, :k * k].reshape(k, k) if nms_enable else None),
                    planes_all[i, 0], info))
    return out


def objs_iou(objs8: np.ndarray, ctx: Optional[Context] = None) -> np.ndarray:
    """BEV IoU matrix (k,k) float32 of objs_nms' float32 boxes (modest_objs_iou)."""
    lib = load()
    objs8 = np.ascontiguousarray(objs8, dtype=np.float64).reshape(-1, 8)
    k = objs8.shape[0]
    out = np.zeros((k, k), dtype=np.float32)
    if k:
        c = ctx if ctx is not None else default_context(torch.cuda.current_device())
        check(lib.modest_objs_iou(c.handle, _np_ptr(objs8), k, _np_ptr(out), _stream()), "modest_objs_iou")
    return out


def objs_iou_batch(objs_list, ctx: Optional[Context] = None):
    """objs_iou of the box sets of a chain of scans: one launch, one round trip (modest_objs_iou_batch)."""
    lib = load()
    B = len(objs_list)
    rows = [np.ascontiguousarray(o, dtype=np.float64).reshape(-1, 8) for o in objs_list]
    outs = [np.zeros((r.shape[0], r.shape[0]), dtype=np.float32) for r in rows]
    if B == 0 or not any(r.shape[0] for r in rows):
        return outs
    k = np.array([r.shape[0] for r in rows], dtype=np.int32)
    ip = np.array([r.ctypes.data if r.shape[0] else 0 for r in rows], dtype=np.uint64)
    op = np.array([o.ctypes.data if o.size else 0 for o in outs], dtype=np.uint64)
    c = ctx if ctx is not None else default_context(torch.cuda.current_device())
    check(lib.modest_objs_iou_batch(c.handle, ip.ctypes.data, k.ctypes.data, B, op.ctypes.data, _stream()), "modest_objs_iou_batch")
    return outs


def label_lines(objs8: np.ndarray, order: Optional[np.ndarray], iou: Optional[np.ndarray], P34, nms_enable: bool,
                nms_threshold: float, fov_only: bool, image_shape) -> tuple:
    """objs_nms' greedy walk in `order`, is_within_fov, objs2label (modest_label_lines): (text, kept indices)."""
    lib = load()
    objs8 = np.ascontiguousarray(objs8, dtype=np.float64).reshape(-1, 8)
    k = objs8.shape[0]
    if k == 0:
        return "", np.zeros(0, dtype=np.int32)
    cs = np.ascontiguousarray(np.stack([np.cos(objs8[:, 6]), np.sin(objs8[:, 6])], axis=1))   # numpy's roty values
    Q = LabelsParams()
    Q.P[:] = [float(x) for x in np.asarray(P34, dtype=np.float64).reshape(12)]
    Q.nms_enable, Q.nms_threshold, Q.fov_only = int(bool(nms_enable)), float(nms_threshold), int(bool(fov_only))
    Q.image_h, Q.image_w = float(image_shape[0]), float(image_shape[1])
    if nms_enable:
        order = np.ascontiguousarray(order, dtype=np.int64)
        iou = np.ascontiguousarray(iou, dtype=np.float32)
        assert order.shape == (k,) and iou.shape == (k, k)
    kept = np.zeros(k, dtype=np.int32)
    nk, tl = C.c_int32(0), C.c_int32(0)
    cap = 256 * k + 16
    for _ in range(2):   # 256 bytes per line hold every sane box; a corner next to the image plane prints longer numbers
        buf = C.create_string_buffer(cap)
        rc = lib.modest_label_lines(_np_ptr(objs8), _np_ptr(cs), k, _np_ptr(order) if nms_enable else None,
                                    _np_ptr(iou) if nms_enable else None, C.byref(Q), _np_ptr(kept), C.byref(nk), buf, cap,
                                    C.byref(tl))
        if rc != -3 or cap >= 4096 * k + 16:   # MODEST_ERR_CAPACITY: once more with the library's own line bound
            break
        cap = 4096 * k + 16
    check(rc, "modest_label_lines")
    return buf.raw[: tl.value].decode("ascii"), kept[: nk.value]


def cluster_stats(pts: torch.Tensor, pp: torch.Tensor, labels: torch.Tensor, n_clusters: int,
                  plane: np.ndarray, quantile: float, ctx: Optional[Context] = None) -> np.ndarray:
    """Per-cluster (count, min dist, max dist, a, b, gamma) for is_valid_cluster; (C,6) float64 host."""
    lib = load()
    _dev(pts, torch.float32, "pts")
    _dev(pp, torch.float32, "pp")
    _dev(labels, torch.int32, "labels")
    out = np.zeros((n_clusters, 6), dtype=np.float64)
    if n_clusters == 0:
        return out
    plane = np.ascontiguousarray(plane, dtype=np.float64).reshape(4)
    c = _ctx(ctx, pts)
    check(lib.modest_cluster_stats(c.handle, pts.data_ptr(), pts.shape[0], pts.shape[1], pp.data_ptr(),
                                   labels.data_ptr(), int(n_clusters), _np_ptr(plane), float(quantile),
                                   _np_ptr(out), _stream()), "modest_cluster_stats")
    return out


def boxes_pp_stats(rect_xyz: torch.Tensor, pp: torch.Tensor, boxes12: np.ndarray, quantile: float,
                   ctx: Optional[Context] = None) -> np.ndarray:
    """filter_by_ppscore statistics (combine_labels.py:41-60): per box (points inside, a, b, gamma);
    (K,4) float64 host.  boxes12 = the twelve float64 scalars of include/modest_hip.h."""
    lib = load()
    _dev(rect_xyz, torch.float64, "rect_xyz")
    _dev(pp, torch.float32, "pp")
    assert rect_xyz.ndim == 2 and rect_xyz.shape[1] == 3 and pp.shape[0] == rect_xyz.shape[0]
    boxes12 = np.ascontiguousarray(boxes12, dtype=np.float64).reshape(-1, 12)
    out = np.zeros((boxes12.shape[0], 4), dtype=np.float64)
    if boxes12.shape[0] == 0:
        return out
    c = _ctx(ctx, rect_xyz)
    check(lib.modest_boxes_pp_stats(c.handle, rect_xyz.data_ptr(), rect_xyz.shape[0], pp.data_ptr(),
                                    _np_ptr(boxes12), boxes12.shape[0], float(quantile), _np_ptr(out),
                                    _stream()), "modest_boxes_pp_stats")
    return out


# --------------------------------------------------------------------------- box fitting
def fit_boxes_closeness(pts_xz: torch.Tensor, offsets: Sequence[int], cossin: np.ndarray, d0: float = 1e-2,
                        return_beta: bool = False, ctx: Optional[Context] = None):
    """Index of the first strict maximum of the closeness criterion per cluster."""
    lib = load()
    _dev(pts_xz, torch.float64, "pts_xz")
    off = np.ascontiguousarray(np.asarray(offsets, dtype=np.int32))
    cs = np.ascontiguousarray(cossin, dtype=np.float64).reshape(-1, 2)
    ncl, na = off.shape[0] - 1, cs.shape[0]
    best = np.full(ncl, -1, dtype=np.int32)
    beta = np.zeros((ncl, na), dtype=np.float64) if return_beta else None
    c = _ctx(ctx, pts_xz)
    check(lib.modest_fit_boxes_closeness(c.handle, pts_xz.data_ptr(), _np_ptr(off), ncl, _np_ptr(cs), na,
                                         float(d0), _np_ptr(best), _np_ptr(beta) if return_beta else None,
                                         _stream()), "modest_fit_boxes_closeness")
    return (best, beta) if return_beta else best


def fit_boxes_closeness_host(pts_xz: np.ndarray, offsets: Sequence[int], cossin: np.ndarray, d0: float = 1e-2,
                             cossin90: Optional[np.ndarray] = None, ctx: Optional[Context] = None):
    """fit_boxes_closeness for cluster points in host memory ((m,2) float64): points and tables go to
    the device in one staged copy.  With ``cossin90`` ((cos, sin) of every table angle + pi/2) also returns
    the (C,8) extents of every cluster at the chosen heading and at heading + pi/2."""
    lib = load()
    pts = np.ascontiguousarray(pts_xz, dtype=np.float64).reshape(-1, 2)
    off = np.ascontiguousarray(np.asarray(offsets, dtype=np.int32))
    assert off[-1] == pts.shape[0]
    cs = np.ascontiguousarray(cossin, dtype=np.float64).reshape(-1, 2)
    ncl, na = off.shape[0] - 1, cs.shape[0]
    best = np.full(ncl, -1, dtype=np.int32)
    ext = None
    if cossin90 is not None:
        cossin90 = np.ascontiguousarray(cossin90, dtype=np.float64).reshape(na, 2)
        ext = np.zeros((ncl, 8), dtype=np.float64)
    c = ctx if ctx is not None else default_context(torch.cuda.current_device())
    check(lib.modest_fit_boxes_closeness_host(c.handle, _np_ptr(pts), _np_ptr(off), ncl, _np_ptr(cs), na, float(d0),
                                              _np_ptr(best), None if ext is None else _np_ptr(cossin90),
                                              None if ext is None else _np_ptr(ext), _stream()),
          "modest_fit_boxes_closeness_host")
    return best if ext is None else (best, ext)


def fit_boxes_variance(pts_xz: torch.Tensor, offsets: Sequence[int], cossin: np.ndarray, return_crit: bool = False,
                       ctx: Optional[Context] = None):
    """Index of the first strict maximum of the variance_to_edge criterion per cluster."""
    lib = load()
    _dev(pts_xz, torch.float64, "pts_xz")
    off = np.ascontiguousarray(np.asarray(offsets, dtype=np.int32))
    cs = np.ascontiguousarray(cossin, dtype=np.float64).reshape(-1, 2)
    ncl, na = off.shape[0] - 1, cs.shape[0]
    best = np.full(ncl, -1, dtype=np.int32)
    crit = np.zeros((ncl, na), dtype=np.float64) if return_crit else None
    c = _ctx(ctx, pts_xz)
    check(lib.modest_fit_boxes_variance(c.handle, pts_xz.data_ptr(), _np_ptr(off), ncl, _np_ptr(cs), na,
                                        _np_ptr(best), _np_ptr(crit) if return_crit else None, _stream()),
          "modest_fit_boxes_variance")
    return (best, crit) if return_crit else best


def fit_boxes_pca(pts_xz: torch.Tensor, offsets: Sequence[int], ctx: Optional[Context] = None) -> np.ndarray:
    """(K,8): PCA components (row-major 2x2) and the extent of every cluster along them."""
    lib = load()
    _dev(pts_xz, torch.float64, "pts_xz")
    off = np.ascontiguousarray(np.asarray(offsets, dtype=np.int32))
    out = np.zeros((off.shape[0] - 1, 8), dtype=np.float64)
    c = _ctx(ctx, pts_xz)
    check(lib.modest_fit_boxes_pca(c.handle, pts_xz.data_ptr(), _np_ptr(off), off.shape[0] - 1, _np_ptr(out),
                                   _stream()), "modest_fit_boxes_pca")
    return out


def lowest_point(pts_rect: torch.Tensor, boxes6: np.ndarray, ctx: Optional[Context] = None) -> np.ndarray:
    lib = load()
    _dev(pts_rect, torch.float64, "pts_rect")
    b = np.ascontiguousarray(boxes6, dtype=np.float64).reshape(-1, 6)
    out = np.zeros(b.shape[0], dtype=np.float64)
    c = _ctx(ctx, pts_rect)
    check(lib.modest_lowest_point(c.handle, pts_rect.data_ptr(), pts_rect.shape[0], _np_ptr(b), b.shape[0],
                                  _np_ptr(out), _stream()), "modest_lowest_point")
    return out


# --------------------------------------------------------------------------- BEV IoU / NMS
def boxes_iou_bev(a: torch.Tensor, b: torch.Tensor, overlap_only: bool = False) -> torch.Tensor:
    lib = load()
    _dev(a, torch.float32, "boxes_a")
    _dev(b, torch.float32, "boxes_b")
    assert a.shape[1] == 7 and b.shape[1] == 7
    out = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    fn = lib.modest_boxes_overlap_bev if overlap_only else lib.modest_boxes_iou_bev
    check(fn(a.data_ptr(), a.shape[0], b.data_ptr(), b.shape[0], out.data_ptr(), _stream()),
          "modest_boxes_iou_bev")
    return out


def boxes_iou_bev_host(a: np.ndarray, b: np.ndarray, ctx: Optional[Context] = None) -> np.ndarray:
    """Rotated BEV IoU of host box arrays (n,7) float32 -> (na, nb) float32 host matrix: boxes and matrix
    go through the context's pinned block, no device tensors and no copies."""
    lib = load()
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 7)
    b = np.ascontiguousarray(b, dtype=np.float32).reshape(-1, 7)
    out = np.zeros((a.shape[0], b.shape[0]), dtype=np.float32)
    c = ctx if ctx is not None else default_context(torch.cuda.current_device())
    check(lib.modest_boxes_iou_bev_host(c.handle, _np_ptr(a), a.shape[0], _np_ptr(b), b.shape[0], _np_ptr(out),
                                        _stream()), "modest_boxes_iou_bev_host")
    return out


def nms(boxes: torch.Tensor, thresh: float, rotated: bool = True, ctx: Optional[Context] = None) -> np.ndarray:
    """Greedy NMS over boxes already sorted by score; returns kept indices (int64, host)."""
    lib = load()
    _dev(boxes, torch.float32, "boxes")
    n = boxes.shape[0]
    keep = np.zeros(n, dtype=np.int64)
    num = C.c_int(0)
    c = _ctx(ctx, boxes)
    fn = lib.modest_nms_bev if rotated else lib.modest_nms_normal
    check(fn(c.handle, boxes.data_ptr(), n, float(thresh), _np_ptr(keep), C.byref(num), _stream()), "modest_nms")
    return keep[: num.value]
