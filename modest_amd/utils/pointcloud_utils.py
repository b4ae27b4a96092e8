"""Geometry helpers of the seed-label path, backed by libmodest_hip.so.

Host-side mirror of the reference's ``generate_cluster_mask/utils/
pointcloud_utils.py`` (same function names and argument meaning; file:line
cited per function).  Arrays may be numpy (uploaded, result returned as numpy)
or PyTorch-ROCm device tensors (result stays on the device).  Every O(N) loop
runs in a HIP kernel; what stays on the host is per-box scalar arithmetic
written with the same numpy expressions the reference uses, so that it rounds
identically.
"""
from __future__ import annotations

import types
from typing import List, Sequence

import numpy as np
import torch

from .. import ops
from .._lib import ModestHipError
from . import kitti_util
from .iou3d_nms import iou3d_nms_utils
from .ransac import ransac_plane


def _device(device=None):
    if device is not None:
        return torch.device(device)
    if not torch.cuda.is_available():
        raise RuntimeError("modest_amd needs an MI355X (no CPU fallback): torch.cuda.is_available() is False")
    return torch.device("cuda", torch.cuda.current_device())


def to_device(a, dtype=torch.float32, device=None) -> torch.Tensor:
    if isinstance(a, torch.Tensor):
        t = a if a.is_cuda else a.to(_device(device))
        return t.to(dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a)).to(_device(device)).to(dtype).contiguous()


def load_velo_scan(velo_filename):
    """(:22-25) raw KITTI scan, (n,4) float32 [x, y, z, intensity]."""
    return np.fromfile(velo_filename, dtype=np.float32).reshape((-1, 4))


def transform_points(pts_3d_ref, Tr, remove_center: bool = False):
    """(:11-19) [p,1] @ Tr^T, float32; optionally drops the ego-vehicle box first
    (pre_compute_pp_score.py:48-52).  Accepts (n,3) or raw (n,4) frames."""
    is_np = not isinstance(pts_3d_ref, torch.Tensor)
    out = ops.transform_points(to_device(pts_3d_ref), np.asarray(Tr, dtype=np.float32), remove_center=remove_center)
    return out.cpu().numpy() if is_np else out


def plane_from_linear_model(coef, intercept):
    """(:53-62) unit normal with +z up from z = c0 x + c1 y + b."""
    w = np.zeros(3)
    w[0] = coef[0]
    w[1] = coef[1]
    w[2] = -1.0
    h = intercept
    norm = np.linalg.norm(w)
    w /= norm
    h = h / norm
    result = np.array((w[0], w[1], w[2], h))
    result *= -1
    return result


def estimate_plane(origin_ptc, max_hs=-1.5, it=1, ptc_range=((-20, 70), (-20, 20)), random_state=None,
                   return_info: bool = False, prepared=None):
    """(:44-65) RANSAC ground plane of the points with z < max_hs inside ptc_range
    (strict bounds) -> (4,) float64 [n, d] with n.z > 0.

    ``random_state``: int seed or numpy RandomState.  The reference draws from
    numpy's global stream (sklearn's default), which makes its result depend on
    processing order; pass ``np.random.mtrand._rand`` to reproduce that.  The
    loop body's trailing above_plane (:63-64) is dead work for it=1 and skipped.
    ``prepared`` = (candidates, MAD threshold) from ``prepare_planes`` for the same arguments.
    """
    assert it == 1, "the reference only ever runs one iteration"
    if prepared is not None:
        cand, thr = prepared
    else:
        cand, _ = ops.plane_candidates(to_device(origin_ptc), max_hs, ptc_range)
        thr = None
    res = ransac_plane(cand, random_state=random_state, thr=thr)
    plane = plane_from_linear_model(res.coef, res.intercept)
    return (plane, res) if return_info else plane


def prepare_planes(pts_dev, specs):
    """The RNG-independent part of several estimate_plane calls on one scan, batched: candidate
    selection per (max_hs, ptc_range) and all MAD thresholds from ONE launch (a workgroup each).
    Returns a list of (candidates, threshold) for ``estimate_plane(..., prepared=...)``; a set with
    fewer than one candidate gets threshold None (the fit itself then raises, where the reference does)."""
    if len(specs) == 2:   # a scan's two fits: one pass over the rows, one round trip
        return ops.plane_prepare(pts_dev, specs)
    cands = [ops.plane_candidates(pts_dev, max_hs, rng)[0] for max_hs, rng in specs]
    live = [c for c in cands if c.shape[0] >= 1]
    thr = iter(ops.mad_threshold_batch(live)) if live else iter(())
    return [(c, next(thr) if c.shape[0] >= 1 else None) for c in cands]


def distance_to_plane(ptc, plane, directional=False):
    """(:76-81) host numpy (used on per-cluster point sets)."""
    d = ptc @ plane[:3] + plane[3]
    if not directional:
        d = np.abs(d)
    d /= np.sqrt((plane[:3] ** 2).sum())
    return d


def above_plane(ptc, plane, offset=0.05, only_range=((-30, 30), (-30, 30))):
    """(:68-74) True for points to KEEP: not (dist < offset and inside only_range)."""
    is_np = not isinstance(ptc, torch.Tensor)
    big = [[-np.inf, np.inf], [-np.inf, np.inf]]
    mask, _, _ = ops.plane_range_mask(to_device(ptc), plane, offset, only_range, big)
    return mask.cpu().numpy() if is_np else mask


def angle_table(delta=0.1):
    """The 901 candidate headings of closeness_rectangle (:170-175) as (cos, sin),
    computed with the host's numpy exactly as the reference computes them."""
    ang = np.arange(0, 90 + delta, delta) / 180. * np.pi
    return ang, np.stack([np.cos(ang), np.sin(ang)], axis=1)


_ANGLES = {}


def _angles(delta):
    if delta not in _ANGLES:
        _ANGLES[delta] = angle_table(delta)
    return _ANGLES[delta]


_ANGLES90 = {}


def _angles90(delta):
    """(cos, sin) of every table angle + pi/2, each evaluated as rectangle_at_angle evaluates it
    (numpy scalar calls on ``angle + np.pi / 2``), once per process."""
    if delta not in _ANGLES90:
        ang, _ = _angles(delta)
        _ANGLES90[delta] = np.array([[np.cos(a + np.pi / 2), np.sin(a + np.pi / 2)] for a in ang])
    return _ANGLES90[delta]


def rectangle_from_extents(choose_angle, ext):
    """rectangle_at_angle given the cluster's extents along the heading (ext[:4]) and along
    heading + pi/2 (ext[4:]) from the device: the scalar tail of (:188-216)."""
    angle = choose_angle
    min_x, max_x, min_y, max_y = ext[0], ext[1], ext[2], ext[3]
    if (max_x - min_x) < (max_y - min_y):
        angle = choose_angle + np.pi / 2
        min_x, max_x, min_y, max_y = ext[4], ext[5], ext[6], ext[7]
    c, s = np.cos(angle), np.sin(angle)
    components = np.array([[c, s], [-s, c]])
    area = (max_x - min_x) * (max_y - min_y)
    rval = np.array([[max_x, min_y], [min_x, min_y], [min_x, max_y], [max_x, max_y]])
    return rval @ components, angle, area


def rectangle_at_angle(cluster_ptc, choose_angle):
    """(:188-216) tight rectangle at the chosen heading; rotated by 90 degrees when
    needed so that the first side is the long one.  Host numpy, per cluster."""
    angle = choose_angle
    c, s = np.cos(angle), np.sin(angle)   # (each is evaluated twice in the reference: same values)
    components = np.array([[c, s], [-s, c]])
    projection = cluster_ptc @ components.T
    px, py = projection[:, 0], projection[:, 1]
    min_x, max_x, min_y, max_y = px.min(), px.max(), py.min(), py.max()
    if (max_x - min_x) < (max_y - min_y):
        angle = choose_angle + np.pi / 2
        c, s = np.cos(angle), np.sin(angle)
        components = np.array([[c, s], [-s, c]])
        projection = cluster_ptc @ components.T
        px, py = projection[:, 0], projection[:, 1]
        min_x, max_x, min_y, max_y = px.min(), px.max(), py.min(), py.max()
    area = (max_x - min_x) * (max_y - min_y)
    rval = np.array([[max_x, min_y], [min_x, min_y], [min_x, max_y], [max_x, max_y]])
    return rval @ components, angle, area


def closeness_rectangles(clusters_xz: Sequence[np.ndarray], delta=0.1, d0=1e-2):
    """Batched closeness_rectangle (:167-216): the 901-angle search of every
    cluster runs in one HIP launch; returns [(corners, angle, area)]."""
    if len(clusters_xz) == 0:
        return []
    ang, cs = _angles(delta)
    off = np.cumsum([0] + [len(c) for c in clusters_xz]).astype(np.int32)
    pts = np.concatenate(clusters_xz).astype(np.float64)
    try:      # heading and the cluster's extents along it from one launch
        best, ext = ops.fit_boxes_closeness_host(pts, off, cs, d0, cossin90=_angles90(delta))
    except ModestHipError:   # a cluster too large for that kernel: extents on the host
        best = ops.fit_boxes_closeness_host(pts, off, cs, d0)
        return [rectangle_at_angle(c, ang[b]) for c, b in zip(clusters_xz, best)]
    return [rectangle_from_extents(ang[b], e) for b, e in zip(best, ext)]


def variance_rectangles(clusters_xz: Sequence[np.ndarray], delta=0.1):
    """Batched variance_rectangle (:218-275): same angle table, criterion
    -var(Dx[Dx<Dy]) - var(Dy[Dy<Dx]) on the device, the rectangle at the chosen angle on the host."""
    if len(clusters_xz) == 0:
        return []
    ang, cs = _angles(delta)
    off = np.cumsum([0] + [len(c) for c in clusters_xz]).astype(np.int32)
    pts = to_device(np.concatenate(clusters_xz).astype(np.float64), dtype=torch.float64)
    best = ops.fit_boxes_variance(pts, off, cs)
    return [rectangle_at_angle(c, ang[b]) for c, b in zip(clusters_xz, best)]


def pca_rectangles(clusters_xz: Sequence[np.ndarray]):
    """Batched PCA_rectangle (:189-206): principal axes and extents on the device."""
    if len(clusters_xz) == 0:
        return []
    off = np.cumsum([0] + [len(c) for c in clusters_xz]).astype(np.int32)
    pts = to_device(np.concatenate(clusters_xz).astype(np.float64), dtype=torch.float64)
    out = []
    for r in ops.fit_boxes_pca(pts, off):
        components = r[:4].reshape(2, 2)
        min_x, max_x, min_y, max_y = r[4], r[5], r[6], r[7]
        area = (max_x - min_x) * (max_y - min_y)
        rval = np.array([[max_x, min_y], [min_x, min_y], [min_x, max_y], [max_x, max_y]]) @ components
        out.append((rval, np.arctan2(components[0, 1], components[0, 0]), area))
    return out


def min_area_rectangles(clusters_xz: Sequence[np.ndarray]):
    """minimum_bounding_rectangle (:88-147, fit_method='min_zx_area_fit') per cluster.

    The reference tries the directions of the hull edges ``hull[1:] - hull[:-1]`` in the vertex order
    scipy's ConvexHull (Qhull) reports and never the closing edge, so its answer depends on the
    vertex Qhull happens to start from.  The hull therefore comes from the same library call on the
    host (tens of vertices; the only O(n) step of this branch), the caliper arithmetic follows the
    reference line by line in numpy; the lowest-point search of get_obj stays on the device."""
    from scipy.spatial import ConvexHull
    half_pi = np.pi / 2.
    out = []
    for pts in clusters_xz:
        hull = pts[ConvexHull(pts).vertices]
        step = hull[1:] - hull[:-1]
        ang = np.unique(np.abs(np.mod(np.arctan2(step[:, 1], step[:, 0]), half_pi)))
        rot = np.vstack([np.cos(ang), np.cos(ang - half_pi), np.cos(ang + half_pi), np.cos(ang)]).T.reshape((-1, 2, 2))
        turned = np.dot(rot, hull.T)
        lo_x, hi_x = np.nanmin(turned[:, 0], axis=1), np.nanmax(turned[:, 0], axis=1)
        lo_y, hi_y = np.nanmin(turned[:, 1], axis=1), np.nanmax(turned[:, 1], axis=1)
        area = (hi_x - lo_x) * (hi_y - lo_y)
        k = np.argmin(area)
        r = rot[k]
        corners = np.zeros((4, 2))
        corners[0] = np.dot([hi_x[k], lo_y[k]], r)
        corners[1] = np.dot([lo_x[k], lo_y[k]], r)
        corners[2] = np.dot([lo_x[k], hi_y[k]], r)
        corners[3] = np.dot([hi_x[k], hi_y[k]], r)
        out.append((corners, ang[k], area[k]))
    return out


def closeness_rectangle(cluster_ptc, delta=0.1, d0=1e-2):
    """(:167-216) single-cluster form."""
    return closeness_rectangles([np.asarray(cluster_ptc, dtype=np.float64)], delta, d0)[0]


def get_lowest_points_rect(full_ptc_dev: torch.Tensor, centers, ls, ws, rys) -> np.ndarray:
    """Batched get_lowest_point_rect (:278-290) over boxes, one launch."""
    rys = np.asarray(rys, dtype=np.float64)
    centers = np.asarray(centers, dtype=np.float64).reshape(-1, 2)
    boxes6 = np.stack([centers[:, 0], centers[:, 1], np.asarray(ls, dtype=np.float64),
                       np.asarray(ws, dtype=np.float64), np.cos(rys), np.sin(rys)], axis=1)
    bottom = ops.lowest_point(full_ptc_dev, boxes6)
    if np.any(np.isinf(bottom)):
        raise ValueError("zero-size array to reduction operation maximum which has no identity")
    return bottom


def get_lowest_point_rect(ptc, xz_center, l, w, ry):
    """(:278-290) single-box form."""
    return get_lowest_points_rect(to_device(ptc, torch.float64), [xz_center], [l], [w], [ry])[0]


def get_objs(clusters_rect: List[np.ndarray], full_ptc, fit_method="min_zx_area_fit"):
    """Batched get_obj (:292-317) for a scan: boxes of all clusters.
    clusters_rect: list of (n_c,3) float64 rect-frame points; full_ptc (N,3) float64."""
    fitters = {"closeness_to_edge": closeness_rectangles, "variance_to_edge": variance_rectangles,
               "PCA": pca_rectangles, "min_zx_area_fit": min_area_rectangles}
    if fit_method not in fitters:
        raise NotImplementedError(fit_method)
    if len(clusters_rect) == 0:
        return []
    fits = fitters[fit_method]([c[:, [0, 2]] for c in clusters_rect])
    ls, ws, cs, rys = [], [], [], []
    for corners, ry, _ in fits:
        ry = ry * -1
        ls.append(np.linalg.norm(corners[0] - corners[1]))
        ws.append(np.linalg.norm(corners[0] - corners[-1]))
        cs.append((corners[0] + corners[2]) / 2)
        rys.append(ry)
    bottoms = get_lowest_points_rect(to_device(full_ptc, torch.float64), cs, ls, ws, rys)
    objs = []
    for ptc, (corners, _, area), l, w, c, ry, bottom in zip(clusters_rect, fits, ls, ws, cs, rys, bottoms):
        h = bottom - ptc[:, 1].min()
        obj = types.SimpleNamespace()
        obj.t = np.array([c[0], bottom, c[1]])
        obj.l, obj.w, obj.h, obj.ry = l, w, h, ry
        obj.volume = area * h
        objs.append(obj)
    return objs


def get_obj(ptc, full_ptc, fit_method="min_zx_area_fit"):
    """(:292-317) single-cluster form."""
    return get_objs([np.asarray(ptc, dtype=np.float64)], full_ptc, fit_method)[0]


def objs_nms(objs, use_score_rank=False, nms_threshold=0.1, after_device=None):
    """(:320-344) BEV NMS of boxes in the rect frame; keeps the original order.
    ``after_device``: called once when the device part (the IoU matrix) is done and only host work is left."""
    boxes = np.array([[obj.t[0], obj.t[2], 0, obj.l, obj.w, obj.h, -obj.ry] for obj in objs])
    # float32 boxes as the reference's .float() makes them; the IoU kernel reads them from and writes
    # the matrix to pinned host memory (same kernel as iou3d_nms_utils.boxes_iou_bev)
    overlaps_bev = ops.boxes_iou_bev_host(boxes.astype(np.float32), boxes.astype(np.float32))
    if after_device is not None:
        after_device()
    mask = np.ones(overlaps_bev.shape[0], dtype=bool)
    if use_score_rank:
        order = np.argsort([obj.score for obj in objs])[::-1]
    else:
        order = np.diag(overlaps_bev).argsort()[::-1]
    for idx in order:
        if not mask[idx]:
            continue
        mask[overlaps_bev[idx] > nms_threshold] = False
        mask[idx] = True
    return [objs[i] for i in range(len(objs)) if mask[i]]


def objs2label(objs, calib, obj_type="Dynamic", with_score=False):
    """(:347-370) KITTI label lines, all fields ``%.4f``, joined by newlines."""
    lines = []
    for obj in objs:
        alpha = -np.arctan2(obj.t[0], obj.t[2]) + obj.ry
        corners_2d = kitti_util.compute_box_3d(obj, calib.P)[0]
        b = np.concatenate([np.min(corners_2d, axis=0), np.max(corners_2d, axis=0)], axis=0)
        fields = [alpha, b[0], b[1], b[2], b[3], obj.h, obj.w, obj.l, obj.t[0], obj.t[1], obj.t[2], obj.ry]
        if with_score:
            fields.append(obj.score if hasattr(obj, "score") else -1)
        lines.append(f"{obj_type} -1 -1 " + " ".join(f"{v:.4f}" for v in fields))
    return "\n".join(lines)


def is_within_fov(obj, calib, image_shape):
    """(:373-379) box centre projects inside the image and lies in front of the camera."""
    center = obj.t.copy()
    center[1] -= obj.h / 2
    uv = calib.project_rect_to_image(center.reshape(1, -1)).squeeze()
    return uv[0] < image_shape[1] and uv[0] >= 0 and uv[1] < image_shape[0] and uv[1] >= 0 and center[2] > 0
