"""ORACLE (test infrastructure, never imported by the product path): CPU restatement of the
self-training label merge, generate_cluster_mask/combine_labels.py of YurongYou/MODEST.
Every function cites the reference lines it restates; numpy does the arithmetic exactly as the
reference does (same dtypes, same calls).  Pinned by tests/golden/combine.npz, which was produced
by importing the reference's own combine_labels module (tools/make_golden_combine.py)."""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np

from oracle import labels as ol


def predicts2objs(preds):
    """combine_labels.py:23-34 -- OpenPCDet prediction dict -> objects (dimensions = l, h, w)."""
    objs = []
    for i in range(preds["location"].shape[0]):
        obj = SimpleNamespace()
        obj.t = preds["location"][i]
        obj.l = preds["dimensions"][i][0]
        obj.h = preds["dimensions"][i][1]
        obj.w = preds["dimensions"][i][2]
        obj.ry = preds["rotation_y"][i]
        obj.score = preds["score"][i]
        objs.append(obj)
    return objs


def add_area_score(objs):
    """combine_labels.py:37-39 -- seed boxes rank below every detection, larger footprint first."""
    for obj in objs:
        obj.score = -999 + obj.w * obj.l


def box_mask(ptc_rect, obj):
    """combine_labels.py:42-57 -- the boolean point mask of filter_by_ppscore."""
    ry, l, w = obj.ry, obj.l, obj.w
    xz_center = obj.t[[0, 2]]
    ptc_xz = ptc_rect[:, [0, 2]] - xz_center
    rot = np.array([[np.cos(ry), -np.sin(ry)], [np.sin(ry), np.cos(ry)]])
    ptc_xz = ptc_xz @ rot.T
    mask = (ptc_xz[:, 0] > -l / 2) & (ptc_xz[:, 0] < l / 2) & (ptc_xz[:, 1] > -w / 2) & (ptc_xz[:, 1] < w / 2)
    y_mask = (ptc_rect[:, 1] > obj.t[1] - obj.h) * (ptc_rect[:, 1] <= obj.t[1])
    return mask * y_mask


def filter_by_ppscore(ptc_rect, pp_score, obj, percentile=50, threshold=0.5):
    """combine_labels.py:41-60."""
    mask = box_mask(ptc_rect, obj)
    if mask.sum() == 0 or np.percentile(pp_score[mask], percentile) > threshold:
        return False
    return True


def combine_scan(ptc, pp_score, calib, det_bbox, gen_obj, percentile=50, threshold=0.5, score_filtering=-1,
                 nms_threshold=0.1, fov_only=True, image_shape=(1024, 1224), with_score=False):
    """combine_labels.py:94-121 for one frame: returns (label text, kept objects, per-box keep flags)."""
    ptc_in_rect = calib.project_velo_to_rect(ptc[:, :3])
    dets = predicts2objs(det_bbox)
    flags = [bool(filter_by_ppscore(ptc_in_rect, pp_score, o, percentile=percentile, threshold=threshold) &
                  (o.score > score_filtering)) for o in dets]
    det_obj = [o for o, f in zip(dets, flags) if f]
    add_area_score(gen_obj)
    objs = det_obj + gen_obj
    if len(objs) > 0:
        objs = ol.objs_nms(objs, nms_threshold=nms_threshold, use_score_rank=True)
    if fov_only:
        objs = [o for o in objs if ol.is_within_fov(o, calib, image_shape)]
    return ol.objs2label(objs, calib, with_score=with_score), objs, flags
