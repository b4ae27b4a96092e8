"""ORACLE — test infrastructure only, never imported by the product path.

CPU restatement of MODEST's label-file stage: ``generate_cluster_mask/
gen_label_files.py`` with ``utils/pointcloud_utils.py:320-379`` and the
calibration/box helpers of ``utils/kitti_util.py`` (reference checkout).
BEV IoU comes from oracle/iou3d_oracle.c (plain-C restatement) or, where
present, from the reference's own iou3d_cpu.cpp build (oracle/_ref).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(HERE, "_build", "libiou3d_oracle.so")
        if not os.path.exists(so):
            subprocess.run(["make", "-C", HERE], check=True, capture_output=True)
        lib = ctypes.CDLL(so)
        lib.modest_oracle_boxes_bev.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                                ctypes.c_void_p, ctypes.c_int]
        lib.modest_oracle_boxes_bev.restype = None
        lib.modest_oracle_nms.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
        lib.modest_oracle_nms.restype = ctypes.c_int
        _LIB = lib
    return _LIB


def boxes_iou_bev(boxes_a, boxes_b, overlap_only=False):
    """iou3d_nms_utils.boxes_iou_bev (utils/iou3d_nms/iou3d_nms_utils.py:37-51)."""
    a = np.ascontiguousarray(boxes_a, dtype=np.float32)
    b = np.ascontiguousarray(boxes_b, dtype=np.float32)
    out = np.zeros((a.shape[0], b.shape[0]), dtype=np.float32)
    _lib().modest_oracle_boxes_bev(a.ctypes.data, a.shape[0], b.ctypes.data, b.shape[0], out.ctypes.data,
                                   1 if overlap_only else 0)
    return out


def boxes_iou_bev_reference(boxes_a, boxes_b):
    """The reference's own iou3d_cpu.cpp (oracle/_ref/iou3d_ref.so), or None if not built."""
    ref_dir = os.path.join(HERE, "_ref")
    if not os.path.exists(os.path.join(ref_dir, "iou3d_ref.so")):
        return None
    import torch
    if ref_dir not in sys.path:
        sys.path.insert(0, ref_dir)
    import iou3d_ref
    a = torch.from_numpy(np.ascontiguousarray(boxes_a, dtype=np.float32))
    b = torch.from_numpy(np.ascontiguousarray(boxes_b, dtype=np.float32))
    out = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.float32)
    iou3d_ref.boxes_iou_bev_cpu(a, b, out)
    return out.numpy()


def nms(boxes_sorted, thresh, rotated=True):
    """nms_gpu / nms_normal_gpu (src/iou3d_nms.cpp:90-186) on score-sorted boxes."""
    b = np.ascontiguousarray(boxes_sorted, dtype=np.float32)
    keep = np.zeros(b.shape[0], dtype=np.int64)
    k = _lib().modest_oracle_nms(b.ctypes.data, b.shape[0], ctypes.c_float(thresh), 1 if rotated else 0,
                                 keep.ctypes.data)
    return keep[:k]


def objs_to_boxes(objs):
    """utils/pointcloud_utils.py:322-324 (float64 -> float32)."""
    return np.array([[o.t[0], o.t[2], 0, o.l, o.w, o.h, -o.ry] for o in objs]).astype(np.float32).reshape(-1, 7)


def nms_from_iou(overlaps_bev, nms_threshold, scores=None):
    """Greedy suppression of utils/pointcloud_utils.py:329-343 given the IoU matrix."""
    mask = np.ones(overlaps_bev.shape[0], dtype=bool)
    if scores is not None:
        order = np.argsort(scores)[::-1]
    else:
        order = np.diag(overlaps_bev).argsort()[::-1]
    for idx in order:
        if not mask[idx]:
            continue
        mask[overlaps_bev[idx] > nms_threshold] = False
        mask[idx] = True
    return mask


def objs_nms(objs, use_score_rank=False, nms_threshold=0.1, iou_fn=boxes_iou_bev):
    """utils/pointcloud_utils.py:320-344."""
    boxes = objs_to_boxes(objs)
    iou = iou_fn(boxes, boxes)
    mask = nms_from_iou(iou, nms_threshold, [o.score for o in objs] if use_score_rank else None)
    return [objs[i] for i in range(len(objs)) if mask[i]]


class Calibration:
    """utils/kitti_util.py:200-371 (the members the hot path uses)."""

    def __init__(self, calib_filepath):
        data = {}
        with open(calib_filepath, "r") as f:
            for line in f.readlines():
                line = line.rstrip()
                if len(line) == 0:
                    continue
                key, value = line.split(":", 1)
                try:
                    data[key] = np.array([float(x) for x in value.split()])
                except ValueError:
                    pass
        self.P = np.reshape(data["P2"], [3, 4])
        self.V2C = np.reshape(data["Tr_velo_to_cam"], [3, 4])
        self.R0 = np.reshape(data["R0_rect"], [3, 3])
        self.P3 = np.reshape(data["P3"], [3, 4])

    @staticmethod
    def cart2hom(pts_3d):
        return np.hstack((pts_3d, np.ones((pts_3d.shape[0], 1))))

    def project_velo_to_rect(self, pts_3d_velo):
        ref = np.dot(self.cart2hom(pts_3d_velo), np.transpose(self.V2C))
        return np.transpose(np.dot(self.R0, np.transpose(ref)))

    def project_rect_to_image(self, pts_3d_rect):
        pts_2d = np.dot(self.cart2hom(pts_3d_rect), np.transpose(self.P))
        pts_2d[:, 0] /= pts_2d[:, 2]
        pts_2d[:, 1] /= pts_2d[:, 2]
        return pts_2d[:, 0:2]


def roty(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def compute_box_3d(obj, P):
    """utils/kitti_util.py:453-488 (no behind-camera early-out)."""
    R = roty(obj.ry)
    l, w, h = obj.l, obj.w, obj.h
    x_corners = [l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2]
    y_corners = [0, 0, 0, 0, -h, -h, -h, -h]
    z_corners = [w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2]
    corners_3d = np.dot(R, np.vstack([x_corners, y_corners, z_corners]))
    corners_3d[0, :] = corners_3d[0, :] + obj.t[0]
    corners_3d[1, :] = corners_3d[1, :] + obj.t[1]
    corners_3d[2, :] = corners_3d[2, :] + obj.t[2]
    pts = np.transpose(corners_3d)
    ext = np.hstack((pts, np.ones((pts.shape[0], 1))))
    pts_2d = np.dot(ext, np.transpose(P))
    pts_2d[:, 0] /= pts_2d[:, 2]
    pts_2d[:, 1] /= pts_2d[:, 2]
    return pts_2d[:, 0:2], pts


def is_within_fov(obj, calib, image_shape):
    """utils/pointcloud_utils.py:373-379."""
    center = obj.t.copy()
    center[1] -= obj.h / 2
    uv = calib.project_rect_to_image(center.reshape(1, -1)).squeeze()
    return uv[0] < image_shape[1] and uv[0] >= 0 and uv[1] < image_shape[0] and uv[1] >= 0 and center[2] > 0


def objs2label(objs, calib, obj_type="Dynamic", with_score=False):
    """utils/pointcloud_utils.py:347-370."""
    lines = []
    for obj in objs:
        alpha = -np.arctan2(obj.t[0], obj.t[2]) + obj.ry
        corners_2d = compute_box_3d(obj, calib.P)[0]
        box = np.concatenate([np.min(corners_2d, axis=0), np.max(corners_2d, axis=0)], axis=0)
        s = (f"{obj_type} -1 -1 {alpha:.4f} {box[0]:.4f} {box[1]:.4f} {box[2]:.4f} {box[3]:.4f} "
             f"{obj.h:.4f} {obj.w:.4f} {obj.l:.4f} {obj.t[0]:.4f} {obj.t[1]:.4f} {obj.t[2]:.4f} {obj.ry:.4f}")
        if with_score:
            s += f" {getattr(obj, 'score', -1):.4f}"
        lines.append(s)
    return "\n".join(lines)


def gen_label_scan(objs, calib, image_shape=(1024, 1224), fov_only=True, nms_enable=True, nms_threshold=0.1,
                   iou_fn=boxes_iou_bev):
    """gen_label_files.py:41-52 for one scan -> label text."""
    if nms_enable and len(objs) > 0:
        objs = objs_nms(objs, nms_threshold=nms_threshold, iou_fn=iou_fn)
    if fov_only:
        objs = [o for o in objs if is_within_fov(o, calib, image_shape)]
    return objs2label(objs, calib), objs
