"""KITTI calibration / box helpers on the hot path (host side, float64 numpy).

Mirror of the members of the reference's ``generate_cluster_mask/utils/
kitti_util.py`` that the seed-label path uses: ``Calibration`` (:200-371:
``project_velo_to_rect`` :327-329, ``project_rect_to_image`` :334-342),
``roty`` :383-389, ``compute_box_3d`` :453-488 (without the behind-camera
early-out, which the reference has commented out at :481-483) and
``project_to_image`` :430-450.  These are per-scan 3x4 products on a handful of
boxes; they stay on the host.
"""
from __future__ import annotations

import numpy as np


def roty(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def inverse_rigid_trans(Tr):
    inv = np.zeros_like(Tr)
    inv[0:3, 0:3] = np.transpose(Tr[0:3, 0:3])
    inv[0:3, 3] = np.dot(-np.transpose(Tr[0:3, 0:3]), Tr[0:3, 3])
    return inv


class Calibration(object):
    """Calibration matrices of one KITTI frame (``calib/NNNNNN.txt``)."""

    def __init__(self, calib_filepath):
        calibs = self.read_calib_file(calib_filepath)
        self.P = np.reshape(calibs["P2"], [3, 4])
        self.V2C = np.reshape(calibs["Tr_velo_to_cam"], [3, 4])
        self.C2V = inverse_rigid_trans(self.V2C)
        self.R0 = np.reshape(calibs["R0_rect"], [3, 3])
        self.P3 = np.reshape(calibs["P3"], [3, 4])
        self.c_u, self.c_v = self.P[0, 2], self.P[1, 2]
        self.f_u, self.f_v = self.P[0, 0], self.P[1, 1]
        self.b_x = self.P[0, 3] / (-self.f_u)
        self.b_y = self.P[1, 3] / (-self.f_v)

    @staticmethod
    def read_calib_file(filepath):
        data = {}
        with open(filepath, "r") as f:
            for line in f.readlines():
                line = line.rstrip()
                if len(line) == 0:
                    continue
                key, value = line.split(":", 1)
                try:
                    data[key] = np.array([float(x) for x in value.split()])
                except ValueError:
                    pass
        return data

    @staticmethod
    def cart2hom(pts_3d):
        n = pts_3d.shape[0]
        return np.hstack((pts_3d, np.ones((n, 1))))

    def project_velo_to_ref(self, pts_3d_velo):
        return np.dot(self.cart2hom(pts_3d_velo), np.transpose(self.V2C))

    def project_ref_to_rect(self, pts_3d_ref):
        return np.transpose(np.dot(self.R0, np.transpose(pts_3d_ref)))

    def project_velo_to_rect(self, pts_3d_velo):
        return self.project_ref_to_rect(self.project_velo_to_ref(pts_3d_velo))

    def project_rect_to_image(self, pts_3d_rect):
        pts_2d = np.dot(self.cart2hom(pts_3d_rect), np.transpose(self.P))
        pts_2d[:, 0] /= pts_2d[:, 2]
        pts_2d[:, 1] /= pts_2d[:, 2]
        return pts_2d[:, 0:2]


def project_to_image(pts_3d, P):
    n = pts_3d.shape[0]
    ext = np.hstack((pts_3d, np.ones((n, 1))))
    pts_2d = np.dot(ext, np.transpose(P))
    pts_2d[:, 0] /= pts_2d[:, 2]
    pts_2d[:, 1] /= pts_2d[:, 2]
    return pts_2d[:, 0:2]


def compute_box_3d(obj, P):
    """(8,2) image corners and (8,3) rect corners of a box with bottom centre obj.t."""
    R = roty(obj.ry)
    l, w, h = obj.l, obj.w, obj.h
    x_corners = [l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2]
    y_corners = [0, 0, 0, 0, -h, -h, -h, -h]
    z_corners = [w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2]
    corners_3d = np.dot(R, np.vstack([x_corners, y_corners, z_corners]))
    corners_3d[0, :] = corners_3d[0, :] + obj.t[0]
    corners_3d[1, :] = corners_3d[1, :] + obj.t[1]
    corners_3d[2, :] = corners_3d[2, :] + obj.t[2]
    corners_2d = project_to_image(np.transpose(corners_3d), P)
    return corners_2d, np.transpose(corners_3d)
