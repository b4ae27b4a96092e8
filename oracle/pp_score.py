"""ORACLE — test infrastructure only, never imported by the product path.

CPU restatement of MODEST's PP-score stage (``generate_cluster_mask/
pre_compute_pp_score.py`` in the reference checkout).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  Each function cites the reference lines it follows.

Third-party arithmetic the reference delegates to and that is NOT under
/root/reference: ``scipy.spatial.cKDTree`` (unpinned by the reference,
``README.md:35``; here scipy 1.15.3) and numpy/OpenBLAS ``np.dot``.  The same
calls are made here, so this oracle is the reference's own CPU path up to the
Hydra/file plumbing.  Pinned by ``tests/golden/pp_*.npz`` which were produced
by importing the reference itself (``tools/make_goldens.py``).
"""
from __future__ import annotations

import numpy as np
from scipy.spatial import cKDTree


def kitti2nu(nusc: bool) -> np.ndarray:
    """``Quaternion(axis=(0,0,1), angle=pi or pi/2).transformation_matrix``
    (pre_compute_pp_score.py:22-24).  pyquaternion 0.9.x is absent from this
    image; its published algorithm is restated: q = (cos(a/2), axis*sin(a/2)),
    R = (Q @ conj(Qbar).T)[1:,1:] with the 4x4 left/right product matrices."""
    angle = np.pi / 2 if nusc else np.pi
    axis = np.array([0.0, 0.0, 1.0])
    w = np.cos(angle / 2.0)
    x, y, z = axis * np.sin(angle / 2.0)
    q_matrix = np.array([[w, -x, -y, -z], [x, w, -z, y], [y, z, w, -x], [z, -y, x, w]])
    q_bar_matrix = np.array([[w, -x, -y, -z], [x, w, z, -y], [y, -z, w, x], [z, y, -x, w]])
    rot = np.dot(q_matrix, q_bar_matrix.conj().transpose())[1:][:, 1:]
    t = np.eye(4)
    t[:3, :3] = rot
    return t


def get_relative_pose(fixed_l2e, fixed_ego, query_l2e, query_ego, KITTI2NU):
    """pre_compute_pp_score.py:27-28."""
    return np.linalg.solve(
        KITTI2NU,
        np.linalg.solve(fixed_l2e, np.linalg.solve(fixed_ego, query_ego @ query_l2e @ KITTI2NU)),
    ).astype(np.float32)


def transform_points(pts_3d_ref, Tr):
    """utils/pointcloud_utils.py:11-19 (float32 BLAS product)."""
    n = pts_3d_ref.shape[0]
    hom = np.hstack((pts_3d_ref, np.ones((n, 1), dtype=np.float32)))
    return np.dot(hom, np.transpose(Tr)).reshape(-1, 4)[:, 0:3]


def transform_points_fma(pts, Tr):
    """Explicit statement of the rounding the float32 BLAS product was observed
    to have on this image: acc = x*a; acc = fma(y,b,acc);
    acc = fma(z,c,acc); acc = acc + d.  fma emulated exactly in float64
    (products of float32 are exact in float64; one rounding to float32)."""
    p = pts.astype(np.float32)
    T = np.asarray(Tr, dtype=np.float32)
    out = np.empty((p.shape[0], 3), dtype=np.float32)
    x, y, z = (p[:, i].astype(np.float64) for i in range(3))
    for r in range(3):
        a, b, c, d = (np.float64(T[r, k]) for k in range(4))
        acc = (x * a).astype(np.float32)
        acc = (y * b + acc.astype(np.float64)).astype(np.float32)
        acc = (z * c + acc.astype(np.float64)).astype(np.float32)
        out[:, r] = acc + np.float32(T[r, 3])
    return out


def remove_center(ptc, x_range=(-1.15, 1.75), y_range=(-0.65, 0.65)):
    """pre_compute_pp_score.py:48-52."""
    mask = (ptc[:, 0] < x_range[1]) & (ptc[:, 0] >= x_range[0]) & (
        ptc[:, 1] < y_range[1]) & (ptc[:, 1] >= y_range[0])
    return ptc[np.logical_not(mask)]


def count_neighbors(ptc, hist_list, max_neighbor_dist=0.3, workers=1):
    """pre_compute_pp_score.py:54-60 + tree build :188-190 -> (N,T) int64.
    ``workers=1`` is the reference's setting (single thread)."""
    cols = []
    for h in hist_list:
        tree = cKDTree(h)
        cols.append(tree.query_ball_point(ptc[:, :3], r=max_neighbor_dist, return_length=True,
                                          workers=workers))
    return np.stack(cols).T


def count_neighbors_bruteforce(ptc, hist_list, max_neighbor_dist=0.3, chunk=256):
    """Definition the KD-tree implements: float64 sum dx^2+dy^2+dz^2 (in that
    order) <= r*r, inclusive.  O(N*M): small cases only."""
    r2 = np.float64(max_neighbor_dist) * np.float64(max_neighbor_dist)
    p = ptc[:, :3].astype(np.float64)
    out = np.zeros((p.shape[0], len(hist_list)), dtype=np.int64)
    for t, h in enumerate(hist_list):
        h64 = h.astype(np.float64)
        for s in range(0, p.shape[0], chunk):
            d = p[s:s + chunk, None, :] - h64[None, :, :]
            d2 = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]
            d2 = d2 + d[..., 2] * d[..., 2]
            out[s:s + chunk, t] = (d2 <= r2).sum(axis=1)
    return out


def compute_ephe_score(count):
    """pre_compute_pp_score.py:68-75 (ephe_type == 'entropy')."""
    N = count.shape[1]
    P = count / (np.expand_dims(count.sum(axis=1), -1) + 1e-8)
    H = (-P * np.log(P + 1e-8)).sum(axis=1) / np.log(N)
    return H


def pp_score(live_xyz, hist_list, max_neighbor_dist=0.3, workers=1):
    """count_neighbors + compute_ephe_score + the float32 cast of
    pre_compute_pp_score.py:193-196."""
    c = count_neighbors(live_xyz, hist_list, max_neighbor_dist, workers=workers)
    return compute_ephe_score(c).astype(np.float32), c


def load_pose(oxts_txt: str) -> np.ndarray:
    """pre_compute_pp_score.py:92-101: oxts line "x y z roll pitch yaw" -> 4x4 float32 ego pose."""
    from scipy.spatial.transform import Rotation as R
    info = np.array([float(x) for x in oxts_txt.split()])
    trans = np.eye(4)
    trans[:3, 3] = info[:3]
    trans[:3, :3] = R.from_euler("xyz", info[3:]).as_matrix()
    return trans.astype(np.float32)


def stack_history(traversals, track_list, load_frame, poses, l2es, nusc=False):
    """pre_compute_pp_score.py:126-150: the stacked, transformed history of one scan -- ``for seq_id, indices in
    valid_idx[origin][2]: for frame in indices:`` load, (remove_center), relative pose to the FIRST listed frame,
    transform, append.  A frame is stacked AS OFTEN AS ``indices`` names it (split_traintest.py:86-101 produces such
    repeats whenever two distance thresholds select the same pose).  ``load_frame(file id) -> (n,>=3) float32``;
    ``poses[seq][frame]`` float32 ego pose, ``l2es[seq][frame]``.  Returns ([stack per traversal], first_pose, first_l2e)."""
    K = kitti2nu(nusc)
    seq_id, indices = traversals[0]
    first_pose, first_l2e = poses[seq_id][indices[0]], l2es[seq_id][indices[0]]
    stacks = []
    for seq_id, indices in traversals:
        queue = []
        for frame in indices:
            ptc = load_frame(track_list[seq_id][frame])[:, :3]
            if nusc:
                ptc = remove_center(ptc)
            rel = get_relative_pose(first_l2e, first_pose, l2es[seq_id][frame], poses[seq_id][frame], K)
            queue.append(transform_points(ptc, rel))
        stacks.append(np.concatenate(queue).astype(np.float32))
    return stacks, first_pose, first_l2e
