"""Clustering helpers of the seed-label path, backed by libmodest_hip.so.

Host-side mirror of the reference's ``generate_cluster_mask/utils/
clustering_utils.py``.  The reference materialises a sparse affinity matrix
(``precompute_affinity_matrix`` :7-60) and hands it to sklearn's DBSCAN
(``generate_mask.py:75-81``); here both are ONE device call on the implicit
graph (``cluster_points``): no N x N matrix, sparse or otherwise, exists.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import ops
from .pointcloud_utils import distance_to_plane, estimate_plane, to_device


def cluster_points(ptc, pp_score, neighbor_type="radius_mutual_knn", affinity_type="l1", n_neighbors=70,
                   radius=2., eps=0.1, min_samples=10):
    """precompute_affinity_matrix(...) + DBSCAN(metric='precomputed', eps, min_samples).labels_
    in one call -> (n,) int64 numpy labels (-1 = noise)."""
    dev = to_device(ptc)
    xyz = dev[:, :3].contiguous()
    pp = to_device(pp_score)
    assert xyz.shape[0] == pp.shape[0]
    n = xyz.shape[0]
    inten = None
    if affinity_type == "3d_l2_distance":   # the reference takes the norm of the full rows it is given
        inten = dev[:, 3].contiguous() if dev.shape[1] > 3 else torch.zeros((n,), dtype=torch.float32, device=dev.device)
    if neighbor_type != "radius" and n and n <= n_neighbors:
        # sklearn raises here as well (kneighbors with n_neighbors > n_samples)
        raise ValueError(f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {n_neighbors + 1}, "
                         f"n_samples_fit = {n}, n_samples = {n}")
    labels, _ = ops.cluster_dbscan(xyz, pp, n_neighbors, radius, eps, min_samples, neighbor_type=neighbor_type,
                                   affinity_type=affinity_type, intensity=inten)
    return labels.cpu().numpy().astype(np.int64)


def is_valid_cluster(ptc, pp_score, plane, min_points=10, max_volume=40, min_volume=0.5, max_min_height=4,
                     min_max_height=0, percentile=10, min_percentile_pp_score=0.7):
    """(:94-117) host numpy on one cluster (max/min_volume are accepted and unused,
    as in the reference)."""
    if ptc.shape[0] < min_points:
        return False
    distance_to_ground = distance_to_plane(ptc, plane, directional=True)
    if distance_to_ground.min() > max_min_height:
        return False
    if distance_to_ground.max() < min_max_height:
        return False
    if np.percentile(pp_score, percentile) > min_percentile_pp_score:
        return False
    return True


def percentile_from_order_stats(a, b, gamma):
    """numpy.percentile(float32 data, q, method='linear') given the two neighbouring order
    statistics a <= b and the fractional part gamma of the virtual index.  numpy (2.x) keeps
    the data dtype throughout: ``_lerp`` = ``a + (b - a) * t``, replaced by
    ``b - (b - a) * (1 - t)`` where t >= 0.5, all in float32."""
    a32, b32 = np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32)
    t = np.asarray(gamma, dtype=np.float32)
    diff = b32 - a32
    lo = a32 + diff * t
    hi = b32 - diff * (np.float32(1) - t)
    return np.where(t >= np.float32(0.5), hi, lo)


def compact_labels(labels):
    """``np.searchsorted(np.unique(labels), labels)`` (clustering_utils.py:133-135,
    generate_mask.py:100-103) for small integer labels >= -1, without the two O(n log n) sorts:
    a presence table over the label range and its prefix sum give the same ranks."""
    labels = np.asarray(labels)
    if labels.size == 0:
        return labels.copy()
    lo = int(labels.min())
    present = np.zeros(int(labels.max()) - lo + 1, dtype=bool)
    present[labels - lo] = True
    rank = np.cumsum(present) - 1
    return rank[labels - lo].astype(labels.dtype)


def members_sorted(labels, n_lab):
    """The members of labels 1..n_lab as one index array in label order (ascending indices inside a
    label) plus the n_lab + 1 cut positions: one stable sort of the labelled points only (most points
    are background; small labels take numpy's O(n) radix sort)."""
    idx = np.flatnonzero(labels > 0)
    sub = labels[idx]
    if n_lab < 32768:
        sub = sub.astype(np.int16)
    order = idx[np.argsort(sub, kind="stable")]
    cuts = np.searchsorted(labels[order], np.arange(1, n_lab + 2), side="left")
    return order, cuts


def members_by_label(labels, n_lab):
    """[np.flatnonzero(labels == i) for i in 1..n_lab] (ascending indices)."""
    order, cuts = members_sorted(labels, n_lab)
    return [order[cuts[k]:cuts[k + 1]] for k in range(n_lab)]


def relabel_after_drop(labels, n_lab, keep):
    """``labels[members of dropped clusters] = 0`` followed by compact_labels (generate_mask.py:94-103) as
    one table look-up.  labels holds 0 (background) and 1..n_lab, every one of which has members;
    keep (n_lab,) bool."""
    keep = np.asarray(keep, dtype=bool)
    has_zero = (not bool(keep.all())) or bool((labels == 0).any())
    table = np.zeros(n_lab + 1, dtype=labels.dtype)
    table[1:][keep] = np.arange(int(keep.sum()), dtype=labels.dtype) + (1 if has_zero else 0)
    return table[labels]


FILTER_PLANE_SPEC = (-1.5, ((-70, 70), (-50, 50)))   # the reference's hard-coded second ground fit (:126)


def filter_labels(ptc, pp_score, labels, random_state=None, plane=None, ptc_dev=None, pp_dev=None,
                  labels_dev=None, plane_prepared=None, min_points=10, max_volume=40, min_volume=0.5, max_min_height=4,
                  min_max_height=0, percentile=10, min_percentile_pp_score=0.7):
    """(:119-135) drop clusters failing is_valid_cluster, relabel to 0 = background, 1..C.
    The second ground plane (hard-coded max_hs=-1.5, range ((-70,70),(-50,50))) and the
    per-cluster statistics (count, height extremes, PP-score percentile) are device work;
    the host keeps the four scalar comparisons per cluster."""
    dev_pts = to_device(ptc) if ptc_dev is None else ptc_dev
    if plane is None:
        plane = estimate_plane(dev_pts, max_hs=FILTER_PLANE_SPEC[0], ptc_range=FILTER_PLANE_SPEC[1],
                               random_state=random_state, prepared=plane_prepared)
    n_lab = int(labels.max()) + 1 if labels.size else 0
    if n_lab > 0:
        dev_pp = to_device(pp_score) if pp_dev is None else pp_dev
        dev_lab = torch.from_numpy(labels.astype(np.int32)).to(dev_pts.device) if labels_dev is None else labels_dev
        q32 = np.true_divide(percentile, np.float32(100))     # numpy divides by a float32 hundred for float32 data
        st = ops.cluster_stats(dev_pts, dev_pp, dev_lab, n_lab, np.asarray(plane, dtype=np.float64), float(q32))
        n, dmin, dmax = st[:, 0], st[:, 1], st[:, 2]
        pct = percentile_from_order_stats(st[:, 3], st[:, 4], st[:, 5])
        valid = (n >= min_points) & ~(dmin > max_min_height) & ~(dmax < min_max_height) & \
            ~(pct > np.float32(min_percentile_pp_score))
        # labels[dropped] = -1 and compact_labels in ONE table look-up over the scan: table[l + 1] = rank of
        # label l among the surviving values.  Labels without members never occur in `labels`; -1
        # survives iff it was there or a cluster is dropped.
        valid &= n >= 1
        has_neg = bool(labels.min() < 0) or bool((~valid & (n >= 1)).any())
        table = np.zeros(n_lab + 1, dtype=labels.dtype)
        table[1:][valid] = np.arange(int(valid.sum()), dtype=labels.dtype) + (1 if has_neg else 0)
        return table[labels + 1]
    return compact_labels(labels)
