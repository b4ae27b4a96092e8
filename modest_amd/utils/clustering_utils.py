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

_DEFAULT_GRAPH = ("radius_mutual_knn", "l1")


def cluster_points(ptc, pp_score, neighbor_type="radius_mutual_knn", affinity_type="l1", n_neighbors=70,
                   radius=2., eps=0.1, min_samples=10):
    """precompute_affinity_matrix(...) + DBSCAN(metric='precomputed', eps, min_samples).labels_
    in one call -> (n,) int64 numpy labels (-1 = noise)."""
    if (neighbor_type, affinity_type) != _DEFAULT_GRAPH:
        raise NotImplementedError(f"graph {neighbor_type}/{affinity_type}: only the configs/generate_mask.yaml "
                                  "default radius_mutual_knn/l1 is built (SURVEY.md §8f-3)")
    xyz = to_device(ptc)[:, :3].contiguous()
    pp = to_device(pp_score)
    assert xyz.shape[0] == pp.shape[0]
    n = xyz.shape[0]
    if n and n <= n_neighbors:
        # sklearn raises here as well (kneighbors with n_neighbors > n_samples)
        raise ValueError(f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {n_neighbors + 1}, "
                         f"n_samples_fit = {n}, n_samples = {n}")
    labels, _ = ops.cluster_dbscan(xyz, pp, n_neighbors, radius, eps, min_samples)
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


def filter_labels(ptc, pp_score, labels, random_state=None, plane=None, ptc_dev=None, **kwargs):
    """(:119-135) drop clusters failing is_valid_cluster, relabel to 0 = background,
    1..C.  The second ground plane (hard-coded max_hs=-1.5, range ((-70,70),(-50,50)))
    is estimated on the device; cluster statistics are a few hundred points each."""
    labels = labels.copy()
    if plane is None:
        plane = estimate_plane(ptc if ptc_dev is None else ptc_dev, max_hs=-1.5, ptc_range=((-70, 70), (-50, 50)),
                               random_state=random_state)
    ptc = np.asarray(ptc)
    order = np.argsort(labels, kind="stable")          # members of a label in ascending index order
    sl = labels[order]
    n_lab = int(labels.max()) + 1 if labels.size else 0
    starts = np.searchsorted(sl, np.arange(n_lab), side="left")
    ends = np.searchsorted(sl, np.arange(n_lab), side="right")
    for i in range(n_lab):
        members = order[starts[i]:ends[i]]
        if not is_valid_cluster(ptc[members, :3], pp_score[members], plane, **kwargs):
            labels[members] = -1
    uniq = np.unique(labels)
    return np.searchsorted(uniq, labels).astype(labels.dtype)
