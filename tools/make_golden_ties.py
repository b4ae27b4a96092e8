#!/usr/bin/env python
"""Golden vectors for duplicated points and exact distance ties in the k-NN graph, produced
by the REFERENCE's own precompute_affinity_matrix + sklearn DBSCAN (clustering_utils.py:32-38,
generate_mask.py:75-81) -> tests/golden/ties.npz.  Build container only.

Cases:
  small   small groups of duplicated points (2..6 copies, some at (0,0,0): dropped LiDAR returns) inside a
          realistic cloud; every group is smaller than k, so no tie reaches the k-th neighbour
  origin  150 copies of (0,0,0) (> k = 70): every copy's k nearest neighbours are copies, which ones is
          decided by sklearn's KD-tree traversal order
  lattice points on an exact 0.25 m lattice: many exact ties AT the k-th distance
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_goldens as mg   # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def clouds():
    rng = np.random.default_rng(77)
    base = np.concatenate([rng.normal([4, 1, 0.3], [0.6, 0.5, 0.3], (700, 3)), rng.normal([9, -2, 0.5], [0.8, 0.4, 0.4], (500, 3)),
                           rng.uniform([-5, -8, -0.5], [15, 8, 2.0], (900, 3))]).astype(np.float32)
    pp = np.clip(0.15 + 0.1 * np.sin(base[:, 0]) + rng.normal(0, 0.02, len(base)), 0, 1).astype(np.float32)
    # small duplicate groups
    dup_src = rng.choice(len(base), 60, replace=False)
    reps = rng.integers(1, 6, 60)
    small = np.concatenate([base, np.repeat(base[dup_src], reps, axis=0), np.zeros((5, 3), np.float32)])
    pp_small = np.concatenate([pp, np.repeat(pp[dup_src], reps), np.full(5, 0.2, np.float32)])
    perm = rng.permutation(len(small))
    yield "small", small[perm], pp_small[perm]
    origin = np.concatenate([base, np.zeros((150, 3), np.float32)])
    pp_origin = np.concatenate([pp, np.full(150, 0.2, np.float32)])
    perm = rng.permutation(len(origin))
    yield "origin", origin[perm], pp_origin[perm]
    gx, gy, gz = np.meshgrid(np.arange(24), np.arange(20), np.arange(3), indexing="ij")
    lat = (np.stack([gx, gy, gz], -1).reshape(-1, 3) * 0.25).astype(np.float32)
    pp_lat = (0.1 + 0.05 * ((gx + gy) % 3).reshape(-1)).astype(np.float32)
    perm = rng.permutation(len(lat))
    yield "lattice", lat[perm], pp_lat[perm]


def main():
    mg._install_stubs()
    from sklearn import cluster
    from utils import clustering_utils as rcu
    out = {}
    for name, xyz, pp in clouds():
        ptc = np.concatenate([xyz, np.zeros((len(xyz), 1), np.float32)], axis=1)   # (n,4) rows as generate_mask.py:66-68
        for k, ms in ((70, 10), (12, 5)):
            G = rcu.precompute_affinity_matrix(ptc, pp, neighbor_type="radius_mutual_knn", affinity_type="l1", n_neighbors=k, radius=2.0)
            lab = cluster.DBSCAN(metric="precomputed", eps=0.1, min_samples=ms, n_jobs=-1).fit(G).labels_
            out[f"{name}_labels_k{k}"] = lab.astype(np.int64)
            deg = np.diff(G.indptr)
            print(name, "k", k, "n", len(xyz), "clusters", int(lab.max()) + 1, "noise", int((lab < 0).sum()), "nnz", G.nnz, "max degree", int(deg.max()))
        out[f"{name}_xyz"], out[f"{name}_pp"] = xyz, pp
    np.savez_compressed(os.path.join(GOLD, "ties.npz"), **out)


if __name__ == "__main__":
    main()
