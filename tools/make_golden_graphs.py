#!/usr/bin/env python
"""Golden vectors for the non-default graph / affinity branches, produced by the
REFERENCE's own precompute_affinity_matrix + sklearn DBSCAN (generate_mask.py:66-81) on the kept
points of tests/golden/mask_stage.npz -> tests/golden/graph_variants.npz.  Build container only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_goldens as mg   # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
# (neighbor_type, affinity_type, radius, eps, min_samples, n_neighbors)
VARIANTS = [("radius", "l1", 1.0, 0.1, 10, 70), ("radius", "l1", 2.0, 0.05, 10, 70),
            ("radius_mutual_knn", "exp", 2.0, 1.002, 10, 70), ("radius", "exp", 1.0, 1.001, 8, 70),
            ("radius_mutual_knn", "3d_l2_distance", 2.0, 0.6, 5, 70), ("radius", "3d_l2_distance", 1.0, 0.3, 6, 70),
            ("knn", "l1", 2.0, 0.1, 10, 70), ("knn", "l1", 2.0, 0.05, 6, 15), ("sym_knn", "l1", 2.0, 0.1, 10, 70),
            ("sym_knn", "3d_l2_distance", 2.0, 0.7, 8, 20), ("mutual_knn", "l1", 2.0, 0.1, 10, 70),
            ("mutual_knn", "exp", 2.0, 1.003, 6, 30)]


def main():
    mg._install_stubs()
    from sklearn import cluster
    from utils import clustering_utils as rcu
    g = np.load(os.path.join(GOLD, "mask_stage.npz"))
    ptc, pp, final_mask = g["ptc"], g["pp"], g["final_mask"].astype(bool)
    kept, ppk = ptc[final_mask], pp[final_mask]          # (n,4) rows, as generate_mask.py:66-68 passes them
    out = dict(kept=kept, pp=ppk)
    for k, (nt, at, radius, eps, ms, nn) in enumerate(VARIANTS):
        G = rcu.precompute_affinity_matrix(kept, ppk, neighbor_type=nt, affinity_type=at, n_neighbors=nn, radius=radius)
        lab = cluster.DBSCAN(metric="precomputed", eps=eps, min_samples=ms, n_jobs=-1).fit(G).labels_
        out[f"labels{k}"] = lab.astype(np.int64)
        print(nt, at, radius, eps, ms, nn, "-> clusters", int(lab.max()) + 1, "noise", int((lab < 0).sum()), "nnz", G.nnz)
    out["variants"] = np.array([f"{a}|{b}|{c}|{d}|{e}|{f}" for a, b, c, d, e, f in VARIANTS])
    np.savez_compressed(os.path.join(GOLD, "graph_variants.npz"), **out)


if __name__ == "__main__":
    main()
