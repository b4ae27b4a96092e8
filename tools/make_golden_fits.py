#!/usr/bin/env python
"""Golden vectors for the non-default box fits: the REFERENCE's variance_rectangle,
PCA_rectangle and minimum_bounding_rectangle, and get_obj(fit_method=...) on the clusters of
tests/golden/mask_stage.npz -> tests/golden/fit_variants.npz.  Build container only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_goldens as mg   # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    mg._install_stubs()
    from utils import pointcloud_utils as rpc
    g = np.load(os.path.join(GOLD, "mask_stage.npz"))
    off, pts, rect, seg = g["cl_offsets"], g["cl_pts"], g["rect"], g["labels_filtered"]
    out = {}
    for name, fn in (("variance", rpc.variance_rectangle), ("pca", rpc.PCA_rectangle), ("minarea", rpc.minimum_bounding_rectangle)):
        rows = []
        for k in range(len(off) - 1):
            corners, angle, area = fn(pts[off[k]:off[k + 1]])
            rows.append(np.concatenate([np.asarray(corners).reshape(-1), [angle, area]]))
        out[name] = np.array(rows)
        print(name, out[name].shape)
    # full objects through the reference's get_obj on the rect-frame cluster points
    ids = [i for i in np.unique(seg) if i > 0]
    for method in ("variance_to_edge", "PCA", "min_zx_area_fit"):
        objs = [rpc.get_obj(rect[seg == i], rect, fit_method=method) for i in ids]
        out["objs_" + method] = np.array([[*o.t, o.l, o.w, o.h, o.ry, o.volume] for o in objs])
    np.savez_compressed(os.path.join(GOLD, "fit_variants.npz"), **out)
    print("clusters", len(off) - 1, "objects", len(ids))


if __name__ == "__main__":
    main()
