"""Unpack tests/golden/e2e_tree.npz into a KITTI-format tree (inputs only)."""
import os
import pickle

import numpy as np


def unpack_tree(golden_dir, dst, name="e2e_tree.npz"):
    g = np.load(os.path.join(golden_dir, name), allow_pickle=False)
    train = os.path.join(dst, "data", "training")
    meta = os.path.join(dst, "meta")
    for d in ("velodyne", "oxts", "l2e", "calib"):
        os.makedirs(os.path.join(train, d), exist_ok=True)
    os.makedirs(meta, exist_ok=True)
    off = g["bin_offsets"]
    for i in range(len(off) - 1):
        g["bins"][off[i]:off[i + 1]].astype(np.float32).tofile(os.path.join(train, "velodyne", f"{i:06d}.bin"))
        open(os.path.join(train, "oxts", f"{i:06d}.txt"), "w").write(str(g["oxts"][i]))
        np.save(os.path.join(train, "l2e", f"{i:06d}.npy"), g["l2e"][i])
        open(os.path.join(train, "calib", f"{i:06d}.txt"), "w").write(str(g["calib"][i]))
    track = pickle.loads(g["track"].tobytes())
    valid = pickle.loads(g["valid"].tobytes())
    paths = dict(track_path=os.path.join(meta, "track_list.pkl"), idx_info=os.path.join(meta, "valid_idx_info.pkl"),
                 idx_list=os.path.join(meta, "train_idx.txt"))
    pickle.dump(track, open(paths["track_path"], "wb"))
    pickle.dump(valid, open(paths["idx_info"], "wb"))
    origins = [int(g["origin"])] if "origin" in g.files else [int(x) for x in g["origins"]]
    open(paths["idx_list"], "w").write("\n".join(f"{o:06d}" for o in origins))
    return g, train, paths
