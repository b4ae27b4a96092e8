#!/usr/bin/env python
"""nuScenes-configuration end-to-end golden (BASELINE config 5 shape in small: nusc=True ->
remove_center on the history frames, KITTI2NU = rot-z pi/2, plane_estimate.max_hs=-1.3,
image_shape=[900,1600]), produced by the REFERENCE's three mains on a synthetic KITTI tree
-> tests/golden/e2e_tree_nusc.npz.  Build container only."""
import io
import os
import pickle
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_goldens as mg   # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    mg._install_stubs()
    import pre_compute_pp_score as rpp
    import generate_mask as rgm
    import gen_label_files as rgl
    from modest_amd import synth

    tmp = tempfile.mkdtemp(prefix="modest_gold_nusc_")
    root, meta, out = os.path.join(tmp, "data"), os.path.join(tmp, "meta"), os.path.join(tmp, "out")
    paths = synth.write_kitti_tree(root, meta, n_seq=4, n_frames=6, n_pts=5000, nusc=True, world_seed=3,
                                   origins=(1,), hist_frames=5)
    train = os.path.join(root, "training")
    dp = dict(paths, load_precomputed_lidars=None, load_save_precomputed_trans_mat=None,
              pp_score_path=f"{out}/pp", seg_save_dst=f"{out}/seg", bbox_info_save_dst=f"{out}/bbox",
              label_file_save_dst=f"{out}/labels")
    a1 = mg.ad(dict(data_paths=dp, total_part=1, part=0, seed=1024, max_neighbor_dist=0.3,
                    remove_ground_plane=False, limit_traversals=-1, data_root=train, nusc=True,
                    add_random_noise=0, skip_ephe=False, ephe_type="entropy"))
    cfg = dict(mg.MASK_CFG)
    cfg["plane_estimate"] = dict(cfg["plane_estimate"], max_hs=-1.3)
    origin = int(open(paths["idx_list"]).read().split()[0])
    SEED = 5 + origin
    stderr, sys.stderr = sys.stderr, io.StringIO()
    try:
        rpp.main(a1)
        a2 = mg.ad(dict(cfg, data_paths=dp, total_part=1, part=0, data_root=train, calib_path=f"{train}/calib",
                        ptc_path=f"{train}/velodyne"))
        np.random.seed(SEED)
        rgm.main(a2)
        a3 = mg.ad(dict(data_paths=dp, total_part=1, part=0, data_root=train, calib_path=f"{train}/calib",
                        ptc_path=f"{train}/velodyne", image_shape=[900, 1600], fov_only=True,
                        nms=dict(enable=True, threshold=0.1)))
        rgl.main(a3)
    finally:
        sys.stderr = stderr
    track = pickle.load(open(paths["track_path"], "rb"))
    valid = pickle.load(open(paths["idx_info"], "rb"))
    nfiles = sum(len(s) for s in track)
    bins = [np.fromfile(f"{train}/velodyne/{i:06d}.bin", dtype=np.float32).reshape(-1, 4) for i in range(nfiles)]
    pack = dict(
        bins=np.concatenate(bins), bin_offsets=np.cumsum([0] + [len(b) for b in bins]),
        oxts=np.array([open(f"{train}/oxts/{i:06d}.txt").read() for i in range(nfiles)]),
        l2e=np.array([np.load(f"{train}/l2e/{i:06d}.npy") for i in range(nfiles)]),
        calib=np.array([open(f"{train}/calib/{i:06d}.txt").read() for i in range(nfiles)]),
        track=np.array(pickle.dumps(track, protocol=2)), valid=np.array(pickle.dumps(valid, protocol=2)),
        origin=origin, seed=SEED,
        pp=np.load(f"{out}/pp/{origin:06d}.npy"), seg=np.load(f"{out}/seg/{origin:06d}.npy"),
        label_txt=np.array(open(f"{out}/labels/{origin:06d}.txt").read()),
    )
    objs = pickle.load(open(f"{out}/bbox/{origin:06d}.pkl", "rb"))
    pack["objs"] = np.array([[*o.t, o.l, o.w, o.h, o.ry, o.volume] for o in objs]).reshape(-1, 8)
    np.savez_compressed(os.path.join(GOLD, "e2e_tree_nusc.npz"), **pack)
    print("nusc e2e: N", len(pack["pp"]), "clusters", int(pack["seg"].max()), "objs", len(objs),
          "label lines", len(str(pack["label_txt"]).splitlines()), "pp mean %.4f" % float(pack["pp"].mean()))


if __name__ == "__main__":
    main()
