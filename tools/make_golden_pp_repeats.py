#!/usr/bin/env python
"""Golden vectors for REPEATED history frames, produced by the REFERENCE's own pre_compute_pp_score.main and its own
count_neighbors -> tests/golden/pp_repeats.npz.  Build container only.

The reference stacks a history frame as often as a traversal's index list names it (`for frame in indices`,
pre_compute_pp_score.py:132-150), and such lists are ordinary: split_traintest.py:86-101 appends, for every distance
threshold 2, 4, ..., 70 m, the first frame beyond it -- two thresholds select the same frame whenever consecutive poses of
the history track lie more than 2 m apart.  The tree: one live sequence with eight consecutive origins, three history
traversals whose lists repeat a frame once (traversal 1), three times (traversal 2) and name the same frame at the first
and at the last position of the window (traversal 3)."""
import io
import os
import pickle
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_goldens as mg   # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def repeat_lists(o, n_frames):
    """index lists of the three history traversals for origin frame o (every list repeats a frame)"""
    c = lambda k: min(o + k, n_frames - 1)
    return [[c(0), c(1), c(1), c(2), c(3)],          # once
            [c(0), c(2), c(2), c(2), c(4)],          # three times
            [c(0), c(1), c(2), c(3), c(0)]]          # first == last


def main():
    mg._install_stubs()
    import pre_compute_pp_score as rpp
    from scipy.spatial import cKDTree
    from modest_amd import synth
    tmp = tempfile.mkdtemp(prefix="modest_gold_ppr_")
    root, meta = os.path.join(tmp, "data"), os.path.join(tmp, "meta")
    n_frames, origins = 13, tuple(range(1, 9))
    paths = synth.write_kitti_tree(root, meta, n_seq=4, n_frames=n_frames, n_pts=1800, origins=origins, hist_frames=5, world_seed=5)
    train = os.path.join(root, "training")
    track = pickle.load(open(paths["track_path"], "rb"))
    valid = pickle.load(open(paths["idx_info"], "rb"))
    for o in origins:
        seq0, fr0, hist = valid[track[0][o]]
        valid[track[0][o]] = (seq0, fr0, [(s, lst) for (s, _), lst in zip(hist, repeat_lists(o, n_frames))])
    pickle.dump(valid, open(paths["idx_info"], "wb"))
    out, lid = os.path.join(tmp, "out"), os.path.join(tmp, "lid")
    dp = dict(paths, load_precomputed_lidars=lid, load_save_precomputed_trans_mat=None, pp_score_path=f"{out}/pp")
    a = dict(data_paths=dp, total_part=1, part=0, seed=1024, max_neighbor_dist=0.3, remove_ground_plane=False,
             limit_traversals=-1, data_root=train, nusc=False, add_random_noise=0, skip_ephe=False, ephe_type="entropy")
    stderr, sys.stderr = sys.stderr, io.StringIO()
    try:
        rpp.main(mg.ad(a))
    finally:
        sys.stderr = stderr
    origin_ids = [int(x) for x in open(paths["idx_list"]).read().split()]
    nfiles = sum(len(s) for s in track)
    bins = [np.fromfile(f"{train}/velodyne/{i:06d}.bin", dtype=np.float32).reshape(-1, 4) for i in range(nfiles)]
    pack = dict(
        bins=np.concatenate(bins), bin_offsets=np.cumsum([0] + [len(b) for b in bins]),
        oxts=np.array([open(f"{train}/oxts/{i:06d}.txt").read() for i in range(nfiles)]),
        l2e=np.array([np.load(f"{train}/l2e/{i:06d}.npy") for i in range(nfiles)]),
        calib=np.array([open(f"{train}/calib/{i:06d}.txt").read() for i in range(nfiles)]),
        track=np.array(pickle.dumps(track, protocol=2)), valid=np.array(pickle.dumps(valid, protocol=2)),
        origins=np.array(origin_ids))
    args = mg.ad(dict(max_neighbor_dist=0.3))
    for o in origin_ids:
        pack[f"pp_{o}"] = np.load(f"{out}/pp/{o:06d}.npy")
        # the counts behind the score: the reference's own count_neighbors on the clouds its main stacked (the dump of :152-155)
        comb = pickle.load(open(os.path.join(lid, f"{o:06d}.pkl"), "rb"))
        seq0, fr0, hist = valid[o]
        first_seq, first_idx = hist[0]
        poses, l2es = [], []
        trees = {s: cKDTree(comb[s]) for s, _ in hist}
        # the live scan in the common frame, by the reference's own functions
        import utils.pointcloud_utils as rpu
        def pose_of(i):
            info = np.array([float(x) for x in open(f"{train}/oxts/{i:06d}.txt").readline().split()])
            from scipy.spatial.transform import Rotation as R
            t = np.eye(4)
            t[:3, 3] = info[:3]
            t[:3, :3] = R.from_euler("xyz", info[3:]).as_matrix()
            return t.astype(np.float32)
        f_id, l_id = track[first_seq][first_idx[0]], track[seq0][fr0]
        tm = rpp.get_relative_pose(fixed_l2e=np.load(f"{train}/l2e/{f_id:06d}.npy"), fixed_ego=pose_of(f_id),
                                   query_l2e=np.load(f"{train}/l2e/{l_id:06d}.npy"), query_ego=pose_of(l_id),
                                   KITTI2NU=rpp._KITTI2NU_lyft)
        live = rpu.transform_points(rpu.load_velo_scan(f"{train}/velodyne/{l_id:06d}.bin")[:, :3], tm)
        cnt = rpp.count_neighbors(live, trees, args)
        H = rpp.compute_ephe_score(cnt, mg.ad(dict(ephe_type="entropy")))
        assert np.array_equal(H.astype(np.float32), pack[f"pp_{o}"]), "count_neighbors replay differs from the reference's main"
        pack[f"counts_{o}"] = np.asarray(cnt).astype(np.int32)
        pack[f"stack_sizes_{o}"] = np.array([len(comb[s]) for s, _ in hist])
    # a repeated frame is really stacked twice by the reference: the stacked cloud of traversal 1 holds 5 frames' points
    o = origin_ids[0]
    seq0, fr0, hist = valid[o]
    n1 = sum(len(bins[track[hist[0][0]][f]]) for f in hist[0][1])
    assert pack[f"stack_sizes_{o}"][0] == n1
    np.savez_compressed(os.path.join(GOLD, "pp_repeats.npz"), **pack)
    print("pp_repeats:", {k: v.shape for k, v in pack.items() if k.startswith(("pp_", "counts_"))})
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
