#!/usr/bin/env python
"""Golden vectors for the self-training label merge, produced by the REFERENCE's
own generate_cluster_mask/combine_labels.py imported in this container (build container only;
/root/reference does not exist on the GPU box).  Inputs: the frame of tests/golden/e2e_tree.npz
(scan, calibration, PP scores, seed boxes) and a synthetic OpenPCDet-style result list
(float32 arrays, like `result.pkl`).  Outputs: per-box filter_by_ppscore decisions, the points
inside every box, and the label files for three configurations -> tests/golden/combine.npz."""
import io
import os
import pickle
import sys
import tempfile
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import make_goldens as mg   # noqa: E402  (stubs + attr-dict config)
from golden_tree import unpack_tree   # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def make_detections(rng, seeds, n_random=24):
    """Seed boxes perturbed (what a detector trained on them predicts), random boxes over the
    scene, a few degenerate ones (empty, far away); everything float32 as OpenPCDet writes it."""
    loc, dim, ry, score = [], [], [], []
    for o in seeds:
        for _ in range(2):
            loc.append(o[:3] + rng.normal(0, [0.25, 0.05, 0.25]))
            dim.append([o[3] * rng.uniform(0.9, 1.2), o[5] * rng.uniform(0.9, 1.2), o[4] * rng.uniform(0.9, 1.2)])  # l, h, w
            ry.append(o[6] + rng.normal(0, 0.1))
            score.append(rng.uniform(0.2, 0.95))
    for _ in range(n_random):
        loc.append([rng.uniform(-25, 25), rng.uniform(1.0, 1.9), rng.uniform(3, 60)])
        dim.append([rng.uniform(0.5, 6.0), rng.uniform(0.8, 2.5), rng.uniform(0.5, 2.5)])
        ry.append(rng.uniform(-np.pi, np.pi))
        score.append(rng.uniform(0.05, 0.9))
    loc.append([0.0, 1.7, 500.0]); dim.append([4.0, 1.5, 1.8]); ry.append(0.3); score.append(0.99)   # no points inside
    loc.append([3.0, -20.0, 20.0]); dim.append([4.0, 1.5, 1.8]); ry.append(0.0); score.append(0.5)   # above everything
    return dict(location=np.array(loc, dtype=np.float32), dimensions=np.array(dim, dtype=np.float32),
                rotation_y=np.array(ry, dtype=np.float32), score=np.array(score, dtype=np.float32))


def main():
    mg._install_stubs()
    import combine_labels as rcl   # the reference module
    from utils import kitti_util as rku
    from utils import pointcloud_utils as rpc

    tmp = tempfile.mkdtemp(prefix="modest_gold_combine_")
    g, train, paths = unpack_tree(GOLD, tmp)
    origin = int(g["origin"])
    out = os.path.join(tmp, "out")
    os.makedirs(f"{out}/pp", exist_ok=True)
    os.makedirs(f"{out}/bbox", exist_ok=True)
    np.save(f"{out}/pp/{origin:06d}.npy", g["pp"])
    seeds = np.asarray(g["objs"], dtype=np.float64)

    def seed_objs():
        return [SimpleNamespace(t=o[:3].copy(), l=o[3], w=o[4], h=o[5], ry=o[6], volume=o[7]) for o in seeds]

    pickle.dump(seed_objs(), open(f"{out}/bbox/{origin:06d}.pkl", "wb"))
    rng = np.random.default_rng(4242)
    det = make_detections(rng, seeds)
    det["frame_id"] = f"{origin:06d}"
    pickle.dump([det], open(f"{out}/result.pkl", "wb"))

    # per-box decisions and masks straight from the reference function
    calib = rku.Calibration(f"{train}/calib/{origin:06d}.txt")
    ptc = rpc.load_velo_scan(f"{train}/velodyne/{origin:06d}.bin")
    rect = calib.project_velo_to_rect(ptc[:, :3])
    pp = np.load(f"{out}/pp/{origin:06d}.npy")
    dets = rcl.predicts2objs(det)
    cfgs = [dict(percentile=50, threshold=0.5), dict(percentile=20, threshold=0.3), dict(percentile=90, threshold=0.8)]
    decisions = np.array([[rcl.filter_by_ppscore(rect, pp, o, **c) for o in dets] for c in cfgs])
    inside = []
    for o in dets:   # mask count, restated inline from combine_labels.py:42-57 with the reference's own expressions
        xz = rect[:, [0, 2]] - o.t[[0, 2]]
        rot = np.array([[np.cos(o.ry), -np.sin(o.ry)], [np.sin(o.ry), np.cos(o.ry)]])
        xz = xz @ rot.T
        m = (xz[:, 0] > -o.l / 2) & (xz[:, 0] < o.l / 2) & (xz[:, 1] > -o.w / 2) & (xz[:, 1] < o.w / 2)
        m = m * ((rect[:, 1] > o.t[1] - o.h) * (rect[:, 1] <= o.t[1]))
        inside.append(int(m.sum()))

    texts = []
    runs = [dict(det_filtering=dict(pp_score_percentile=50, pp_score_threshold=0.5, score_filtering=-1), with_score=False, fov_only=True, bbox=True),
            dict(det_filtering=dict(pp_score_percentile=20, pp_score_threshold=0.3, score_filtering=0.3), with_score=True, fov_only=True, bbox=True),
            dict(det_filtering=dict(pp_score_percentile=90, pp_score_threshold=0.8, score_filtering=-1), with_score=True, fov_only=False, bbox=False)]
    for k, r in enumerate(runs):
        dp = dict(paths, pp_score_path=f"{out}/pp", bbox_info_save_dst=f"{out}/bbox" if r["bbox"] else None)
        args = mg.ad(dict(data_paths=dp, total_part=1, part=0, data_root=train, calib_path=f"{train}/calib",
                          ptc_path=f"{train}/velodyne", det_result_path=f"{out}/result.pkl",
                          save_path=f"{out}/combined{k}", image_shape=[1024, 1224], fov_only=r["fov_only"],
                          det_filtering=r["det_filtering"], nms=dict(enable=True, threshold=0.1),
                          with_score=r["with_score"]))
        stderr, sys.stderr = sys.stderr, io.StringIO()
        try:
            rcl.main(args)
        finally:
            sys.stderr = stderr
        texts.append(open(f"{out}/combined{k}/{origin:06d}.txt").read())
        print("run", k, "lines", len(texts[-1].splitlines()))

    np.savez_compressed(os.path.join(GOLD, "combine.npz"), location=det["location"], dimensions=det["dimensions"],
                        rotation_y=det["rotation_y"], score=det["score"], decisions=decisions,
                        inside=np.array(inside), cfg_percentile=np.array([c["percentile"] for c in cfgs]),
                        cfg_threshold=np.array([c["threshold"] for c in cfgs]),
                        label_txt=np.array(texts))
    print("boxes", len(dets), "kept per cfg", decisions.sum(1), "inside min/max", min(inside), max(inside))


if __name__ == "__main__":
    main()
