"""Self-training label merge (reference generate_cluster_mask/combine_labels.py):
oracle vs the golden vectors produced by the reference itself (CPU), product path vs both (GPU)."""
import os
import tempfile
from types import SimpleNamespace

import numpy as np
import pytest

RUNS = [dict(percentile=50, threshold=0.5, score_filtering=-1, with_score=False, fov_only=True, bbox=True),
        dict(percentile=20, threshold=0.3, score_filtering=0.3, with_score=True, fov_only=True, bbox=True),
        dict(percentile=90, threshold=0.8, score_filtering=-1, with_score=True, fov_only=False, bbox=False)]


def _frame(golden_dir):
    from golden_tree import unpack_tree
    tmp = tempfile.mkdtemp(prefix="modest_combine_")
    g, train, _ = unpack_tree(golden_dir, tmp)
    origin = int(g["origin"])
    c = np.load(os.path.join(golden_dir, "combine.npz"), allow_pickle=False)
    det = dict(location=c["location"], dimensions=c["dimensions"], rotation_y=c["rotation_y"], score=c["score"],
               frame_id=f"{origin:06d}")
    ptc = np.fromfile(f"{train}/velodyne/{origin:06d}.bin", dtype=np.float32).reshape(-1, 4)
    seeds = np.asarray(g["objs"], dtype=np.float64)

    def seed_objs():
        return [SimpleNamespace(t=o[:3].copy(), l=o[3], w=o[4], h=o[5], ry=o[6], volume=o[7]) for o in seeds]

    return c, det, ptc, np.asarray(g["pp"]), f"{train}/calib/{origin:06d}.txt", seed_objs


def test_oracle_combine_matches_reference_golden(golden_dir):
    from oracle import combine as oc
    from oracle import labels as ol
    c, det, ptc, pp, calib_path, seed_objs = _frame(golden_dir)
    calib = ol.Calibration(calib_path)
    rect = calib.project_velo_to_rect(ptc[:, :3])
    dets = oc.predicts2objs(det)
    for k, (pc, th) in enumerate(zip(c["cfg_percentile"], c["cfg_threshold"])):
        got = [oc.filter_by_ppscore(rect, pp, o, percentile=int(pc), threshold=float(th)) for o in dets]
        assert np.array_equal(np.array(got), c["decisions"][k])
    assert np.array_equal(np.array([int(oc.box_mask(rect, o).sum()) for o in dets]), c["inside"])
    for k, r in enumerate(RUNS):
        text, _, _ = oc.combine_scan(ptc, pp, calib, det, seed_objs() if r["bbox"] else [], percentile=r["percentile"],
                                     threshold=r["threshold"], score_filtering=r["score_filtering"],
                                     fov_only=r["fov_only"], with_score=r["with_score"])
        assert text == str(c["label_txt"][k])


@pytest.mark.gpu
def test_combine_labels_matches_reference_golden(gpu, golden_dir):
    import torch
    from modest_amd import combine_labels as cl
    from modest_amd import config, ops
    from modest_amd.utils import kitti_util
    c, det, ptc, pp, calib_path, seed_objs = _frame(golden_dir)
    calib = kitti_util.Calibration(calib_path)
    rect = torch.from_numpy(np.ascontiguousarray(calib.project_velo_to_rect(ptc[:, :3]))).to(gpu)
    ppd = torch.from_numpy(pp).to(gpu)
    dets = cl.predicts2objs(det)
    st = ops.boxes_pp_stats(rect, ppd, np.array([cl._box_scalars(o) for o in dets], dtype=np.float64), 0.5)
    assert np.array_equal(st[:, 0].astype(np.int64), c["inside"])          # point masks: exact
    for k, (pc, th) in enumerate(zip(c["cfg_percentile"], c["cfg_threshold"])):
        got = cl.filter_by_ppscore_batch(rect, ppd, dets, percentile=int(pc), threshold=float(th))
        assert np.array_equal(np.array(got), c["decisions"][k])
    for k, r in enumerate(RUNS):
        args = config.compose("combine_labels", [
            "data_root=/unused", f"det_filtering.pp_score_percentile={r['percentile']}",
            f"det_filtering.pp_score_threshold={r['threshold']}", f"det_filtering.score_filtering={r['score_filtering']}",
            f"fov_only={r['fov_only']}", f"with_score={r['with_score']}"])
        text, _, _ = cl.combine_scan(ptc, pp, calib, det, seed_objs() if r["bbox"] else [], args)
        assert text == str(c["label_txt"][k])


@pytest.mark.gpu
def test_boxes_pp_stats_vs_numpy(gpu):
    """Random boxes over a random cloud: counts and percentiles equal numpy's on the same masks
    (float64 rotation, float32 percentile), incl. empty boxes, one-point boxes and q = 0 / 1."""
    import torch
    from modest_amd import combine_labels as cl
    from modest_amd import ops
    from modest_amd.utils.clustering_utils import percentile_from_order_stats
    from oracle import combine as oc
    rng = np.random.default_rng(7)
    rect = rng.standard_normal((40000, 3)) * [20, 1.0, 20]
    pp = rng.uniform(0, 1, 40000).astype(np.float32)
    objs = []
    for _ in range(200):
        objs.append(SimpleNamespace(t=(rng.standard_normal(3) * [15, 0.5, 15]).astype(np.float32), l=np.float32(rng.uniform(0.05, 8)),
                                    w=np.float32(rng.uniform(0.05, 4)), h=np.float32(rng.uniform(0.1, 3)),
                                    ry=np.float32(rng.uniform(-4, 4))))
    rd, pd = torch.from_numpy(rect).to(gpu), torch.from_numpy(pp).to(gpu)
    b12 = np.array([cl._box_scalars(o) for o in objs], dtype=np.float64)
    for q in (0.0, 0.2, 0.5, 0.9, 1.0):
        q32 = np.true_divide(q * 100, np.float32(100))
        st = ops.boxes_pp_stats(rd, pd, b12, float(q32))
        pct = percentile_from_order_stats(st[:, 1], st[:, 2], st[:, 3])
        for k, o in enumerate(objs):
            m = oc.box_mask(rect, o)
            assert int(st[k, 0]) == int(m.sum())
            if m.sum():
                assert pct[k] == np.percentile(pp[m], q * 100), (k, q)
