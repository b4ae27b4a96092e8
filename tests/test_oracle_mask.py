"""CPU: mask/cluster + label oracles against fixtures generated from the reference."""
import os
import types

import numpy as np
import pytest

from oracle import labels as ol
from oracle import mask as om


@pytest.fixture(scope="module")
def ms(golden_dir):
    return np.load(os.path.join(golden_dir, "mask_stage.npz"))


def test_plane_and_masks(ms):
    rs = np.random.RandomState(int(ms["seed"]))
    plane, reg, cand = om.estimate_plane(ms["ptc"][:, :3], max_hs=-1.5, ptc_range=[[-70, 70], [-20, 20]],
                                         random_state=rs, return_reg=True)
    assert np.array_equal(plane, ms["plane"])          # same RNG stream as np.random.seed(seed)
    assert reg.n_trials_ == len(ms["triplets1"])
    pm = om.above_plane(ms["ptc"][:, :3], plane, offset=0.05, only_range=[[-70, 70], [-20, 20]])
    assert np.array_equal(pm, ms["plane_mask"])
    rs7 = np.random.RandomState(int(ms["seed"]) + 7)
    p2 = om.estimate_plane(ms["ptc"], max_hs=-1.5, ptc_range=((-70, 70), (-50, 50)), random_state=rs7)
    assert np.array_equal(p2, ms["plane2_seed7"])


def test_graph_and_dbscan(ms):
    fm = ms["final_mask"]
    g = om.precompute_affinity_matrix(ms["ptc"][fm], ms["pp"][fm])
    assert np.array_equal(g.indptr, ms["graph_indptr"]) and np.array_equal(g.indices, ms["graph_indices"])
    assert np.array_equal(g.data, ms["graph_data"])
    lab = om.dbscan_labels(g)
    assert np.array_equal(lab, ms["dbscan"])


def test_closed_form_equals_sklearn(ms):
    """The implicit-graph definition implemented in cluster.hip == sklearn's two calls."""
    fm = ms["final_mask"]
    lab, kth = om.dbscan_closed_form(ms["ptc"][fm][:, :3], ms["pp"][fm])
    assert np.array_equal(lab, ms["dbscan"])
    # a second geometry: dense blobs + borders, k and min_samples exercised at small n
    rng = np.random.default_rng(3)
    pts = np.concatenate([rng.normal([0, 0, 0], 0.4, (300, 3)), rng.normal([3, 0, 0], 0.5, (200, 3)),
                          rng.uniform(-6, 6, (150, 3))]).astype(np.float32)
    pp = np.concatenate([rng.uniform(0, 0.12, 300), rng.uniform(0.5, 0.7, 200), rng.uniform(0, 1, 150)]).astype(np.float32)
    for k, ms_ in ((20, 10), (70, 10), (8, 5)):
        g = om.precompute_affinity_matrix(pts, pp, n_neighbors=k, radius=2.0)
        ref = om.dbscan_labels(g, 0.1, ms_)
        lab, _ = om.dbscan_closed_form(pts, pp, k, 2.0, 0.1, ms_)
        assert np.array_equal(lab, ref), (k, ms_)


def test_filter_and_boxes(ms):
    labels = np.zeros(len(ms["ptc"]), dtype=int) - 1
    labels[ms["final_mask"]] = ms["dbscan"]
    # plane #2 of the golden run came from the continuing RNG stream: replay it
    rs = np.random.RandomState(int(ms["seed"]))
    om.estimate_plane(ms["ptc"][:, :3], max_hs=-1.5, ptc_range=[[-70, 70], [-20, 20]], random_state=rs)
    lf = om.filter_labels(ms["ptc"], ms["pp"], labels, random_state=rs, **om.DEFAULT_CFG["filtering"])
    assert np.array_equal(lf, ms["labels_filtered"])
    rect = ms["rect"]
    off = ms["cl_offsets"]
    for c in range(len(off) - 1):
        cp = rect[lf == c + 1]
        assert np.array_equal(cp[:, [0, 2]], ms["cl_pts"][off[c]:off[c + 1]])
        corners, angle, area, idx = om.closeness_rectangle(cp[:, [0, 2]], return_index=True)
        f = ms["fits"][c]
        assert angle == f[0] and area == f[1] and np.array_equal(corners.ravel(), f[2:10])
        o = om.get_obj(cp, rect)
        assert np.array_equal(np.array([*o.t, o.l, o.w, o.h, o.ry, o.volume]), f[10:18])


def test_e2e_scan_matches_reference_files(golden_dir, tmp_path):
    from tests.golden_tree import unpack_tree
    g, train, paths = unpack_tree(golden_dir, str(tmp_path))
    origin = int(g["origin"])
    ptc = np.fromfile(os.path.join(train, "velodyne", f"{origin:06d}.bin"), dtype=np.float32).reshape(-1, 4)
    calib = ol.Calibration(os.path.join(train, "calib", f"{origin:06d}.txt"))
    res = om.generate_mask_scan(ptc, g["pp"], calib, random_state=np.random.RandomState(int(g["seed"])))
    assert np.array_equal(res["labels"], g["seg"])
    got = np.array([[*o.t, o.l, o.w, o.h, o.ry, o.volume] for o in res["objs"]]).reshape(-1, 8)
    assert np.array_equal(got, g["objs"])
    txt, _ = ol.gen_label_scan(res["objs"], calib)
    assert txt == str(g["label_txt"])


def test_iou_oracle(golden_dir):
    g = np.load(os.path.join(golden_dir, "boxes_iou.npz"))
    iou = ol.boxes_iou_bev(g["boxes"], g["boxes"])
    assert np.array_equal(iou, g["iou"])               # C restatement == reference iou3d_cpu.cpp
    ref = ol.boxes_iou_bev_reference(g["boxes"], g["boxes"])
    if ref is not None:
        assert np.array_equal(ref, g["iou"])
        rng = np.random.default_rng(1)
        b = np.c_[rng.uniform(-5, 5, (300, 2)), np.zeros(300), rng.uniform(0.3, 6, (300, 2)), np.ones(300),
                  rng.uniform(-4, 4, 300)].astype(np.float32)
        assert np.array_equal(ol.boxes_iou_bev(b, b[:40]), ol.boxes_iou_bev_reference(b, b[:40]))
    objs = []
    for b, s in zip(g["boxes"], g["scores"]):
        o = types.SimpleNamespace(t=np.array([b[0], 0.0, b[1]], dtype=np.float64), l=float(b[3]), w=float(b[4]),
                                  h=float(b[5]), ry=float(-b[6]), score=float(s))
        objs.append(o)
    ident = {id(o): i for i, o in enumerate(objs)}
    assert [ident[id(o)] for o in ol.objs_nms(objs, False, 0.1)] == list(g["keep_diag"])
    assert [ident[id(o)] for o in ol.objs_nms(objs, True, 0.1)] == list(g["keep_score"])
    # analytic known answers: identical boxes -> 1, disjoint -> 0, half-overlap axis-aligned -> 1/3
    a = np.array([[0, 0, 0, 2, 2, 1, 0], [1, 0, 0, 2, 2, 1, 0], [9, 9, 0, 1, 1, 1, 0.3]], dtype=np.float32)
    m = ol.boxes_iou_bev(a, a)
    assert abs(m[0, 0] - 1) < 1e-5 and m[0, 2] == 0 and abs(m[0, 1] - 1 / 3) < 1e-5
    # nms on score-sorted boxes agrees with the greedy definition
    order = np.argsort(-g["scores"])
    keep = ol.nms(g["boxes"][order], 0.1)
    mask = ol.nms_from_iou(g["iou"][np.ix_(order, order)], 0.1, scores=-np.arange(len(order), dtype=float))
    assert list(keep) == list(np.nonzero(mask)[0])


def test_graph_variants_match_reference_golden(golden_dir):
    """Non-default neighbor_type / affinity_type branches: the oracle's restatement
    of precompute_affinity_matrix + DBSCAN reproduces the labels the reference itself produced."""
    import os
    from oracle import mask as om
    g = np.load(os.path.join(golden_dir, "graph_variants.npz"))
    for k, v in enumerate(g["variants"]):
        nt, at, radius, eps, ms, nn = str(v).split("|")
        G = om.precompute_affinity_matrix(g["kept"], g["pp"], n_neighbors=int(nn), radius=float(radius),
                                          neighbor_type=nt, affinity_type=at)
        lab = om.dbscan_labels(G, eps=float(eps), min_samples=int(ms))
        assert np.array_equal(lab, g[f"labels{k}"]), v


def test_fit_variants_match_reference_golden(golden_dir):
    """variance_to_edge / PCA / min_zx_area_fit: the oracle's restatements reproduce
    the rectangles and objects the reference's own functions produced."""
    import os
    from oracle import mask as om
    g = np.load(os.path.join(golden_dir, "mask_stage.npz"))
    f = np.load(os.path.join(golden_dir, "fit_variants.npz"))
    off, pts, rect, seg = g["cl_offsets"], g["cl_pts"], g["rect"], g["labels_filtered"]
    for name, fn in (("variance", om.variance_rectangle), ("pca", om.PCA_rectangle), ("minarea", om.minimum_bounding_rectangle)):
        for k in range(len(off) - 1):
            corners, angle, area = fn(pts[off[k]:off[k + 1]])
            got = np.concatenate([np.asarray(corners).reshape(-1), [angle, area]])
            assert np.array_equal(got, f[name][k]), (name, k)
    ids = [i for i in np.unique(seg) if i > 0]
    for method in ("variance_to_edge", "PCA", "min_zx_area_fit"):
        got = np.array([[*o.t, o.l, o.w, o.h, o.ry, o.volume] for o in
                        (om.get_obj(rect[seg == i], rect, fit_method=method) for i in ids)])
        assert np.array_equal(got, f["objs_" + method]), method
