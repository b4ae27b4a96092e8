"""GPU parity of every stage after the PP score: HIP path (through the C ABI)
vs the oracle and the reference-generated fixtures.

Bars: integer / index / mask outputs bit exact; plane coefficients <= 1e-4
relative (BASELINE.json); box parameters <= 1e-9 relative; BEV IoU <= 2e-6
absolute (float32 trig, see csrc/trig_f32.h)."""
import os
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ms(golden_dir):
    return np.load(os.path.join(golden_dir, "mask_stage.npz"))


def test_transform_points(gpu, golden_dir):
    import torch
    from modest_amd import ops
    t = np.load(os.path.join(golden_dir, "transform.npz"))
    pts = torch.from_numpy(t["pts"]).to(gpu)
    out = ops.transform_points(pts, t["T"]).cpu().numpy()
    assert np.array_equal(out, t["out"])                      # bit exact vs the reference's BLAS product
    pts4 = torch.cat([pts, torch.ones((pts.shape[0], 1), device=gpu)], 1).contiguous()
    assert np.array_equal(ops.transform_points(pts4, t["T"]).cpu().numpy(), t["out"])
    from oracle import pp_score as opp
    kept = ops.transform_points(pts4, t["T"], remove_center=True).cpu().numpy()
    assert np.array_equal(kept, opp.transform_points(t["kept"], t["T"]))
    # many blocks, order preserved
    rng = np.random.default_rng(0)
    big = (rng.standard_normal((300_001, 4)) * [2, 1, 1, 1]).astype(np.float32)
    ref = opp.transform_points_fma(opp.remove_center(big[:, :3]), t["T"])
    got = ops.transform_points(torch.from_numpy(big).to(gpu), t["T"], remove_center=True).cpu().numpy()
    assert np.array_equal(got, ref)
    assert ops.transform_points(pts[:0], t["T"]).shape == (0, 3)


def test_plane_candidates_mad_and_scoring(gpu, ms):
    import torch
    from modest_amd import ops
    from oracle import mask as om
    ptc = ms["ptc"]
    dev = torch.from_numpy(ptc).to(gpu)
    rng_ = [[-70, 70], [-20, 20]]
    cand, idx = ops.plane_candidates(dev, -1.5, rng_)
    m = om.plane_candidate_mask(ptc, -1.5, rng_)
    assert np.array_equal(idx.cpu().numpy(), np.nonzero(m)[0])
    assert np.array_equal(cand.cpu().numpy(), ptc[m][:, :3])
    z = ptc[m][:, 2]
    mad = np.median(np.abs(z - np.median(z)))
    assert ops.mad_threshold(cand) == mad                       # exact float32 medians
    for n in (1, 2, 3, 10, 11, 4097):
        sub = cand[:n].contiguous()
        zz = ptc[m][:n, 2]
        assert ops.mad_threshold(sub) == np.median(np.abs(zz - np.median(zz))), n
    # trial scoring vs a float32 numpy statement with the same fma-chain prediction
    rs = np.random.default_rng(1)
    models = np.c_[rs.normal(0, 0.01, 40), rs.normal(0, 0.01, 40), rs.normal(-1.7, 0.02, 40)].astype(np.float32)
    n_in, sse, sy, syy = ops.ransac_score_trials(cand, models, mad)
    X = ptc[m][:, :2].astype(np.float64)
    for k in range(40):
        c0, c1, b = (np.float64(v) for v in models[k])
        acc = (X[:, 0] * c0).astype(np.float32).astype(np.float64)
        pred = (X[:, 1] * c1 + acc).astype(np.float32) + models[k, 2]
        res = np.abs(z - pred)
        inl = res <= mad
        assert n_in[k] == inl.sum()
        assert abs(sse[k] - (res[inl].astype(np.float64) ** 2).sum()) <= 1e-9 * max(1.0, sse[k])
        assert abs(sy[k] - z[inl].astype(np.float64).sum()) <= 1e-9 * max(1.0, abs(sy[k]))


def test_ransac_native_driver_equals_python_loop(gpu, ms):
    """modest_ransac_plane (generator + batches + accept rule + refit in the library) against the Python
    loop of utils/ransac.py on the same candidates: same triplets, same number of trials, same plane, and
    the caller's generator ends in the same state (the next fit of the scan draws from it)."""
    import torch
    from modest_amd import ops
    from modest_amd.utils import ransac
    dev = torch.from_numpy(ms["ptc"]).to(gpu)
    for max_hs, rng_ in ((-1.5, [[-70, 70], [-20, 20]]), (-1.5, [[-70, 70], [-50, 50]]), (-1.2, [[0, 40], [-10, 10]])):
        cand, _ = ops.plane_candidates(dev, max_hs, rng_)
        thr = ops.mad_threshold(cand)
        for seed in range(12):
            a, b = np.random.RandomState(seed), np.random.RandomState(seed)
            nat = ransac.ransac_plane(cand, random_state=a, thr=thr)
            ransac.NATIVE_DRIVER = False   # the Python statement of the same loop
            try:
                ref = ransac.ransac_plane(cand, random_state=b, thr=thr)
            finally:
                ransac.NATIVE_DRIVER = True
            assert np.array_equal(nat.triplets, ref.triplets), seed
            assert nat.n_trials == ref.n_trials and nat.n_inliers == ref.n_inliers
            assert np.array_equal(nat.coef, ref.coef) and nat.intercept == ref.intercept
            assert np.array_equal(nat.best_model, ref.best_model)
            sa, sb = a.get_state(), b.get_state()
            assert np.array_equal(sa[1], sb[1]) and sa[2:] == sb[2:], seed


def test_plane_prepare_equals_separate_calls(gpu, ms):
    """modest_plane_prepare (both candidate sets from one pass + both MADs, one sync) against
    plane_candidates x 2 + mad_threshold, incl. an empty set and repeated calls."""
    import torch
    from modest_amd import ops
    from modest_amd.utils.clustering_utils import FILTER_PLANE_SPEC
    dev = torch.from_numpy(ms["ptc"]).to(gpu)
    for specs in ([(-1.5, [[-70, 70], [-20, 20]]), FILTER_PLANE_SPEC], [(-1.3, [[0, 70], [-40, 40]]), (-100.0, [[-1, 1], [-1, 1]])],
                  [(-1.5, [[-70, 70], [-20, 20]]), FILTER_PLANE_SPEC]):
        got = ops.plane_prepare(dev, specs)
        for (cand, thr), (max_hs, rng_) in zip(got, specs):
            ref, _ = ops.plane_candidates(dev, max_hs, rng_)
            assert torch.equal(cand, ref)
            if ref.shape[0]:
                assert thr == ops.mad_threshold(ref)
            else:
                assert thr is None
    e = ops.plane_prepare(dev[:0], [(-1.5, [[-70, 70], [-20, 20]]), FILTER_PLANE_SPEC])
    assert e[0][0].shape[0] == 0 and e[0][1] is None and e[1][1] is None


def test_ransac_trials_batch_paths_agree(gpu, ms):
    """One round trip per batch of trials: batches of up to 64 triplets travel as a kernel argument and
    are fitted inside the scoring kernel, larger ones through the separate fit kernel; both must give
    the same planes and sums as the explicit-model scoring entry point, trial by trial."""
    import torch
    from modest_amd import ops
    dev = torch.from_numpy(ms["ptc"]).to(gpu)
    cand, _ = ops.plane_candidates(dev, -1.5, [[-70, 70], [-20, 20]])
    n = cand.shape[0]
    trip = np.random.RandomState(2).randint(0, n, size=(150, 3))
    trip[7] = [5, 5, 9]                                # degenerate triplet -> lstsq's minimum-norm model
    thr, models_big, n_big, sse_big, sy_big, syy_big = ops.ransac_trials(cand, trip)          # K = 150
    assert thr == ops.mad_threshold(cand)
    for lo, hi in ((0, 48), (48, 112), (112, 150), (7, 8)):                                     # K <= 64
        t2, m2, n2, sse2, sy2, syy2 = ops.ransac_trials(cand, trip[lo:hi], thr)
        assert t2 == thr and np.array_equal(m2, models_big[lo:hi], equal_nan=True)
        assert np.array_equal(n2, n_big[lo:hi]) and np.array_equal(sse2, sse_big[lo:hi])
        assert np.array_equal(sy2, sy_big[lo:hi]) and np.array_equal(syy2, syy_big[lo:hi])
    ok = ~np.isnan(models_big[:, 0])
    assert ok.all()
    from modest_amd.utils import ransac
    c = cand.cpu().numpy()
    assert np.allclose(models_big[7], ransac.planes_through_triplets(c[trip[7]][None])[0], rtol=1e-6, atol=1e-7)
    n3, sse3, sy3, syy3 = ops.ransac_score_trials(cand, models_big[ok], thr)
    assert np.array_equal(n3, n_big[ok]) and np.array_equal(sse3, sse_big[ok]) and np.array_equal(syy3, syy_big[ok])


def test_ransac_plane_vs_sklearn(gpu, ms):
    import torch
    from modest_amd.utils import pointcloud_utils as pcu
    from oracle import mask as om
    ptc = ms["ptc"]
    for seed, kw in ((int(ms["seed"]), dict(max_hs=-1.5, ptc_range=[[-70, 70], [-20, 20]])),
                     (int(ms["seed"]) + 7, dict(max_hs=-1.5, ptc_range=((-70, 70), (-50, 50))))):
        ref = om.estimate_plane(ptc, random_state=np.random.RandomState(seed), **kw)
        got, info = pcu.estimate_plane(torch.from_numpy(ptc).to(gpu), random_state=np.random.RandomState(seed),
                                       return_info=True, **kw)
        assert np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3)) <= 1e-4, (got, ref)
        assert got[2] > 0
    # the triplet stream is the one sklearn consumed for the reference's run
    got, info = pcu.estimate_plane(torch.from_numpy(ptc).to(gpu), max_hs=-1.5, ptc_range=[[-70, 70], [-20, 20]],
                                   random_state=np.random.RandomState(int(ms["seed"])), return_info=True)
    nt = min(len(info.triplets), len(ms["triplets1"]))
    assert np.array_equal(info.triplets[:nt], ms["triplets1"][:nt])
    assert abs(info.n_trials - len(ms["triplets1"])) <= 2
    assert np.max(np.abs(got - ms["plane"]) / np.maximum(np.abs(ms["plane"]), 1e-3)) <= 1e-4


def test_plane_range_mask_exact(gpu, ms):
    import torch
    from modest_amd import ops
    dev = torch.from_numpy(ms["ptc"]).to(gpu)
    mask, kept, idx = ops.plane_range_mask(dev, ms["plane"], 0.05, [[-70, 70], [-20, 20]], [[-70, 70], [-40, 40]])
    assert np.array_equal(mask.cpu().numpy(), ms["final_mask"])
    assert np.array_equal(idx.cpu().numpy(), np.nonzero(ms["final_mask"])[0])
    assert np.array_equal(kept.cpu().numpy(), ms["ptc"][ms["final_mask"]][:, :3])
    from modest_amd.utils import pointcloud_utils as pcu
    pm = pcu.above_plane(ms["ptc"][:, :3].copy(), ms["plane"], offset=0.05, only_range=[[-70, 70], [-20, 20]])
    assert np.array_equal(pm, ms["plane_mask"])


def test_compaction_many_blocks_and_reuse(gpu):
    """Order-preserving compaction across hundreds of blocks (the look-back sums predecessors 64 at
    a time) and back-to-back launches on one context (the state words reset themselves): candidate
    selection and the plane/range mask against numpy boolean indexing, ragged sizes included."""
    import torch
    from modest_amd import ops
    rng = np.random.RandomState(5)
    plane = np.array([0.01, -0.02, 1.0, 1.6])
    for n in (700_001, 1, 1023, 1025, 131_072, 70_000):
        pts = np.empty((n, 4), dtype=np.float32)
        pts[:, 0] = rng.uniform(-80, 80, n)
        pts[:, 1] = rng.uniform(-50, 50, n)
        pts[:, 2] = rng.uniform(-3, 1, n)
        pts[:, 3] = rng.rand(n)
        dev = torch.from_numpy(pts).to(gpu)
        for _ in range(2):
            cand, idx = ops.plane_candidates(dev, -1.5, ((-20, 70), (-20, 20)))
            m = (pts[:, 2] < -1.5) & (pts[:, 0] > -20) & (pts[:, 0] < 70) & (pts[:, 1] > -20) & (pts[:, 1] < 20)
            assert np.array_equal(idx.cpu().numpy(), np.nonzero(m)[0])
            assert np.array_equal(cand.cpu().numpy(), pts[m][:, :3])
            mask, kept, kidx = ops.plane_range_mask(dev, plane, 0.05, [[-70, 70], [-20, 20]], [[-70, 70], [-40, 40]])
            mk = mask.cpu().numpy()
            assert np.array_equal(kidx.cpu().numpy(), np.nonzero(mk)[0])
            assert np.array_equal(kept.cpu().numpy(), pts[mk][:, :3])
            d = (pts[:, :3].astype(np.float64) @ plane[:3] + plane[3]) / np.sqrt((plane[:3] ** 2).sum())
            below = (d < 0.05) & (pts[:, 0] < 70) & (pts[:, 0] > -70) & (pts[:, 1] < 20) & (pts[:, 1] > -20)
            rng_ok = (pts[:, 0] <= 70) & (pts[:, 0] > -70) & (pts[:, 1] <= 40) & (pts[:, 1] > -40)
            ref = ~below & rng_ok
            # the float64 dot product may differ from numpy's in the last bit: only points off the threshold
            sure = np.abs(d - 0.05) > 1e-9
            assert np.array_equal(mk[sure], ref[sure])


def test_mad_threshold_exact(gpu):
    """MAD(z) = numpy.median(|z - median(z)|) in float32, bit for bit: odd / even counts, heavy
    duplicates, a single value, more candidates than the register-resident path holds."""
    import torch
    from modest_amd import ops
    rng = np.random.RandomState(11)
    cases = [rng.normal(-1.7, 0.05, 14935), rng.normal(-1.7, 0.05, 14936), np.full(5000, -1.625),
             np.round(rng.normal(-1.7, 0.05, 20001), 2), rng.uniform(-3, 3, 3), rng.normal(0, 1, 40_001),
             rng.normal(0, 1e-3, 32_768), np.array([0.5]),
             # the key range drives the bins: both signs, huge spread, denormals, two distinct values
             np.concatenate([rng.normal(0, 1, 9000), [1e30, -1e30, 1e-40, -1e-40, 0.0, -0.0]]),
             np.concatenate([np.full(3000, 2.0), np.full(3001, 2.0000002)]), rng.normal(0, 1, 2) * 1e-38,
             np.concatenate([np.full(700, -1.5), rng.normal(-1.6, 0.3, 23_000)])]
    for z in cases:
        z = z.astype(np.float32)
        cand = np.zeros((z.size, 3), dtype=np.float32)
        cand[:, 2] = z
        got = np.float32(ops.mad_threshold(torch.from_numpy(cand).to(gpu)))
        ref = np.median(np.abs(z - np.median(z)))
        assert got == ref, (z.size, got, ref)


def test_cluster_dbscan_exact(gpu, ms):
    import torch
    from modest_amd import ops
    from oracle import mask as om
    fm = ms["final_mask"]
    xyz = torch.from_numpy(np.ascontiguousarray(ms["ptc"][fm][:, :3])).to(gpu)
    pp = torch.from_numpy(np.ascontiguousarray(ms["pp"][fm])).to(gpu)
    labels, ncl, kth = ops.cluster_dbscan(xyz, pp, 70, 2.0, 0.1, 10, return_kth=True)
    assert np.array_equal(labels.cpu().numpy().astype(np.int64), ms["dbscan"])
    assert ncl == ms["dbscan"].max() + 1
    _, kref = om.dbscan_closed_form(ms["ptc"][fm][:, :3], ms["pp"][fm])
    assert np.array_equal(kth.cpu().numpy(), kref)
    # second geometry incl. border points between clusters, small k / min_samples
    rng = np.random.default_rng(3)
    pts = np.concatenate([rng.normal([0, 0, 0], 0.4, (300, 3)), rng.normal([3, 0, 0], 0.5, (200, 3)),
                          rng.uniform(-6, 6, (150, 3))]).astype(np.float32)
    ppv = np.concatenate([rng.uniform(0, 0.12, 300), rng.uniform(0.5, 0.7, 200), rng.uniform(0, 1, 150)]).astype(np.float32)
    for k, msmp in ((20, 10), (70, 10), (8, 5)):
        g = om.precompute_affinity_matrix(pts, ppv, n_neighbors=k, radius=2.0)
        ref = om.dbscan_labels(g, 0.1, msmp)
        lab, _ = ops.cluster_dbscan(torch.from_numpy(pts).to(gpu), torch.from_numpy(ppv).to(gpu), k, 2.0, 0.1, msmp)
        assert np.array_equal(lab.cpu().numpy().astype(np.int64), ref), (k, msmp)
    e, n0 = ops.cluster_dbscan(xyz[:0], pp[:0])
    assert e.shape == (0,) and n0 == 0


def test_cluster_dbscan_vs_sklearn_full_size(gpu):
    """Lyft-shape scan: ~15 k kept points, sklearn's two graph calls + DBSCAN vs the implicit-graph kernel."""
    import torch
    from modest_amd import ops, synth
    from oracle import mask as om
    s = synth.make_scan(21, n_live=30000, n_trav=2, n_frames=1)
    ptc = s.live_raw
    keep = (ptc[:, 2] > -1.55) & (np.abs(ptc[:, 1]) < 40) & (np.abs(ptc[:, 0]) < 70)
    xyz = np.ascontiguousarray(ptc[keep][:, :3])
    rng = np.random.default_rng(2)
    ppv = np.clip(0.5 + 0.5 * np.sin(xyz[:, 0] * 0.3) + rng.normal(0, 0.03, len(xyz)), 0, 1).astype(np.float32)
    g = om.precompute_affinity_matrix(xyz, ppv)
    ref = om.dbscan_labels(g)
    lab, ncl = ops.cluster_dbscan(torch.from_numpy(xyz).to(gpu), torch.from_numpy(ppv).to(gpu))
    assert np.array_equal(lab.cpu().numpy().astype(np.int64), ref)
    assert ncl == ref.max() + 1 and ncl > 3


def test_mask_cluster_fused_equals_separate_calls(gpu):
    """modest_mask_cluster (mask + grid count + DBSCAN + label scatter in one call) against
    plane_range_mask -> cluster_dbscan -> labels[mask] = ..., every graph / weight variant, repeated calls
    on one context (the persistent cell counters must come back zeroed), ranges that cut the grid."""
    import torch
    from modest_amd import ops, synth
    s = synth.make_scan(23, n_live=30000, n_trav=2, n_frames=1)
    rng = np.random.default_rng(5)
    for rep, (n, lim, onl) in enumerate(((30000, [[-70, 70], [-40, 40]], [[-20, 70], [-20, 20]]),
                                         (9000, [[0, 70], [-40, 40]], None),
                                         (30000, [[-1000, 1000], [-1000, 1000]], [[-30, 30], [-30, 30]]),
                                         (4000, [[-8, 8], [-8, 8]], None))):
        ptc = np.ascontiguousarray(s.live_raw[rng.permutation(len(s.live_raw))[:n]])
        ppv = np.clip(0.5 + 0.5 * np.sin(ptc[:, 0] * 0.3) + rng.normal(0, 0.03, n), 0, 1).astype(np.float32)
        plane = np.array([0.01, -0.02, 1.0, 1.6])
        d, p = torch.from_numpy(ptc).to(gpu), torch.from_numpy(ppv).to(gpu)
        for nt, at, k, radius, eps in (("radius_mutual_knn", "l1", 70, 2.0, 0.1), ("radius", "3d_l2_distance", 70, 1.0, 0.45),
                                       ("knn", "l1", 25, 1.0, 0.03), ("radius_mutual_knn", "exp", 70, 2.0, 1.0005)):
            _, kept, idx = ops.plane_range_mask(d, plane, 0.05, onl, lim)
            ref = torch.full((n,), -1, dtype=torch.int32, device=gpu)
            if kept.shape[0]:
                inten = d[idx.long(), 3].contiguous() if at == "3d_l2_distance" else None
                lab, _ = ops.cluster_dbscan(kept, p[idx.long()].contiguous(), k, radius, eps, 10, neighbor_type=nt,
                                            affinity_type=at, intensity=inten)
                ref[idx.long()] = lab
            got, n_kept = ops.mask_cluster(d, p, plane, 0.05, onl, lim, k, radius, eps, 10, neighbor_type=nt,
                                           affinity_type=at)
            assert n_kept == kept.shape[0]
            assert torch.equal(got, ref), (rep, nt, at, int((got != ref).sum()))
            assert int(ref.max()) >= 1 or n < 5000
    # fewer kept rows than neighbours: the error sklearn raises, and the context stays usable
    few = torch.from_numpy(np.ascontiguousarray(s.live_raw[:50])).to(gpu)
    with pytest.raises(ValueError):
        ops.mask_cluster(few, torch.zeros(50, device=gpu), np.array([0, 0, 1.0, 100.0]), 0.05, None, [[-70, 70], [-40, 40]])
    lab, nk = ops.mask_cluster(few, torch.zeros(50, device=gpu), np.array([0, 0, 1.0, -100.0]), 0.05, None,
                               [[-70, 70], [-40, 40]])
    assert nk == 0 and bool((lab == -1).all())
    got2, _ = ops.mask_cluster(d, p, plane, 0.05, onl, lim)
    _, kept, idx = ops.plane_range_mask(d, plane, 0.05, onl, lim)
    lab2, _ = ops.cluster_dbscan(kept, p[idx.long()].contiguous())
    assert torch.equal(got2[idx.long()], lab2)


def test_project_velo_to_rect_bit_exact(gpu, ms, golden_dir):
    """The device copy of the rect-frame scan equals Calibration.project_velo_to_rect (numpy: two dgemm
    calls) bit for bit -- on THIS host's BLAS, which is the one the host half of get_obj uses."""
    import tempfile
    import torch
    from modest_amd import ops, synth
    from modest_amd.utils import kitti_util
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "c.txt"), "w").write(synth.CALIB_TXT)
        calib = kitti_util.Calibration(os.path.join(d, "c.txt"))
    rng = np.random.default_rng(4)
    for ptc in (ms["ptc"], (rng.normal(0, 40, (50_001, 4))).astype(np.float32), ms["ptc"][:1], ms["ptc"][:0]):
        got = ops.project_velo_to_rect(torch.from_numpy(np.ascontiguousarray(ptc)).to(gpu), calib.V2C, calib.R0)
        ref = calib.project_velo_to_rect(ptc[:, :3])
        assert np.array_equal(got.cpu().numpy(), ref)
    # a calibration with a full rotation in both matrices
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    calib.V2C = np.hstack([q, rng.normal(0, 1, (3, 1))])
    calib.R0 = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    ptc = (rng.normal(0, 40, (20_000, 4))).astype(np.float32)
    got = ops.project_velo_to_rect(torch.from_numpy(ptc).to(gpu), calib.V2C, calib.R0)
    assert np.array_equal(got.cpu().numpy(), calib.project_velo_to_rect(ptc[:, :3]))


def test_filter_and_boxes(gpu, ms):
    from modest_amd.utils import clustering_utils as cu
    from modest_amd.utils import pointcloud_utils as pcu
    from oracle import mask as om
    labels = np.zeros(len(ms["ptc"]), dtype=int) - 1
    labels[ms["final_mask"]] = ms["dbscan"]
    rs = np.random.RandomState(int(ms["seed"]))
    om.estimate_plane(ms["ptc"][:, :3], max_hs=-1.5, ptc_range=[[-70, 70], [-20, 20]], random_state=rs)
    plane2 = om.estimate_plane(ms["ptc"], max_hs=-1.5, ptc_range=((-70, 70), (-50, 50)), random_state=rs)
    lf = cu.filter_labels(ms["ptc"], ms["pp"], labels, plane=plane2, **om.DEFAULT_CFG["filtering"])
    assert np.array_equal(lf, ms["labels_filtered"])
    rect, off = ms["rect"], ms["cl_offsets"]
    clusters = [rect[lf == c + 1] for c in range(len(off) - 1)]
    fits = pcu.closeness_rectangles([c[:, [0, 2]] for c in clusters])
    for (corners, angle, area), f in zip(fits, ms["fits"]):
        assert angle == f[0] and area == f[1] and np.array_equal(corners.ravel(), f[2:10])
    objs = pcu.get_objs(clusters, rect, fit_method="closeness_to_edge")
    got = np.array([[*o.t, o.l, o.w, o.h, o.ry, o.volume] for o in objs])
    assert np.array_equal(got, ms["fits"][:, 10:18])
    # beta table vs the oracle's numpy loop on one cluster (pairwise summation order)
    import torch
    from modest_amd import ops
    ang, cs = pcu.angle_table(0.1)
    c0 = clusters[0][:, [0, 2]]
    best, beta = ops.fit_boxes_closeness(torch.from_numpy(np.ascontiguousarray(c0)).to(gpu), [0, len(c0)], cs,
                                         return_beta=True)
    ref = []
    for a in ang:
        comp = np.array([[np.cos(a), np.sin(a)], [-np.sin(a), np.cos(a)]])
        pr = c0 @ comp.T
        dx = np.minimum(pr[:, 0] - pr[:, 0].min(), pr[:, 0].max() - pr[:, 0])
        dy = np.minimum(pr[:, 1] - pr[:, 1].min(), pr[:, 1].max() - pr[:, 1])
        ref.append((1 / np.maximum(np.minimum(dx, dy), 1e-2)).sum())
    assert np.array_equal(beta[0], np.array(ref))


def test_rectangle_extents_from_device_equal_host_numpy(gpu):
    """closeness_rectangles takes the extents of every cluster along the chosen heading (and heading + pi/2)
    from the kernel that picked the heading; the rectangles must equal rectangle_at_angle (numpy: dgemm
    projection + column min / max) bit for bit, incl. elongated clusters that take the rotated branch."""
    from modest_amd import ops
    from modest_amd.utils import pointcloud_utils as pu
    rng = np.random.default_rng(9)
    clusters = []
    for k in range(60):
        n = int(rng.choice([10, 11, 64, 65, 129, 500, 2049, 5000]))
        th = rng.uniform(0, np.pi)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        ext = rng.uniform(0.2, 6.0, 2) * (rng.random(2) < 0.9)          # some degenerate (zero-width) clusters
        clusters.append((rng.uniform(-0.5, 0.5, (n, 2)) * ext) @ R.T + rng.uniform(-60, 60, 2))
    got = pu.closeness_rectangles(clusters)
    ang, cs = pu._angles(0.1)
    off = np.cumsum([0] + [len(c) for c in clusters]).astype(np.int32)
    best = ops.fit_boxes_closeness_host(np.concatenate(clusters), off, cs, 1e-2)
    rotated = 0
    for c, b, (corners, angle, area) in zip(clusters, best, got):
        rc, ra, rarea = pu.rectangle_at_angle(c, ang[b])
        assert np.array_equal(corners, rc) and angle == ra and area == rarea
        rotated += ra != ang[b]
    assert 5 < rotated < 55


def test_closeness_criterion_tree_sizes(gpu):
    """The closeness criterion of every (cluster, angle) equals numpy's pairwise sum bit for bit at
    the sizes where numpy's summation tree changes shape (8-accumulator leaves, 128-element leaves,
    halves split at multiples of 8), with 16 and with 64 lanes per angle, and on the one-lane-per-
    angle kernel that clusters too large for the LDS leaf table fall back to."""
    import torch
    from modest_amd import ops
    from modest_amd.utils import pointcloud_utils as pcu
    rng = np.random.RandomState(3)
    ang, cs = pcu.angle_table(0.1)
    sub = np.arange(0, len(ang), 60)          # 16 of the 901 angles keep the numpy reference loop short
    cs_sub = np.ascontiguousarray(cs[sub])

    def reference(c):
        out = []
        for a in ang[sub]:
            comp = np.array([[np.cos(a), np.sin(a)], [-np.sin(a), np.cos(a)]])
            pr = c @ comp.T
            dx = np.minimum(pr[:, 0] - pr[:, 0].min(), pr[:, 0].max() - pr[:, 0])
            dy = np.minimum(pr[:, 1] - pr[:, 1].min(), pr[:, 1].max() - pr[:, 1])
            out.append((1 / np.maximum(np.minimum(dx, dy), 1e-2)).sum())
        return np.array(out)

    def check(sizes):
        cl = [rng.normal(0, 1, (n, 2)) * [2.0, 0.8] + rng.uniform(-30, 30, 2) for n in sizes]
        off = np.cumsum([0] + list(sizes))
        pts = torch.from_numpy(np.ascontiguousarray(np.concatenate(cl))).to(gpu)
        best, beta = ops.fit_boxes_closeness(pts, off, cs_sub, return_beta=True)
        for k, c in enumerate(cl):
            ref = reference(c)
            assert np.array_equal(beta[k], ref), (sizes[k], np.abs(beta[k] - ref).max())
            assert best[k] == int(np.argmax(ref))

    check([1, 2, 7, 8, 9, 15, 16, 17, 127, 128, 129, 130, 136, 255, 256, 257, 263, 264, 1000, 4096])   # 16 lanes
    check([5000, 3, 20_011, 129, 8192, 8193])   # 64 lanes; beyond 8192 elements numpy reduces buffer by buffer
    check([100_003, 40])                         # fallback

    # variance_to_edge criterion: np.var of data-dependent subsets, the same summation rules
    sizes = [1, 9, 300, 20_011]
    cl = [rng.normal(0, 1, (n, 2)) * [2.0, 0.8] + rng.uniform(-30, 30, 2) for n in sizes]
    pts = torch.from_numpy(np.ascontiguousarray(np.concatenate(cl))).to(gpu)
    best, crit = ops.fit_boxes_variance(pts, np.cumsum([0] + sizes), cs_sub, return_crit=True)
    for k, c in enumerate(cl):
        ref = []
        for a in ang[sub]:
            comp = np.array([[np.cos(a), np.sin(a)], [-np.sin(a), np.cos(a)]])
            pr = c @ comp.T
            dx = np.vstack((pr[:, 0] - pr[:, 0].min(), pr[:, 0].max() - pr[:, 0])).min(axis=0)
            dy = np.vstack((pr[:, 1] - pr[:, 1].min(), pr[:, 1].max() - pr[:, 1])).min(axis=0)
            v = 0
            if (dx < dy).sum() > 0:
                v += -np.var(dx[dx < dy])
            if (dy < dx).sum() > 0:
                v += -np.var(dy[dy < dx])
            ref.append(v)
        assert np.array_equal(crit[k], np.array(ref, dtype=np.float64)), (sizes[k], np.abs(crit[k] - ref).max())


def test_bev_iou_and_nms(gpu, golden_dir):
    import torch
    from modest_amd import ops
    from modest_amd.utils import pointcloud_utils as pcu
    from modest_amd.utils.iou3d_nms import iou3d_nms_utils as iu
    from oracle import labels as ol
    g = np.load(os.path.join(golden_dir, "boxes_iou.npz"))
    b = torch.from_numpy(g["boxes"]).to(gpu)
    iou = iu.boxes_iou_bev(b, b).cpu().numpy()
    assert np.max(np.abs(iou - g["iou"])) <= 2e-6
    ov = ops.boxes_iou_bev(b, b, overlap_only=True).cpu().numpy()
    assert np.max(np.abs(ov - ol.boxes_iou_bev(g["boxes"], g["boxes"], overlap_only=True))) <= 2e-5
    assert np.max(np.abs(iu.boxes_bev_iou_cpu(g["boxes"], g["boxes"][:5]) - g["iou"][:, :5])) <= 2e-6
    objs = [types.SimpleNamespace(t=np.array([x[0], 0.0, x[1]], dtype=np.float64), l=float(x[3]), w=float(x[4]),
                                  h=float(x[5]), ry=float(-x[6]), score=float(s)) for x, s in zip(g["boxes"], g["scores"])]
    ident = {id(o): i for i, o in enumerate(objs)}
    assert [ident[id(o)] for o in pcu.objs_nms(objs, True, 0.1)] == list(g["keep_score"])
    assert [ident[id(o)] for o in pcu.objs_nms(objs, False, 0.1)] == list(g["keep_diag"])
    # sorted-box NMS entry points against the oracle (rotated and axis-aligned), > 64 boxes
    rng = np.random.default_rng(4)
    big = np.c_[rng.uniform(-20, 20, (300, 2)), np.zeros(300), rng.uniform(1, 5, (300, 2)), np.ones(300),
                rng.uniform(-3.2, 3.2, 300)].astype(np.float32)
    scores = torch.from_numpy(rng.uniform(size=300).astype(np.float32)).to(gpu)
    order = np.argsort(-scores.cpu().numpy(), kind="stable")
    for rotated, fn in ((True, iu.nms_gpu), (False, iu.nms_normal_gpu)):
        keep, _ = fn(torch.from_numpy(big).to(gpu), scores, 0.1)
        ref = order[ol.nms(big[order], 0.1, rotated=rotated)]
        # the contract is score ORDER, not the set: `order[keep]` of iou3d_nms_utils.py:97-105, where
        # src/iou3d_nms.cpp:116-135 lists the kept positions of the score-sorted boxes in ascending order
        assert np.array_equal(keep.cpu().numpy(), ref)
    iou3d = iu.boxes_iou3d_gpu(b, b).cpu().numpy()
    assert np.all(np.abs(np.diag(iou3d) - 1) < 1e-4)


@pytest.mark.gpu
def test_graph_variants_vs_reference_golden(gpu, golden_dir):
    """radius / radius_mutual_knn graphs with l1, exp and 3d_l2_distance weights:
    labels bit-exact vs the reference's precompute_affinity_matrix + sklearn DBSCAN."""
    import os
    import torch
    from modest_amd import ops
    from modest_amd.utils import clustering_utils as cu
    g = np.load(os.path.join(golden_dir, "graph_variants.npz"))
    kept, pp = g["kept"], g["pp"]
    for k, v in enumerate(g["variants"]):
        nt, at, radius, eps, ms, nn = str(v).split("|")
        lab = cu.cluster_points(kept, pp, neighbor_type=nt, affinity_type=at, n_neighbors=int(nn), radius=float(radius),
                                eps=float(eps), min_samples=int(ms))
        assert np.array_equal(lab, g[f"labels{k}"]), v
    with pytest.raises(NotImplementedError):
        cu.cluster_points(kept, pp, neighbor_type="no_such_graph")
    # larger random cloud against the oracle (sklearn): radius graph, l2 weights incl. intensity
    from oracle import mask as om
    rng = np.random.default_rng(3)
    pts = np.concatenate([rng.standard_normal((6000, 3)) * [8, 8, 0.5], rng.uniform(0, 1, (6000, 1))], axis=1).astype(np.float32)
    ppr = rng.uniform(0, 1, 6000).astype(np.float32)
    for nt, at, eps, nn in (("radius", "3d_l2_distance", 0.45, 70), ("radius", "l1", 0.02, 70),
                            ("radius_mutual_knn", "exp", 1.0005, 70), ("knn", "l1", 0.03, 25), ("knn", "l1", 0.2, 12),
                            ("sym_knn", "l1", 0.03, 25), ("mutual_knn", "l1", 0.05, 40),
                            ("knn", "3d_l2_distance", 0.5, 16)):
        G = om.precompute_affinity_matrix(pts, ppr, n_neighbors=nn, radius=1.0, neighbor_type=nt, affinity_type=at)
        ref = om.dbscan_labels(G, eps=eps, min_samples=6)
        lab = cu.cluster_points(pts, ppr, neighbor_type=nt, affinity_type=at, n_neighbors=nn, radius=1.0, eps=eps,
                                min_samples=6)
        assert np.array_equal(lab, ref), (nt, at, eps, nn, int((lab != ref).sum()), int(ref.max()))


@pytest.mark.gpu
def test_fit_variants_vs_reference_golden(gpu, golden_dir):
    """fit_method variance_to_edge: chosen angle, rectangle and object bit exact vs the reference;
    PCA: within 1e-9 relative (sklearn's SVD and the closed form differ in the last bits)."""
    import os
    import torch
    from modest_amd.utils import pointcloud_utils as pu
    g = np.load(os.path.join(golden_dir, "mask_stage.npz"))
    f = np.load(os.path.join(golden_dir, "fit_variants.npz"))
    off, pts, rect, seg = g["cl_offsets"], g["cl_pts"], g["rect"], g["labels_filtered"]
    clusters = [pts[off[k]:off[k + 1]] for k in range(len(off) - 1)]
    got = pu.variance_rectangles(clusters)
    for k, (corners, angle, area) in enumerate(got):
        assert np.array_equal(np.concatenate([corners.reshape(-1), [angle, area]]), f["variance"][k]), k
    got = pu.pca_rectangles(clusters)
    for k, (corners, angle, area) in enumerate(got):
        np.testing.assert_allclose(np.concatenate([corners.reshape(-1), [angle, area]]), f["pca"][k], rtol=1e-9, atol=1e-9)
    ids = [i for i in np.unique(seg) if i > 0]
    members = [rect[seg == i] for i in ids]
    o = pu.get_objs(members, rect, fit_method="variance_to_edge")
    assert np.array_equal(np.array([[*x.t, x.l, x.w, x.h, x.ry, x.volume] for x in o]), f["objs_variance_to_edge"])
    # PCA objects: the box edges pass exactly through the cluster's extreme points, so whether such a
    # point counts as "inside" in get_lowest_point_rect (strict tests) is decided by the last bits of
    # the axes -- sklearn's and ours differ there.  Continuous fields must agree; the bottom must lie
    # between the lowest points of the box shrunk / grown by 1e-7, like the reference's own value.
    from oracle import mask as om
    o = pu.get_objs(members, rect, fit_method="PCA")
    ref = f["objs_PCA"]
    for x, r in zip(o, ref):
        np.testing.assert_allclose([x.t[0], x.t[2], x.l, x.w, x.ry], [r[0], r[2], r[3], r[4], r[6]], rtol=1e-9, atol=1e-9)
        c = np.array([x.t[0], x.t[2]])
        b_in = om.get_lowest_point_rect(rect, c, x.l - 1e-7, x.w - 1e-7, x.ry)
        b_out = om.get_lowest_point_rect(rect, c, x.l + 1e-7, x.w + 1e-7, x.ry)
        assert b_in <= x.t[1] <= b_out and b_in <= r[1] <= b_out
    # min_zx_area_fit: hull from the same Qhull call as the reference, everything else ours
    got = pu.min_area_rectangles(clusters)
    for k, (corners, angle, area) in enumerate(got):
        assert np.array_equal(np.concatenate([corners.reshape(-1), [angle, area]]), f["minarea"][k]), k
    o = pu.get_objs(members, rect, fit_method="min_zx_area_fit")
    assert np.array_equal(np.array([[*x.t, x.l, x.w, x.h, x.ry, x.volume] for x in o]), f["objs_min_zx_area_fit"])
    with pytest.raises(NotImplementedError):
        pu.get_objs(members, rect, fit_method="no_such_fit")
    # variance criterion values against the oracle on random clusters of awkward sizes (pairwise-sum edges)
    from modest_amd import ops
    from oracle import mask as om
    rng = np.random.default_rng(1)
    cl = [rng.standard_normal((n, 2)) * [2.0, 0.8] for n in (3, 7, 8, 9, 127, 128, 129, 300, 1031)]
    ang, cs = pu._angles(0.1)
    offs = np.cumsum([0] + [len(c) for c in cl]).astype(np.int32)
    best = ops.fit_boxes_variance(torch.from_numpy(np.concatenate(cl)).to(gpu), offs, cs)
    for c, b in zip(cl, best):
        assert om.variance_rectangle(c, return_index=True)[3] == int(b)
