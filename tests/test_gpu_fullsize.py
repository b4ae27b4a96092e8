"""GPU parity at the sizes bench.py measures: the whole path of one scan
against the oracle at BASELINE config 3 (30 k live points vs 10 x 36 frames = 10.8 M history points)
and config 5 (nuScenes shape: 35 k points, 20 traversals x 16 frames, remove_center, max_hs=-1.3,
900x1600 images), the NMS keep set on a thousand random box sets, and the RANSAC deviation (float64
fits vs sklearn's float32 LAPACK) on 150 seeded scans."""
import copy
import os
import tempfile
import types

import os

os.environ.setdefault("MODEST_PP4_CHECK", "1")   # modest_pp_score_block reads back b4_plan's overflow words (blocking): no task list may overflow
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _calib(mod):
    from modest_amd import synth
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "c.txt"), "w").write(synth.CALIB_TXT)
        return mod.Calibration(os.path.join(d, "c.txt"))


def _full_scan(gpu, sid, n_live, T, F, nusc, overrides, image_shape, max_hs):
    import torch
    from modest_amd import config, synth
    from modest_amd.frame_store import FrameStore
    from modest_amd.gen_label_files import gen_label_scan
    from modest_amd.generate_mask import generate_mask_scan
    from modest_amd.utils import kitti_util
    from oracle import labels as ol
    from oracle import mask as om
    from oracle import pp_score as opp
    s = synth.make_scan(sid, n_live=n_live, n_trav=T, n_frames=F, nusc=nusc, keep_frames=True)
    M = sum(len(h) for h in s.hist)
    # PP stage: frame store + descriptor table vs cKDTree (the reference's calls, all host threads)
    st = FrameStore(gpu, 0.3)
    items, hist, rels = [], [], []
    for t, fr in enumerate(s.frames):
        for f, (raw, rel, W) in enumerate(fr):
            items.append(((t, f), torch.from_numpy(raw).to(gpu), W))
            hist.append(((t, f), t))
            rels.append(rel)
    live_dev = torch.from_numpy(s.live_raw).to(gpu)
    items.append(("live", live_dev, s.live_W))
    st.insert_many(items)
    H, c = st.pp_score("live", s.live_rel, hist, np.stack(rels), s.world_from_ref, T, remove_center=nusc,
                       return_counts=True)
    Href, cref = opp.pp_score(s.live_xyz, s.hist, 0.3, workers=-1)
    assert np.array_equal(c.cpu().numpy().astype(np.int64), cref)
    pp = H.cpu().numpy()
    assert np.max(np.abs(pp.astype(np.float64) - Href.astype(np.float64))) <= 1e-6
    # mask / box / label stages on the device scores vs the oracle on the same scores
    margs = config.compose("generate_mask", ["data_root=/unused"] + overrides)
    largs = config.compose("generate_label_files", ["data_root=/unused", f"image_shape={list(image_shape)}"])
    calib = _calib(kitti_util)
    labels, objs, _ = generate_mask_scan(s.live_raw, pp, calib, margs, random_state=np.random.RandomState(3),
                                         ptc_dev=live_dev, pp_dev=H)
    text, _ = gen_label_scan(objs, calib, largs)
    cfg = copy.deepcopy(om.DEFAULT_CFG)
    cfg["plane_estimate"]["max_hs"] = max_hs
    ocalib = _calib(ol)
    ref = om.generate_mask_scan(s.live_raw, pp, ocalib, cfg=cfg, random_state=np.random.RandomState(3), n_jobs=-1)
    ref_text, _ = ol.gen_label_scan(ref["objs"], ocalib, image_shape=tuple(image_shape))
    assert np.array_equal(labels, ref["labels"])
    assert len(objs) == len(ref["objs"]) and len(objs) >= 3
    for o, r in zip(objs, ref["objs"]):
        np.testing.assert_allclose([*o.t, o.l, o.w, o.h, o.ry, o.volume], [*r.t, r.l, r.w, r.h, r.ry, r.volume],
                                   rtol=1e-9, atol=1e-9)
    assert text == ref_text
    return M


def test_config3_full_size_scan_vs_oracle(gpu):
    """BASELINE config 3: 30 000 live points vs 10 traversals x 36 frames (10.8 M history points)."""
    M = _full_scan(gpu, 7, 30_000, 10, 36, False, [], (1024, 1224), -1.5)
    assert M == 10_800_000


def test_config5_nuscenes_shape_vs_oracle(gpu):
    """BASELINE config 5 at its real size: 35 k points, 20 traversals x 16 frames, nuScenes KITTI2NU,
    remove_center on the history, plane_estimate.max_hs=-1.3, image_shape=[900,1600] (README.md:68-69)."""
    M = _full_scan(gpu, 8, 35_000, 20, 16, True, ["plane_estimate.max_hs=-1.3"], (900, 1600), -1.3)
    assert 10_500_000 < M <= 11_200_000   # remove_center drops the ego-vehicle returns of every history frame


def test_objs_nms_keep_sets_on_random_box_sets(gpu):
    """objs_nms orders by the diagonal of the IoU matrix (self-IoU = 1 +- float noise), so
    the keep set depends on last-bit behaviour of the kernel.  1200 box sets of 5-60 boxes, with
    near-duplicates, touching and nested boxes: keep sets equal the oracle's in both order modes."""
    from modest_amd.utils import pointcloud_utils as pcu
    from oracle import labels as ol
    rng = np.random.default_rng(77)
    n_diff_iou = 0
    for case in range(1200):
        k = int(rng.integers(5, 61))
        ctr = rng.uniform(-30, 30, (k, 2))
        lw = np.c_[rng.uniform(0.5, 6.0, k), rng.uniform(0.4, 2.5, k)]
        ry = rng.uniform(-np.pi, np.pi, k)
        for j in range(k // 3):            # near-duplicates, touching neighbours, nested boxes
            src, mode = int(rng.integers(0, k)), int(rng.integers(0, 3))
            dst = int(rng.integers(0, k))
            if mode == 0:
                ctr[dst] = ctr[src] + rng.normal(0, 0.02, 2)
                lw[dst] = lw[src] * (1 + rng.normal(0, 0.01, 2))
                ry[dst] = ry[src] + rng.normal(0, 0.01)
            elif mode == 1:
                d = np.array([np.cos(ry[src]), -np.sin(ry[src])]) * lw[src, 0]
                ctr[dst], lw[dst], ry[dst] = ctr[src] + d, lw[src], ry[src]
            else:
                ctr[dst], lw[dst], ry[dst] = ctr[src], lw[src] * 0.5, ry[src] + 0.3
        objs = [types.SimpleNamespace(t=np.array([ctr[i, 0], 1.0, ctr[i, 1]]), l=float(lw[i, 0]), w=float(lw[i, 1]),
                                      h=1.5, ry=float(ry[i]), score=float(rng.uniform())) for i in range(k)]
        ident = {id(o): i for i, o in enumerate(objs)}
        for by_score in (False, True):
            got = [ident[id(o)] for o in pcu.objs_nms(objs, by_score, 0.1)]
            ref = [ident[id(o)] for o in ol.objs_nms(objs, by_score, 0.1)]
            if got != ref:
                n_diff_iou += 1
    assert n_diff_iou == 0, f"{n_diff_iou} of 2400 keep sets differ from the oracle"


def test_ransac_deviation_does_not_change_labels(gpu):
    """DESIGN.md deviation (i): the device fits RANSAC trial planes in float64 -> float32, sklearn in
    float32 LAPACK.  50 seeds x 3 scans: the final labels equal the oracle's (sklearn) labels."""
    import torch
    from modest_amd import config, ops, synth
    from modest_amd.generate_mask import generate_mask_scan
    from modest_amd.utils import kitti_util
    from oracle import labels as ol
    from oracle import mask as om
    margs = config.compose("generate_mask", ["data_root=/unused"])
    calib, ocalib = _calib(kitti_util), _calib(ol)
    bad = []
    for sid in (41, 42, 43):
        s = synth.make_scan(sid, n_live=9000, n_trav=3, n_frames=4)
        off = np.cumsum([0] + [len(h) for h in s.hist]).astype(np.int64)
        live_dev = torch.from_numpy(s.live_raw).to(gpu)
        H = ops.pp_score(torch.from_numpy(s.live_xyz).to(gpu), torch.from_numpy(np.concatenate(s.hist)).to(gpu), off, 0.3)
        pp = H.cpu().numpy()
        for seed in range(50):
            labels, objs, _ = generate_mask_scan(s.live_raw, pp, calib, margs, random_state=np.random.RandomState(seed),
                                                 ptc_dev=live_dev, pp_dev=H)
            ref = om.generate_mask_scan(s.live_raw, pp, ocalib, random_state=np.random.RandomState(seed), n_jobs=-1)
            if not np.array_equal(labels, ref["labels"]) or len(objs) != len(ref["objs"]):
                bad.append((sid, seed))
    assert not bad, bad


def _block_full_size(gpu, n_live, T, F, nusc, matched=None, S=32, presence=None):
    """S (default 32 = configs/pp_score.yaml pp_batch, bench.py --pp-batch) consecutive scans of a shard at the benchmarked size
    through ONE block call (what the CLI and bench.py run, FrameStore.pp_score_batch(block=True)) -- and the same scans forced
    through two blocks of half the scans and through the join variant that reads the poses from memory instead of LDS (what a
    union of more than 2 048 entries takes): all three identical on every scan, and scans 0 / S/2-1 / S-1 against scipy's
    cKDTree on the stacked, transformed history (pre_compute_pp_score.py:132-150,188-193; all host threads)."""
    import os
    import torch
    from modest_amd import synth
    from modest_amd.frame_store import FrameStore
    from oracle import pp_score as opp
    if matched is None:
        sh = synth.make_shard(S, n_live=n_live, n_trav=T, n_frames=F, nusc=nusc, seed=4, presence=presence)
    else:
        sh = synth.make_shard_matched(S, n_live=n_live, n_trav=T, nusc=nusc, live_speed=matched[0], hist_speeds=matched[1:], seed=4,
                                      presence=presence)
    store = FrameStore(gpu, 0.3)
    items, ids = [], {}
    for t, tr in enumerate(sh.tracks):
        for j, (raw, W) in enumerate(tr):
            ids[(t, j)] = len(ids)
            items.append((ids[(t, j)], torch.from_numpy(raw).to(gpu), W))
    lives = []
    for sc in sh.scans:
        items.append((10 ** 6 + sc.index, torch.from_numpy(sc.live_raw).to(gpu), sc.live_W))
        lives.append(10 ** 6 + sc.index)
    store.insert_many(items)
    descs = [store.describe(lives[i], sc.live_rel, [ids[h] for h in sc.hist], sc.trav_list(), sc.rels, nusc)
             for i, sc in enumerate(sh.scans)]
    Ts = [sc.n_trav for sc in sh.scans]
    n0 = getattr(store, "block_calls", 0)
    Hs, cs = store.pp_score_batch(lives, descs, Ts, return_counts=True, block=True)
    assert getattr(store, "block_calls", 0) == n0 + 1, "the block path declined the benchmarked shape"
    torch.cuda.synchronize()
    for i in (0, S // 2 - 1, S - 1):
        lv, hist = sh.stacked(i)
        Href, cref = opp.pp_score(lv, hist, 0.3, workers=-1)
        assert cs[i].shape[1] == Ts[i]
        assert np.array_equal(cs[i].cpu().numpy().astype(np.int64), cref), i
        assert np.max(np.abs(Hs[i].cpu().numpy().astype(np.float64) - Href)) <= 1e-6   # compute_ephe_score's tolerance (measured 0)
        assert int(cref.sum()) > 1_000_000
    # two blocks of half the scans (what the split rule does past 3 x / 2 048 entries / the block window)
    h = S // 2
    lo = store.pp_score_batch(lives[:h], descs[:h], Ts[:h], return_counts=True, block=True)
    hi = store.pp_score_batch(lives[h:], descs[h:], Ts[h:], return_counts=True, block=True)
    assert getattr(store, "block_calls", 0) == n0 + 3
    for i in range(S):
        Hh, ch = (lo[0][i], lo[1][i]) if i < h else (hi[0][i - h], hi[1][i - h])
        assert torch.equal(ch, cs[i]) and torch.equal(Hh, Hs[i]), ("halves", i)
    # the poses read from memory (MODEST_PP4_DBG=256: the kernel instantiation of unions with more than 2 048 entries)
    old = os.environ.get("MODEST_PP4_DBG")
    os.environ["MODEST_PP4_DBG"] = "256"
    try:
        Hm, cm = store.pp_score_batch(lives, descs, Ts, return_counts=True, block=True)
        torch.cuda.synchronize()
    finally:
        if old is None:
            del os.environ["MODEST_PP4_DBG"]
        else:
            os.environ["MODEST_PP4_DBG"] = old
    for i in range(S):
        assert torch.equal(cm[i], cs[i]) and torch.equal(Hm[i], Hs[i]), ("poses from memory", i)
    return sh


def test_config3_block_of_32_scans_full_size(gpu):
    """BASELINE config 3 through the block path at the default block size: 32 consecutive scans x (30 000 live points vs 10 x 36
    frames), 670 union entries."""
    sh = _block_full_size(gpu, 30_000, 10, 36, False)
    assert sum(len(sh.tracks[t][j][0]) for t, j in sh.scans[0].hist) == 10_800_000


def test_config3_block_of_16_scans_full_size(gpu):
    """... and in blocks of 16 (the halves of a split block; rounds 4-5's default)."""
    _block_full_size(gpu, 30_000, 10, 36, False, S=16)


def test_config5_block_of_32_scans_full_size(gpu):
    """BASELINE config 5 through the block path: nuScenes shape, remove_center on the history, 35 k x 20 x 16 -- 940 union entries."""
    _block_full_size(gpu, 35_000, 20, 16, True)


def test_config3_block_on_reference_rule_windows_full_size(gpu):
    """... and on windows chosen as the reference chooses them (split_traintest.py:79-101: repeated frames at the fast traversals,
    a third of the frames shared at the slow ones; live 8 m/s, history 3-15 m/s): ~935 union entries for 32 scans."""
    sh = _block_full_size(gpu, 30_000, 10, 36, False, matched=(8.0, 3.0, 15.0))
    assert any(len(set(sc.hist)) < len(sc.hist) for sc in sh.scans)   # the lists do repeat frames


def test_block_of_32_scans_with_traversals_entering_and_leaving_full_size(gpu):
    """The reference accepts a traversal PER SCAN (closest pose within 3 m, split_traintest.py:17,79) and needs two (:111): T
    changes along a sequence.  Reference-rule windows, 12 tracks that enter and leave (T between 5 and 11 inside the block)."""
    from modest_amd import synth
    pr = synth.presence_ramp(32, 12, seed=3)
    sh = _block_full_size(gpu, 30_000, 12, 36, False, matched=(8.0, 3.0, 15.0), presence=pr)
    assert len({sc.n_trav for sc in sh.scans}) >= 3
