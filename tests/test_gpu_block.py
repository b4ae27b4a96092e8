"""Block path of the PP neighbour count (modest_pp_score_block, csrc/pp_v4.hip): consecutive scans of a shard in ONE
call -- the union of their frames binned once on the world lattice -- against the oracle (scipy cKDTree on the stacked,
transformed history: pre_compute_pp_score.py:132-150,188-193) and against the per-scan chain, bit for bit."""
import os

os.environ.setdefault("MODEST_PP4_CHECK", "1")   # modest_pp_score_block reads back b4_plan's overflow words (blocking): no task list may overflow
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _load(store, sh, dev, torch, key0=0):
    items, ids = [], {}
    for t, tr in enumerate(sh.tracks):
        for j, (raw, W) in enumerate(tr):
            ids[(t, j)] = key0 + len(ids)
            items.append((ids[(t, j)], torch.from_numpy(raw).to(dev), W))
    lives = []
    for sc in sh.scans:
        k = key0 + 100000 + sc.index
        items.append((k, torch.from_numpy(sc.live_raw).to(dev), sc.live_W))
        lives.append(k)
    store.insert_many(items)
    descs = [store.describe(lives[i], sc.live_rel, [ids[h] for h in sc.hist], [t for t, _ in sc.hist], sc.rels, sh.nusc)
             for i, sc in enumerate(sh.scans)]
    return lives, descs, ids


def _oracle_counts(sh, i, radius=0.3):
    from oracle import pp_score as opp
    lv, hist = sh.stacked(i)
    return opp.pp_score(lv, hist, radius, workers=-1)


@pytest.mark.parametrize("nusc,T,F,S", [(False, 3, 6, 7), (True, 4, 5, 6), (False, 33, 2, 6)])
def test_block_equals_oracle_and_chain(gpu, nusc, T, F, S):
    """Lyft shape, nuScenes shape (remove_center on the history, KITTI2NU = rot-z pi/2) and more than 32 traversals (the sixth
    traversal bit of the segmented popcount): counts == oracle, H <= 1e-6, block == per-scan chain."""
    import torch
    from modest_amd import synth
    from modest_amd.frame_store import FrameStore
    sh = synth.make_shard(S, n_live=5000, n_trav=T, n_frames=F, nusc=nusc, seed=3)
    store = FrameStore(gpu, 0.3)
    lives, descs, _ = _load(store, sh, gpu, torch)
    Hb, cb = store.pp_score_batch(lives, descs, T, return_counts=True, block=True)
    assert getattr(store, "block_calls", 0) == 1
    Hv, cv = store.pp_score_batch(lives, descs, T, return_counts=True, block=False)
    for i in range(S):
        Href, cref = _oracle_counts(sh, i)
        assert np.array_equal(cb[i].cpu().numpy().astype(np.int64), cref), i
        assert np.max(np.abs(Hb[i].cpu().numpy().astype(np.float64) - Href)) <= 1e-6   # tolerance of compute_ephe_score (measured 0)
        assert torch.equal(cb[i], cv[i]) and torch.equal(Hb[i], Hv[i])
    assert int(sum(int(c.sum()) for c in cb)) > 0


def test_block_choice_and_fallbacks(gpu):
    """The store takes the block path on its own from four scans that share most of their frames (>= 12 frames per
    traversal, union <= 4 x a scan's frames); shorter windows, few scans, a frame
    with points outside its table, poses that disagree with the lattice and a radius other than the store's all take the
    per-scan chain -- with identical results."""
    import torch
    from modest_amd import synth
    from modest_amd.frame_store import FrameStore
    sh = synth.make_shard(8, n_live=1500, n_trav=2, n_frames=24, n_per_frame=1500, seed=5)
    store = FrameStore(gpu, 0.3)
    lives, descs, ids = _load(store, sh, gpu, torch)
    ref = store.pp_score_batch(lives, descs, 2, return_counts=True, block=False)[1]
    n0 = getattr(store, "block_calls", 0)
    got = store.pp_score_batch(lives, descs, 2, return_counts=True)[1]   # automatic: 8 scans sharing 23 of 24 frames
    assert getattr(store, "block_calls", 0) == n0 + 1 and all(torch.equal(a, b) for a, b in zip(got, ref))
    got = store.pp_score_batch(lives[:3], descs[:3], 2, return_counts=True)[1]   # too few scans for the block to pay
    assert getattr(store, "block_calls", 0) == n0 + 1 and all(torch.equal(a, b) for a, b in zip(got, ref[:3]))
    # a pose that disagrees with the lattice (a caller may hand in anything): the per-scan chain takes it
    lv, arr, sl = descs[2]
    arr2 = arr.copy()
    arr2["rel"][0][3] += 0.01
    bad = [descs[0], descs[1], (lv, arr2, sl)] + list(descs[3:])
    assert store.block_tables(bad, 2, force=True) is None
    # a frame with points outside its table (200 m from the sensor): known to the store, the block path declines
    raw = sh.tracks[0][0][0].copy()
    raw[:50, 0] += 400.0
    store.insert_many([(777777, torch.from_numpy(raw).to(gpu), sh.tracks[0][0][1])])
    lv, arr, sl = descs[0]
    sl2 = sl.copy()
    sl2[0] = store.frames[777777].slot
    arr3 = arr.copy()
    arr3["xyz_dev"][0], arr3["tab_dev"][0] = store.frames[777777].xyz.data_ptr(), store.frames[777777].tab.data_ptr()
    assert store.block_tables([(lv, arr3, sl2)] + list(descs[1:]), 2, force=True) is None


def test_block_ragged(gpu):
    """Scans of different sizes, a scan without history, a scan without live points, frames shared by some scans only, and
    a block of ONE scan: every scan equals its own per-scan call."""
    import torch
    from modest_amd import synth
    from modest_amd.frame_store import FrameStore, PP_FRAME
    sh = synth.make_shard(6, n_live=2500, n_trav=3, n_frames=4, seed=9)
    store = FrameStore(gpu, 0.3)
    lives, descs, ids = _load(store, sh, gpu, torch)
    # scan 1 loses a whole traversal and a few frames, scan 4 has no history at all
    lv, arr, sl = descs[1]
    keep = np.array([k for k in range(len(arr)) if arr["trav"][k] != 1 and k % 5 != 0])
    a1 = arr[keep].copy()
    a1["trav"] = np.where(a1["trav"] == 2, 1, a1["trav"])
    descs[1] = (lv, a1, np.concatenate([sl[:-1][keep], sl[-1:]]))
    lv4, arr4, sl4 = descs[4]
    descs[4] = (lv4, np.zeros(1, dtype=PP_FRAME), sl4[-1:])
    ref = []
    for k, d in zip(lives, descs):
        T = 2 if d is descs[1] else 3
        ref.append(store.pp_score_batch([k], [d], T, return_counts=True, block=False)[1][0])
    # (one traversal count per call: scan 1 is checked in its own block of one)
    idx = [0, 2, 3, 4, 5]
    got = store.pp_score_batch([lives[i] for i in idx], [descs[i] for i in idx], 3, return_counts=True, block=True)[1]
    for g, i in zip(got, idx):
        assert torch.equal(g, ref[i]), i
    assert int(ref[4].abs().sum()) == 0 and int(ref[0].sum()) > 0
    one = store.pp_score_batch([lives[1]], [descs[1]], 2, return_counts=True, block=True)[1][0]
    assert torch.equal(one, ref[1])


def test_block_dense_window_bands(gpu):
    """Live points piled up next to the sensor (hundreds per lattice cell: more than a join workgroup keeps in LDS for three
    cell rows) exercise the column-split bands and the global-memory walk of a single overfull cell."""
    import torch
    from modest_amd import synth
    from modest_amd.frame_store import FrameStore
    sh = synth.make_shard(6, n_live=6000, n_trav=3, n_frames=4, seed=11)
    rng = np.random.default_rng(0)
    for sc in sh.scans:   # 3000 live points inside half a metre, 1500 of them inside one cell
        sc.live_raw[:3000, :2] = rng.uniform(-0.25, 0.25, (3000, 2)).astype(np.float32) + np.float32(3.0)
        sc.live_raw[:1500, :2] = rng.uniform(-0.1, 0.1, (1500, 2)).astype(np.float32) + np.float32(3.0)
        sc.live_raw[:3000, 2] = rng.uniform(-1.8, -1.6, 3000).astype(np.float32)
    for tr in sh.tracks:   # ... and history there
        for raw, W in tr:
            raw[:800, :2] = rng.uniform(-0.4, 0.4, (800, 2)).astype(np.float32) + np.float32(3.0)
            raw[:800, 2] = rng.uniform(-1.8, -1.6, 800).astype(np.float32)
    store = FrameStore(gpu, 0.3)
    lives, descs, _ = _load(store, sh, gpu, torch)
    cb = store.pp_score_batch(lives, descs, 3, return_counts=True, block=True)[1]
    cv = store.pp_score_batch(lives, descs, 3, return_counts=True, block=False)[1]
    assert all(torch.equal(a, b) for a, b in zip(cb, cv))
    _, cref = _oracle_counts(sh, 2)
    assert np.array_equal(cb[2].cpu().numpy().astype(np.int64), cref) and cref.max() > 200


def _tree_descs(store, g, gpu, torch):
    """the descriptor tables the CLI builds (modest_amd/pre_compute_pp_score.py: plans + relative_poses + describe) for every origin
    of an unpacked golden tree"""
    import pickle
    from modest_amd import pre_compute_pp_score as pps
    from oracle import pp_score as opp
    off = g["bin_offsets"]
    bins = [g["bins"][off[i]:off[i + 1]] for i in range(len(off) - 1)]
    track, valid = pickle.loads(g["track"].tobytes()), pickle.loads(g["valid"].tobytes())
    poses = [[opp.load_pose(str(g["oxts"][i])) for i in seq] for seq in track]
    l2es = [[g["l2e"][i] for i in seq] for seq in track]
    K = pps._KITTI2NU_lyft
    world = pps.frame_world_matrices(track, poses, l2es, K)
    store.insert_many([(i, torch.from_numpy(np.ascontiguousarray(b)).to(gpu), world[i]) for i, b in enumerate(bins)])
    lives, descs = [], []
    for o in g["origins"]:
        seq0, fr0, trav = valid[int(o)]
        hist_ids = [track[s][f] for s, ix in trav for f in ix]
        travs = [t for t, (s, ix) in enumerate(trav) for _ in ix]
        live = track[seq0][fr0]
        fs, fi = trav[0]
        rels = pps.relative_poses(l2es[fs][fi[0]], poses[fs][fi[0]], world.stack(hist_ids + [live]), K)
        lives.append(live)
        descs.append(store.describe(live, rels[-1], hist_ids, travs, rels[:-1], False))
    return lives, descs


def test_repeated_frames_block_chain_and_golden(gpu, golden_dir):
    """A history frame counts as often as the scan's index list names it (pre_compute_pp_score.py:132-150; lists with repeats:
    split_traintest.py:86-101).  tests/golden/pp_repeats.npz = the reference's own main + count_neighbors on eight
    consecutive scans whose lists repeat a frame once / three times / at the first and last position: the block path
    (one union entry per occurrence), the per-scan chain and the single-scan call all reproduce its counts bit for bit."""
    import torch
    from modest_amd.frame_store import FrameStore
    g = np.load(f"{golden_dir}/pp_repeats.npz")
    store = FrameStore(gpu, 0.3)
    lives, descs = _tree_descs(store, g, gpu, torch)
    assert all(len(np.unique(sl[:-1])) < len(sl) - 1 for _, _, sl in descs)   # every scan repeats frames
    n0 = getattr(store, "block_calls", 0)
    Hb, cb = store.pp_score_batch(lives, descs, 3, return_counts=True, block=True)
    assert getattr(store, "block_calls", 0) == n0 + 1
    Hv, cv = store.pp_score_batch(lives, descs, 3, return_counts=True, block=False)
    for i, o in enumerate(g["origins"]):
        assert np.array_equal(cb[i].cpu().numpy(), g[f"counts_{o}"]), ("block", int(o))
        assert np.array_equal(cv[i].cpu().numpy(), g[f"counts_{o}"]), ("chain", int(o))
        assert np.max(np.abs(Hb[i].cpu().numpy() - g[f"pp_{o}"])) <= 1e-6 and torch.equal(Hb[i], Hv[i])
    one = store.pp_score_batch(lives[3:4], descs[3:4], 3, return_counts=True, block=False)[1][0]
    assert np.array_equal(one.cpu().numpy(), g[f"counts_{g['origins'][3]}"])


def test_repeated_frames_on_a_shard(gpu):
    """make_shard windows with repeats spliced in (a frame twice in a row, a frame three times, the window's first frame again
    at its end; different scans repeat different frames): block == chain == oracle, and the union holds an entry per
    occurrence."""
    import torch
    from modest_amd import synth
    from modest_amd.frame_store import FrameStore
    T, F, S = 3, 6, 8
    sh = synth.make_shard(S, n_live=4000, n_trav=T, n_frames=F, seed=21)
    for sc in sh.scans:
        hist, rels = list(sc.hist), list(sc.rels)
        per = [[k for k, (t, _) in enumerate(hist) if t == tt] for tt in range(T)]
        new = []
        for tt, ks in enumerate(per):
            if tt == 0:
                ks = ks[:2] + [ks[1]] + ks[2:]                    # once (only on even scans: the union mixes both kinds)
                if sc.index % 2:
                    ks = per[tt]
            elif tt == 1:
                ks = ks[:3] + [ks[2], ks[2]] + ks[3:]             # three times
            else:
                ks = ks + [ks[0]]                                 # first == last
            new += ks
        sc.hist = [hist[k] for k in new]
        sc.rels = np.stack([rels[k] for k in new])
    store = FrameStore(gpu, 0.3)
    lives, descs, _ = _load(store, sh, gpu, torch)
    tabs = store.block_tables(descs, T, force=True)
    assert tabs is not None
    fr, sc_, _ = tabs
    assert len(fr) > len(np.unique(np.concatenate([sl[:-1] for _, _, sl in descs])))   # occurrences, not frames
    Hb, cb = store.pp_score_batch(lives, descs, T, return_counts=True, block=True)
    Hv, cv = store.pp_score_batch(lives, descs, T, return_counts=True, block=False)
    for i in range(S):
        Href, cref = _oracle_counts(sh, i)
        assert np.array_equal(cb[i].cpu().numpy().astype(np.int64), cref), i
        assert torch.equal(cb[i], cv[i]) and torch.equal(Hb[i], Hv[i])


def test_block_refuses_a_member_list_that_names_a_slot_twice(gpu):
    """the C ABI's own guard (a caller other than FrameStore): one pose entry exists per (scan, union slot)"""
    import torch
    from modest_amd import synth, _lib
    from modest_amd.frame_store import FrameStore
    sh = synth.make_shard(2, n_live=1000, n_trav=2, n_frames=3, seed=2)
    store = FrameStore(gpu, 0.3)
    lives, descs, _ = _load(store, sh, gpu, torch)
    fr, sc, keep = store.block_tables(descs, 2, force=True)
    keep[0][1] = keep[0][0]   # scan 0 now names its first union slot twice
    H = [torch.empty((1000,), dtype=torch.float32, device=gpu) for _ in range(2)]
    sc["H_dev"] = [h.data_ptr() for h in H]
    lib = _lib.load()
    rc = lib.modest_pp_score_block(store._ctx().handle, fr.ctypes.data, len(fr), sc.ctypes.data, 2, 2, 0.3, store.cell,
                                   torch.cuda.current_stream().cuda_stream)
    assert rc != 0 and b"twice" in lib.modest_last_error()


@pytest.mark.parametrize("T,F,S", [(40, 20, 8), (64, 26, 8)])
def test_block_with_large_union_tables(gpu, T, F, S):
    """1 080 union entries (a 53 KB pose table in the join's LDS) and 2 112 (more than the LDS table holds: the variant that reads
    the poses from memory): == the per-scan chain, and == the oracle on the first and the last scan."""
    import torch
    from modest_amd import synth
    from modest_amd.frame_store import FrameStore
    sh = synth.make_shard(S, n_live=2000, n_trav=T, n_frames=F, n_per_frame=300, seed=21)
    store = FrameStore(gpu, 0.3)
    lives, descs, _ = _load(store, sh, gpu, torch)
    tabs = store.block_tables(descs, T, force=True)
    assert tabs is not None and len(tabs[0]) == T * (F + S - 1)
    Hb, cb = store.pp_score_batch(lives, descs, T, return_counts=True, block=True)
    Hv, cv = store.pp_score_batch(lives, descs, T, return_counts=True, block=False)
    for i in range(S):
        assert torch.equal(cb[i], cv[i]) and torch.equal(Hb[i], Hv[i]), i
    for i in (0, S - 1):
        assert np.array_equal(cb[i].cpu().numpy().astype(np.int64), _oracle_counts(sh, i)[1]), i
    assert int(sum(int(c.sum()) for c in cb)) > 0


@pytest.mark.parametrize("tk,lf", [(1, 0), (3, 4), (2, 16), (7, 1)])
def test_block_join_deal_does_not_change_counts(gpu, monkeypatch, tk, lf):
    """The join deals its tasks dynamically (tickets of MODEST_PP4_TK tasks, MODEST_PP4_LF wavefronts of a workgroup start with
    the four-cell queue): every count is an integer sum over tasks, whatever the deal."""
    import torch
    from modest_amd import synth
    from modest_amd.frame_store import FrameStore
    sh = synth.make_shard(9, n_live=6000, n_trav=4, n_frames=8, seed=13)
    store = FrameStore(gpu, 0.3)
    lives, descs, _ = _load(store, sh, gpu, torch)
    ref = store.pp_score_batch(lives, descs, 4, return_counts=True, block=False)[1]
    monkeypatch.setenv("MODEST_PP4_TK", str(tk))
    monkeypatch.setenv("MODEST_PP4_LF", str(lf))
    got = store.pp_score_batch(lives, descs, 4, return_counts=True, block=True)[1]
    assert all(torch.equal(a, b) for a, b in zip(got, ref))


def test_blocks_of_32_and_the_split_rule(gpu):
    """Many consecutive scans in ONE call: 32 scans with long windows (40 frames per traversal, union 1.78 x a scan's entries) and with
    short ones (12 frames: 3.6 x) go as one block; 48 scans with 12 frames (4.9 x, past the rule's 4 x) go as two blocks of 24
    (FrameStore.block_tables: SPLIT_BLOCK) -- every scan equals the per-scan chain either way, and the first / last scan the oracle."""
    import torch
    from modest_amd import synth
    from modest_amd.frame_store import FrameStore
    for S, F, calls in ((32, 40, 1), (32, 12, 1), (48, 12, 2)):
        sh = synth.make_shard(S, n_live=1500, n_trav=2, n_frames=F, n_per_frame=1200, seed=31 + F, frame_gap=0.8)
        store = FrameStore(gpu, 0.3)
        lives, descs, _ = _load(store, sh, gpu, torch)
        ref = store.pp_score_batch(lives, descs, 2, return_counts=True, block=False)[1]
        n0 = getattr(store, "block_calls", 0)
        Hb, cb = store.pp_score_batch(lives, descs, 2, return_counts=True)   # the automatic rule
        assert getattr(store, "block_calls", 0) - n0 == calls, (F, getattr(store, "block_calls", 0) - n0)
        assert all(torch.equal(a, b) for a, b in zip(cb, ref))
        for i in (0, S - 1):
            assert np.array_equal(cb[i].cpu().numpy().astype(np.int64), _oracle_counts(sh, i)[1]), (S, F, i)


def test_block_with_different_traversal_counts_per_scan(gpu):
    """The reference accepts a traversal PER SCAN (closest pose within 3 m, data_preprocessing/lyft/split_traintest.py:17,79) and only
    asks for two (:111; pre_compute_pp_score.py:125-126): T changes along a sequence, and so does the traversal index a track has in a
    scan's list.  One block of scans with at least three distinct T == the per-scan chain (one call per run of equal T) == the oracle
    (counts (n, T_scan), H normalised by ln T_scan, pre_compute_pp_score.py:68-75) -- on windows i..i+F-1 and on reference-rule windows."""
    import torch
    from modest_amd import synth
    from modest_amd.frame_store import FrameStore
    S = 12
    for matched in (False, True):
        pr = synth.presence_ramp(S, 7, seed=2)
        if matched:
            sh = synth.make_shard_matched(S, n_live=4000, n_trav=7, n_per_frame=4000, live_speed=8.0, hist_speeds=(3.0, 15.0), seed=5, presence=pr)
        else:
            sh = synth.make_shard(S, n_live=4000, n_trav=7, n_frames=5, n_per_frame=4000, seed=5, presence=pr)
        Ts = [sc.n_trav for sc in sh.scans]
        assert len(set(Ts)) >= 3, Ts
        store = FrameStore(gpu, 0.3)
        lives, _, ids = _load(store, sh, gpu, torch)
        descs = [store.describe(lives[i], sc.live_rel, [ids[h] for h in sc.hist], sc.trav_list(), sc.rels, sh.nusc)
                 for i, sc in enumerate(sh.scans)]
        n0 = getattr(store, "block_calls", 0)
        Hb, cb = store.pp_score_batch(lives, descs, Ts, return_counts=True, block=True)
        assert getattr(store, "block_calls", 0) == n0 + 1
        Hv, cv = store.pp_score_batch(lives, descs, Ts, return_counts=True, block=False)
        assert getattr(store, "block_calls", 0) == n0 + 1
        for i in range(S):
            Href, cref = _oracle_counts(sh, i)
            assert cref.shape[1] == Ts[i] and tuple(cb[i].shape) == cref.shape
            assert np.array_equal(cb[i].cpu().numpy().astype(np.int64), cref), (matched, i)
            assert np.max(np.abs(Hb[i].cpu().numpy().astype(np.float64) - Href)) <= 1e-6
            assert torch.equal(cb[i], cv[i]) and torch.equal(Hb[i], Hv[i]), (matched, i)
        # a scan alone, with its own T, through the single-scan entry point
        k = int(np.argmin(Ts))
        one = store.pp_score_batch(lives[k:k + 1], descs[k:k + 1], Ts[k], return_counts=True, block=False)[1][0]
        assert torch.equal(one, cb[k])


def test_block_path_on_random_shard_shapes(gpu):
    """tools/r06_block_fuzz.py: random live / frame sizes, traversal counts (constant and per scan), window lengths, block sizes, radii,
    Lyft and nuScenes shape, sliding and reference-rule windows -- block == per-scan chain on every scan, == the oracle on two scans per
    shard (profiles/r06_block_fuzz.txt holds a 40-case run)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "r06_block_fuzz.py"), "12", "11"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "12 cases, 0 mismatches" in r.stdout, r.stdout[-3000:]
