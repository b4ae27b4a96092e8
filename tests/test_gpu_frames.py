"""GPU parity of the frame-store PP path (modest_frame_sort + modest_pp_score_frames, through the
C ABI) against the oracle (the reference's scipy calls) on the same synthetic frames."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _store_scan(gpu, s, radius=0.3, nusc=False):
    import torch
    from modest_amd.frame_store import FrameStore
    st = FrameStore(gpu, radius)
    items, hist, rels = [], [], []
    for t, fr in enumerate(s.frames):
        for f, (raw, rel, W) in enumerate(fr):
            items.append(((t, f), torch.from_numpy(raw).to(gpu), W))
            hist.append(((t, f), t))
            rels.append(rel)
    items.append(("live", torch.from_numpy(s.live_raw).to(gpu), s.live_W))
    st.insert_many(items)
    return st, hist, np.stack(rels)


@pytest.mark.parametrize("nusc", [False, True])
def test_frames_vs_oracle(gpu, nusc):
    from modest_amd import synth
    from oracle import pp_score as opp
    s = synth.make_scan(21, n_live=20000, n_trav=5, n_frames=8, n_per_frame=20000, keep_frames=True, nusc=nusc)
    Href, cref = opp.pp_score(s.live_xyz, s.hist, 0.3, workers=-1)
    st, hist, rels = _store_scan(gpu, s)
    H, c = st.pp_score("live", s.live_rel, hist, rels, s.world_from_ref, 5, remove_center=nusc, return_counts=True)
    assert np.array_equal(c.cpu().numpy().astype(np.int64), cref)
    assert np.max(np.abs(H.cpu().numpy().astype(np.float64) - Href.astype(np.float64))) <= 1e-6
    # the stacked fallback (V3 kernels on transformed store frames) gives the same answer
    H2, c2 = st.pp_score("live", s.live_rel, hist, rels, s.world_from_ref, 5, remove_center=nusc, return_counts=True,
                         force_stacked=True)
    assert np.array_equal(c2.cpu().numpy(), c.cpu().numpy())
    assert np.array_equal(H2.cpu().numpy(), H.cpu().numpy())


def test_frame_sort_layout(gpu):
    """The store's sort is a permutation of the frame, tile runs follow the table."""
    import torch
    from modest_amd import synth
    from modest_amd.frame_store import FrameStore
    s = synth.make_scan(3, n_live=5000, n_trav=2, n_frames=1, keep_frames=True)
    raw, rel, W = s.frames[0][0]
    st = FrameStore(gpu, 0.3)
    f = st.insert("a", torch.from_numpy(raw).to(gpu), W)
    xyz, perm, tab = f.xyz.cpu().numpy(), f.perm.cpu().numpy().astype(np.int64), f.tab.cpu().numpy().astype(np.int64)
    assert f.n_inside == f.n == raw.shape[0]
    assert np.array_equal(np.sort(perm), np.arange(raw.shape[0]))
    assert np.array_equal(xyz, raw[perm, :3])
    assert np.array_equal(f.original_order().cpu().numpy(), raw[:, :3])
    rows = st.lattice_rows(W).reshape(2, 4)
    lat = raw[:, :3].astype(np.float64) @ rows[:, :3].T + rows[:, 3]
    tile = (np.floor(lat[:, 1]).astype(np.int64) // 8 - f.TY0) * st.ntf + (np.floor(lat[:, 0]).astype(np.int64) // 8 - f.TX0)
    assert np.all(np.diff(tab) >= 0) and tab[0] == 0 and tab[-1] == f.n
    ts = tile[perm]
    assert np.all(np.diff(ts) >= 0)
    assert np.array_equal(np.searchsorted(ts, np.arange(st.ntf * st.ntf + 1)), tab)


def _rigid(rng, yaw_range=3.1, shift=20.0):
    from scipy.spatial.transform import Rotation as R
    t = np.eye(4)
    t[:3, :3] = R.from_euler("xyz", [rng.uniform(-0.02, 0.02), rng.uniform(-0.02, 0.02), rng.uniform(-yaw_range, yaw_range)]).as_matrix()
    t[:3, 3] = [rng.uniform(-shift, shift), rng.uniform(-shift, shift), rng.uniform(-0.3, 0.3)]
    return t


@pytest.mark.parametrize("T,radius", [(1, 0.3), (7, 0.5), (33, 0.3), (64, 0.3), (65, 0.3)])
def test_frames_edge_cases(gpu, T, radius):
    """Ragged inputs through the frame API: empty frames, a frame far outside the live window, a frame
    with points beyond its own table (outliers), duplicated points, arbitrary rigid poses with roll/pitch, 1..65 traversals (65 takes the
    stacked path), another radius.  Counts are compared with the brute-force oracle on the
    reference-transformed points."""
    import torch
    from modest_amd.frame_store import FrameStore
    from oracle import pp_score as opp
    rng = np.random.default_rng(1000 * T + int(radius * 10))
    world0 = _rigid(rng, shift=500.0)                      # the common (first history) frame in the world

    def transform(p, M):
        hom = np.hstack((p.astype(np.float32), np.ones((p.shape[0], 1), dtype=np.float32)))
        return np.dot(hom, M.astype(np.float32).T)[:, :3]

    live_W = world0 @ _rigid(rng, shift=5.0)
    live_raw = np.concatenate([rng.standard_normal((1500, 3)) * [8, 8, 0.4], rng.standard_normal((300, 3)) * [0.5, 0.5, 0.2]]
                              ).astype(np.float32)
    live_rel = np.linalg.solve(world0, live_W).astype(np.float32)
    st = FrameStore(gpu, radius)
    items, hist, rels, stacks = [("live", torch.from_numpy(live_raw).to(gpu), live_W)], [], [], [[] for _ in range(T)]
    n_fr = 0
    for t in range(T):
        for f in range(int(rng.integers(0, 4)) if T > 8 else 3):
            W = world0 @ _rigid(rng, shift=15.0)
            kind = (t + f) % 5
            n = 0 if kind == 0 else int(rng.integers(50, 2500))
            raw = (rng.standard_normal((n, 3)) * [10, 10, 0.4]).astype(np.float32)
            if kind == 1 and n:
                raw[: n // 4] = raw[0]                      # duplicates
            if kind == 2:
                W = world0 @ _rigid(rng, shift=15.0)
                W[:3, 3] += [4000.0, -3000.0, 0.0]          # nowhere near the live scan
            if kind == 3 and n:
                raw[:5] += np.float32(900.0)                # outside the frame's own table
            rel = np.linalg.solve(world0, W).astype(np.float32)
            key = (t, f)
            items.append((key, torch.from_numpy(raw).to(gpu), W))
            hist.append((key, t))
            rels.append(rel)
            stacks[t].append(transform(raw, rel))
            n_fr += 1
    st.insert_many(items)
    rels = np.stack(rels) if rels else np.zeros((0, 4, 4), np.float32)
    H, c = st.pp_score("live", live_rel, hist, rels, world0, T, return_counts=True)
    live_xyz = transform(live_raw, live_rel)
    ref_hist = [np.concatenate(s) if s else np.zeros((0, 3), np.float32) for s in stacks]
    cref = opp.count_neighbors_bruteforce(live_xyz, ref_hist, radius)
    assert np.array_equal(c.cpu().numpy().astype(np.int64), cref), (T, radius, n_fr)
    if T > 1:
        Href = opp.compute_ephe_score(cref)
        assert np.max(np.abs(H.cpu().numpy().astype(np.float64) - Href)) <= 1e-6


@pytest.mark.gpu
def test_batched_chain_equals_separate_calls(gpu):
    """modest_pp_score_frames_batch: several scans through one chain of launches (scan = blockIdx.y) give the
    counts and scores of separate calls bit for bit -- scans of different size, traversal count below the
    batch's, one scan without history (falls back to separate calls), more scans than one group takes."""
    import torch
    from modest_amd.frame_store import FrameStore
    rng = np.random.default_rng(11)
    T = 5
    st = FrameStore(gpu, 0.3)
    keys, descs, lives = [], [], []
    key = 0
    for sid in range(11):
        n_live = int(rng.integers(800, 6000))
        n_frames = int(rng.integers(1, 9)) if sid != 4 else 0
        yaw = rng.uniform(-0.2, 0.2)
        base = np.eye(4)
        base[:2, :2] = [[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]]
        base[:3, 3] = rng.uniform(-3, 3, 3) * [1, 1, 0.05]
        items, hist_keys, travs, rels = [], [], [], []
        cloud = rng.uniform(-25, 25, (20000, 3)).astype(np.float32) * np.float32([1, 1, 0.04])
        for f in range(n_frames):
            sel = rng.choice(len(cloud), int(rng.integers(500, 9000)), replace=False)
            raw = cloud[sel] + rng.normal(0, 0.05, (len(sel), 3)).astype(np.float32)
            items.append((key, torch.from_numpy(np.ascontiguousarray(raw)).to(gpu), base.copy()))
            hist_keys.append(key)
            travs.append(int(rng.integers(0, T)))
            rels.append(np.eye(4, dtype=np.float32))
            key += 1
        live_raw = cloud[rng.choice(len(cloud), n_live, replace=False)]
        items.append((key, torch.from_numpy(np.ascontiguousarray(live_raw)).to(gpu), base.copy()))
        lk = key
        key += 1
        st.insert_many(items)
        rel = np.eye(4, dtype=np.float32)
        desc = st.describe(lk, rel, hist_keys, travs, np.stack(rels) if rels else np.zeros((0, 4, 4), np.float32), sid % 2 == 1)
        keys.append(lk)
        descs.append(desc)
        lives.append((rel, list(zip(hist_keys, travs)), np.stack(rels) if rels else np.zeros((0, 4, 4), np.float32), base, sid % 2 == 1))
    single = []
    for lk, d, (rel, hist, rels, base, rc) in zip(keys, descs, lives):
        single.append(st.pp_score(lk, rel, hist, rels, base, T, remove_center=rc, return_counts=True, desc=d))
    for group in (list(range(11)), [0, 1, 2, 3], [5, 6]):
        Hs, cs = st.pp_score_batch([keys[i] for i in group], [descs[i] for i in group], T, return_counts=True)
        torch.cuda.synchronize()
        for i, H, c in zip(group, Hs, cs):
            assert torch.equal(c, single[i][1]), i
            assert torch.equal(H, single[i][0]), i
    assert any(int(c.sum()) > 0 for _, c in single)


def test_store_capacity_counts_slabs_and_reader_streams(gpu):
    """the store's capacity is what it really holds -- whole slabs, which go back to the allocator only when
    their last frame is gone -- and a slab freed by eviction is not reused before the kernels of the streams that read its
    frames have finished (every reading stream is recorded on every slab)."""
    import torch
    from modest_amd.frame_store import FrameStore
    rng = np.random.default_rng(5)
    store = FrameStore(gpu, 0.3, capacity_bytes=3 << 20)
    store.slab_bytes = 1 << 20   # small slabs: ~2 blocks of 3 frames each
    def block(base):
        raws = [np.c_[rng.uniform(-30, 30, 9000), rng.uniform(-30, 30, 9000), rng.uniform(-2, 1, 9000), np.zeros(9000)].astype(np.float32)
                for _ in range(3)]
        offs = np.cumsum([0] + [len(r) for r in raws])
        Ws = np.stack([np.eye(4) for _ in raws])
        store.insert_block([base + k for k in range(3)], torch.from_numpy(np.concatenate(raws)).to(gpu), offs, Ws)
    side = torch.cuda.Stream(device=gpu)
    store.note_reader(side)
    for b in range(12):
        block(10 * b)
        assert store.footprint() <= store.cap + store.slab_bytes   # never more than one slab over (the one being carved)
    assert side.cuda_stream in store._readers and len(store.frames) < 36 and len(store.frames) >= 3
    assert store.footprint() == sum(ent[1] for ent in store._slabs.values()) and store.bytes <= store.footprint()
    torch.cuda.synchronize()
    for key, f in store.frames.items():   # the survivors are intact
        assert f.n_inside == f.n == 9000
    # an insertion whose first carve nearly fills a fresh slab: its other carves open the next slab while the first one
    # has no counted frame yet (it must stay in the table)
    store = FrameStore(gpu, 0.3, capacity_bytes=64 << 20)
    store.slab_bytes = 340 << 10   # 27 000 points x 12 bytes = 324 KB
    for b in range(3):
        block(10 * b)
    assert len(store.frames) == 9 and all(f.n_inside == 9000 for f in store.frames.values())
    assert store.footprint() == sum(ent[1] for ent in store._slabs.values()) >= store.bytes


def test_frame_loader_takes_files_of_any_size(gpu, tmp_path):
    """FrameLoader (the PP CLI's ingest: modest_host_read_files -> pinned ring -> one copy + one sort launch per group) on `.bin`
    frames whose sizes differ by a factor of 300 -- the staging ring is sized from ONE probed file (pre_compute_pp_score.py) and has
    to grow in the middle of a group: every stored frame, put back into file order, is the file's xyz bit for bit (load_velo_scan,
    utils/pointcloud_utils.py:22-25), on the asynchronous path the ingest thread uses and on the blocking one; a missing file is an
    IOError that names it."""
    import torch
    from modest_amd import _lib
    from modest_amd.frame_store import FrameStore
    from modest_amd.pre_compute_pp_score import FrameLoader, WorldTable
    rng = np.random.default_rng(8)
    d = tmp_path / "velodyne"
    d.mkdir()
    sizes = [40] * 6 + [int(x) for x in rng.integers(16, 12000, size=70)] + [12000, 17, 9000]
    world = WorldTable()
    files = {}
    for i, n in enumerate(sizes):
        pts = np.concatenate([rng.uniform(-60, 60, size=(n, 2)), rng.uniform(-3, 3, size=(n, 1)), rng.uniform(0, 1, size=(n, 1))], axis=1).astype(np.float32)
        pts.tofile(d / f"{i:06d}.bin")
        files[i] = pts
        W = np.eye(4)
        W[:2, 3] = rng.uniform(-5, 5, size=2)
        world[i] = W
    world.freeze()
    dev = torch.device("cuda:0")
    for blocking in (False, True):
        store = FrameStore(dev, 0.3)
        loader = FrameLoader(str(d), store, world, readers=3, ctx=_lib.Context(0), frame_bytes=40 * 16)   # (probe = a small file)
        cap0 = loader.cap
        ids = list(range(len(sizes)))
        loader.ensure(ids[:6], blocking=blocking)
        assert loader.cap == cap0
        loader.ensure(ids, blocking=blocking)      # 73 new frames in pieces of 32: every piece is larger than the ring
        loader.ensure(ids[10:20], blocking=blocking)   # (all resident: nothing to read)
        torch.cuda.synchronize()
        assert loader.cap > cap0 and loader.n_frames == len(sizes) and loader.read_bytes == 16 * sum(sizes)
        for i in ids:
            f = store.frames[i]
            got = f.original_order().cpu().numpy()
            assert got.shape == (sizes[i], 3) and np.array_equal(got, files[i][:, :3]), (blocking, i, sizes[i])
            assert np.array_equal(np.sort(f.perm.cpu().numpy()), np.arange(sizes[i]))
    world[999] = np.eye(4)
    with pytest.raises(IOError, match="000999.bin"):
        loader.ensure([999], blocking=False)
