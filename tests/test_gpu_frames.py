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


@pytest.fixture(params=["stream", "gather", "gather-fused", "gather-wave"])
def frames_path(request, monkeypatch):
    """The three device paths behind modest_pp_score_frames: the V3 streaming kernels over the
    descriptor table (default), the split / fused gather-join and its wave-autonomous variant (MODEST_PP_FRAMES_PATH)."""
    if request.param == "stream":
        monkeypatch.delenv("MODEST_PP_FRAMES_PATH", raising=False)
    else:
        monkeypatch.setenv("MODEST_PP_FRAMES_PATH", request.param)
    return request.param


@pytest.mark.parametrize("nusc", [False, True])
def test_frames_vs_oracle(gpu, nusc, frames_path):
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
