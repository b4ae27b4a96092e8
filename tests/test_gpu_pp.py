"""GPU parity: HIP PP-score path (through the C ABI) vs the oracle and the
reference-generated fixtures.  Counts are integers -> bit exact; H within 1e-6."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    g = np.load(f"{golden_dir}/{name}.npz")
    return g, g["offsets"]


@pytest.mark.parametrize("name", ["pp_lyft", "pp_nusc"])
def test_pp_counts_and_entropy_golden(gpu, golden_dir, name):
    import torch
    from modest_amd import ops
    g, off = _load(golden_dir, name)
    live = torch.from_numpy(g["live"]).to(gpu)
    hist = torch.from_numpy(g["hist"]).to(gpu)
    counts = ops.pp_count(live, hist, off, 0.3)
    assert np.array_equal(counts.cpu().numpy().astype(np.int64), g["count"])
    H = ops.pp_entropy(counts).cpu().numpy()
    assert H.dtype == np.float32
    assert np.max(np.abs(H.astype(np.float64) - g["H"])) <= 1e-6
    H2, c2 = ops.pp_score(live, hist, off, 0.3, return_counts=True)
    assert np.array_equal(c2.cpu().numpy(), counts.cpu().numpy())
    assert np.array_equal(H2.cpu().numpy(), H)
    H3 = ops.pp_score(live, hist, off, 0.3)
    assert np.array_equal(H3.cpu().numpy(), H)


def test_pp_vs_oracle_seeded(gpu):
    """Mid-size seeded scene: oracle (scipy cKDTree, the reference's calls) vs HIP."""
    import torch
    from modest_amd import ops, synth
    from oracle import pp_score as opp
    s = synth.make_scan(11, n_live=20000, n_trav=5, n_frames=8, n_per_frame=20000)
    Href, cref = opp.pp_score(s.live_xyz, s.hist, 0.3, workers=-1)
    off = np.cumsum([0] + [len(h) for h in s.hist])
    live = torch.from_numpy(s.live_xyz).to(gpu)
    hist = torch.from_numpy(np.concatenate(s.hist)).to(gpu)
    H, c = ops.pp_score(live, hist, off, 0.3, return_counts=True)
    assert np.array_equal(c.cpu().numpy().astype(np.int64), cref)
    assert np.max(np.abs(H.cpu().numpy().astype(np.float64) - Href.astype(np.float64))) <= 1e-6


def test_pp_edge_cases(gpu):
    import torch
    from modest_amd import ops
    from oracle import pp_score as opp
    rng = np.random.default_rng(5)
    live = (rng.standard_normal((257, 3)) * [3, 3, 0.5]).astype(np.float32)
    # unaligned history start, empty traversal, ragged sizes, points exactly on the radius
    h0 = (rng.standard_normal((1001, 3)) * [3, 3, 0.5]).astype(np.float32)
    h1 = np.zeros((0, 3), dtype=np.float32)
    h2 = live[:50] + np.array([0.3, 0, 0], dtype=np.float32)        # |d| ~ 0.3 boundary pairs
    h3 = np.repeat(live[:3], 7, axis=0)                                # duplicates of live points
    hist = [h0, h1, h2, h3]
    cref = opp.count_neighbors_bruteforce(live, hist, 0.3)
    assert np.array_equal(cref, opp.count_neighbors(live, [h0, np.full((1, 3), 1e6, np.float32), h2, h3], 0.3))
    off = np.cumsum([0] + [len(h) for h in hist])
    allh = np.concatenate(hist)
    lt = torch.from_numpy(live).to(gpu)
    c = ops.pp_count(lt, torch.from_numpy(allh).to(gpu), off, 0.3)
    assert np.array_equal(c.cpu().numpy().astype(np.int64), cref)
    # same through an unaligned view (base pointer + 12 bytes)
    pad = torch.from_numpy(np.concatenate([np.zeros((1, 3), np.float32), allh])).to(gpu)
    c2 = ops.pp_count(lt, pad[1:], off, 0.3)
    assert np.array_equal(c2.cpu().numpy().astype(np.int64), cref)
    # far-away live points are clamped into border cells, still exact
    far = live.copy()
    far[:5] += np.float32(5000.0)
    cref3 = opp.count_neighbors_bruteforce(far, hist, 0.3)
    c3 = ops.pp_count(torch.from_numpy(far).to(gpu), torch.from_numpy(allh).to(gpu), off, 0.3)
    assert np.array_equal(c3.cpu().numpy().astype(np.int64), cref3)
    # empty live scan / empty history
    e = ops.pp_count(lt[:0], torch.from_numpy(allh).to(gpu), off, 0.3)
    assert e.shape == (0, 4)
    z = ops.pp_count(lt, torch.zeros((0, 3), device=gpu), [0, 0, 0], 0.3)
    assert int(z.abs().sum()) == 0
    Hz = ops.pp_entropy(z).cpu().numpy()
    assert np.all(Hz == 0.0)


def test_pp_linearity_full_size(gpu):
    """Size-independent property at BASELINE config-2 scale: counting against
    the union of two histories equals the sum of the counts (per traversal)."""
    import torch
    from modest_amd import ops
    g = torch.Generator(device="cpu").manual_seed(0)
    n, m = 30000, 2_000_000
    live = (torch.randn((n, 3), generator=g) * torch.tensor([20.0, 20.0, 0.3])).to(gpu)
    ha = (torch.randn((m, 3), generator=g) * torch.tensor([20.0, 20.0, 0.3])).to(gpu)
    hb = (torch.randn((m, 3), generator=g) * torch.tensor([20.0, 20.0, 0.3])).to(gpu)
    ca = ops.pp_count(live, ha, [0, m], 0.3)
    cb = ops.pp_count(live, hb, [0, m], 0.3)
    cab = ops.pp_count(live, torch.cat([ha, hb]), [0, 2 * m], 0.3)
    assert torch.equal(ca + cb, cab)
    two = ops.pp_count(live, torch.cat([ha, hb]), [0, m, 2 * m], 0.3)
    assert torch.equal(two[:, 0], ca[:, 0]) and torch.equal(two[:, 1], cb[:, 0])
    assert int(cab.sum()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("T", [17, 32, 33, 40, 64, 65])
def test_pp_many_traversals(gpu, T):
    """Up to 64 traversals use the routed path (one lane per traversal in the segmented popcount,
    a sixth traversal bit beyond 32), 65 the direct path; ragged traversal sizes, some of them empty."""
    import torch
    from modest_amd import ops
    from oracle import pp_score as opp
    rng = np.random.default_rng(100 + T)
    live = (rng.standard_normal((3000, 3)) * [6, 6, 0.4]).astype(np.float32)
    hist = []
    for t in range(T):
        m = 0 if t % 7 == 3 else int(rng.integers(500, 6000))
        hist.append((rng.standard_normal((m, 3)) * [6, 6, 0.4]).astype(np.float32))
    cref = opp.count_neighbors_bruteforce(live, hist, 0.3)
    off = np.cumsum([0] + [len(h) for h in hist])
    H, c = ops.pp_score(torch.from_numpy(live).to(gpu), torch.from_numpy(np.concatenate(hist)).to(gpu), off, 0.3,
                        return_counts=True)
    assert np.array_equal(c.cpu().numpy().astype(np.int64), cref)
    Href = opp.compute_ephe_score(cref)
    assert np.max(np.abs(H.cpu().numpy().astype(np.float64) - Href)) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("spread", [0.25, 1.0, 4.0])
def test_pp_dense_clusters(gpu, spread):
    """Live scans far denser than LiDAR data: thousands of live points in a few cells force the
    row-band split of a block (spread 1.0), the per-record fallback for bands that do not fit
    the LDS even as a single row (spread 0.25), and dense-block quadrant lists (spread 4.0)."""
    import torch
    from modest_amd import ops
    from oracle import pp_score as opp
    rng = np.random.default_rng(int(spread * 100))
    live = np.concatenate([(rng.standard_normal((9000, 3)) * [spread, spread, 0.3]),
                           (rng.standard_normal((1000, 3)) * [15, 15, 0.3])]).astype(np.float32)
    hist = [(rng.standard_normal((20000, 3)) * [spread * 1.5, spread * 1.5, 0.3]).astype(np.float32),
            (rng.standard_normal((15000, 3)) * [10, 10, 0.3]).astype(np.float32),
            (rng.standard_normal((9000, 3)) * [spread, spread, 0.3] + [0.2, 0.1, 0.0]).astype(np.float32)]
    cref = opp.count_neighbors(live, hist, 0.3)
    off = np.cumsum([0] + [len(h) for h in hist])
    c = ops.pp_count(torch.from_numpy(live).to(gpu), torch.from_numpy(np.concatenate(hist)).to(gpu), off, 0.3)
    assert np.array_equal(c.cpu().numpy().astype(np.int64), cref)


@pytest.mark.gpu
def test_pp_permutation_invariance_full_size(gpu):
    """Size-independent properties at BASELINE config-2 scale (30 k live points, 10 x 1.08 M
    history points): counts do not depend on the order of the history points inside a traversal
    (the scatter pass fills lists in arrival order), they follow a permutation of the live points,
    and swapping two traversals swaps the two count columns."""
    import torch
    from modest_amd import ops, synth
    s = synth.make_scan(5, n_live=30000, n_trav=10, n_frames=36)
    off = np.cumsum([0] + [len(h) for h in s.hist])
    live = torch.from_numpy(s.live_xyz).to(gpu)
    hist = torch.from_numpy(np.concatenate(s.hist)).to(gpu)
    base = ops.pp_count(live, hist, off, 0.3)
    assert int(base.sum()) > 0
    g = torch.Generator(device="cpu").manual_seed(1)
    # (1) shuffle the history inside every traversal
    parts = []
    for t in range(10):
        seg = hist[off[t]:off[t + 1]]
        parts.append(seg[torch.randperm(seg.shape[0], generator=g).to(gpu)])
    assert torch.equal(ops.pp_count(live, torch.cat(parts), off, 0.3), base)
    # (2) permute the live points
    perm = torch.randperm(live.shape[0], generator=g).to(gpu)
    assert torch.equal(ops.pp_count(live[perm].contiguous(), hist, off, 0.3), base[perm])
    # (3) swap traversals 2 and 7
    order = [0, 1, 7, 3, 4, 5, 6, 2, 8, 9]
    hs = torch.cat([hist[off[t]:off[t + 1]] for t in order])
    offs = np.cumsum([0] + [int(off[t + 1] - off[t]) for t in order])
    assert torch.equal(ops.pp_count(live, hs, offs, 0.3), base[:, order])


@pytest.mark.gpu
@pytest.mark.parametrize("radius", [0.05, 0.5, 1.25])
def test_pp_other_radii(gpu, radius):
    """max_neighbor_dist other than the config default (the grid cell follows the radius)."""
    import torch
    from modest_amd import ops
    from oracle import pp_score as opp
    rng = np.random.default_rng(int(radius * 100))
    live = (rng.standard_normal((4000, 3)) * [10, 10, 0.5]).astype(np.float32)
    hist = [(rng.standard_normal((m, 3)) * [10, 10, 0.5]).astype(np.float32) for m in (30000, 1, 12345)]
    cref = opp.count_neighbors(live, hist, radius)
    off = np.cumsum([0] + [len(h) for h in hist])
    c = ops.pp_count(torch.from_numpy(live).to(gpu), torch.from_numpy(np.concatenate(hist)).to(gpu), off, radius)
    assert np.array_equal(c.cpu().numpy().astype(np.int64), cref)
