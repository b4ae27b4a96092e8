"""Pair-test statistics of one C3-shaped scan on the CPU (no GPU): how many (64-record chunk x candidate) tests the join
runs under different record orders, and what share of the pair tests are hits.  Diagnostics only."""
import sys, json
import numpy as np
sys.path.insert(0, ".")
from modest_amd import synth

def main(n_live=30000, T=10, F=36, nusc=False):
    sc = synth.make_scan(0, n_live=n_live, n_trav=T, n_frames=F, nusc=nusc)
    r = 0.3
    c = r * (1 + 1 / 256)
    live = sc.live_xyz.astype(np.float64)
    o = np.floor(live[:, :2].min(0) / c) - 2
    lc = np.floor(live[:, :2] / c) - o
    W = int(lc[:, 0].max()) + 4
    Hh = int(lc[:, 1].max()) + 4
    lcell = (lc[:, 1] * W + lc[:, 0]).astype(np.int64)
    lcount = np.bincount(lcell, minlength=W * Hh).reshape(Hh, W)
    # candidates of a cell: live points in the 3x3 cells around it
    pad = np.pad(lcount, 1)
    cand = sum(pad[1 + dy:1 + dy + Hh, 1 + dx:1 + dx + W] for dy in (-1, 0, 1) for dx in (-1, 0, 1))
    nct = np.zeros((T, Hh, W), dtype=np.int64)
    hits = 0
    from scipy.spatial import cKDTree
    tree = cKDTree(live)
    for t in range(T):
        h = sc.hist[t].astype(np.float64)
        hc = np.floor(h[:, :2] / c) - o
        ok = (hc[:, 0] >= 0) & (hc[:, 0] < W) & (hc[:, 1] >= 0) & (hc[:, 1] < Hh)
        cell = (hc[ok, 1] * W + hc[ok, 0]).astype(np.int64)
        nct[t] = np.bincount(cell, minlength=W * Hh).reshape(Hh, W)
        hsel = h[ok][cand.reshape(-1)[cell] > 0]
        hits += int(tree.query_ball_point(hsel, r, return_length=True, workers=-1).sum())
    nc = nct.sum(0)
    act = cand > 0
    pairs = int((nc * cand)[act].sum())
    heavy = act & (nc >= 64)
    light = act & (nc < 64)
    res = {
        "records_near_live": int(nc[act].sum()), "active_cells": int(act.sum()), "pairs_tested": pairs, "hits": hits,
        "hit_rate": hits / pairs,
        "heavy_cells": int(heavy.sum()), "light_cells": int(light.sum()),
        "pairs_heavy": int((nc * cand)[heavy].sum()), "pairs_light": int((nc * cand)[light].sum()),
        # chunk tests (64 records x 1 candidate)
        "chunktests_now_heavy": int((((nc + 63) // 64) * cand)[heavy].sum()),
        "chunktests_travuniform_heavy": int(sum((((nct[t] + 63) // 64) * cand)[heavy].sum() for t in range(T))),
        "chunktests_travuniform_all": int(sum((((nct[t] + 63) // 64) * cand)[act].sum() for t in range(T))),
    }
    # heavy threshold variants for the traversal-uniform order: a (cell, traversal) run with >= thr records gets chunks of its own,
    # the rest is packed (lane walks its candidates)
    for thr in (16, 24, 32, 48):
        big = (nct >= thr) & act[None]
        res[f"tu_thr{thr}_chunktests"] = int(sum((((nct[t] + 63) // 64) * cand)[big[t]].sum() for t in range(T)))
        res[f"tu_thr{thr}_lane_pairs"] = int(sum((nct[t] * cand)[(~big[t]) & act].sum() for t in range(T)))
    # z culling: share of pairs with |dz| <= r (upper bound of what slabs could keep)
    print(json.dumps(res, indent=1))
    q = np.percentile(nc[act], [10, 50, 90, 99])
    print("records per active cell p10/50/90/99", q, "cand per active cell", np.percentile(cand[act], [10, 50, 90, 99]))
    w = (nc * cand)[act]
    order = np.argsort(nc[act])
    cw = np.cumsum(w[order]) / w.sum()
    for thr in (64, 128, 256, 640, 1280, 2560):
        k = np.searchsorted(nc[act][order], thr)
        print(f"share of pair tests in cells with < {thr} records: {cw[k - 1] if k else 0:.3f}")

if __name__ == "__main__":
    main()
