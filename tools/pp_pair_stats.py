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

def fill_and_pairing():
    """Round 5: chunk fill of the one-cell tasks (what half-empty chunk pairs cost, what single-chunk steps for one- and
    three-chunk remainders save) and what two horizontally neighbouring sparse cells per chunk would save in four-cell chunk steps."""
    sc = synth.make_scan(0, n_live=30000, n_trav=10, n_frames=36)
    r=0.3; c=r*(1+1/256)
    live=sc.live_xyz.astype(np.float64)
    o=np.floor(live[:,:2].min(0)/c)-2
    o = o - (o % 8)   # tile aligned
    lc=np.floor(live[:,:2]/c)-o
    W=int(lc[:,0].max())+12; W+= (-W)%8; Hh=int(lc[:,1].max())+12; Hh+=(-Hh)%8
    lcell=(lc[:,1]*W+lc[:,0]).astype(np.int64)
    lcount=np.bincount(lcell,minlength=W*Hh).reshape(Hh,W)
    pad=np.pad(lcount,1)
    cand=sum(pad[1+dy:1+dy+Hh,1+dx:1+dx+W] for dy in (-1,0,1) for dx in (-1,0,1))
    nc=np.zeros((Hh,W),dtype=np.int64)
    for t in range(10):
        h=sc.hist[t].astype(np.float64); hc=np.floor(h[:,:2]/c)-o
        ok=(hc[:,0]>=0)&(hc[:,0]<W)&(hc[:,1]>=0)&(hc[:,1]<Hh)
        nc+=np.bincount((hc[ok,1]*W+hc[ok,0]).astype(np.int64),minlength=W*Hh).reshape(Hh,W)
    act=(cand>0)&(nc>0)
    light=act&(nc<64); heavy=act&(nc>=64)
    print("light cells",light.sum(),"heavy",heavy.sum(),"light records mean",nc[light].mean())
    # steps now: light: cand per cell (one chunk-step each, 13 VALU); heavy: ceil(n/256) tasks * cand * (37 or 19)
    steps_light=cand[light].sum()
    print("light chunk-steps",steps_light, "heavy steps (4-chunk)", (np.ceil(nc[heavy]/256)*cand[heavy]).sum())
    # pairing (even, odd) horizontally
    L=light.copy(); n=nc
    even=L[:,0::2]; odd=L[:,1::2]
    pair=even&odd&((n[:,0::2]+n[:,1::2])<=64)
    # candidates for a pair: 4 cells wide
    pad2=np.pad(lcount,((1,1),(1,2)))
    cand4=sum(pad2[1+dy:1+dy+Hh, 0+dx:0+dx+W] for dy in (-1,0,1) for dx in (0,1,2,3))  # for even cx: cells cx-1..cx+2 -> pad offset
    c4=cand4[:,0::2]
    steps_pair=c4[pair].sum()
    unp_even=even&~pair; unp_odd=odd&~pair
    steps_new=steps_pair+cand[:,0::2][unp_even].sum()+cand[:,1::2][unp_odd].sum()
    print("pairs",pair.sum(),"unpaired",unp_even.sum()+unp_odd.sum(),"light steps now",steps_light,"new",steps_new, "ratio",steps_new/steps_light)
    # quads: 4 cells in a row (aligned), total<=64, all light or empty-inactive

    # ---- chunk fill of heavy cells
    H = heavy
    n = nc[H].astype(np.int64); c = cand[H].astype(np.int64)
    full_tasks = n // 256; rem = n % 256
    rem_chunks = (rem + 63) // 64
    cost = full_tasks * 37 * c + np.where(rem_chunks == 0, 0, np.where(rem_chunks <= 2, 19, 37)) * c
    ideal = n * c * 37.0 / 256.0
    print("heavy: VALU(pair phase) now %.2f M, ideal (all lanes full) %.2f M, efficiency %.3f" % (cost.sum()/1e6, ideal.sum()/1e6, ideal.sum()/cost.sum()))
    # alternative: 3-chunk remainders as NP=1 + one single-chunk step (13)
    alt = full_tasks * 37 * c + np.select([rem_chunks == 0, rem_chunks == 1, rem_chunks == 2, rem_chunks == 3], [0, 13, 19, 19 + 13], 37) * c
    print("with single-chunk steps for 1- and 3-chunk remainders: %.2f M" % (alt.sum()/1e6))
    for lo, hi in ((64,128),(128,192),(192,256),(256,512),(512,1024),(1024,10**9)):
        m=(n>=lo)&(n<hi); print(lo,hi,"cells",m.sum(),"cost share %.3f"%(cost[m].sum()/cost.sum()))
    print("one-cell tasks (members only):", int((full_tasks + (rem > 0)).sum()), "heavy cells", int(H.sum()), "records in heavy cells", int(n.sum()), "in light", int(nc[light].sum()))
    print("candidate steps (members only):", int(((full_tasks + (rem > 0)) * c).sum()))


if __name__ == "__main__":
    main()
    fill_and_pairing()
