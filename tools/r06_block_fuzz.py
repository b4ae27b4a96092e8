"""Randomised parity sweep of the block path (modest_pp_score_block_mixed) against the per-scan chain and, on two scans per shard, the
oracle (scipy cKDTree on the stacked history): random live / frame sizes, traversal counts, window lengths, block sizes, radii, Lyft and
nuScenes shape, sliding and reference-rule windows, traversals that enter and leave.  `python tools/r06_block_fuzz.py [cases] [seed]`"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MODEST_PP4_CHECK", "1")
from modest_amd import _lib, synth   # noqa: E402
from modest_amd.frame_store import FrameStore   # noqa: E402
from oracle import pp_score as opp   # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 6)
_lib.load()
dev = torch.device("cuda:0")
bad = 0
t00 = time.time()
for case in range(n_cases):
    S = int(rng.integers(4, 41))
    T = int(rng.integers(2, 17))
    n_live = int(rng.choice([700, 3000, 9000, 20000]))
    n_pf = int(rng.choice([500, 3000, 12000]))
    nusc = bool(rng.integers(0, 2))
    radius = float(rng.choice([0.15, 0.3, 0.3, 0.5, 0.9]))
    matched = bool(rng.integers(0, 2))
    pres = synth.presence_ramp(S, T, t_min=2, seed=case) if (T >= 4 and rng.integers(0, 2)) else None
    if matched:
        sh = synth.make_shard_matched(S, n_live=n_live, n_trav=T, n_per_frame=n_pf, nusc=nusc, live_speed=float(rng.uniform(3, 12)),
                                      hist_speeds=(3.0, float(rng.uniform(5, 15))), seed=case, presence=pres)
    else:
        sh = synth.make_shard(S, n_live=n_live, n_trav=T, n_frames=int(rng.integers(2, 25)), n_per_frame=n_pf, nusc=nusc, seed=case, presence=pres)
    store = FrameStore(dev, radius)
    items, ids = [], {}
    for t, tr in enumerate(sh.tracks):
        for j, (raw, W) in enumerate(tr):
            ids[(t, j)] = len(ids)
            items.append((ids[(t, j)], torch.from_numpy(raw).to(dev), W))
    lives = []
    for sc in sh.scans:
        items.append((10 ** 6 + sc.index, torch.from_numpy(sc.live_raw).to(dev), sc.live_W))
        lives.append(10 ** 6 + sc.index)
    store.insert_many(items)
    descs = [store.describe(lives[i], sc.live_rel, [ids[h] for h in sc.hist], sc.trav_list(), sc.rels, nusc) for i, sc in enumerate(sh.scans)]
    Ts = [sc.n_trav for sc in sh.scans]
    c0 = getattr(store, "block_calls", 0)
    Hb, cb = store.pp_score_batch(lives, descs, Ts, return_counts=True, block=True)
    took_block = getattr(store, "block_calls", 0) > c0
    Hv, cv = store.pp_score_batch(lives, descs, Ts, return_counts=True, block=False)
    torch.cuda.synchronize()
    ok = all(torch.equal(a, b) for a, b in zip(cb, cv)) and all(torch.equal(a, b) for a, b in zip(Hb, Hv))
    for i in (0, S - 1):
        lv, hist = sh.stacked(i)
        Href, cref = opp.pp_score(lv, hist, radius, workers=-1)
        ok &= bool(np.array_equal(cb[i].cpu().numpy().astype(np.int64), cref)) and float(np.max(np.abs(Hb[i].cpu().numpy().astype(np.float64) - Href))) <= 1e-6
    bad += not ok
    print(f"case {case:2d}: scans {S:2d} T {min(Ts)}..{max(Ts)} live {n_live:5d} frame {n_pf:5d} r {radius} nusc {int(nusc)} "
          f"{'rule windows' if matched else 'sliding'}{' +presence' if pres else ''} block path {'yes' if took_block else 'refused -> chain'}: "
          f"pairs {int(sum(int(c.sum()) for c in cb))} {'OK' if ok else 'MISMATCH'}", flush=True)
    del store
print(f"{n_cases} cases, {bad} mismatches, {time.time() - t00:.0f} s")
