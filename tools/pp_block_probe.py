"""Block path (modest_pp_score_block) against the per-scan chain and the oracle on a synthetic shard; timing.
    python tools/pp_block_probe.py [--scans 8] [--n 30000] [--trav 10] [--frames 36] [--oracle] [--reps 5]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modest_amd import _lib, synth   # noqa: E402
from modest_amd.frame_store import FrameStore   # noqa: E402


def load_shard(store, sh, dev, key0=0):
    items, nf = [], {}
    for t, tr in enumerate(sh.tracks):
        for j, (raw, W) in enumerate(tr):
            nf[(t, j)] = key0 + len(nf)
            items.append((nf[(t, j)], torch.from_numpy(raw).to(dev), W))
    lives = []
    for sc in sh.scans:
        k = key0 + 100000 + sc.index
        items.append((k, torch.from_numpy(sc.live_raw).to(dev), sc.live_W))
        lives.append(k)
    store.insert_many(items)
    descs = [store.describe(lives[i], sc.live_rel, [nf[h] for h in sc.hist], sc.trav_list(), sc.rels, sh.nusc)
             for i, sc in enumerate(sh.scans)]
    return lives, descs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=8)
    ap.add_argument("--n", type=int, default=30000)
    ap.add_argument("--trav", type=int, default=10)
    ap.add_argument("--frames", type=int, default=36)
    ap.add_argument("--nusc", action="store_true")
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--shards", type=int, default=2, help="distinct shards cycled through in the timing loop")
    ap.add_argument("--auto", action="store_true", help="also time the store's own choice (block=None: the split rule for large blocks)")
    ap.add_argument("--halves", action="store_true", help="also time every shard as TWO forced blocks of half the scans")
    ap.add_argument("--stream", action="store_true", help="run on a created stream instead of the null stream")
    ap.add_argument("--json-out", type=str, default="", help="append one JSON line with the run's numbers to this file")
    ap.add_argument("--matched", type=str, default="", help="live_speed,hist_lo,hist_hi: windows chosen by the reference's rule "
                    "(synth.make_shard_matched, split_traintest.py:79-101) instead of frames i..i+F-1")
    ap.add_argument("--presence", type=int, default=-1, help="seed: --trav + 3 tracks that enter and leave along the shard (synth.presence_ramp: the "
                    "reference accepts a traversal per scan, split_traintest.py:17,79) -- T changes inside the block, as in bench.py's realistic shard")
    a = ap.parse_args()
    _lib.load()
    dev = torch.device("cuda:0")
    store = FrameStore(dev, 0.3)
    ctx = _lib.Context(0)
    shards, tabs = [], []
    for q in range(a.shards):
        t0 = time.time()
        if a.matched:
            ls, h0, h1 = (float(x) for x in a.matched.split(","))
            pres = synth.presence_ramp(a.scans, a.trav + 3, t_min=max(2, a.trav - 3), seed=a.presence) if a.presence >= 0 else None
            sh = synth.make_shard_matched(a.scans, n_live=a.n, n_trav=a.trav + (3 if pres else 0), nusc=a.nusc, live_speed=ls, hist_speeds=(h0, h1),
                                          seed=q, x0=40.0 * q, presence=pres)
            print("sharing:", {k: round(v, 2) for k, v in synth.sharing_stats(sh, a.scans).items()}, flush=True)
        else:
            sh = synth.make_shard(a.scans, n_live=a.n, n_trav=a.trav, n_frames=a.frames, nusc=a.nusc, seed=q, x0=40.0 * q)
        lives, descs = load_shard(store, sh, dev, key0=1000000 * q)
        shards.append(sh)
        tabs.append((lives, descs))
        print(f"shard {q}: generated + loaded in {time.time() - t0:.1f} s", flush=True)
    ok = True
    Ts = [[sc.n_trav for sc in sh.scans] for sh in shards]   # (per scan: the scan's own number of traversals)
    for q, (lives, descs) in enumerate(tabs):
        T = Ts[q]
        Hb, cb = store.pp_score_batch(lives, descs, T, return_counts=True, ctx=ctx, block=True)
        Hv, cv = store.pp_score_batch(lives, descs, T, return_counts=True, ctx=ctx, block=False)
        torch.cuda.synchronize()
        for i in range(len(lives)):
            same = torch.equal(cb[i], cv[i])
            if not same:
                d = (cb[i] != cv[i])
                print(f"shard {q} scan {i}: counts DIFFER from the per-scan chain: {int(d.sum())} of {d.numel()} "
                      f"(block sum {int(cb[i].sum())}, chain sum {int(cv[i].sum())})")
            ok &= same
            ok &= bool(torch.equal(Hb[i], Hv[i]))
        if a.oracle:
            from oracle import pp_score as opp
            for i in (0, len(lives) - 1):
                lv, hist = shards[q].stacked(i)
                Href, cref = opp.pp_score(lv, hist, 0.3, workers=-1)
                e = np.array_equal(cb[i].cpu().numpy().astype(np.int64), cref)
                print(f"shard {q} scan {i}: block == oracle: {e}")
                ok &= e
    print("PARITY", "OK" if ok else "FAILED", "block calls", getattr(store, "block_calls", 0), flush=True)
    side = torch.cuda.Stream(device=dev) if a.stream else torch.cuda.current_stream(dev)
    report = dict(scans=a.scans, n_live=a.n, traversals=a.trav, traversals_per_scan=Ts[0], nusc=bool(a.nusc), matched=a.matched or None, parity_ok=bool(ok),
                  sharing=(synth.sharing_stats(shards[0], a.scans) if a.matched else
                           dict(members_per_scan=a.trav * a.frames, union_over_members=(a.frames + a.scans - 1) / a.frames)))
    for mode in ((True, False, None) if a.auto else (True, False)):
        c0 = getattr(store, "block_calls", 0)
        ctx.profile_begin(8 * a.reps * a.shards + 4)
        with torch.cuda.stream(side):
            for r in range(a.reps):
                for (lives, descs), T in zip(tabs, Ts):
                    store.pp_score_batch(lives, descs, T, ctx=ctx, block=mode)
        torch.cuda.synchronize()
        ms = np.asarray(ctx.profile_collect(8 * a.reps * a.shards + 4))
        calls = a.reps * a.shards
        if len(ms) > calls and len(ms) % calls == 0:   # (the per-scan chain splits a call of more than 8 scans into several marks)
            ms = ms.reshape(calls, -1).sum(axis=1)
        per = float(np.mean(ms[a.shards:])) / a.scans
        members = float(np.mean([len(sc.hist) for sh in shards for sc in sh.scans]))
        alg = 12 * members * a.n + 16 * a.n   # (a repeated frame is stacked, and counted, as often as it is listed)
        if mode is None:
            print(f"auto: {(getattr(store, 'block_calls', 0) - c0) / calls:.1f} block calls per call of {a.scans} scans", flush=True)
        print(f"{'auto' if mode is None else 'block' if mode else 'chain'}: {np.mean(ms[a.shards:]):.3f} ms per call of {a.scans} scans = {per * 1e3:.1f} us/scan "
              f"-> {alg / per / 1e6:.0f} GB/s = {alg / per / 1e6 / 8000 * 100:.1f} % of 8 TB/s", flush=True)
        tag = "auto" if mode is None else "block" if mode else "chain"
        report[tag + "_us_per_scan"] = per * 1e3
        report[tag + "_frac"] = alg / per / 1e6 / 8000
        report["algorithmic_bytes_per_scan"] = alg
    if a.halves:
        h = a.scans // 2
        ctx.profile_begin(16 * a.reps * a.shards + 4)
        with torch.cuda.stream(side):
            for r in range(a.reps):
                for (lives, descs), T in zip(tabs, Ts):
                    store.pp_score_batch(lives[:h], descs[:h], T[:h], ctx=ctx, block=True)
                    store.pp_score_batch(lives[h:], descs[h:], T[h:], ctx=ctx, block=True)
        torch.cuda.synchronize()
        ms = np.asarray(ctx.profile_collect(16 * a.reps * a.shards + 4))
        per = float(ms[2 * a.shards:].sum()) / ((a.reps - 1) * a.shards) / a.scans
        print(f"two halves: {per * 1e3:.1f} us/scan", flush=True)
        report["halves_us_per_scan"] = per * 1e3
    if a.json_out:
        import json
        with open(a.json_out, "a") as fh:
            fh.write(json.dumps(report) + "\n")


if __name__ == "__main__":
    main()
