"""Times the frame-store PP path on a full-size synthetic scan with HIP events (GPU box) and
checks its counts against the stacked V3 path."""
import sys, time, json, os
import numpy as np, torch
sys.path.insert(0, ".")
from modest_amd import ops, synth
from modest_amd.frame_store import FrameStore

def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 36
    nscan = int(os.environ.get("NSCAN", "3"))
    dev = torch.device("cuda:0")
    scans = []
    t0 = time.time()
    for sid in range(nscan):
        s = synth.make_scan(sid, n_live=30000, n_trav=T, n_frames=F, keep_frames=True)
        st = FrameStore(dev, 0.3)
        items, hist, rels = [], [], []
        for t, fr in enumerate(s.frames):
            for f, (raw, rel, W) in enumerate(fr):
                items.append(((t, f), torch.from_numpy(raw).to(dev), W))
                hist.append(((t, f), t)); rels.append(rel)
        items.append(("live", torch.from_numpy(s.live_raw).to(dev), s.live_W))
        torch.cuda.synchronize(); ts = time.time()
        st.insert_many(items)
        torch.cuda.synchronize(); te = time.time()
        scans.append((s, st, hist, np.stack(rels)))
    print("gen %.1fs; last insert_many of %d frames %.1f ms" % (time.time() - t0, len(items), (te - ts) * 1e3), flush=True)
    s, st, hist, rels = scans[0]
    H, c = st.pp_score("live", s.live_rel, hist, rels, s.world_from_ref, T, return_counts=True)
    off = np.cumsum([0] + [len(h) for h in s.hist])
    H3, c3 = ops.pp_score(torch.from_numpy(s.live_xyz).to(dev), torch.from_numpy(np.concatenate(s.hist)).to(dev), off, 0.3,
                          return_counts=True)
    torch.cuda.synchronize()
    print("counts equal V3:", bool(torch.equal(c, c3)), "H equal:", bool(torch.equal(H, H3)), "sum", int(c.sum()), flush=True)
    M = int(off[-1])
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        K = 12
        torch.cuda.synchronize()
        e0.record()
        for k in range(K):
            s, st, hist, rels = scans[k % nscan]
            H = st.pp_score("live", s.live_rel, hist, rels, s.world_from_ref, T)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / K
        bytes_alg = 12 * M + 16 * 30000
        print(json.dumps(dict(M=M, ms_per_scan=ms, alg_GBps=bytes_alg / ms / 1e6, frac_of_8TBps=bytes_alg / ms / 1e6 / 8000)), flush=True)

    # several scans per chain of launches (modest_pp_score_frames_batch)
    descs = []
    for (s, st, hist, rels) in scans:
        descs.append(st.describe("live", s.live_rel, [k for k, _ in hist], [t for _, t in hist], rels, False))
    if all(sc[1] is scans[0][1] for sc in scans) or True:
        pass
    for B in (1, 2, 3, 4, 6, 8):
        if B > nscan:
            break
        # one store per scan here: the batch call only needs resident frames, so borrow the first store's method
        st0 = scans[0][1]
        outs = [torch.empty((30000,), dtype=torch.float32, device=dev) for _ in range(B)]
        for i in range(B):   # the live frames of the other stores must be found by key
            st0.frames[("live", i)] = scans[i][1].frames["live"]
        keys = [("live", i) for i in range(B)]
        Hs = st0.pp_score_batch(keys, descs[:B], T, outs=outs)
        torch.cuda.synchronize()
        ok = all(bool(torch.equal(Hs[i], scans[i][1].pp_score("live", scans[i][0].live_rel, scans[i][2], scans[i][3],
                                                               scans[i][0].world_from_ref, T))) for i in range(B))
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            K = 6
            torch.cuda.synchronize()
            e0.record()
            for k in range(K):
                st0.pp_score_batch(keys, descs[:B], T, outs=outs)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / K / B
        print(json.dumps(dict(batch=B, equal_single=ok, ms_per_scan=ms, frac_of_8TBps=(12 * M + 16 * 30000) / ms / 1e6 / 8000)), flush=True)

if __name__ == "__main__":
    main()
