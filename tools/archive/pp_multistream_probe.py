"""GPU box: how much of the PP chain's time is latency that other scans' kernels can fill -- the same
scans issued round-robin on S streams (one context each) from ONE host thread, HIP events around all."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from modest_amd import synth, _lib
from modest_amd.frame_store import FrameStore

dev = torch.device("cuda:0")
nscan = int(os.environ.get("NSCAN", "4"))
scans = []
for sid in range(nscan):
    s = synth.make_scan(sid, n_live=30000, n_trav=10, n_frames=36, keep_frames=True)
    st = FrameStore(dev, 0.3)
    items, hist, rels = [], [], []
    for t, fr in enumerate(s.frames):
        for f, (raw, rel, W) in enumerate(fr):
            items.append(((t, f), torch.from_numpy(raw).to(dev), W))
            hist.append(((t, f), t)); rels.append(rel)
    items.append(("live", torch.from_numpy(s.live_raw).to(dev), s.live_W))
    st.insert_many(items)
    rels = np.stack(rels)
    desc = st.describe("live", s.live_rel, [k for k, _ in hist], [t for _, t in hist], rels, False)
    scans.append((s, st, hist, rels, desc))
torch.cuda.synchronize()
M = 10 * 36 * 30000
for S in (1, 2, 3, 4, 6, 8):
    streams = [torch.cuda.Stream() for _ in range(S)]
    ctxs = [_lib.Context(0) for _ in range(S)]
    outs = [torch.empty((30000,), dtype=torch.float32, device=dev) for _ in range(S)]
    K = 48
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            s, st, hist, rels, desc = scans[k % nscan]
            with torch.cuda.stream(streams[k % S]):
                st.pp_score("live", s.live_rel, hist, rels, s.world_from_ref, 10, out=outs[k % S], ctx=ctxs[k % S], desc=desc)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(json.dumps(dict(streams=S, us_per_scan=(t2 - t0) / K * 1e6, host_issue_us=(t1 - t0) / K * 1e6,
                          frac=(12 * M + 16 * 30000) / ((t2 - t0) / K) / 8e12)), flush=True)
