"""Host wall time of the stages of one chain on an idle GPU: mask stage (modest_mask_stage_batch), box tail
(modest_scan_boxes_batch), label stage -- for chains of B scans (bench.py's Runner, one process, resident scans)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
a = bench.parse(["--procs", "1", "--cpu-scans", "0", "--cli-scans", "0", "--scans", "16"])
r = bench.Runner(a, 0, 0, 0)
for B in (int(x) for x in (sys.argv[1:] or ["4", "10", "16"])):
    js = list(range(B))
    Hs = r.pp_many([r.scan_of(j) for j in js], 0)
    torch.cuda.synchronize()
    for rep in range(6):
        r.trace = []
        t0 = time.perf_counter()
        out = r.steps(js, 0, Hs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tr, r.trace = r.trace, None
    print(f"chain of {B}: {dt * 1e3:.2f} ms: " + " ".join(f"{k} {v * 1e3:.2f}" for k, v in tr), flush=True)
    # inside the mask part: stage vs boxes
    from modest_amd import generate_mask as gm, ops
    scs = [r.scan_of(j) for j in js]
    items = [(sc.live_raw, H, np.random.RandomState(i)) for i, (sc, H) in enumerate(zip(scs, Hs))]
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = ops.mask_stage_batch(items, gm._stage_params(r.margs), ctxs=r.chain_ctxs[0][:B])
        t1 = time.perf_counter()
    print(f"   mask_stage_batch alone: {(t1 - t0) * 1e3:.2f} ms", flush=True)
if os.environ.get("CHAIN_PROFILE"):
    import cProfile, pstats
    B = 16
    js = list(range(B))
    Hs = r.pp_many([r.scan_of(j) for j in js], 0)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for rep in range(10):
        out = r.steps(js, 0, Hs)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
