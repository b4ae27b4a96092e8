"""Host wall time of one block call of the PP stage: FrameStore.block_tables (numpy) vs the library call's host part
(staging + launches, asynchronous) -- for blocks of B scans (bench.py's Runner, one process, resident scans)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
a = bench.parse(["--procs", "1", "--cpu-scans", "0", "--cli-scans", "0", "--scans", "16"])
r = bench.Runner(a, 0, 0, 0)
store = r.store
for B in (int(x) for x in (sys.argv[1:] or ["4", "7", "16"])):
    scs = [r.scan_of(j) for j in range(B)]
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        Hs = r.pp_many(scs, 0)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    descs = [sc.desc for sc in scs]
    tb = None
    if all(d is not None for d in descs):
        for rep in range(5):
            t3 = time.perf_counter()
            tabs = store.block_tables(descs, scs[0].T)
            tb = time.perf_counter() - t3
    print(f"block of {B}: pp_many host {1e3 * (t1 - t0):.2f} ms, + device tail {1e3 * (t2 - t1):.2f} ms; "
          f"describe+block_tables alone {('%.2f ms' % (1e3 * tb)) if tb is not None else 'n/a'}", flush=True)
if os.environ.get("PP_HOST_PROFILE"):
    import cProfile, pstats
    scs = [r.scan_of(j) for j in range(4)]
    pr = cProfile.Profile()
    pr.enable()
    for rep in range(50):
        r.pp_many(scs, 0)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)
