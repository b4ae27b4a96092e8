"""Where the ingest-inclusive step of bench.py spends its time (one process): host seconds of _ingest per block, split by part."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
a = bench.parse(["--procs", "1", "--cpu-scans", "0", "--cli-scans", "0"])
r = bench.Runner(a, 0, 0, 0)
from modest_amd.pre_compute_pp_score import relative_poses
scs = r.scans[:16]
ctx = r.ctxs[0]
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r._ingest(scs, ctx)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"_ingest host {1e3*(t1-t0):.2f} ms, device tail {1e3*(t2-t1):.2f} ms")
# parts
sh = scs[0].shard
t0 = time.perf_counter()
for sc in scs: relative_poses(sc.fixed_l2e, sc.fixed_ego, sc.W_stack, sc.K)
t1 = time.perf_counter()
for sc in sh.scans: sc.describe()
t2 = time.perf_counter()
dev = torch.empty((sum(int(sc.new_pinned.shape[0]) for sc in scs), 4), dtype=torch.float32, device=r.dev)
torch.cuda.synchronize(); t3 = time.perf_counter()
base = 0
for sc in scs:
    n = int(sc.new_pinned.shape[0]); dev[base:base+n].copy_(sc.new_pinned, non_blocking=True); base += n
t4 = time.perf_counter(); torch.cuda.synchronize(); t5 = time.perf_counter()
print(f"16 x relative_poses {1e3*(t1-t0):.2f} ms | 16 x describe {1e3*(t2-t1):.2f} ms | 16 copies enqueue {1e3*(t4-t3):.2f} ms, done {1e3*(t5-t3):.2f} ms ({base*16/1e6:.0f} MB)")
tabs0 = time.perf_counter()
tb = r.store.block_tables([sc.desc for sc in scs], scs[0].T)
print(f"block_tables {1e3*(time.perf_counter()-tabs0):.2f} ms", tb is not None)
t0 = time.perf_counter(); r.store.pp_score_batch([sc.live_key for sc in scs], [sc.desc for sc in scs], scs[0].T, ctx=ctx); t1 = time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print(f"pp_score_batch host {1e3*(t1-t0):.2f} ms, device tail {1e3*(t2-t1):.2f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    r.store.block_tables([sc.desc for sc in scs], scs[0].T)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
