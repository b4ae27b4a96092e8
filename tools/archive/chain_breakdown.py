"""GPU box, one process, one stream: wall time of a block of scans by part (no profiler; wrappers with perf_counter).
    python tools/chain_breakdown.py [blocks]
Parts: PP call (host side of the enqueue), the 16 score read-backs, the mask stage library call, the box tail, the
generate_mask_scan glue, the IoU launch, label text; the rest is the bench's own loop."""
import collections, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from modest_amd import ops, generate_mask as gm, gen_label_files as gl

acc, cnt = collections.defaultdict(float), collections.defaultdict(int)
def wrap(mod, name, label=None):
    f = getattr(mod, name)
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label or name] += time.perf_counter() - t; cnt[label or name] += 1
    setattr(mod, name, w)
    return w
a = bench.parse(["--procs", "1", "--streams", "1", "--cpu-scans", "0", "--cli-scans", "0"] + sys.argv[2:])
r = bench.Runner(a, 0, 0, 0)
for mod, name in ((ops, "mask_stage_batch"), (ops, "scan_boxes_batch"), (ops, "objs_iou_batch"), (ops, "label_lines")):
    wrap(mod, name)
r._generate_mask_chain = wrap(gm, "generate_mask_chain")
r._gen_label_chain = wrap(gl, "gen_label_chain")
pm = r.pp_many
def pp_many(scs, w):
    t = time.perf_counter(); out = pm(scs, w); acc["pp_many(host enqueue)"] += time.perf_counter() - t; cnt["pp_many(host enqueue)"] += 1; return out
r.pp_many = pp_many
_cpu = torch.Tensor.cpu
def cpu(self, *x, **k):
    t = time.perf_counter(); o = _cpu(self, *x, **k); acc["Tensor.cpu (score read-backs)"] += time.perf_counter() - t; cnt["Tensor.cpu (score read-backs)"] += 1; return o
torch.Tensor.cpu = cpu
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = nb * r.PB
r.run(r.n_warm, r.n_warm + n); torch.cuda.synchronize()
acc.clear(); cnt.clear()
t0 = time.perf_counter(); r.run(r.n_warm, r.n_warm + n); torch.cuda.synchronize(); tot = time.perf_counter() - t0
print("%d scans, %.3f ms per scan (%.0f scans/s), blocks of %d, mask chains of %d" % (n, 1e3 * tot / n, n / tot, r.PB, r.MB))
inner = {"mask_stage_batch", "scan_boxes_batch"}
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-34s %8.3f ms/scan  (%d calls, %.3f ms each)%s" % (k, 1e3 * v / n, cnt[k], 1e3 * v / max(cnt[k], 1), "   [inside generate_mask_chain]" if k in inner else ("   [inside gen_label_chain]" if k in ("objs_iou_batch", "label_lines") else "")))
