"""HIP-event timings of small stages at Lyft shape (GPU box): boxes_pp_stats (combine_labels)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from modest_amd import ops

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
rect = torch.from_numpy(rng.standard_normal((30000, 3)) * [25, 1.0, 25]).to(dev)
pp = torch.from_numpy(rng.uniform(0, 1, 30000).astype(np.float32)).to(dev)
b12 = []
for _ in range(100):
    cx, cz, ry = rng.normal(0, 15), rng.normal(0, 15), rng.uniform(-3, 3)
    l, w = rng.uniform(3, 5), rng.uniform(1.5, 2.2)
    b12.append([cx, cz, np.cos(ry), -np.sin(ry), np.sin(ry), np.cos(ry), -l / 2, l / 2, -w / 2, w / 2, -0.5, 1.5])
b12 = np.array(b12)
for _ in range(3):
    ops.boxes_pp_stats(rect, pp, b12, 0.5)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    st = ops.boxes_pp_stats(rect, pp, b12, 0.5)
e1.record(); torch.cuda.synchronize()
print("boxes_pp_stats 100 boxes x 30k points: %.1f us per call (incl. H2D/D2H + sync), mean inside %.0f" % (e0.elapsed_time(e1) / 20 * 1e3, st[:, 0].mean()))
