"""GPU box: time the MAD-threshold kernel alone on a realistic candidate set (run under rocprofv3)."""
import numpy as np
import torch

from modest_amd import ops, synth
from modest_amd.utils.pointcloud_utils import to_device

scan = synth.make_scan(3)
pts = to_device(np.ascontiguousarray(scan.live_raw, dtype=np.float32))
cand, _ = ops.plane_candidates(pts, -1.5, ((-20, 70), (-20, 20)))
print("candidates", cand.shape[0])
for c in (cand, torch.cat([cand] * 5)[: 70000].contiguous()):      # register-resident path / > 32768 candidates
    vals = [float(ops.mad_threshold(c)) for _ in range(100)]
    z = c[:, 2].cpu().numpy()
    ref = np.median(np.abs(z - np.median(z)))
    print("n", c.shape[0], "mad", vals[0], "numpy", float(ref), "equal", vals[0] == float(ref), "stable", len(set(vals)) == 1)
