"""GPU box: latency of the mask stage of a chain of scans, RANSAC trial loops on the device against the host loop."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modest_amd import config, generate_mask as gm, ops, synth
dev = torch.device("cuda:0")
args = config.compose("generate_mask", ["data_root=/unused"])
params = gm._stage_params(args)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
scans = []
for k in range(B):
    raw = np.ascontiguousarray(synth.make_scan(500 + k, n_live=30000, n_trav=2, n_frames=1).live_raw)
    pp = np.clip(0.5 + 0.5 * np.sin(raw[:, 0] * 0.3), 0, 1).astype(np.float32)
    scans.append((torch.from_numpy(raw).to(dev), torch.from_numpy(pp).to(dev)))
for mode in ("host", "device", "host", "device"):
    if mode == "host": os.environ["MODEST_RANSAC_HOST"] = "1"
    else: os.environ.pop("MODEST_RANSAC_HOST", None)
    ts = []
    for rep in range(40):
        items = [(p, q, np.random.RandomState(rep * 16 + k)) for k, (p, q) in enumerate(scans)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = ops.mask_stage_batch(items, params)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts[5:]) * 1e3
    print(f"{mode:7s} chain of {B}: median {np.median(ts):.3f} ms  min {ts.min():.3f}  max {ts.max():.3f}  trials {[int(o[3][6]) for o in out if o is not None]}")
