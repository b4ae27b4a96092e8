"""GPU box: where the host time of one scan's mask + label stage goes (cProfile over repeated scans)."""
import cProfile
import os
import pstats
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modest_amd import config, synth  # noqa: E402
from modest_amd.gen_label_files import gen_label_scan  # noqa: E402
from modest_amd.generate_mask import generate_mask_scan  # noqa: E402
from modest_amd.utils import kitti_util  # noqa: E402

dev = torch.device("cuda:0")
s = synth.make_scan(7, n_live=30000, n_trav=2, n_frames=1)
with tempfile.TemporaryDirectory() as d:
    open(os.path.join(d, "c.txt"), "w").write(synth.CALIB_TXT)
    calib = kitti_util.Calibration(os.path.join(d, "c.txt"))
margs = config.compose("generate_mask", ["data_root=/unused"])
largs = config.compose("generate_label_files", ["data_root=/unused"])
rng = np.random.default_rng(0)
pp = np.clip(0.5 + 0.5 * np.sin(s.live_raw[:, 0] * 0.3) + rng.normal(0, 0.03, len(s.live_raw)), 0, 1).astype(np.float32)
ptc_dev, pp_dev = torch.from_numpy(s.live_raw).to(dev), torch.from_numpy(pp).to(dev)


def one(i):
    labels, objs, _ = generate_mask_scan(s.live_raw, pp, calib, margs, random_state=np.random.RandomState(i),
                                         ptc_dev=ptc_dev, pp_dev=pp_dev, as_rows=True)
    return gen_label_scan(objs, calib, largs)


for i in range(5):
    one(i)
t0 = time.perf_counter()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for i in range(N):
    one(i)
print("%.3f ms per scan (mask + label stage, one process, one stream)" % ((time.perf_counter() - t0) / N * 1e3))
pr = cProfile.Profile()
pr.enable()
for i in range(N):
    one(i)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats(sys.argv[2] if len(sys.argv) > 2 else "cumulative").print_stats(55)
