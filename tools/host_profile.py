"""cProfile of the full-pipeline step on the GPU box (host-side hot spots)."""
import cProfile, pstats, sys, os, tempfile, io
import numpy as np, torch
sys.path.insert(0, ".")
from modest_amd import _lib, config, ops, synth
from modest_amd.gen_label_files import gen_label_scan
from modest_amd.generate_mask import generate_mask_scan
from modest_amd.utils import kitti_util

dev = torch.device("cuda:0")
with tempfile.TemporaryDirectory() as d:
    open(os.path.join(d, "c.txt"), "w").write(synth.CALIB_TXT)
    calib = kitti_util.Calibration(os.path.join(d, "c.txt"))
margs = config.compose("generate_mask", ["data_root=/unused"])
largs = config.compose("generate_label_files", ["data_root=/unused"])
s = synth.make_scan(0, n_live=30000, n_trav=10, n_frames=4)
off = np.cumsum([0] + [len(h) for h in s.hist])
live_raw = torch.from_numpy(s.live_raw).to(dev); live = torch.from_numpy(s.live_xyz).to(dev)
hist = torch.from_numpy(np.concatenate(s.hist)).to(dev)

def step(i):
    H = ops.pp_score(live, hist, off, 0.3)
    pp = H.cpu().numpy()
    labels, objs, _ = generate_mask_scan(s.live_raw, pp, calib, margs, random_state=np.random.RandomState(i), ptc_dev=live_raw, pp_dev=H)
    return gen_label_scan(objs, calib, largs)

for i in range(3): step(i)
pr = cProfile.Profile(); pr.enable()
for i in range(20): step(i)
torch.cuda.synchronize(); pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(40); print(st.getvalue()[:9000])
