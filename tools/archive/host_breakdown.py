"""Wall-clock breakdown of one full-pipeline step, single thread, no profiler (GPU box)."""
import sys, os, time, tempfile, collections
import numpy as np, torch
sys.path.insert(0, ".")
from modest_amd import config, ops, synth
from modest_amd import generate_mask as gm
from modest_amd import gen_label_files as gl
from modest_amd.utils import kitti_util, clustering_utils as cu, pointcloud_utils as pu

acc = collections.defaultdict(float)
def timed(mod, name, label=None):
    f = getattr(mod, name)
    def w(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[label or name] += time.perf_counter() - t; return r
    setattr(mod, name, w)
for mod, name in ((gm, "estimate_plane"), (gm, "filter_labels"), (gm, "get_objs"), (gm, "members_by_label"),
                  (gm, "compact_labels"), (ops, "plane_range_mask"), (ops, "cluster_dbscan"), (ops, "cluster_stats"),
                  (gl, "objs_nms"), (gl, "objs2label"), (gl, "is_within_fov")):
    timed(mod, name)
dev = torch.device("cuda:0")
with tempfile.TemporaryDirectory() as d:
    open(os.path.join(d, "c.txt"), "w").write(synth.CALIB_TXT)
    calib = kitti_util.Calibration(os.path.join(d, "c.txt"))
_p = calib.project_velo_to_rect
def pvr(x):
    t = time.perf_counter(); r = _p(x); acc["project_velo_to_rect"] += time.perf_counter() - t; return r
calib.project_velo_to_rect = pvr
margs = config.compose("generate_mask", ["data_root=/unused"]); largs = config.compose("generate_label_files", ["data_root=/unused"])
s = synth.make_scan(0, n_live=30000, n_trav=10, n_frames=4)
off = np.cumsum([0] + [len(h) for h in s.hist])
live_raw = torch.from_numpy(s.live_raw).to(dev); live = torch.from_numpy(s.live_xyz).to(dev)
hist = torch.from_numpy(np.concatenate(s.hist)).to(dev)
def step(i):
    t = time.perf_counter(); H = ops.pp_score(live, hist, off, 0.3); pp = H.cpu().numpy(); acc["pp_score+cpu"] += time.perf_counter() - t
    t = time.perf_counter()
    labels, objs, _ = gm.generate_mask_scan(s.live_raw, pp, calib, margs, random_state=np.random.RandomState(i), ptc_dev=live_raw, pp_dev=H)
    acc["generate_mask_scan(total)"] += time.perf_counter() - t
    t = time.perf_counter(); r = gl.gen_label_scan(objs, calib, largs); acc["gen_label_scan(total)"] += time.perf_counter() - t
    return len(objs)
for i in range(5): step(i)
acc.clear(); K = 40; t0 = time.perf_counter(); no = 0
for i in range(K): no += step(i)
tot = time.perf_counter() - t0
print("step %.3f ms, %.1f objs/scan" % (tot / K * 1e3, no / K))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]): print("  %-28s %.3f ms" % (k, v / K * 1e3))
