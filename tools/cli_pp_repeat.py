"""GPU box: the PP CLI with workers=N on ONE synthetic tree, repeated: the rank's summary line of every run (looks for
intermittent slow runs)."""
import os
import shutil
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modest_amd import synth  # noqa: E402

n_scan, W, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
F, T = 36, 10
with tempfile.TemporaryDirectory(dir="/dev/shm") as root:
    paths = synth.write_kitti_tree(os.path.join(root, "kitti"), os.path.join(root, "meta"), n_seq=T + 1, n_frames=n_scan + F,
                                   n_pts=30000, origins=tuple(range(n_scan)), hist_frames=F, max_range=80.0)
    for rep in range(reps):
        shutil.rmtree(f"{root}/pp", ignore_errors=True)
        cmd = [sys.executable, "-m", "modest_amd.pre_compute_pp_score", f"data_root={root}/kitti/training",
               f"data_paths.track_path={paths['track_path']}", f"data_paths.idx_info={paths['idx_info']}",
               f"data_paths.idx_list={paths['idx_list']}", f"data_paths.pp_score_path={root}/pp", f"workers={W}"] + sys.argv[4:]
        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        lines = [l[:260] for l in r.stderr.splitlines() if l.startswith("[pp_score]")]
        print("run %d: %.1f s wall | %s" % (rep, time.time() - t0, lines[-1] if lines else r.stderr[-300:]), flush=True)
