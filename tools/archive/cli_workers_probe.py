"""GPU box: the PP CLI with workers=N on a synthetic tree, every worker's own summary line."""
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modest_amd import synth  # noqa: E402

n_scan, W = int(sys.argv[1]), int(sys.argv[2])
F, T = 36, 10
with tempfile.TemporaryDirectory(dir="/dev/shm") as root:
    paths = synth.write_kitti_tree(os.path.join(root, "kitti"), os.path.join(root, "meta"), n_seq=T + 1, n_frames=n_scan + F,
                                   n_pts=30000, origins=tuple(range(n_scan)), hist_frames=F, max_range=80.0)
    cmd = [sys.executable, "-m", "modest_amd.pre_compute_pp_score", f"data_root={root}/kitti/training",
           f"data_paths.track_path={paths['track_path']}", f"data_paths.idx_info={paths['idx_info']}",
           f"data_paths.idx_list={paths['idx_list']}", f"data_paths.pp_score_path={root}/pp", f"workers={W}"] + sys.argv[3:]
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, MODEST_PP_TRACE=os.environ.get("TRACE_LEVEL", "1")), cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    print("\n".join(l[:330] for l in r.stderr.splitlines() if l.startswith("[pp_score")))
    if r.returncode:
        print(r.stderr[-2000:])
