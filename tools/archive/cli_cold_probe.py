"""The cold start of ONE PP CLI process (what every worker of workers=N pays): allocation trace + per-scan submit times."""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modest_amd import synth
n_scan, F, T = int(sys.argv[1]) if len(sys.argv) > 1 else 96, 36, 10
with tempfile.TemporaryDirectory(dir="/dev/shm") as root:
    paths = synth.write_kitti_tree(os.path.join(root, "kitti"), os.path.join(root, "meta"), n_seq=T + 1, n_frames=n_scan + F,
                                   n_pts=30000, origins=tuple(range(n_scan)), hist_frames=F, max_range=80.0)
    cmd = [sys.executable, "-m", "modest_amd.pre_compute_pp_score", f"data_root={root}/kitti/training",
           f"data_paths.track_path={paths['track_path']}", f"data_paths.idx_info={paths['idx_info']}",
           f"data_paths.idx_list={paths['idx_list']}", f"data_paths.pp_score_path={root}/pp"] + sys.argv[2:]
    env = dict(os.environ, MODEST_PP_TRACE="1", MODEST_ALLOC_TRACE="1")
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    lines = [l for l in r.stderr.splitlines() if l.startswith("[pp_score") or l.startswith("[modest alloc]")]
    print("\n".join(l[:300] for l in lines[:60]))
    if r.returncode:
        print(r.stderr[-2000:])
