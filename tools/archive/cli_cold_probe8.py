"""workers=8 PP CLI: the allocation + submit trace of ONE worker (the first pid seen), times relative to its first line."""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modest_amd import synth
n_scan, F, T = 768, 36, 10
with tempfile.TemporaryDirectory(dir="/dev/shm") as root:
    paths = synth.write_kitti_tree(os.path.join(root, "kitti"), os.path.join(root, "meta"), n_seq=T + 1, n_frames=n_scan + F,
                                   n_pts=30000, origins=tuple(range(n_scan)), hist_frames=F, max_range=80.0)
    cmd = [sys.executable, "-m", "modest_amd.pre_compute_pp_score", f"data_root={root}/kitti/training",
           f"data_paths.track_path={paths['track_path']}", f"data_paths.idx_info={paths['idx_info']}",
           f"data_paths.idx_list={paths['idx_list']}", f"data_paths.pp_score_path={root}/pp", "workers=8"] + sys.argv[1:]
    env = dict(os.environ, MODEST_PP_TRACE="1", MODEST_ALLOC_TRACE="1", MODEST_PP_WALL="1")
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    lines = r.stderr.splitlines()
    al = [l for l in lines if l.startswith("[modest alloc]") and "staging" not in l]
    pid = re.search(r"pid (\d+)", al[0]).group(1) if al else None
    t0 = None
    for l in al:
        if f"pid {pid}" in l:
            t = float(re.search(r"t=([\d.]+)", l).group(1))
            t0 = t if t0 is None else t0
            print(f"+{(t - t0) * 1e3:8.1f} ms  {l.split(': ', 1)[1]}")
    print("\n".join(l[:200] for l in lines if l.startswith("[pp_score]") or "wall" in l)[:3000])
