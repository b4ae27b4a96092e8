"""GPU box: the PP CLI with workers=N on ONE synthetic tree, repeated: every worker's loop seconds per run (a straggling
worker shows as one value far above the others) and, with MODEST_ALLOC_TRACE=1, the allocation lines of the slowest worker."""
import os, re, shutil, subprocess, sys, tempfile, time
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
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        secs = [float(m.group(1)) for m in re.finditer(r"\[pp_score\] \d+ scans, [^,]+ history points, ([0-9.]+) s", r.stderr)]
        phases = [l[:230] for l in r.stderr.splitlines() if "trace" in l and "pp_score" in l]
        print("run %d: workers' loop seconds %s" % (rep, " ".join("%.2f" % s for s in sorted(secs))), flush=True)
        if secs and max(secs) > 1.8 * sorted(secs)[len(secs) // 2]:
            print("   STRAGGLER; last lines of stderr with 'alloc' or 'trace':")
            for l in [l for l in r.stderr.splitlines() if "modest alloc" in l or "trace" in l][-40:]:
                print("   " + l[:200])
