"""GPU box: where the PP CLI and the fused CLI spend a worker's time on one rank's share of the Lyft train set (1 485 scans,
workers=8).  Prints every worker's summary line (ingest thread: read / upload + sort / bookkeeping; loop thread: wait / poses /
tables / pp / submit / post) and a raw read probe (N processes x 4 threads, readinto pinned memory, no GPU work)."""
import argparse
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def raw_probe(vel, n_proc, n_files, threads=4):
    code = r'''
import os, sys, time, torch
from concurrent.futures import ThreadPoolExecutor
vel, w, n, per, threads = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
files = sorted(os.listdir(vel))[w * per:(w + 1) * per]
sz = os.path.getsize(os.path.join(vel, files[0]))
if sys.argv[6] == "2":   # pageable, first touched here, then pinned in place
    torch.zeros(1, device="cuda")
    t = torch.empty((32, sz), dtype=torch.uint8); t.zero_()
    rc = torch.cuda.cudart().cudaHostRegister(t.data_ptr(), t.numel(), 0)
    assert int(rc) == 0, rc
    buf = t.numpy()
else:
    buf = torch.empty((32, sz), dtype=torch.uint8, pin_memory=(sys.argv[6] == "1")).numpy()
pool = ThreadPoolExecutor(threads)
def rd(k):
    with open(os.path.join(vel, files[k]), "rb", buffering=0) as f:
        f.readinto(memoryview(buf[k % 32])[:os.path.getsize(os.path.join(vel, files[k]))])
list(pool.map(rd, range(8)))
t0 = time.perf_counter()
for c in range(0, len(files), 32):
    list(pool.map(rd, range(c, min(c + 32, len(files)))))
dt = time.perf_counter() - t0
print("raw read worker %d: %d files, %.2f GB/s" % (w, len(files), len(files) * sz / dt / 1e9), flush=True)
'''
    for pinned in ("0", "1", "2"):
        t0 = time.perf_counter()
        ps = [subprocess.Popen([sys.executable, "-c", code, vel, str(w), str(n_proc), str(n_files // n_proc), str(threads), pinned]) for w in range(n_proc)]
        for p in ps:
            p.wait()
        print("raw probe: %d processes x %d threads, pinned=%s, wall %.2f s (with start-up)" % (n_proc, threads, pinned, time.perf_counter() - t0), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=1485)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--modes", default="raw,pp,fused")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--variants", default="", help="';'-separated lists of extra overrides, e.g. 'ingest_depth=4;ingest_depth=20'")
    a = ap.parse_args()
    from modest_amd import pre_compute_pp_score, seed_labels, synth
    F, T = 36, 10
    with tempfile.TemporaryDirectory(dir="/dev/shm") as root:
        paths = synth.write_kitti_tree(os.path.join(root, "kitti"), os.path.join(root, "meta"), n_seq=T + 1, n_frames=a.scans + F, n_pts=30000,
                                       origins=tuple(range(a.scans)), hist_frames=F, max_range=80.0)
        data = f"data_root={root}/kitti/training"
        idx = f"data_paths.idx_list={paths['idx_list']}"
        common = [data, f"data_paths.track_path={paths['track_path']}", f"data_paths.idx_info={paths['idx_info']}", idx, "device=0"]
        vel = f"{root}/kitti/training/velodyne"
        modes = a.modes.split(",")
        if "raw" in modes:
            n = len(os.listdir(vel))
            print("velodyne files:", n, flush=True)
            raw_probe(vel, 1, min(n, 2400))
            raw_probe(vel, a.workers, min(n, 2400 * a.workers))
        run = 0
        for rep in range(a.reps):
          for var in a.variants.split(";"):
            extra = [v for v in var.split(",") if v]
            run += 1
            if "pp" in modes:
                print("[pp_score] variant", extra, file=sys.stderr, flush=True)
                tot = pre_compute_pp_score.main(argv=common + [f"data_paths.pp_score_path={root}/pp{run}", f"workers={a.workers}"] + extra)
                print("PP CLI workers=%d %s: %.0f scans/s (slowest worker's loop)" % (a.workers, extra, tot["scans"] / tot.get("max_worker_seconds", tot["max_seconds"])), flush=True)
            if "fused" in modes:
                print("[seed_labels] variant", extra, file=sys.stderr, flush=True)
                tot = seed_labels.main(argv=common + [f"data_paths.pp_score_path={root}/fpp{run}", f"data_paths.seg_save_dst={root}/fseg{run}",
                                                      f"data_paths.bbox_info_save_dst={root}/fbbox{run}", f"data_paths.label_file_save_dst={root}/flab{run}",
                                                      f"workers={a.workers}"] + [("pp." + v) if v.startswith("ingest") or v.startswith("readers") else v for v in extra])
                print("fused CLI workers=%d %s: %.0f scans/s (slowest worker's loop)" % (a.workers, extra, tot["scans"] / tot.get("max_worker_seconds", tot["max_seconds"])), flush=True)


if __name__ == "__main__":
    main()
