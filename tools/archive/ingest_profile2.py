"""cProfile of the ingest thread's work (FrameLoader.ensure scan by scan, blocking=False) on a synthetic tree: where the host
time of 'upload + sort' goes."""
import cProfile, os, pickle, pstats, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modest_amd import _lib, synth, pre_compute_pp_score as pps
from modest_amd.frame_store import FrameStore
n_scan, F, T = 96, 36, 10
with tempfile.TemporaryDirectory(dir="/dev/shm") as root:
    paths = synth.write_kitti_tree(os.path.join(root, "kitti"), os.path.join(root, "meta"), n_seq=T + 1, n_frames=n_scan + F,
                                   n_pts=30000, origins=tuple(range(n_scan)), hist_frames=F, max_range=80.0)
    train = f"{root}/kitti/training"
    track = pickle.load(open(paths["track_path"], "rb")); valid = pickle.load(open(paths["idx_info"], "rb"))
    poses, l2es = pps.load_poses(track, f"{train}/oxts", f"{train}/l2e")
    world = pps.frame_world_matrices(track, poses, l2es, pps._KITTI2NU_lyft)
    dev = torch.device("cuda:0")
    store = FrameStore(dev, 0.3)
    store.reserve(2 * 2 ** 30)
    ld = pps.FrameLoader(f"{train}/velodyne", store, world, readers=4, ctx=_lib.Context(0), frame_bytes=os.path.getsize(f"{train}/velodyne/000000.bin"))
    plans = []
    for o in sorted(valid):
        s0, f0, trav = valid[o]
        plans.append([track[s][f] for s, ix in trav for f in ix] + [track[s0][f0]])
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        tc = time.perf_counter()
        pc = cProfile.Profile(); pc.enable()
        ld.ensure(plans[0], blocking=False)   # cold
        pc.disable()
        tc1 = time.perf_counter(); torch.cuda.synchronize(); tc2 = time.perf_counter()
        print(f"cold scan (361 frames): host {1e3 * (tc1 - tc):.1f} ms, device tail {1e3 * (tc2 - tc1):.1f} ms; read {ld.t_read:.3f} insert {ld.t_insert:.3f}")
        pstats.Stats(pc).sort_stats("tottime").print_stats(10)
        pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
        for p in plans[1:]:
            ld.ensure(p, protect=p, blocking=False)
        pr.disable(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{len(plans) - 1} scans: host {1e3 * (t1 - t0):.1f} ms ({1e3 * (t1 - t0) / (len(plans) - 1):.2f} ms per scan), device tail {1e3 * (t2 - t1):.1f} ms; read {ld.t_read:.3f} insert {ld.t_insert:.3f} touch {ld.t_touch:.3f}")
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
