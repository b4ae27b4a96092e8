"""GPU box: cProfile of the PP CLI's compute loop (main thread) on a synthetic Lyft-shaped tree."""
import cProfile
import os
import pstats
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modest_amd import pre_compute_pp_score, generate_mask, synth  # noqa: E402

n_scan, F, T = int(sys.argv[1]) if len(sys.argv) > 1 else 96, 36, 10
which = sys.argv[2] if len(sys.argv) > 2 else "pp"
with tempfile.TemporaryDirectory(dir="/dev/shm") as root:
    paths = synth.write_kitti_tree(os.path.join(root, "kitti"), os.path.join(root, "meta"), n_seq=T + 1, n_frames=n_scan + F,
                                   n_pts=30000, origins=tuple(range(n_scan)), hist_frames=F, max_range=80.0)
    data = f"data_root={root}/kitti/training"
    common = [data, f"data_paths.track_path={paths['track_path']}", f"data_paths.idx_info={paths['idx_info']}",
              f"data_paths.idx_list={paths['idx_list']}"]
    import io
    sys.stderr = cap = io.StringIO()
    if which == "pp":
        pre_compute_pp_score.main(argv=common + [f"data_paths.pp_score_path={root}/warm"])     # warm-up (kernels, allocator)
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        pr.enable()
        tot = pre_compute_pp_score.main(argv=common + [f"data_paths.pp_score_path={root}/pp"])
        pr.disable()
    else:
        pre_compute_pp_score.main(argv=common + [f"data_paths.pp_score_path={root}/pp"])
        generate_mask.main(argv=common[:1] + common[3:] + [f"data_paths.pp_score_path={root}/pp", f"data_paths.seg_save_dst={root}/segw",
                                                         f"data_paths.bbox_info_save_dst={root}/bboxw"])
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        pr.enable()
        tot = generate_mask.main(argv=common[:1] + common[3:] + [f"data_paths.pp_score_path={root}/pp", f"data_paths.seg_save_dst={root}/seg",
                                                               f"data_paths.bbox_info_save_dst={root}/bbox"])
        pr.disable()
    dt = time.perf_counter() - t0
    sys.stderr = sys.__stderr__
    print("\n".join(l for l in cap.getvalue().splitlines() if l.startswith("[pp_score]") or l.startswith("[generate_mask]")))
    print("%s CLI: %d scans in %.3f s = %.1f scans/s (loop clock %.3f s)" % (which, tot["scans"], dt, tot["scans"] / dt, tot["max_seconds"]))
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
