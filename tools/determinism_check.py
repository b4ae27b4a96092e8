"""GPU box: generate_mask twice in one process + once with workers=8 on the same tree; report differing files."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modest_amd import generate_mask, pre_compute_pp_score, synth  # noqa: E402

n_scan = int(sys.argv[1]) if len(sys.argv) > 1 else 300
F, T = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (12, 4)
with tempfile.TemporaryDirectory(dir="/dev/shm") as root:
    paths = synth.write_kitti_tree(os.path.join(root, "kitti"), os.path.join(root, "meta"), n_seq=T + 1, n_frames=n_scan + F,
                                   n_pts=30000, origins=tuple(range(n_scan)), hist_frames=F, max_range=80.0)
    data = f"data_root={root}/kitti/training"
    common = [data, f"data_paths.track_path={paths['track_path']}", f"data_paths.idx_info={paths['idx_info']}",
              f"data_paths.idx_list={paths['idx_list']}"]
    err = sys.stderr
    sys.stderr = open(os.devnull, "w")
    pre_compute_pp_score.main(argv=common + [f"data_paths.pp_score_path={root}/pp"])
    pre_compute_pp_score.main(argv=common + [f"data_paths.pp_score_path={root}/ppW", "workers=8"])
    for tag, w in (("a", 1), ("b", 1), ("w", 8)):
        generate_mask.main(argv=[data, common[3], f"data_paths.pp_score_path={root}/pp", f"data_paths.seg_save_dst={root}/seg{tag}",
                                 f"data_paths.bbox_info_save_dst={root}/bbox{tag}", f"workers={w}"])
    sys.stderr = err
    files = sorted(f for f in os.listdir(f"{root}/sega") if f.endswith(".npy"))
    ppd = [f for f in files if not np.array_equal(np.load(f"{root}/pp/{f}"), np.load(f"{root}/ppW/{f}"))]
    print("scans", len(files), "pp files differing single vs workers:", len(ppd), ppd[:5])
    for other in ("b", "w"):
        bad = []
        for f in files:
            x, y = np.load(f"{root}/sega/{f}"), np.load(f"{root}/seg{other}/{f}")
            if not np.array_equal(x, y):
                bad.append((f, int((x != y).sum()), int(x.max()), int(y.max())))
        print(f"seg a vs {other}: {len(bad)} differing", bad[:8])
