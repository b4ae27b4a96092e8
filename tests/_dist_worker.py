"""Worker for tests/test_dist_cpu.py: scan sharding + counters over gloo (no GPU)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modest_amd import dist  # noqa: E402


def main():
    out_dir = sys.argv[1]
    rank, ws, local = dist.init(backend="gloo")
    idx = np.arange(100, 100 + 37)
    shard = dist.shard(idx, total_part=1, part=0, rank=rank, ws=ws)
    shard2 = dist.shard(idx, total_part=3, part=1, rank=rank, ws=ws)
    dist.barrier()
    tot = dist.reduce_counters(dict(scans=len(shard), hist_points=1000 * (rank + 1), max_seconds=0.5 + rank))
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(dict(rank=rank, ws=ws, shard=[int(x) for x in shard], shard2=[int(x) for x in shard2], tot=tot), f)
    dist.barrier()
    dist.finalize()


if __name__ == "__main__":
    main()
