"""Worker for tests/test_dist_cpu.py: the chunked dynamic work queue over gloo (no GPU)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modest_amd import config, dist  # noqa: E402


def main():
    out_dir = sys.argv[1]
    rank, ws, local = dist.init(backend="gloo")
    cfg = config.ConfigNode(dict(total_part=1, part=0, work_queue="dynamic", queue_chunk=5, workers=1))
    idx = np.arange(1000, 1000 + 93)
    dist.barrier()
    t0 = time.perf_counter()
    got = []
    for x in dist.scans_of(idx, cfg, rank, ws, "t"):
        got.append(int(x))
        time.sleep(0.004 if rank == 0 else 0.001)     # rank 0 is the slow one: the queue gives it less
    tot = dist.rank_report("queue-test", len(got), t0, rank, ws)
    # a second queue under another name starts from zero again; the static mode ignores the store
    cfg2 = config.ConfigNode(dict(total_part=3, part=1, work_queue="dynamic", queue_chunk=4, workers=1))
    got2 = [int(x) for x in dist.scans_of(idx, cfg2, rank, ws, "t2")]
    cfg3 = config.ConfigNode(dict(total_part=1, part=0, work_queue="static", workers=1))
    got3 = [int(x) for x in dist.scans_of(idx, cfg3, rank, ws, "t3")]
    with open(os.path.join(out_dir, f"q{rank}.json"), "w") as f:
        json.dump(dict(got=got, got2=got2, got3=got3, tot=tot), f)
    dist.barrier()
    dist.finalize()


if __name__ == "__main__":
    main()
