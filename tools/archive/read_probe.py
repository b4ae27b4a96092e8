"""GPU box: aggregate /dev/shm read rate of P processes into pageable vs pinned host buffers."""
import multiprocessing as mp
import os
import sys
import tempfile
import time

import numpy as np

NFILES = int(os.environ.get("NFILES", "800"))


def work(args):
    d, p, n, mode = args
    import torch
    torch.cuda.init()
    size = os.path.getsize(os.path.join(d, "000000.bin"))
    if mode == "pinned":
        buf = torch.empty((size * 8,), dtype=torch.uint8, pin_memory=True).numpy()
    else:
        buf = np.empty(size * 8, dtype=np.uint8)
    buf[:] = 0
    t0 = time.perf_counter()
    tot = 0
    for k in range(n):
        i = (p * n + k) % NFILES
        with open(os.path.join(d, f"{i:06d}.bin"), "rb", buffering=0) as f:
            tot += f.readinto(memoryview(buf[(k % 8) * size:(k % 8 + 1) * size]))
    return tot, time.perf_counter() - t0


if __name__ == "__main__":
    with tempfile.TemporaryDirectory(dir="/dev/shm") as root:
        rng = np.random.default_rng(0)
        a = rng.standard_normal((30000, 4)).astype(np.float32)
        for i in range(NFILES):
            (a + i).tofile(os.path.join(root, f"{i:06d}.bin"))
        for mode in ("pageable", "pinned"):
            for P in (1, 4, 8):
                with mp.get_context("spawn").Pool(P) as pool:
                    res = pool.map(work, [(root, p, NFILES // P, mode) for p in range(P)])
                tot = sum(r[0] for r in res)
                print(f"{mode:9s} {P} processes: {tot / max(r[1] for r in res) / 1e9:6.1f} GB/s aggregate, per process {min(r[0] / r[1] for r in res) / 1e9:5.1f}..{max(r[0] / r[1] for r in res) / 1e9:5.1f}")
