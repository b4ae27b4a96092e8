"""GPU box: cost of the frame ingest (read -> pinned -> H2D -> sort) per batch size, one process."""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modest_amd import _lib, synth  # noqa: E402
from modest_amd.frame_store import FrameStore  # noqa: E402
from modest_amd.pre_compute_pp_score import FrameLoader  # noqa: E402

dev = torch.device("cuda:0")
with tempfile.TemporaryDirectory(dir="/dev/shm") as root:
    d = os.path.join(root, "velodyne")
    os.makedirs(d)
    rng = np.random.default_rng(0)
    N = 800
    for i in range(N):
        (rng.standard_normal((30000, 4)) * [20, 20, 1, 1]).astype(np.float32).tofile(os.path.join(d, f"{i:06d}.bin"))
    world = {i: np.eye(4) for i in range(N)}
    for readers in (1, 4, 8):
        store = FrameStore(dev, 0.3)
        ld = FrameLoader(d, store, world, readers=readers, ctx=_lib.Context(0))
        ld.ensure(list(range(0, 40)))            # warm-up: pinned buffer, kernels
        torch.cuda.synchronize()
        for batch in (361, 11, 11, 11):
            ids = list(range(len(store.frames), len(store.frames) + batch))
            t0 = time.perf_counter()
            host, offs = ld._read_batch(ids)
            t1 = time.perf_counter()
            devb = torch.empty(host.shape, dtype=torch.float32, device=dev)
            devb.copy_(host, non_blocking=True)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            store.insert_many([(i, devb[offs[k]:offs[k + 1]], world[i]) for k, i in enumerate(ids)], ctx=ld.ctx)
            t3 = time.perf_counter()
            print(f"readers {readers} batch {batch:3d}: read {1e3 * (t1 - t0):6.2f} ms ({host.numel() * 4 / (t1 - t0) / 1e9:5.1f} GB/s)  "
                  f"H2D {1e3 * (t2 - t1):5.2f} ms  sort+bookkeeping {1e3 * (t3 - t2):5.2f} ms", flush=True)
