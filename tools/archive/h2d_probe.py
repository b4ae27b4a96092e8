"""GPU box: aggregate pinned->device copy rate of P processes (5.3 MB pieces, like one scan's new frames)."""
import multiprocessing as mp
import sys
import time


def work(args):
    p, n, mb = args
    import torch
    dev = torch.device("cuda:0")
    h = torch.empty((int(mb * 2 ** 20),), dtype=torch.uint8, pin_memory=True)
    d = torch.empty_like(h, device=dev)
    for _ in range(5):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    return n * mb * 2 ** 20, time.perf_counter() - t0


if __name__ == "__main__":
    for mb in (5.3, 0.03):
        for P in (1, 2, 4, 8):
            with mp.get_context("spawn").Pool(P) as pool:
                res = pool.map(work, [(p, 400, mb) for p in range(P)])
            print(f"{mb} MB pieces, {P} processes: {sum(r[0] for r in res) / max(r[1] for r in res) / 1e9:6.1f} GB/s aggregate, "
                  f"{1e3 * max(r[1] for r in res) / 400:.3f} ms per copy per process")
