"""Scan sharding across the GPUs of a node.

The path is embarrassingly parallel: every scan's PP score, mask, boxes and
label file depend only on that scan and its history frames, and the reference
itself shards with ``np.array_split(idx_list, total_part)[part]``
(``pre_compute_pp_score.py:114-116``, ``generate_mask.py:35-37``,
``gen_label_files.py:36-38``).  One process per GPU (torchrun / torch.distributed,
backend "nccl" = RCCL over xGMI on ROCm, "gloo" on CPU-only hosts for tests);
rank r of W takes ``np.array_split(part_list, W)[r]`` so the union of all ranks'
output files equals a single-process run.  RCCL carries only a start/end
barrier and one all-reduce of a few counters -- there is no data-path
collective because there is no cross-scan data dependency.
"""
from __future__ import annotations

import os
import time
from typing import Dict, Optional

import numpy as np
import torch


def world() -> tuple:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def runtime_defaults() -> None:
    """Process-wide ROCm runtime settings of the CLIs and bench.py; call before the first HIP call of the process (the
    environment wins: nothing set there is overridden).

    HSA_ENABLE_INTERRUPT=0: completion signals are polled instead of interrupt driven.  With several processes on one GPU
    a stream synchronise that sleeps on the interrupt wakes late now and then -- measured on the 8-process bench: in 5 of
    10 runs of a 20-scan window one process finished its chain at 14-17 ms instead of 7 (and in 1 of 10 all of them at
    50-70 ms); 0 of 10 with polling, steady-state throughput equal or better.  The price is one busy host thread per
    process that waits, which a GPU host has to spare."""
    if not torch.cuda.is_initialized():
        os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")


def init(backend: Optional[str] = None, poll_wait: bool = True) -> tuple:
    """Initialise torch.distributed when launched with WORLD_SIZE > 1 -- or, with MODEST_DIST_FORCE=1, at world size 1
    too: the RCCL branches of barrier() / reduce_counters() (device_ids, CUDA tensors) then run on a box with a single
    GPU exactly as they do on a node (a launcher must have set MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE, or the
    defaults below apply).  A forced run prints the result of selfcheck() on stderr."""
    rank, ws, local = world()
    if ws == 1 and poll_wait:   # (config key `poll_wait: false`: a host without a spare core per waiting process keeps interrupt waits)
        # a rank of a multi-GPU job keeps the runtime's own wait mode for its RCCL traffic (polling under RCCL with
        runtime_defaults()   # several ranks has not been on hardware yet); its worker / helper processes are single-rank
    force = os.environ.get("MODEST_DIST_FORCE", "") == "1"
    if (ws > 1 or force) and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            # MODEST_DIST_BACKEND=gloo: ranks that SHARE a GPU (more ranks than devices: RCCL refuses two
            # ranks on one device) still get the barrier / counter all-reduce / work queue, over TCP
            backend = os.environ.get("MODEST_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            if local >= torch.cuda.device_count():
                raise RuntimeError(f"rank {rank}: LOCAL_RANK {local} has no GPU of its own ({torch.cuda.device_count()} "
                                   "visible); set MODEST_DIST_BACKEND=gloo to let ranks share devices")
            torch.cuda.set_device(local)
        torch.distributed.init_process_group(backend=backend, rank=rank, world_size=ws)
        if force:
            import sys
            print("[dist] " + " ".join(f"{k}={v}" for k, v in selfcheck().items()), file=sys.stderr, flush=True)
    return rank, ws, local


def selfcheck() -> Dict[str, object]:
    """What the process group actually does (not what the environment says): backend, world size, a barrier, and an
    all-reduce of one 1 per rank -- `ranks_seen` must equal the world size.  Without a process group: ranks_seen = 1,
    backend = none."""
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return dict(backend="none", world_size=1, ranks_seen=1, barrier="skipped")
    barrier()
    seen = reduce_counters(dict(ones=1.0))["ones"]
    return dict(backend=torch.distributed.get_backend(), world_size=torch.distributed.get_world_size(),
                ranks_seen=int(round(seen)), barrier="ok")


def device_index(local: int, ws: int, cfg) -> int:
    """The GPU of this process: LOCAL_RANK under a multi-rank launch (wrapped around the visible devices when
    ranks share GPUs, see init), the config's `device` otherwise."""
    if ws > 1:
        n = max(torch.cuda.device_count(), 1)
        return local % n
    return int(cfg.get("device", 0))


class WorkQueue:
    """Chunked dynamic work queue over the scans of a shard (optional there): per-scan cost varies
    (cluster count, RANSAC trials, new frames to load), a static contiguous split leaves the fast ranks idle.
    Chunks of `chunk` consecutive scans (consecutive scans share history frames: the frame store stays warm)
    are handed out by an atomic counter in the process group's key-value store (served by rank 0, the same TCP
    store the rendezvous used) -- no collective, so ranks may finish at different times.  Which rank writes
    which file changes, the set of files does not.  `static` (default) is the reference's split."""

    def __init__(self, items, rank: int, ws: int, mode: str = "static", chunk: int = 32, name: str = "q"):
        self.items = np.asarray(items)
        self.dynamic = (mode == "dynamic" and ws > 1 and torch.distributed.is_available()
                        and torch.distributed.is_initialized() and not os.environ.get("MODEST_WORKER"))
        self.chunk = max(int(chunk), 1)
        if self.dynamic:
            from torch.distributed import distributed_c10d as c10d
            self.store, self.key = c10d._get_default_store(), f"modest_wq_{name}"
        else:
            self.mine = np.array_split(self.items, ws)[rank] if ws > 1 else self.items
        self.taken = 0

    def __iter__(self):
        if not self.dynamic:
            for x in self.mine:
                self.taken += 1
                yield x
            return
        n_chunks = (len(self.items) + self.chunk - 1) // self.chunk
        while True:
            c = int(self.store.add(self.key, 1)) - 1     # atomic fetch-and-add on the store
            if c >= n_chunks:
                return
            for x in self.items[c * self.chunk:(c + 1) * self.chunk]:
                self.taken += 1
                yield x


def shard(idx_list, total_part: int = 1, part: int = 0, rank: Optional[int] = None, ws: Optional[int] = None):
    """The reference's manual ``total_part/part`` split, then the per-rank split, then -- inside a
    worker process of a rank (``workers=N``, see run_workers) -- the per-worker split.  All three
    are contiguous ``np.array_split`` pieces: consecutive scans share history frames."""
    idx_list = np.asarray(idx_list)
    if total_part > 1:
        idx_list = np.array_split(idx_list, total_part)[part]
    if rank is None or ws is None:
        rank, ws, _ = world()
    if os.environ.get("MODEST_WORKER") and "MODEST_PARENT_WS" in os.environ:
        # A worker process of a rank (run_workers strips RANK / WORLD_SIZE from its environment, so its
        # own dist.init() says (0, 1)): the rank split is its PARENT's, whatever the caller passed.
        rank, ws = int(os.environ["MODEST_PARENT_RANK"]), int(os.environ["MODEST_PARENT_WS"])
    if ws > 1:
        idx_list = np.array_split(idx_list, ws)[rank]
    w = os.environ.get("MODEST_WORKER")
    if w:
        wp, wt = (int(x) for x in w.split("/"))
        idx_list = np.array_split(idx_list, wt)[wp]
    return idx_list


def scans_of(idx_list, cfg, rank: int, ws: int, name: str):
    """The scans this process works through: the reference's total_part/part piece, then either the static
    per-rank split (`work_queue: static`, the default) or the chunked dynamic queue shared by the ranks
    (`work_queue: dynamic`, `queue_chunk`), then -- inside a worker -- the worker's piece."""
    mode = str(cfg.get("work_queue", "static"))
    if mode == "dynamic" and ws > 1 and int(cfg.get("workers", 1) or 1) > 1 and not os.environ.get("MODEST_WORKER") and rank == 0:
        import sys
        print(f"[{name}] work_queue=dynamic is ignored with workers > 1 (the workers of a rank take contiguous pieces of the "
              "rank's static share): static split", file=sys.stderr, flush=True)
    if mode == "dynamic" and ws > 1 and not os.environ.get("MODEST_WORKER") and int(cfg.get("workers", 1) or 1) <= 1:
        part = shard(idx_list, cfg.total_part, cfg.part, rank=0, ws=1)
        return WorkQueue(part, rank, ws, "dynamic", int(cfg.get("queue_chunk", 32)), name)
    return WorkQueue(shard(idx_list, cfg.total_part, cfg.part, rank, ws), 0, 1)


def rank_report(tag: str, done: int, t0: float, rank: int, ws: int, extra: Optional[Dict[str, float]] = None) -> Dict[str, float]:
    """End of a CLI loop: this rank's busy time (before the barrier), then barrier + all-reduce.  Reports the
    spread between the ranks (scans and busy seconds): what a dynamic queue would have to absorb."""
    busy = time.perf_counter() - t0
    barrier()
    c = dict(scans=done, max_seconds=time.perf_counter() - t0, max_busy_seconds=busy, min_busy_seconds=busy,
             max_rank_scans=done, min_rank_scans=done)
    c.update(extra or {})
    tot = reduce_counters(c)
    tot["imbalance"] = (tot["max_busy_seconds"] - tot["min_busy_seconds"]) / max(tot["max_busy_seconds"], 1e-9)
    if rank == 0 and ws > 1:
        import sys
        print("[%s] ranks: scans %d..%d, busy %.2f..%.2f s (imbalance %.1f %%)"
              % (tag, tot["min_rank_scans"], tot["max_rank_scans"], tot["min_busy_seconds"], tot["max_busy_seconds"],
                 100.0 * tot["imbalance"]), file=sys.stderr, flush=True)
    return tot


def run_workers(module: str, cfg, rank: int, ws: int, local: int) -> Optional[Dict[str, float]]:
    """``workers=N`` of the CLIs: the host side of a scan (Python + ~60 HIP calls + blocking round
    trips) saturates one process long before the GPU, which is why the reference is run as several
    ``total_part/part`` jobs by hand (generate_mask.py:33-37).  Here a rank does that itself: N
    child processes on the rank's GPU, each taking a contiguous piece of the rank's shard and
    writing its own output files.  Returns the summed counters of the children (None when this
    process should do the work itself: workers <= 1, or it IS a worker)."""
    import json
    import subprocess
    import sys
    import tempfile
    from . import config as _config
    n = int(cfg.get("workers", 1) or 1)
    if n <= 1 or os.environ.get("MODEST_WORKER"):
        return None
    with tempfile.TemporaryDirectory() as d:
        cpath = os.path.join(d, "cfg.yaml")
        c2 = _config.ConfigNode(cfg.to_container(resolve=True))
        c2["workers"] = 1
        c2["device"] = device_index(local, ws, cfg)
        _config.save(c2, cpath, resolve=False)
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
        env.update(MODEST_PARENT_RANK=str(rank), MODEST_PARENT_WS=str(ws))
        # processes that share a GPU size their persistent PP grids for half of it: a launch that fills all
        # CUs keeps the other workers' small kernels waiting (bench.py --pp-cus has the measurements)
        env.setdefault("MODEST_NUM_CUS", "128")
        # N interpreters x (BLAS pool of one thread per core, spinning) oversubscribe the host: the per-scan
        # linear algebra here is 4x4 (poses) and a few dozen box corners
        for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
            env.setdefault(k, "1")
        procs = []
        for w in range(n):
            e = dict(env, MODEST_WORKER=f"{w}/{n}", MODEST_WORKER_RESULT=os.path.join(d, f"w{w}.json"))
            procs.append(subprocess.Popen([sys.executable, "-m", "modest_amd.dist", module, cpath], env=e))
        rcs = [p.wait() for p in procs]
        if any(rcs):
            raise RuntimeError(f"{module}: worker processes failed with exit codes {rcs}")
        tot: Dict[str, float] = {}
        for w in range(n):
            for k, v in json.load(open(os.path.join(d, f"w{w}.json"))).items():
                if k.startswith("max_"):
                    tot[k] = max(tot.get(k, 0.0), v)
                elif k.startswith("min_"):
                    tot[k] = min(tot.get(k, v), v)
                else:
                    tot[k] = tot.get(k, 0.0) + v
    return tot


def _worker_entry() -> None:
    """``python -m modest_amd.dist <module> <config.yaml>``: one worker of run_workers."""
    import importlib
    import json
    import sys
    import yaml
    module, cpath = sys.argv[1], sys.argv[2]
    cfg = yaml.safe_load(open(cpath))
    tot = importlib.import_module(module).main(cfg)
    with open(os.environ["MODEST_WORKER_RESULT"], "w") as f:
        json.dump({k: float(v) for k, v in (tot or {}).items()}, f)


def barrier() -> None:
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        if torch.distributed.get_backend() == "nccl":
            torch.distributed.barrier(device_ids=[torch.cuda.current_device()])
        else:
            torch.distributed.barrier()


def reduce_counters(counters: Dict[str, float]) -> Dict[str, float]:
    """Sum scalar counters over ranks ("max_" / "min_" prefixed keys take the maximum / minimum)."""
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return dict(counters)
    keys = sorted(counters)
    dev = "cuda" if torch.distributed.get_backend() == "nccl" else "cpu"
    out = {}
    for prefix, op in (("max_", torch.distributed.ReduceOp.MAX), ("min_", torch.distributed.ReduceOp.MIN),
                       (None, torch.distributed.ReduceOp.SUM)):
        ks = [k for k in keys if (k.startswith(prefix) if prefix else not k.startswith(("max_", "min_")))]
        if not ks:
            continue
        t = torch.tensor([float(counters[k]) for k in ks], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=op)
        out.update({k: float(v) for k, v in zip(ks, t.tolist())})
    return out


def finalize() -> None:
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


class StageTimer:
    """Wall-clock accumulator per stage (the reference has only tqdm bars)."""

    def __init__(self):
        self.t: Dict[str, float] = {}
        self._t0 = None
        self._name = None

    def start(self, name):
        self._name, self._t0 = name, time.perf_counter()

    def stop(self):
        self.t[self._name] = self.t.get(self._name, 0.0) + time.perf_counter() - self._t0


if __name__ == "__main__":
    _worker_entry()
