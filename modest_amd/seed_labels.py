"""The three seed-label stages in ONE process per GPU share: `python -m modest_amd.seed_labels data_root=... [key=value ...]`.

Not a CLI of the reference -- the reference runs pre_compute_pp_score.py, generate_mask.py and gen_label_files.py one after the
other (README.md:52-70) and hands scans over through files -- but a drop-in for that SEQUENCE: every file the three CLIs write is
written, with the same names and the same bytes (pp_score_path/NNNNNN.npy, seg_save_dst/NNNNNN.npy + configs.yaml,
bbox_info_save_dst/NNNNNN.pkl + configs.yaml, label_file_save_dst/NNNNNN.txt; tests/test_gpu_e2e.py compares the trees).
What it saves is what the hand-over costs: the live scan is read once instead of twice, the PP score never leaves the device
between the stages, there is one interpreter / library / frame-store start-up instead of three, and stages 2 + 3 of a batch run on
the host while the GPU works on the next batch's neighbour count.

Overrides: `key=value` applies to every stage config that HAS the key (data_root, data_paths=..., data_paths.*, total_part, part,
workers, device, nusc ...); `pp.key=value`, `mask.key=value`, `labels.key=value` address one stage (configs/pp_score.yaml,
generate_mask.yaml, generate_label_files.yaml)."""
from __future__ import annotations

import os
import os.path as osp
import pickle
import queue
import sys
import threading
import time
from typing import List, Optional

import numpy as np
import torch

from . import config, dist
from .gen_label_files import gen_label_chain
from .generate_mask import generate_mask_chain, generate_mask_scan, objs_from_rows
from .pre_compute_pp_score import eprint
from .utils import kitti_util
from .utils.pointcloud_utils import load_velo_scan

STAGES = (("pp", "pp_score"), ("mask", "generate_mask"), ("labels", "generate_label_files"))


TOP_KEYS = ("workers", "device", "total_part", "part", "work_queue", "queue_chunk", "poll_wait")   # what modest_amd.dist reads


class FusedConfig:
    """The three stage configs (each a ConfigNode of its own: interpolations resolve inside their stage) + the keys the work split
    reads, taken from the PP stage's config.  Quacks like a ConfigNode as far as modest_amd.dist needs it."""

    def __init__(self, stages):
        self.stages = dict(stages)

    def __getattr__(self, k):
        if k in ("pp", "mask", "labels"):
            return self.stages[k]
        if k in TOP_KEYS and k in self.stages["pp"]:
            return self.stages["pp"][k]
        raise AttributeError(k)

    def __contains__(self, k):
        return k in self.stages or (k in TOP_KEYS and k in self.stages["pp"])

    def __getitem__(self, k):
        return getattr(self, k)

    def get(self, k, default=None):
        return getattr(self, k) if k in self else default

    def to_container(self, resolve: bool = True) -> dict:
        out = {tag: node.to_container(resolve) for tag, node in self.stages.items()}
        out.update({k: self.stages["pp"][k] for k in TOP_KEYS if k in self.stages["pp"]})
        return out


def compose_all(overrides: List[str]) -> FusedConfig:
    """key=value goes to every stage config that has the key (or the group); pp. / mask. / labels. prefixes address one stage."""
    overrides = list(overrides or [])
    prefixed = {tag: [] for tag, _ in STAGES}
    common = []
    for ov in overrides:
        body = ov.lstrip("+~")
        sign = ov[: len(ov) - len(body)]
        key = body.split("=", 1)[0]
        head = key.split(".", 1)[0]
        if head in prefixed and "." in key:
            prefixed[head].append(sign + body.split(".", 1)[1])
        else:
            common.append(ov)
    used = [False] * len(common)
    stages = {}
    for tag, name in STAGES:
        mine = []
        for k, ov in enumerate(common):
            try:
                config.compose(name, [ov])
            except (KeyError, FileNotFoundError):
                continue
            mine.append(ov)
            used[k] = True
        stages[tag] = config.compose(name, mine + prefixed[tag])
    missing = [ov for ov, u in zip(common, used) if not u]
    if missing:
        raise KeyError(f"no stage config has the key of {missing} (use pp. / mask. / labels. prefixes, or +key=value)")
    return FusedConfig(stages)


class _FileWriter:
    """seg / bbox / label files leave through a writer thread: pickling 20 SimpleNamespace boxes and two file creations cost more
    host time per scan than the scan's stage-2 launches."""

    def __init__(self):
        self.q = queue.Queue(maxsize=256)
        self.error = None
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _run(self):
        while True:
            job = self.q.get()
            if job is None:
                return
            try:
                job()
            except BaseException as e:   # noqa: BLE001 (reported by close())
                self.error = e

    def submit(self, job):
        self.q.put(job)

    def close(self):
        self.q.put(None)
        self.thread.join()
        if self.error is not None:
            raise self.error


class FusedStages:
    """Stages 2 + 3 (generate_mask.py:52-109, gen_label_files.py:40-52) of the scans of a flushed PP batch, one batch late."""

    def __init__(self, margs, largs, rank: int):
        self.margs, self.largs = margs, largs
        dp = margs.data_paths
        self.seg_dst, self.bbox_dst, self.pp_dir = dp.seg_save_dst, dp.get("bbox_info_save_dst", "None"), dp.pp_score_path
        self.label_dst = largs.data_paths.label_file_save_dst
        for d, cfg in ((self.seg_dst, margs), (self.bbox_dst, margs)):
            if d is not None:
                os.makedirs(d, exist_ok=True)
                if rank == 0 and not osp.exists(osp.join(d, "configs.yaml")):   # (generate_mask.py:38-46)
                    config.save(config=cfg, f=osp.join(d, "configs.yaml"))
        os.makedirs(self.label_dst, exist_ok=True)
        if self.bbox_dst is None:
            raise ValueError("the fused mode hands the boxes to the label stage: data_paths.bbox_info_save_dst must be set")
        self.seed = int(margs.get("ransac_seed", 0))
        self.n_chain = max(1, int(margs.get("mask_batch", 4)))
        self.pending = None
        self.host_scores = [None, None]   # pinned read-back buffers of a batch's scores: one waiting, one being processed
        self.turn = 0
        self.writer = _FileWriter()
        self.scans = 0
        self.t_host = 0.0
        self.ph = dict(scores=0.0, files=0.0, mask=0.0, labels=0.0)   # seconds of _process by phase (the summary line)

    def done(self, idx: int) -> bool:
        return all(osp.exists(p) for p in (osp.join(self.seg_dst, f"{idx:06d}.npy"), osp.join(self.bbox_dst, f"{idx:06d}.pkl"),
                                            osp.join(self.label_dst, f"{idx:06d}.txt")))

    def __call__(self, batch):
        # The batch's PP scores start their way to the host NOW -- behind the batch's own kernels, ahead of the next batch's -- into a
        # pinned buffer (the mask stage's host statement and the rare scans the library hands back read them).  Read back inside
        # _process, one batch later, the copy queued behind the NEXT batch's PP kernels and the loop waited for all of them.
        Hdev = torch.cat([H for _, _, H in batch])
        k = self.turn = 1 - self.turn
        if self.host_scores[k] is None or self.host_scores[k].shape[0] < Hdev.shape[0]:
            self.host_scores[k] = torch.empty((int(Hdev.shape[0] * 1.25) + 1024,), dtype=Hdev.dtype, pin_memory=True)
        host = self.host_scores[k][:Hdev.shape[0]]
        host.copy_(Hdev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        prev, self.pending = self.pending, (batch, host, ev)
        if prev:
            self._process(*prev)

    def close(self):
        prev, self.pending = self.pending, None
        if prev:
            self._process(*prev)
        self.writer.close()

    def _process(self, batch, host, ev):
        t0 = time.perf_counter()
        margs, largs = self.margs, self.largs
        ev.synchronize()
        Hcat = host.numpy()
        offs = np.cumsum([0] + [int(H.shape[0]) for _, _, H in batch])
        t1 = time.perf_counter()
        self.ph["scores"] += t1 - t0
        items, calibs = [], []
        for k, (idx, _live, H) in enumerate(batch):
            ptc = load_velo_scan(osp.join(margs.ptc_path, f"{idx:06d}.bin"))
            items.append(dict(idx=idx, ptc=ptc, pp_score=Hcat[offs[k]:offs[k + 1]], pp_dev=H,
                              random_state=np.random.RandomState(self.seed + idx)))
            calibs.append(kitti_util.Calibration(osp.join(margs.calib_path, f"{idx:06d}.txt")))
        self.ph["files"] += time.perf_counter() - t1
        for c0 in range(0, len(items), self.n_chain):
            t2 = time.perf_counter()
            chunk, cal = items[c0:c0 + self.n_chain], calibs[c0:c0 + self.n_chain]
            if len(chunk) == 1:
                q = chunk[0]
                res = [generate_mask_scan(q["ptc"], q["pp_score"], cal[0], margs, random_state=q["random_state"], pp_dev=q["pp_dev"], as_rows=True)]
            else:
                res = generate_mask_chain(chunk, cal, margs, as_rows=True, with_iou=bool(largs.nms.enable))
            rows = [np.asarray(r[1], dtype=np.float64).reshape(-1, 8) for r in res]
            t3 = time.perf_counter()
            self.ph["mask"] += t3 - t2
            lab = gen_label_chain(rows, cal, largs, ious=[r[3] if len(r) > 3 else None for r in res])
            for q, r, rw, (text, _kept) in zip(chunk, res, rows, lab):
                self.writer.submit(lambda i=q["idx"], labels=r[0], rw=rw, text=text: self._write(i, labels, rw, text))
            self.ph["labels"] += time.perf_counter() - t3
        self.scans += len(batch)
        self.t_host += time.perf_counter() - t0

    def _write(self, idx, labels, rows, text):
        with open(osp.join(self.bbox_dst, f"{idx:06d}.pkl"), "wb") as f:
            pickle.dump(objs_from_rows(rows), f)
        np.save(osp.join(self.seg_dst, f"{idx:06d}.npy"), labels)
        with open(osp.join(self.label_dst, f"{idx:06d}.txt"), "w") as f:
            f.write(text)


def _pooled(cfg, rank, ws, local):
    dist.barrier()
    t0 = time.perf_counter()
    tot = dist.run_workers("modest_amd.seed_labels", cfg, rank, ws, local)
    dist.barrier()
    tot["max_worker_seconds"] = tot.get("max_seconds", 0.0)
    tot["max_seconds"] = time.perf_counter() - t0
    tot = dist.reduce_counters(tot)
    if rank == 0:
        eprint("[seed_labels] %d scans, %.2f s, %.2f scans/s on %d GPU(s) x %d worker processes"
               % (tot["scans"], tot["max_seconds"], tot["scans"] / max(tot["max_seconds"], 1e-9), ws, int(cfg.workers)))
    return tot


def main(cfg=None, argv: Optional[List[str]] = None):
    from . import pre_compute_pp_score
    if cfg is None:
        cfg = compose_all(sys.argv[1:] if argv is None else argv)
    elif not isinstance(cfg, FusedConfig):   # a worker of workers=N: the parent's resolved container (dist.run_workers), workers = 1, its device
        top = dict(cfg)
        cfg = FusedConfig({tag: config.ConfigNode(dict(top[tag])) for tag, _ in STAGES})
        for k in TOP_KEYS:
            if k in top:
                for a in cfg.stages.values():
                    if k in a:
                        a[k] = top[k]
    pargs, margs, largs = cfg.pp, cfg.mask, cfg.labels
    rank, ws, local = dist.init(poll_wait=bool(pargs.get("poll_wait", True)))
    if int(cfg.get("workers", 1) or 1) > 1 and not os.environ.get("MODEST_WORKER"):
        torch.cuda.set_device(torch.device("cuda", dist.device_index(local, ws, pargs)))
        for d in (pargs.data_paths.pp_score_path, margs.data_paths.seg_save_dst, largs.data_paths.label_file_save_dst):
            os.makedirs(d, exist_ok=True)
        FusedStages(margs, largs, rank).close()   # (directories + configs.yaml once, by the parent)
        return _pooled(cfg, rank, ws, local)
    post = FusedStages(margs, largs, rank)
    tot = pre_compute_pp_score.run(pargs, post=post)
    if rank == 0:
        eprint("[seed_labels] stages 2 + 3: %d scans, %.2f s of host time in this process (under the next batch's PP kernels): %s"
               % (post.scans, post.t_host, ", ".join("%s %.3f s" % kv for kv in post.ph.items())))
    return tot


if __name__ == "__main__":
    main()
