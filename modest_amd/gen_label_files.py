"""``python -m modest_amd.gen_label_files data_root=...`` -- KITTI label CLI.

Drop-in for the reference's ``generate_cluster_mask/gen_label_files.py`` (its
README calls it ``generate_label_files.py``; that alias exists too): reads
``bbox_info_save_dst/NNNNNN.pkl``, runs BEV NMS on the device, keeps boxes in
the camera FOV and writes ``label_file_save_dst/NNNNNN.txt`` in the KITTI text
format OpenPCDet's ``get_objects_from_label`` parses (class ``Dynamic``).
"""
from __future__ import annotations

import os
import os.path as osp
import pickle
import sys
import time

import numpy as np
import torch

from . import config, dist, ops
from .utils import kitti_util
from .utils.pointcloud_utils import is_within_fov, objs2label, objs_nms


def eprint(*args, **kwargs):
    print(*args, file=sys.stderr, **kwargs)


def display_args(args):
    eprint("========== kitti_label gen info ==========")
    eprint("host: {}".format(os.getenv("HOSTNAME")))
    eprint(config.to_yaml(args))
    eprint("==========================================")


NATIVE_LABELS = True   # tests switch it off to compare the library's label stage with the Python statement


def gen_label_scan(objs, calib, args, after_device=None, iou=None):
    """gen_label_files.py:44-52 for one scan -> (label text, kept objs).
    ``after_device`` (optional callable) runs as soon as the stage has no device work left -- its only
    device call is the IoU matrix of the NMS; a pipeline uses it to enqueue the next scan's device work
    under the host tail of this one (greedy NMS, FOV filter, label text)."""
    if NATIVE_LABELS:
        # the float32 boxes + IoU matrix (device), ONE numpy call for the order of the walk (its tie order is
        # numpy's: SURVEY H6), then greedy walk + FOV filter + label text in the library
        rows = objs if isinstance(objs, np.ndarray) else \
            np.array([[*o.t, o.l, o.w, o.h, o.ry, o.volume] for o in objs], dtype=np.float64).reshape(-1, 8)
        nms = bool(args.nms.enable) and len(rows) > 0
        if nms and iou is None:   # (iou given: computed for a chain of scans, gen_label_chain)
            iou = ops.objs_iou(rows)
        if after_device is not None:
            after_device()
        order = np.diag(iou).argsort()[::-1] if nms else None
        text, kept = ops.label_lines(rows, order, iou, calib.P, nms, args.nms.threshold if nms else 0.0,
                                     bool(args.fov_only), args.image_shape)
        return text, (rows[kept] if isinstance(objs, np.ndarray) else [objs[i] for i in kept])
    if isinstance(objs, np.ndarray):
        from .generate_mask import objs_from_rows
        objs = objs_from_rows(objs)
    if args.nms.enable and len(objs) > 0:
        objs = objs_nms(objs, nms_threshold=args.nms.threshold, after_device=after_device)
    elif after_device is not None:
        after_device()
    if args.fov_only:
        objs = [obj for obj in objs if is_within_fov(obj, calib, args.image_shape)]
    return objs2label(objs, calib), objs


def gen_label_chain(objs_list, calibs, args, after_device=None, ious=None):
    """gen_label_scan for a CHAIN of scans: the IoU matrices of all of them from ONE launch and one round trip
    (modest_objs_iou_batch), then order, walk, FOV filter and label text scan by scan.  after_device runs once the
    chain has no device work left.  Returns [(text, kept objs)] -- identical to separate calls."""
    calibs = calibs if isinstance(calibs, (list, tuple)) else [calibs] * len(objs_list)
    # (ious: the matrices modest_seed_chain already produced -- generate_mask_chain(with_iou=True); a None entry is computed here)
    if ious is None:
        ious = [None] * len(objs_list)
    if NATIVE_LABELS and bool(args.nms.enable) and all(isinstance(o, np.ndarray) for o in objs_list):
        todo = [i for i, m in enumerate(ious) if m is None]
        if todo:
            ious = list(ious)
            for i, m in zip(todo, ops.objs_iou_batch([objs_list[i] for i in todo])):
                ious[i] = m
    if after_device is not None:
        after_device()
    return [gen_label_scan(o, cb, args, iou=(m if len(o) > 0 else None)) for o, cb, m in zip(objs_list, calibs, ious)]


def _pooled(args, rank, ws, local):
    """workers=N: N child processes on this rank's GPU (dist.run_workers), same barrier + counter
    all-reduce around them as around the in-process loop."""
    dist.barrier()
    t0 = time.perf_counter()
    tot = dist.run_workers("modest_amd.gen_label_files", args, rank, ws, local)
    dist.barrier()
    tot["max_worker_seconds"] = tot.get("max_seconds", 0.0)   # the workers' own loop clocks (no start-up)
    tot["max_seconds"] = time.perf_counter() - t0
    tot = dist.reduce_counters(tot)
    if rank == 0:
        eprint("[gen_label_files] %d scans, %.2f s, %.2f scans/s on %d GPU(s) x %d worker processes"
               % (tot["scans"], tot["max_seconds"], tot["scans"] / max(tot["max_seconds"], 1e-9), ws, int(args.workers)))
    return tot


@config.main(config_name="generate_label_files.yaml")
def main(args):
    rank, ws, local = dist.init(poll_wait=bool(args.get("poll_wait", True)))
    if rank == 0:
        display_args(args)
    torch.cuda.set_device(torch.device("cuda", dist.device_index(local, ws, args)))
    dp = args.data_paths
    idx_list = np.array([int(x) for x in open(dp.idx_list).readlines()])
    shard = dist.scans_of(idx_list, args, rank, ws, "labels")
    os.makedirs(dp.label_file_save_dst, exist_ok=True)
    if int(args.get("workers", 1) or 1) > 1 and not os.environ.get("MODEST_WORKER"):
        return _pooled(args, rank, ws, local)
    t0, done = time.perf_counter(), 0
    dist.barrier()
    for idx in shard:
        idx = int(idx)
        objs = pickle.load(open(osp.join(dp.bbox_info_save_dst, f"{idx:06d}.pkl"), "rb"))
        calib = kitti_util.Calibration(osp.join(args.calib_path, f"{idx:06d}.txt"))
        text, _ = gen_label_scan(objs, calib, args)
        with open(osp.join(dp.label_file_save_dst, f"{idx:06d}.txt"), "w") as f:
            f.write(text)
        done += 1
    tot = dist.rank_report("gen_label_files", done, t0, rank, ws)
    if rank == 0:
        eprint("[gen_label_files] %d scans, %.2f s on %d GPU(s)" % (tot["scans"], tot["max_seconds"], ws))
    return tot


if __name__ == "__main__":
    main()
