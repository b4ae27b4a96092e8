"""``python -m modest_amd.generate_mask data_root=...`` -- mask / cluster CLI.

Drop-in for the reference's ``generate_cluster_mask/generate_mask.py``: same
config keys, same outputs (``seg_save_dst/NNNNNN.npy`` int64 labels with 0 =
background, ``bbox_info_save_dst/NNNNNN.pkl`` pickled list of
``types.SimpleNamespace(t, l, w, h, ry, volume)``, ``configs.yaml`` dumps).
Per scan: RANSAC ground plane -> plane/range mask -> PP-weighted mutual-kNN
DBSCAN -> cluster filter -> box fit, each heavy loop in a HIP kernel.
"""
from __future__ import annotations

import os
import os.path as osp
import pickle
import sys
import time

import numpy as np
import torch

from . import _lib, config, dist, ops
from .utils import kitti_util
from .utils.clustering_utils import FILTER_PLANE_SPEC, compact_labels, filter_labels, members_sorted, relabel_after_drop
from .utils.pointcloud_utils import estimate_plane, get_objs, load_velo_scan, prepare_planes, to_device


def eprint(*args, **kwargs):
    print(*args, file=sys.stderr, **kwargs)


def display_args(args):
    eprint("========== clustering info ==========")
    eprint("host: {}".format(os.getenv("HOSTNAME")))
    eprint(config.to_yaml(args))
    eprint("=====================================")


def objs_from_rows(rows):
    """(k,8) rows {t0, t1, t2, l, w, h, ry, volume} -> the reference's SimpleNamespace objects (get_obj,
    pointcloud_utils.py:292-317; what bbox_info_save_dst pickles hold)."""
    import types
    out = []
    for r in rows:
        o = types.SimpleNamespace()
        o.t = np.array([r[0], r[1], r[2]])
        o.l, o.w, o.h, o.ry, o.volume = r[3], r[4], r[5], r[6], r[7]    # numpy float64 scalars, as in the reference
        out.append(o)
    return out


_UNSET = object()


def generate_mask_scan(ptc, pp_score, calib, args, random_state=None, planes=None, ptc_dev=None,
                       pp_dev=None, as_rows=False, staged=_UNSET, boxed=_UNSET):
    """The body of the reference's per-scan loop (generate_mask.py:52-103).

    ptc (N,4) float32 numpy, pp_score (N,) float32 numpy, calib a Calibration.
    ``random_state`` feeds both RANSAC calls in order (the reference consumes
    numpy's global stream in that order); ``planes`` = (plane1, plane2) injects
    the two ground planes instead (stage-wise parity tests).
    Returns (labels (N,) int64 with 0 = background, objs list, info dict); ``as_rows``: the boxes as a
    (k,8) float64 array {t0, t1, t2, l, w, h, ry, volume} instead of objects (in-memory pipelines)."""
    pe = args.plane_estimate
    ptc_dev = to_device(ptc) if ptc_dev is None else ptc_dev   # (N,4) float32 resident copy
    pp_dev = to_device(pp_score) if pp_dev is None else pp_dev
    if args.clustering.method != "DBSCAN":
        raise NotImplementedError(args.clustering.method)
    g = args.graph
    if g.neighbor_type not in ops.GRAPH_TYPES or g.affinity_type not in ops.AFFINITY_TYPES:
        raise NotImplementedError(f"graph {g.neighbor_type}/{g.affinity_type} (SURVEY.md §8f-3)")
    if staged is _UNSET:
        staged = None
        if planes is None and NATIVE_STAGE and ptc.shape[0] >= 1:
            rs = np.random.mtrand._rand if random_state is None else random_state
            if isinstance(rs, np.random.RandomState) and rs.get_state()[0] == "MT19937":
                # both ground fits, mask, graph + DBSCAN, cluster statistics, validity rules and the relabelling
                # behind one library call (no interpreter between the device round trips of these steps)
                staged = ops.mask_stage(ptc_dev, pp_dev, _stage_params(args), rs)
    # else: the stage ran in a chain of scans (generate_mask_chain): its result, or None = host statement
    if staged is not None:
        labels_filtered, plane, _, info = staged[:4]
        n_kept = int(info[0])
    else:
        labels_filtered, plane, n_kept = _mask_stage_host(ptc, pp_score, args, random_state, planes, ptc_dev, pp_dev)
    n_lab = int(labels_filtered.max()) if labels_filtered.size else 0
    lo, hi = args.filtering.min_volume, args.filtering.max_volume
    if boxed is not _UNSET or _native_boxes_ok(ptc, n_lab, args):
        # members, rect points, closeness fit, get_obj, volume gate and relabelling behind one library call
        if boxed is not _UNSET:
            done = boxed   # the box tail ran in a chain of scans (generate_mask_chain): its result, or None = host statement
        else:
            from .utils.pointcloud_utils import _angles, _angles90
            ang, cs = _angles(0.1)
            done = ops.scan_boxes(ptc_dev, ptc, labels_filtered, n_lab, calib.V2C, calib.R0, ang, cs, _angles90(0.1), 1e-2,
                                  lo, hi)
        if done is not None:
            labels_final, rows, keep = done
            rows = rows[keep]
            return labels_final, (rows if as_rows else objs_from_rows(rows)), dict(plane=plane, n_kept=n_kept, dbscan_kept=None)
    order, cuts = members_sorted(labels_filtered, n_lab)
    # rect-frame points: the clusters' rows on the host (per-cluster numpy arithmetic of get_obj), the whole
    # scan on the device, where the lowest-point search reads it (same rounding: one fma chain per element)
    if n_lab:
        rect_m = np.ascontiguousarray(calib.project_velo_to_rect(ptc[order, :3]))
        clusters = [rect_m[cuts[k]:cuts[k + 1]] for k in range(n_lab)]
        rect_dev = ops.project_velo_to_rect(ptc_dev, calib.V2C, calib.R0)
    else:
        clusters, rect_dev = [], None
    cand = get_objs(clusters, rect_dev, fit_method=args.bbox_gen.fit_method)
    keep = [bool(obj.volume > lo and obj.volume < hi) for obj in cand]
    objs = [obj for obj, k in zip(cand, keep) if k]
    labels_filtered = relabel_after_drop(labels_filtered, n_lab, keep) if n_lab else compact_labels(labels_filtered)
    if as_rows:
        objs = np.array([[*o.t, o.l, o.w, o.h, o.ry, o.volume] for o in objs], dtype=np.float64).reshape(-1, 8)
    return labels_filtered, objs, dict(plane=plane, n_kept=n_kept, dbscan_kept=None)


def _native_boxes_ok(ptc, n_lab, args) -> bool:
    return bool(NATIVE_BOXES and n_lab and args.bbox_gen.fit_method == "closeness_to_edge" and ptc.dtype == np.float32
                and ptc.flags.c_contiguous)


ONE_CALL_CHAIN = True   # tests switch it off to compare modest_seed_chain with the three separate chain calls


def _one_call_chain(scans, calibs, args, as_rows, ctxs, want_iou):
    """The chain through modest_seed_chain (mask stage + box tail + IoU matrices of the kept boxes: ONE library call), or None when
    some scan is not eligible (then the separate calls below decide scan by scan)."""
    if not (ONE_CALL_CHAIN and NATIVE_STAGE and NATIVE_BOXES and len(scans) > 1 and args.bbox_gen.fit_method == "closeness_to_edge"):
        return None
    c0 = calibs[0]
    if any(not (np.array_equal(c.V2C, c0.V2C) and np.array_equal(c.R0, c0.R0)) for c in calibs[1:]):
        return None
    items = []
    for sc in scans:
        rs = sc.get("random_state")
        rs = np.random.mtrand._rand if rs is None else rs
        p = sc["ptc"]
        if not (p.shape[0] >= 1 and p.dtype == np.float32 and p.flags.c_contiguous and isinstance(rs, np.random.RandomState)
                and rs.get_state()[0] == "MT19937"):
            return None
        if sc.get("ptc_dev") is None:
            sc["ptc_dev"] = to_device(p)
        if sc.get("pp_dev") is None:
            sc["pp_dev"] = to_device(sc["pp_score"])
        items.append((sc["ptc_dev"], p, sc["pp_dev"], rs))
    if len({id(it[3]) for it in items}) != len(items):   # one generator per scan: no draw order between scans
        return None
    from .utils.pointcloud_utils import _angles, _angles90
    ang, cs = _angles(0.1)
    res = ops.seed_chain(items, _stage_params(args), c0.V2C, c0.R0, ang, cs, _angles90(0.1), 1e-2, args.filtering.min_volume,
                         args.filtering.max_volume, want_iou, ctxs=ctxs)
    out = []
    for sc, cb, (status, labels, rows, iou, plane1, info) in zip(scans, calibs, res):
        if status == 0:
            out.append((labels, (rows if as_rows else objs_from_rows(rows)), dict(plane=plane1, n_kept=int(info[0]), dbscan_kept=None), iou))
            continue
        # 1: the stage handed the scan back (generator untouched) -> the host statement; 2 / 3: labels_filtered is there, the box tail
        # takes the separate call / the host statement
        staged = None if status == 1 else (labels, plane1, None, info[:8])
        r = generate_mask_scan(sc["ptc"], sc["pp_score"], cb, args, random_state=sc.get("random_state"), ptc_dev=sc.get("ptc_dev"),
                               pp_dev=sc.get("pp_dev"), as_rows=as_rows, staged=staged)
        out.append((r[0], r[1], r[2], None))
    return out


def generate_mask_chain(scans, calib, args, as_rows=False, ctxs=None, with_iou=False):
    """generate_mask_scan for a CHAIN of scans: stages 2 + 3 of all of them behind ONE library call (modest_seed_chain: the mask
    stage with one launch per kernel for the whole chain, the box tail, and -- `with_iou` -- the BEV IoU matrices the label stage's
    NMS needs), or, for chains that call does not cover (another fit method, different calibrations, numpy's global generator shared
    between scans), the mask stage as one call and boxes per scan.  scans: [dict(ptc=, pp_score=, random_state=, ptc_dev=, pp_dev=)];
    calib: one Calibration or one per scan.  Returns [generate_mask_scan's result] -- identical to separate calls; with_iou:
    4-tuples whose last entry is the (k,k) float32 IoU matrix of the scan's boxes (None: gen_label_scan computes it)."""
    calibs = calib if isinstance(calib, (list, tuple)) else [calib] * len(scans)
    one = _one_call_chain(scans, calibs, args, as_rows, ctxs, bool(with_iou))
    if one is not None:
        return one if with_iou else [r[:3] for r in one]
    res = _generate_mask_chain_calls(scans, calibs, args, as_rows, ctxs)
    return [r + (None,) for r in res] if with_iou else res


def _generate_mask_chain_calls(scans, calibs, args, as_rows, ctxs):
    staged = [_UNSET] * len(scans)
    if NATIVE_STAGE and len(scans) > 1:
        items, who = [], []
        for i, sc in enumerate(scans):
            rs = sc.get("random_state")
            rs = np.random.mtrand._rand if rs is None else rs
            if sc["ptc"].shape[0] >= 1 and isinstance(rs, np.random.RandomState) and rs.get_state()[0] == "MT19937":
                if sc.get("ptc_dev") is None:
                    sc["ptc_dev"] = to_device(sc["ptc"])
                if sc.get("pp_dev") is None:
                    sc["pp_dev"] = to_device(sc["pp_score"])
                items.append((sc["ptc_dev"], sc["pp_dev"], rs))
                who.append(i)
        if len(items) > 1 and len({id(it[2]) for it in items}) == len(items):   # one generator per scan: no draw order between scans
            for i, res in zip(who, ops.mask_stage_batch(items, _stage_params(args), ctxs=ctxs)):
                staged[i] = res
    # the box tail of the chain: scans whose stage result is known here (the library's) and that share one calibration
    boxed = [_UNSET] * len(scans)
    cand = [i for i in range(len(scans)) if staged[i] is not _UNSET and staged[i] is not None]
    if len(cand) > 1:
        c0 = calibs[cand[0]]
        same = [i for i in cand if np.array_equal(calibs[i].V2C, c0.V2C) and np.array_equal(calibs[i].R0, c0.R0)]
        items, who = [], []
        for i in same:
            labels_filtered = staged[i][0]
            n_lab = int(staged[i][3][2]) if labels_filtered.size else 0   # (info[2]: the largest final label = labels_filtered.max())
            if _native_boxes_ok(scans[i]["ptc"], n_lab, args):
                items.append((scans[i]["ptc_dev"], scans[i]["ptc"], labels_filtered, n_lab, staged[i][4] if len(staged[i]) > 4 else None))
                who.append(i)
        if len(items) > 1:
            from .utils.pointcloud_utils import _angles, _angles90
            ang, cs = _angles(0.1)
            res = ops.scan_boxes_batch(items, c0.V2C, c0.R0, ang, cs, _angles90(0.1), 1e-2, args.filtering.min_volume,
                                       args.filtering.max_volume, ctxs=None if ctxs is None else [ctxs[i] for i in who])
            for i, r in zip(who, res):
                boxed[i] = r
    return [generate_mask_scan(sc["ptc"], sc["pp_score"], cb, args, random_state=sc.get("random_state"),
                               ptc_dev=sc.get("ptc_dev"), pp_dev=sc.get("pp_dev"), as_rows=as_rows, staged=staged[i],
                               boxed=boxed[i])
            for i, (sc, cb) in enumerate(zip(scans, calibs))]


NATIVE_STAGE = True   # tests switch it off to compare the library's stage driver with the Python statement
NATIVE_BOXES = True   # ... and the library's box tail (modest_scan_boxes) with get_objs + the volume gate


def _stage_params(args) -> "ops.MaskParams":
    """the config keys of the stage as the library's parameter block.  Cached on the VALUES it is built
    from (a config node mutated in place -- a parameter sweep, a test changing clustering.DBSCAN.eps --
    must not run the native stage with the previous values while the host path reads the live ones)."""
    pe, g, f = args.plane_estimate, args.graph, args.filtering
    key = (pe.max_hs, repr(pe.range), pe.offset, repr(args.limit_range), g.neighbor_type, g.affinity_type,
           g.n_neighbors, g.radius, args.clustering.DBSCAN.min_samples, args.clustering.DBSCAN.eps,
           f.get("min_points", 10), f.get("max_min_height", 4), f.get("min_max_height", 0), f.get("percentile", 10),
           f.get("min_percentile_pp_score", 0.7))
    cached = _PARAMS.get("last")
    if cached is not None and cached[0] == key:
        return cached[1]
    P = ops.MaskParams()
    P.max_hs1 = pe.max_hs
    P.range1[:] = [pe.range[0][0], pe.range[0][1], pe.range[1][0], pe.range[1][1]]
    P.max_hs2 = FILTER_PLANE_SPEC[0]
    r2 = FILTER_PLANE_SPEC[1]
    P.range2[:] = [r2[0][0], r2[0][1], r2[1][0], r2[1][1]]
    P.offset = pe.offset
    P.use_only_range = 0 if pe.range is None else 1
    if pe.range is not None:
        P.only_range[:] = [pe.range[0][0], pe.range[0][1], pe.range[1][0], pe.range[1][1]]
    lim = np.asarray(args.limit_range, dtype=np.float64).reshape(4)
    P.limit_range[:] = list(lim)
    P.neighbor_type, P.affinity_type = ops.GRAPH_TYPES[g.neighbor_type], ops.AFFINITY_TYPES[g.affinity_type]
    P.k_neighbors, P.min_samples = int(g.n_neighbors), int(args.clustering.DBSCAN.min_samples)
    P.radius, P.eps = float(g.radius), float(args.clustering.DBSCAN.eps)
    P.min_points = int(f.get("min_points", 10))
    P.max_min_height, P.min_max_height = float(f.get("max_min_height", 4)), float(f.get("min_max_height", 0))
    P.quantile = float(np.true_divide(f.get("percentile", 10), np.float32(100)))
    P.min_percentile_pp_score = float(np.float32(f.get("min_percentile_pp_score", 0.7)))
    P.max_trials, P.batch, P.stop_probability = 100, 48, 0.99
    _PARAMS["last"] = (key, P)
    return P


_PARAMS = {}


def _mask_stage_host(ptc, pp_score, args, random_state, planes, ptc_dev, pp_dev):
    """The stage as separate calls with the interpreter in between (also: injected planes, integer
    seeds, the rare inputs the library hands back).  Returns (labels_filtered, plane, n_kept)."""
    pe, g = args.plane_estimate, args.graph
    # both ground fits of the scan (here and in filter_labels) need their candidates and MAD
    # thresholds before any random draw: selected and computed together, one launch for the two MADs
    prep = prepare_planes(ptc_dev, [(pe.max_hs, pe.range), FILTER_PLANE_SPEC]) if planes is None else (None, None)
    plane = planes[0] if planes is not None else estimate_plane(
        ptc_dev, max_hs=pe.max_hs, ptc_range=pe.range, random_state=random_state, prepared=prep[0])
    # mask, graph, DBSCAN and labels[ptc_mask] = ... in one device call (generate_mask.py:57-88)
    labels_dev, n_kept = ops.mask_cluster(ptc_dev, pp_dev, plane, pe.offset, pe.range, args.limit_range,
                                          g.n_neighbors, g.radius, args.clustering.DBSCAN.eps,
                                          args.clustering.DBSCAN.min_samples, neighbor_type=g.neighbor_type,
                                          affinity_type=g.affinity_type)
    labels = labels_dev.cpu().numpy().astype(int)
    labels_filtered = filter_labels(ptc, pp_score, labels, random_state=random_state,
                                    plane=None if planes is None else planes[1], ptc_dev=ptc_dev,
                                    pp_dev=pp_dev, labels_dev=labels_dev, plane_prepared=prep[1], **args.filtering)
    return labels_filtered, plane, n_kept


def _pooled(args, rank, ws, local):
    """workers=N: N child processes on this rank's GPU (dist.run_workers), same barrier + counter
    all-reduce around them as around the in-process loop."""
    dist.barrier()
    t0 = time.perf_counter()
    tot = dist.run_workers("modest_amd.generate_mask", args, rank, ws, local)
    dist.barrier()
    tot["max_worker_seconds"] = tot.get("max_seconds", 0.0)   # the workers' own loop clocks (no start-up)
    tot["max_seconds"] = time.perf_counter() - t0
    tot = dist.reduce_counters(tot)
    if rank == 0:
        eprint("[generate_mask] %d scans, %.2f s, %.2f scans/s on %d GPU(s) x %d worker processes"
               % (tot["scans"], tot["max_seconds"], tot["scans"] / max(tot["max_seconds"], 1e-9), ws, int(args.workers)))
    return tot


@config.main(config_name="generate_mask.yaml")
def main(args):
    rank, ws, local = dist.init(poll_wait=bool(args.get("poll_wait", True)))
    if rank == 0:
        display_args(args)
    torch.cuda.set_device(torch.device("cuda", dist.device_index(local, ws, args)))
    dp = args.data_paths
    pooled = int(args.get("workers", 1) or 1) > 1 and not os.environ.get("MODEST_WORKER")
    idx_list = np.array([int(x) for x in open(dp.idx_list).readlines()])
    shard = dist.scans_of(idx_list, args, rank, ws, "mask")
    os.makedirs(dp.seg_save_dst, exist_ok=True)
    if rank == 0 and not osp.exists(osp.join(dp.seg_save_dst, "configs.yaml")):
        config.save(config=args, f=osp.join(dp.seg_save_dst, "configs.yaml"))
    bbox_dst = dp.get("bbox_info_save_dst", "None")
    if bbox_dst is not None:
        os.makedirs(bbox_dst, exist_ok=True)
        if rank == 0 and not osp.exists(osp.join(bbox_dst, "configs.yaml")):
            config.save(config=args, f=osp.join(bbox_dst, "configs.yaml"))
    if pooled:
        return _pooled(args, rank, ws, local)
    seed = int(args.get("ransac_seed", 0))
    _lib.default_context(torch.cuda.current_device()).warmup()   # (the library's device code: loaded before the clock)
    t0, done = time.perf_counter(), 0
    dist.barrier()
    # mask_batch scans go through the stage as ONE chain of kernel launches (generate_mask_chain); the reads of a
    # chain precede its device work, outputs are written scan by scan
    n_batch = max(1, int(args.get("mask_batch", 4)))
    pend = []

    def flush():
        if not pend:
            return
        if len(pend) == 1:
            q = pend[0]
            res = [generate_mask_scan(q["ptc"], q["pp_score"], q["calib"], args, random_state=q["random_state"])]
        else:
            res = generate_mask_chain(pend, [q["calib"] for q in pend], args)
        for q, (labels, objs, _) in zip(pend, res):
            if bbox_dst is not None:
                pickle.dump(objs, open(osp.join(bbox_dst, f"{q['idx']:06d}.pkl"), "wb"))
            np.save(osp.join(dp.seg_save_dst, f"{q['idx']:06d}.npy"), labels)
        pend.clear()

    for idx in shard:
        idx = int(idx)
        if osp.exists(osp.join(dp.seg_save_dst, f"{idx:06d}.npy")) and \
                (bbox_dst is None or osp.exists(osp.join(bbox_dst, f"{idx:06d}.pkl"))):
            continue
        pend.append(dict(idx=idx, ptc=load_velo_scan(osp.join(args.ptc_path, f"{idx:06d}.bin")),
                         pp_score=np.load(osp.join(dp.pp_score_path, f"{idx:06d}.npy")),
                         calib=kitti_util.Calibration(osp.join(args.calib_path, f"{idx:06d}.txt")),
                         random_state=np.random.RandomState(seed + idx)))
        done += 1
        if len(pend) >= n_batch:
            flush()
    flush()
    torch.cuda.synchronize()
    tot = dist.rank_report("generate_mask", done, t0, rank, ws)
    if rank == 0:
        eprint("[generate_mask] %d scans, %.2f s, %.2f scans/s on %d GPU(s)"
               % (tot["scans"], tot["max_seconds"], tot["scans"] / max(tot["max_seconds"], 1e-9), ws))
    return tot


if __name__ == "__main__":
    main()
