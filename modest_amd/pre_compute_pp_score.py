"""``python -m modest_amd.pre_compute_pp_score data_root=...`` -- PP-score CLI.

Drop-in for the reference's ``generate_cluster_mask/pre_compute_pp_score.py``
(same config keys, same ``pp_score_path/NNNNNN.npy`` float32 outputs).  What
changes is where the work happens: raw ``.bin`` frames are uploaded once into
the frame store (``frame_store.py``: tile-sorted, resident in HBM, LRU), a scan
names its history frames through a descriptor table carrying the reference's
float32 relative poses (one batched ``np.linalg.solve`` per scan), and the
neighbour-count kernels read the frames through that table with
``transform_points`` + ``remove_center`` fused (reference :132-150) -- no
stacked history is built; the KD-tree build + ball query of the reference
(:188-193) is replaced by the streamed neighbour count.
"""
from __future__ import annotations

import os
import os.path as osp
import pickle
import sys
import time

import numpy as np
import torch
from scipy.spatial.transform import Rotation as R

from . import _lib, config, dist, ops
from .frame_store import FrameStore
from .utils.pointcloud_utils import load_velo_scan


TRACE = [] if os.environ.get("MODEST_PP_TRACE") == "2" else None


def _tr(tag, k):
    if TRACE is not None:
        TRACE.append((tag, k, time.perf_counter()))


def eprint(*args, **kwargs):
    print(*args, file=sys.stderr, **kwargs)


def _rot_z(angle):
    """``Quaternion(axis=(0,0,1), angle).transformation_matrix`` of the reference
    (:22-24) restated: unit quaternion (cos a/2, 0, 0, sin a/2) -> rotation matrix."""
    w, z = np.cos(angle / 2.0), np.sin(angle / 2.0)
    t = np.eye(4)
    t[0, 0] = w * w - z * z
    t[0, 1] = -2.0 * w * z
    t[1, 0] = 2.0 * w * z
    t[1, 1] = w * w - z * z
    return t


_KITTI2NU_lyft = _rot_z(np.pi)
_KITTI2NU_nusc = _rot_z(np.pi / 2)


def get_relative_pose(fixed_l2e, fixed_ego, query_l2e, query_ego, KITTI2NU=_KITTI2NU_lyft):
    """(:27-28) K^-1 L_f^-1 E_f^-1 E_q L_q K as three nested solves, float32 result."""
    return np.linalg.solve(KITTI2NU, np.linalg.solve(fixed_l2e, np.linalg.solve(
        fixed_ego, query_ego @ query_l2e @ KITTI2NU))).astype(np.float32)


def load_poses(track_list, oxts_path, l2e_path):
    """(:92-106) oxts "x y z roll pitch yaw" -> 4x4 float32 ego pose; l2e 4x4 .npy."""
    poses, l2es = [], []
    for seq in track_list:
        poses.append([])
        l2es.append([])
        for idx in seq:
            with open(osp.join(oxts_path, f"{idx:06d}.txt"), "r") as f:
                info = np.array([float(x) for x in f.readline().split()])
            trans = np.eye(4)
            trans[:3, 3] = info[:3]
            trans[:3, :3] = R.from_euler("xyz", info[3:]).as_matrix()
            poses[-1].append(trans.astype(np.float32))
            l2es[-1].append(np.load(osp.join(l2e_path, f"{idx:06d}.npy")))
    return poses, l2es


class FrameLoader:
    """Feeds the frame store (modest_amd/frame_store.py): a `.bin` frame is read, uploaded and
    tile-sorted once, then serves every scan that names it (a Lyft frame is history of ~70 scans).
    A Lyft training split is ~12.7 k frames x ~0.55 MB: it fits in one MI355X many times over.

    The missing frames of a scan travel as ONE batch: reader threads fill one pinned buffer (readinto
    releases the GIL), one asynchronous copy takes it to the device, one launch sorts all of them."""

    def __init__(self, velodyne_dir, store: FrameStore, world, readers: int = 4, ctx=None, frame_bytes: int = 0):
        from . import _lib
        self.dir, self.store, self.world, self.ctx = velodyne_dir, store, world, ctx
        self.lib, self.readers = _lib.load(), max(1, int(readers))
        self.chunk = 32   # frames per staged piece
        self.slots = 3    # ring: a slot is refilled only after its last upload AND the sort that read it have run
        self.turn = 0
        self.read_bytes = 0
        self.t_read = self.t_insert = self.t_touch = 0.0   # seconds of the ingest thread, by phase
        self.n_frames = 0
        # Staging, host (pinned) and device, taken ONCE here -- outside every loop clock, in two driver calls: a cold scan
        # brings 361 frames in twelve pieces, and with the buffers allocated on first use 145 of its 182 ms went into
        # hipHostMalloc / hipMalloc (three pinned buffers, twelve device blocks; under eight workers those calls serialise).
        # `frame_bytes`: size of a typical .bin (0: the buffers are made on first use, as before).
        self.cap = 0
        self.pinned_all = self.dev_all = None
        self.slot_ev = [None] * self.slots
        if frame_bytes > 0:
            self._make_staging(int(self.chunk * (frame_bytes // 16) * 1.15) + 1024)

    def _make_staging(self, points: int):
        """(re)allocate the ring for pieces of up to `points` points: one pinned and one device tensor, `slots` views each"""
        for ev in self.slot_ev:
            if ev is not None:
                ev.synchronize()
        self.cap = int(points)
        self.pinned_all = torch.empty((self.slots, self.cap, 4), dtype=torch.float32, pin_memory=True)
        self.dev_all = torch.empty((self.slots, self.cap, 4), dtype=torch.float32, device=self.store.device)
        self.slot_ev = [None] * self.slots

    def _read_batch(self, ids):
        """the .bin files of `ids` -> one pinned float32 buffer; returns (slot, buffer (P,4), point offsets).  One library call
        (modest_host_read_files: stat + read by `readers` host threads, interpreter lock released for the whole of it)."""
        import ctypes as C
        n = len(ids)
        paths = (C.c_char_p * n)(*[osp.join(self.dir, f"{i:06d}.bin").encode() for i in ids])
        sizes = np.zeros(n, dtype=np.uint64)
        if self.pinned_all is None:   # (no staging yet: the size probe failed)
            need = int(self.lib.modest_host_read_files(paths, n, None, 0, sizes.ctypes.data, self.readers))
            if need < 0:
                raise IOError(f"cannot read {paths[-need - 2].decode()}" if need < -1 else "modest_host_read_files: bad arguments")
            self._make_staging(max(need // 16 + need // 128, 1 << 16))
        r = self.turn = (self.turn + 1) % self.slots
        while True:
            if self.slot_ev[r] is not None:
                self.slot_ev[r].synchronize()
            rc = int(self.lib.modest_host_read_files(paths, n, self.pinned_all[r].data_ptr(), self.cap * 16, sizes.ctypes.data, self.readers))
            if rc == 0:
                break
            if rc < 0:
                raise IOError(f"cannot read {paths[-rc - 2].decode()}" if rc < -1 else "modest_host_read_files: bad arguments")
            self._make_staging(rc // 16 + rc // 128 + 1024)   # the piece needs a larger ring (rc bytes): rebuild it and read again
            r = self.turn
        assert not (sizes % 16).any(), "velodyne .bin files hold (n,4) float32 rows"
        offs = np.concatenate([[0], np.cumsum(sizes // 16)]).astype(np.int64)
        need = int(offs[-1])
        self.read_bytes += int(sizes.sum())
        return r, self.pinned_all[r][:need], offs

    def ensure(self, file_ids, protect=None, blocking=True):
        """Read + upload + sort (one copy, one launch) every frame of `file_ids` that is not resident, on
        the CURRENT stream.  blocking=False: upload and sort are only enqueued (work ordered behind them on the
        stream, or behind an event recorded after this call, sees the frames)."""
        t0 = time.perf_counter()
        todo = self.store.missing(file_ids)
        # a cold scan (hundreds of frames) goes in pieces: the pinned staging buffers stay small (pinning
        # 170 MB at once takes tens of milliseconds inside the driver, serialised across worker processes)
        # and the first piece is uploading while the next one is read
        for c0 in range(0, len(todo), self.chunk):
            missing = todo[c0:c0 + self.chunk]
            r, host, offs = self._read_batch(missing)
            t1 = time.perf_counter()
            dev = self.dev_all[r][:host.shape[0]]
            dev.copy_(host, non_blocking=True)
            if blocking:
                items = [(i, dev[offs[k]:offs[k + 1]], self.world[i]) for k, i in enumerate(missing)]
                self.store.insert_many(items, ctx=self.ctx, protect=protect if protect is not None else file_ids)
            else:
                self.store.insert_block(missing, dev, offs, self.world.stack(missing), ctx=self.ctx,
                                        protect=protect if protect is not None else file_ids)
            ev = torch.cuda.Event()
            ev.record()   # behind the copy AND the sort that reads the device slot
            self.slot_ev[r] = ev
            t2 = time.perf_counter()
            self.t_read += t1 - t0
            self.t_insert += t2 - t1
            if os.environ.get("MODEST_PP_TRACE") and self.n_frames < 400:
                eprint("[pp_score ingest] %d frames: read %.2f ms, copy + insert %.2f ms" % (len(missing), 1e3 * (t1 - t0), 1e3 * (t2 - t1)))
            self.n_frames += len(missing)
            t0 = t2
        self.store.touch(file_ids)   # LRU order + hit statistics
        self.t_touch += time.perf_counter() - t0


class IngestPipeline:
    """The frames of the next `depth` scans load under the kernels of the current one: a background thread
    pulls scan plans from `plans` (an iterator: the shard may be a dynamic work queue), makes their frames
    resident (FrameLoader.ensure on its own stream and library context) and hands the plan over together with
    a HIP event; iterating the pipeline yields the plans in order once the compute stream waits on that event;
    `done()` releases the window behind a finished scan.  Frames named by a scan inside the window are never
    evicted."""

    def __init__(self, loader: FrameLoader, plans, device, depth: int = 4, own_stream: bool = True, stream=None):
        import collections
        import queue
        import threading
        self.loader, self.plans, self.depth, self.device = loader, plans, max(1, int(depth)), device
        # own_stream=False: uploads and sorts are enqueued on the compute stream itself (the hand-over of a plan
        # already orders them before the scan's kernels).  A GPU shared by several worker processes has 8
        # hardware queues in all: a second stream per worker halves everybody's rate.
        self.stream = stream if stream is not None else (torch.cuda.Stream(device=device) if own_stream else torch.cuda.current_stream(device))
        self.window = threading.Semaphore(self.depth)
        self.inflight = collections.deque()
        self.lock = threading.Lock()
        self.q = queue.Queue()
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _run(self):
        try:
            torch.cuda.set_device(self.device)
            group = max(1, int(os.environ.get("MODEST_INGEST_GROUP", "4")))
            it, held, first = iter(self.plans), None, True
            with torch.cuda.stream(self.stream):
                while True:
                    # up to `group` scans whose window slots are free RIGHT NOW travel as one read round + one copy + one sort
                    # launch (a scan's 11 new files alone keep four reader threads busy for a fraction of the round trip through
                    # the pool); the first scan of a group waits for its slot, the others only join if theirs is free.  The
                    # first scan of the process goes alone: it is the cold one.
                    batch = []
                    while len(batch) < (1 if first else group):
                        plan = held if held is not None else next(it, None)
                        held = None
                        if plan is None:
                            break
                        _tr("L.want", plan["origin"])
                        if not batch:
                            self.window.acquire()
                        elif not self.window.acquire(blocking=False):
                            held = plan
                            break
                        batch.append(plan)
                    if not batch:
                        break
                    first = False
                    with self.lock:
                        for plan in batch:
                            self.inflight.append(plan["frames"])
                        protect = [i for ids in self.inflight for i in ids]
                    _tr("L.start", batch[0]["origin"])
                    self.loader.ensure([i for plan in batch for i in plan["frames"]], protect=protect, blocking=False)
                    ev = torch.cuda.Event()
                    ev.record(self.stream)
                    for plan in batch:
                        _tr("L.done", plan["origin"])
                        self.q.put((plan, ev))
            self.q.put(None)
        except BaseException as e:   # surfaced by the iterator
            self.q.put(e)

    def __iter__(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            plan, ev = item
            torch.cuda.current_stream().wait_event(ev)
            _tr("M.got", plan["origin"])
            yield plan

    def done(self):
        with self.lock:
            self.inflight.popleft()
        self.window.release()


class OutputWriter:
    """Scores leave through a ring of pinned host buffers: an asynchronous copy + event per scan on the compute
    stream, a writer thread that waits for the event and writes the .npy (atomically).  The compute loop
    never blocks on a read-back or on the file system."""

    def __init__(self, n_max: int, slots: int = 8):
        import queue
        import threading
        self.buf = [torch.empty((max(n_max, 1),), dtype=torch.float32, pin_memory=True) for _ in range(slots)]
        self.free = queue.Queue()
        for k in range(slots):
            self.free.put(k)
        self.q = queue.Queue()
        self.error = None
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _run(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            ev, slot, n, path = item
            try:
                ev.synchronize()
                _tr("W.gpu", path[-10:-4])
                save_npy_atomic(path, self.buf[slot][:n].numpy())
                _tr("W.saved", path[-10:-4])
            except BaseException as e:
                self.error = e
            self.free.put(slot)

    def submit(self, H: torch.Tensor, path: str):
        n = int(H.shape[0])
        slot = self.free.get()
        if self.buf[slot].shape[0] < n:
            self.buf[slot] = torch.empty((n,), dtype=torch.float32, pin_memory=True)
        self.buf[slot][:n].copy_(H, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.q.put((ev, slot, n, path))

    def close(self):
        self.q.put(None)
        self.thread.join()
        if self.error is not None:
            raise self.error


class WorldTable(dict):
    """file id -> W_f; `.stack(ids)` gathers (F,4,4) with one fancy index when the ids are small integers"""

    def freeze(self):
        ids = [k for k in self if isinstance(k, (int, np.integer))]
        self.arr = None
        if ids and len(ids) == len(self) and min(ids) >= 0 and max(ids) < 4 * len(ids) + 1024:
            self.arr = np.zeros((max(ids) + 1, 4, 4))
            for k, v in self.items():
                self.arr[k] = v
        return self

    def stack(self, ids):
        if getattr(self, "arr", None) is not None:
            return self.arr[np.asarray(ids, dtype=np.int64)]
        return np.stack([self[i] for i in ids])


def frame_world_matrices(track_list, poses, l2es, K):
    """W_f = E_f @ L_f @ K for every frame -- the right-hand factor of get_relative_pose (:27-28),
    the same three-matrix product in the same order -- computed once: it does not depend on the scan."""
    world = WorldTable()
    for seq, ps, ls in zip(track_list, poses, l2es):
        for idx, E, L in zip(seq, ps, ls):
            world[idx] = E @ L @ K
    return world.freeze()


def relative_poses(fixed_l2e, fixed_ego, world_stack, K):
    """get_relative_pose (:27-28) for a stack of frames: the three nested solves run once, with the 4 F
    columns of the (F,4,4) stack of E_q @ L_q @ K products as right-hand sides of ONE gesv each -- the same
    LU factors and the same forward / back substitution per column as F separate calls (bit-identical:
    tests/test_abi_and_host.py), without F factorisations of the same 4x4 matrix (0.46 -> 0.19 ms for 361)."""
    F = world_stack.shape[0]
    B = np.ascontiguousarray(world_stack.transpose(1, 0, 2).reshape(4, 4 * F))
    X = np.linalg.solve(K, np.linalg.solve(fixed_l2e, np.linalg.solve(fixed_ego, B)))
    return X.reshape(4, F, 4).transpose(1, 0, 2).astype(np.float32)


_POSE_POOL = None


def relative_poses_block(fixed_l2es, fixed_egos, world_stacks, K, threads: int = 4):
    """relative_poses for the scans of a block, a few scans per thread: the 4x4 solves against 4 F right-hand sides are
    ~90 us of LAPACK per scan (dgetrs over 1 444 columns; numpy's stacked solve runs them one after another), LAPACK releases
    the interpreter lock, and a block of 16 scans is 1.5 ms on one thread.  Per scan exactly relative_poses (bit-identical
    by construction).  fixed_l2es / fixed_egos: sequences of (4,4); world_stacks: sequence of (F_i,4,4) -> [(F_i,4,4) float32]."""
    global _POSE_POOL
    n = len(world_stacks)
    if n <= 2 or threads <= 1:
        return [relative_poses(fixed_l2es[i], fixed_egos[i], world_stacks[i], K) for i in range(n)]
    if _POSE_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POSE_POOL = ThreadPoolExecutor(max_workers=threads)
    return list(_POSE_POOL.map(lambda i: relative_poses(fixed_l2es[i], fixed_egos[i], world_stacks[i], K), range(n)))


def save_npy_atomic(path, arr):
    """np.save through a temporary file + os.replace: a killed rank never leaves a truncated .npy
    that the skip-if-exists test would count as done."""
    tmp = f"{path}.tmp{os.getpid()}.npy"
    np.save(tmp, arr)
    os.replace(tmp, path)


def display_args(args):
    eprint("========== ephemerality info ==========")
    eprint("host: {}".format(os.getenv("HOSTNAME")))
    eprint(config.to_yaml(args))
    eprint("=======================================")


def _pooled(args, rank, ws, local):
    """workers=N: N child processes on this rank's GPU (dist.run_workers), same barrier + counter
    all-reduce around them as around the in-process loop."""
    dist.barrier()
    t0 = time.perf_counter()
    tot = dist.run_workers("modest_amd.pre_compute_pp_score", args, rank, ws, local)
    dist.barrier()
    tot["max_worker_seconds"] = tot.get("max_seconds", 0.0)   # the workers' own loop clocks (no start-up)
    tot["max_seconds"] = time.perf_counter() - t0
    tot = dist.reduce_counters(tot)
    if rank == 0:
        eprint("[pp_score] %d scans, %.2f s, %.2f scans/s on %d GPU(s) x %d worker processes"
               % (tot["scans"], tot["max_seconds"], tot["scans"] / max(tot["max_seconds"], 1e-9), ws, int(args.workers)))
    return tot


def run(args, post=None, module="modest_amd.pre_compute_pp_score"):
    """The PP-score CLI's body.  `post` (modest_amd/seed_labels.py: the fused mode) is an object with
    done(idx) -> bool (every output of the scan exists: skip it), __call__(batch) (batch = [(idx, live file id, H device tensor)] of one
    flushed PP batch, in plan order) and close(); None: the reference's stage-by-stage pipeline."""
    rank, ws, local = dist.init(poll_wait=bool(args.get("poll_wait", True)))
    if rank == 0:
        display_args(args)
    device = torch.device("cuda", dist.device_index(local, ws, args))
    torch.cuda.set_device(device)
    dp = args.data_paths
    if int(args.get("workers", 1) or 1) > 1 and not os.environ.get("MODEST_WORKER") and post is None:   # (the fused CLI pools its own workers)
        os.makedirs(dp.pp_score_path, exist_ok=True)
        return _pooled(args, rank, ws, local)
    track_list = pickle.load(open(dp.track_path, "rb"))
    valid_idx = pickle.load(open(dp.idx_info, "rb"))
    os.makedirs(dp.pp_score_path, exist_ok=True)
    poses, l2es = load_poses(track_list, osp.join(args.data_root, "oxts"), osp.join(args.data_root, "l2e"))
    if dp.idx_list is not None:
        idx_list = [int(x) for x in open(dp.idx_list).readlines()]
    else:
        idx_list = [x for x in valid_idx]
    shard = dist.scans_of(idx_list, args, rank, ws, "pp")
    for d in (dp.load_save_precomputed_trans_mat, dp.load_precomputed_lidars):
        if d is not None:
            os.makedirs(d, exist_ok=True)
    K = _KITTI2NU_nusc if args.nusc else _KITTI2NU_lyft
    if args.ephe_type != "entropy":
        raise NotImplementedError(args.ephe_type)
    radius = float(args.max_neighbor_dist)
    store = FrameStore(device, radius, float(args.get("frame_cache_gb", 64)) * 2 ** 30)
    # device memory for the frames this process will hold, taken from the driver in one call (frame_store.py:
    # reserve): every frame of the data set, bounded by the cache size and by frame_prealloc_gb (0: no reservation,
    # slabs of 256 MB on demand)
    try:
        n_all = sum(len(seq) for seq in track_list)
        try:   # ... of this process's share of the scans (a worker of workers=N holds an N-th of the data set, not all of it)
            mine = getattr(shard, "mine", None)   # (a static share; a dynamic queue hands out chunks: reserve for the data set)
            if mine is not None and 0 < len(mine) <= 4096:
                named = set()
                for o in mine:
                    s0, f0, trav = valid_idx[int(o)]
                    named.add(track_list[s0][f0])
                    for sq, ix in trav:
                        named.update(track_list[sq][f] for f in ix)
                n_all = min(n_all, len(named) + 64)
        except (TypeError, KeyError, IndexError):
            pass
        per = os.path.getsize(osp.join(args.data_root, "velodyne", f"{track_list[0][0]:06d}.bin")) + 4 * (store.ntf ** 2 + 1) + 512
        store.reserve(min(n_all * per * 1.05, float(args.get("frame_prealloc_gb", 16.0)) * 2 ** 30))
    except (OSError, IndexError):
        pass
    world = frame_world_matrices(track_list, poses, l2es, K)
    # ingest: reader threads -> pinned buffer -> one copy + one sort launch per scan, `ingest_depth` scans ahead
    # of the kernels (own stream, own library context); scores leave through a writer thread
    try:
        frame_bytes = os.path.getsize(osp.join(args.data_root, "velodyne", f"{track_list[0][0]:06d}.bin"))
    except (OSError, IndexError):
        frame_bytes = 0
    loader = FrameLoader(osp.join(args.data_root, "velodyne"), store, world, readers=int(args.get("ingest_readers", 0) or 0) or (2 if os.environ.get("MODEST_WORKER") else 4),
                         ctx=_lib.Context(device.index or 0), frame_bytes=frame_bytes)

    def plans():
        for origin_idx in shard:
            origin_idx = int(origin_idx)
            out_path = osp.join(dp.pp_score_path, f"{origin_idx:06d}.npy")
            # the reference tests the name without ".npy" (:123-124) and so never skips; here finished
            # scans are skipped unless overwrite=True (required after changing max_neighbor_dist,
            # limit_traversals or add_random_noise: the outputs carry no config hash)
            if osp.exists(out_path) and not args.get("overwrite", False) and (post is None or post.done(origin_idx)):
                continue
            traversals = valid_idx[origin_idx][2]
            assert len(traversals) > 1, origin_idx
            origin_seq, origin_frame = valid_idx[origin_idx][0], valid_idx[origin_idx][1]
            live_id = track_list[origin_seq][origin_frame]
            # history frames of the scan (:132-150): file ids and traversal index
            hist_ids, travs = [], []
            for t, (seq_id, indices) in enumerate(traversals):
                for frame in indices:
                    hist_ids.append(track_list[seq_id][frame])
                    travs.append(t)
            yield dict(origin=origin_idx, out=out_path, traversals=traversals, live=live_id, hist=hist_ids, travs=travs,
                       frames=hist_ids + [live_id])

    # the scratch arena of a full batch on the block path, taken once, before the clock: the first batch of a process is short
    # (below) and a later grow is a device synchronise + free + allocate -- 100 ms alone on the GPU, several times that under
    # eight workers.  Estimate: 32 B per point of the batch's union of frames (two copies of the 16-byte records) + ~45 MB per scan
    # (live index over the block window, task lists, counts); an underestimate only means the arena grows as before.
    n_batch = max(1, int(args.get("pp_batch", 32)))
    _lib.default_context(device.index or 0).warmup()   # (device code of the library: loaded before the clock, not inside the first scan)
    try:
        mine = getattr(shard, "mine", getattr(shard, "items", []))
        first = valid_idx[int(mine[0])][2] if len(mine) else []
        if first and n_batch > 1 and not args.skip_ephe:
            per_frame = frame_bytes // 16
            union_frames = sum(len(ix) + n_batch - 1 for _, ix in first)
            want = int(1.1 * 32 * per_frame * union_frames) + n_batch * (45 << 20)
            # an optimisation must not be able to end the run: never more than this worker's share of what the device has free
            # (the library adds 25 % head room), and a refused allocation only means the arena grows on demand as before
            free_b, _total = torch.cuda.mem_get_info(device)
            n_workers = max(1, int(os.environ.get("MODEST_WORKER", "0/1").split("/")[-1]))
            want = min(want, int(0.5 * free_b / n_workers / 1.25))
            _lib.default_context(device.index or 0).reserve_arena(want)
    except (KeyError, IndexError, TypeError, ValueError, RuntimeError) as e:   # (RuntimeError: _lib.ModestHipError, torch)
        if os.environ.get("MODEST_ALLOC_TRACE"):
            eprint("[pp_score] arena reservation skipped: %r" % (e,))
    # the ingest stream exists, and has carried one copy, before the clock: the FIRST host-to-device copy of a stream sets up
    # its hardware queue and the copy engine's signals (142 ms of the 170 ms a cold scan's ingest took on the host)
    ingest_stream = torch.cuda.current_stream(device) if os.environ.get("MODEST_WORKER") else torch.cuda.Stream(device=device)
    if loader.pinned_all is not None:
        with torch.cuda.stream(ingest_stream):
            loader.dev_all[0][:64].copy_(loader.pinned_all[0][:64], non_blocking=True)
        ingest_stream.synchronize()
    t0, done, pts = time.perf_counter(), 0, 0
    trace = bool(os.environ.get("MODEST_PP_TRACE")) and os.environ.get("MODEST_WORKER", "0/1").startswith("0/")
    dist.barrier()
    # the ingest / writer threads hand the GIL back and forth with this loop: CPython's default forced-switch
    # interval (5 ms) is longer than a whole scan
    sys.setswitchinterval(float(os.environ.get("MODEST_SWITCH_INTERVAL", "0.0002")))
    # pp_batch consecutive scans go through ONE call (FrameStore.pp_score_batch: modest_pp_score_block for scans that share
    # their history frames, modest_pp_score_frames_batch otherwise): the loop below collects their descriptor tables and
    # flushes; the ingest window must hold a whole batch plus the scans ahead -- at least one scan ahead, whatever
    # ingest_depth says (a window smaller than a batch would leave this loop waiting for a scan the ingest thread may
    # not load)
    pipe = IngestPipeline(loader, plans(), device, depth=max(1, int(args.get("ingest_depth", 36))) + (n_batch - 1),
                          stream=ingest_stream)
    # (a slot per scan of a batch + the writer's backlog: with fewer slots than scans per PP call the loop waited in submit() for the
    # batch's own kernels, and the ingest window behind it stayed closed for as long)
    writer = OutputWriter(1 << 16, slots=int(os.environ.get("MODEST_OUT_SLOTS", str(n_batch + 8))))
    pend = []   # the scans waiting for the flush: (live frame, history ids, traversal of every frame, raw pose factors of the
    #             history frames + the live scan, fixed l2e, fixed ego, output path, scan id, traversals)
    pose_threads = max(1, int(args.get("pose_threads", 4)))
    flushed = False
    ph = dict(wait=0.0, poses=0.0, tables=0.0, pp=0.0, submit=0.0, post=0.0)   # seconds of THIS thread, by phase (the summary line)

    def flush():
        if not pend:
            return
        f0 = time.perf_counter()
        # the relative poses of the whole batch (get_relative_pose :27-28 per frame; a few scans per thread) and its descriptor
        # tables (one gather from the store's slot tables) -- then ONE PP call
        rels = relative_poses_block([q[4] for q in pend], [q[5] for q in pend], [q[3] for q in pend], K, threads=pose_threads)
        f1 = time.perf_counter()
        descs = store.describe_many([q[0] for q in pend], [r[-1] for r in rels], [q[1] for q in pend], [q[2] for q in pend],
                                    [r[:-1] for r in rels], bool(args.nusc))
        # (a scan's own number of traversals: the reference accepts a traversal per scan, data_preprocessing/lyft/split_traintest.py:79,111,
        # so T changes along a sequence -- a block takes the scans as they come)
        f2 = time.perf_counter()
        Hs = store.pp_score_batch([q[0] for q in pend], descs, [q[8] for q in pend])
        f3 = time.perf_counter()
        for q, H in zip(pend, Hs):
            _tr("M.enq", q[7])
            writer.submit(H, q[6])
            _tr("M.sub", q[7])
        f4 = time.perf_counter()
        if post is not None:   # fused mode: stages 2 + 3 of the batch (deferred by one batch: they run under the NEXT batch's PP kernels)
            post([(q[7], q[0], H) for q, H in zip(pend, Hs)])
        f5 = time.perf_counter()
        for k, v in zip(("poses", "tables", "pp", "submit", "post"), (f1 - f0, f2 - f1, f3 - f2, f4 - f3, f5 - f4)):
            ph[k] += v
        for _ in pend:
            pipe.done()
        pend.clear()

    pipe_it = iter(pipe)
    while True:
        w0 = time.perf_counter()
        plan = next(pipe_it, None)
        ph["wait"] += time.perf_counter() - w0
        if plan is None:
            break
        origin_idx, out_path, traversals = plan["origin"], plan["out"], plan["traversals"]
        live_id, hist_ids, travs = plan["live"], plan["hist"], plan["travs"]
        first_seq, first_indices = traversals[0]
        first_pose, first_l2e = poses[first_seq][first_indices[0]], l2es[first_seq][first_indices[0]]
        n_trav = len(traversals)
        if args.limit_traversals > 1:   # (:181-186)
            n_trav = min(n_trav, int(args.limit_traversals))
        dumps = dp.load_save_precomputed_trans_mat is not None or dp.load_precomputed_lidars is not None
        batched = not (args.add_random_noise > 0) and n_trav <= 64 and not dumps and not args.skip_ephe
        if batched:   # the default path: poses and tables wait for the flush
            keep = [k for k, t in enumerate(travs) if t < n_trav] if n_trav < len(traversals) else None
            h_ids = hist_ids if keep is None else [hist_ids[k] for k in keep]
            h_tr = travs if keep is None else [travs[k] for k in keep]
            pend.append((live_id, h_ids, h_tr, world.stack(h_ids + [live_id]), first_l2e, first_pose, out_path, origin_idx, n_trav))
            done += 1
            pts += store.points_of(h_ids)
            # the FIRST batch of a process is short: its kernels start once four scans' files are in, not sixteen (a cold
            # worker reads 361 frames for its first scan alone); four is where the block path starts to pay
            if len(pend) >= (n_batch if flushed else min(4, n_batch)):
                flush()
                flushed = True
            if trace and (done <= 4 or done % 8 == 0):
                eprint("[pp_score trace] scan %d submitted at %.1f ms" % (done, 1e3 * (time.perf_counter() - t0)))
            continue
        flush()   # scans leave the ingest window in plan order
        rels = relative_poses(first_l2e, first_pose, world.stack(plan["frames"]), K)
        trans_mat = rels[-1]
        if dp.load_save_precomputed_trans_mat is not None:
            np.save(osp.join(dp.load_save_precomputed_trans_mat, f"{origin_idx:06d}.npy"), trans_mat)
        if dp.load_precomputed_lidars is not None:   # (:152-155) dump of the stacked, transformed history
            combined = {}
            for t, (sq, _) in enumerate(traversals):
                parts = [ops.transform_points(store.frames[hist_ids[k]].original_order(), rels[k],
                                              remove_center=bool(args.nusc)).cpu().numpy()
                         for k, tt in enumerate(travs) if tt == t]
                combined[sq] = np.concatenate(parts) if parts else np.zeros((0, 3), np.float32)
            pickle.dump(combined, open(osp.join(dp.load_precomputed_lidars, f"{origin_idx:06d}.pkl"), "wb"))
        if args.skip_ephe:
            pipe.done()
            continue
        keep = [k for k, t in enumerate(travs) if t < n_trav]
        hist = [(hist_ids[k], travs[k]) for k in keep]
        if args.add_random_noise > 0:            # (:175-179) host draw, same numpy expressions; stacked path
            noise = np.random.randn(3)
            noise /= np.linalg.norm(noise)
            noise *= (args.add_random_noise * np.random.uniform())
            live_np = ops.transform_points(store.frames[live_id].original_order(), trans_mat).cpu().numpy()
            live_np += noise.reshape(-1, 3)
            parts = [[] for _ in range(n_trav)]
            for k in keep:
                parts[travs[k]].append(ops.transform_points(store.frames[hist_ids[k]].xyz, rels[k],
                                                            remove_center=bool(args.nusc)))
            offsets = np.cumsum([0] + [sum(int(q.shape[0]) for q in tp) for tp in parts]).astype(np.int64)
            H = ops.pp_score(torch.from_numpy(live_np).to(device), torch.cat([q for tp in parts for q in tp]),
                             offsets, radius)
        else:
            A44 = first_pose.astype(np.float64) @ np.asarray(first_l2e, dtype=np.float64) @ K
            H = store.pp_score(live_id, trans_mat, hist, rels[keep], A44, n_trav, remove_center=bool(args.nusc))
        _tr("M.enq", origin_idx)
        writer.submit(H, out_path)
        if post is not None:
            post([(origin_idx, live_id, H)])
        _tr("M.sub", origin_idx)
        done += 1
        if trace and (done <= 4 or done % 8 == 0):
            eprint("[pp_score trace] scan %d submitted at %.1f ms" % (done, 1e3 * (time.perf_counter() - t0)))
        pts += store.points_of([i for i, _ in hist])
        pipe.done()
    flush()
    if post is not None:
        post.close()
    writer.close()
    if TRACE is not None:
        base = TRACE[0][2]
        for tag, k, t in TRACE:
            if int(os.environ.get("MODEST_PP_TRACE_FROM", "150")) <= int(k) < int(os.environ.get("MODEST_PP_TRACE_FROM", "150")) + 4:
                eprint("[pp_score trace2] %-8s scan %s at %.3f ms" % (tag, k, 1e3 * (t - base)))
    torch.cuda.synchronize()
    if os.environ.get("MODEST_PP_TRACE_PATHS"):   # (tests: which neighbour-count path the batches took)
        eprint("[pp_score] pp paths: block calls %d, chain calls %d" % (getattr(store, "block_calls", 0), getattr(store, "chain_calls", 0)))
    tot = dist.rank_report("pp_score", done, t0, rank, ws, dict(hist_points=pts))
    if rank == 0:
        eprint("[pp_score] %d scans, %.3g history points, %.2f s, %.2f scans/s on %d GPU(s); frame store %d hits / %d misses; "
               "ingest thread of rank 0: %d frames, %.1f MB read in %.3f s, upload + sort %.3f s, bookkeeping %.3f s; "
               "loop thread: %s"
               % (tot["scans"], tot["hist_points"], tot["max_seconds"], tot["scans"] / max(tot["max_seconds"], 1e-9),
                  ws, store.hits, store.misses, loader.n_frames, loader.read_bytes / 1e6, loader.t_read, loader.t_insert,
                  loader.t_touch, ", ".join("%s %.3f s" % (k, v) for k, v in ph.items())))
    return tot


@config.main(config_name="pp_score.yaml")
def main(args):
    return run(args)


if __name__ == "__main__":
    main()
