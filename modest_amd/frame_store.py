"""Frame store: raw LiDAR frames resident in HBM in the layout the PP kernels read.

The reference re-reads, transforms and stacks every history frame for every scan
(``pre_compute_pp_score.py:132-150``).  Here a frame is uploaded ONCE, sorted by the 8x8-cell tile
of a world lattice (``modest_frame_sort``; cell edge ``c = r (1 + 2^-10)``, the lattice is fixed by
the store's anchor and shared by every frame), and kept with its tile prefix table.  A scan names
frames by key; ``pp_score`` hands descriptors (store buffers + traversal + the reference's float32
relative pose) to ``modest_pp_score_frames`` -- no stacked history is ever built.

World lattice: ``lattice = (world_xy - anchor_xy) / c`` with ``world = W @ [p, 1]`` for the frame's
raw->world matrix ``W = E @ L @ K`` (ego pose, lidar-to-ego, KITTI2NU; the factors of
``get_relative_pose``, ``pre_compute_pp_score.py:27-28``).  ``consistent()`` checks a scan's relative
poses against the lattice (``A = fixed frame -> world``); the streaming kernels do not depend on it.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from collections import OrderedDict
from dataclasses import dataclass
from typing import Hashable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, ops
from ._lib import check, load


class SortJob(C.Structure):
    _fields_ = [("raw_dev", C.c_void_p), ("n", C.c_int32), ("stride", C.c_int32), ("TX0", C.c_int32),
                ("TY0", C.c_int32), ("W", C.c_double * 8), ("xyz_dev", C.c_void_p), ("perm_dev", C.c_void_p),
                ("tab_dev", C.c_void_p)]


# modest_pp_frame (include/modest_hip.h) as a numpy record: descriptor tables are filled vectorised
PP_FRAME = np.dtype([("xyz_dev", "u8"), ("tab_dev", "u8"), ("n", "i4"), ("TX0", "i4"), ("TY0", "i4"),
                     ("trav", "i4"), ("flags", "i4"), ("rel", "f4", (12,))], align=True)
assert PP_FRAME.itemsize == 88

REMOVE_CENTER = 1

# modest_pp_block_frame / modest_pp_block_scan (include/modest_hip.h) as numpy records
BLOCK_FRAME = np.dtype([("xyz_dev", "u8"), ("tab_dev", "u8"), ("n", "i4"), ("TX0", "i4"), ("TY0", "i4"), ("flags", "i4"),
                        ("lat", "f8", (8,))], align=True)
assert BLOCK_FRAME.itemsize == 96
BLOCK_SCAN = np.dtype([("xyz_dev", "u8"), ("perm_dev", "u8"), ("tab_dev", "u8"), ("n", "i4"), ("TX0", "i4"), ("TY0", "i4"),
                       ("n_members", "i4"), ("lat", "f8", (8,)), ("rel", "f4", (12,)), ("member_slot", "u8"),
                       ("member_trav", "u8"), ("member_rel", "u8"), ("counts_dev", "u8"), ("H_dev", "u8")], align=True)
assert BLOCK_SCAN.itemsize == 192

# modest_frame_sort_job as a numpy record: the ingest thread fills a batch of jobs vectorised
SORT_JOB = np.dtype([("raw_dev", "u8"), ("n", "i4"), ("stride", "i4"), ("TX0", "i4"), ("TY0", "i4"), ("W", "f8", (8,)),
                     ("xyz_dev", "u8"), ("perm_dev", "u8"), ("tab_dev", "u8")], align=True)
assert SORT_JOB.itemsize == C.sizeof(SortJob)


class StoreView(C.Structure):
    """modest_pp_store_view (include/modest_hip.h)"""
    _fields_ = [("records", C.c_void_p), ("W", C.c_void_p), ("lat", C.c_void_p), ("perm_dev", C.c_void_p), ("clean", C.c_void_p),
                ("n_slots", C.c_int64), ("radius", C.c_double), ("cell", C.c_double), ("window_span", C.c_int32)]


class BlockFrame:
    """A stored frame that is a slice of a batch's blocks (insert_block): the tensor views are made on demand
    (the compute path only needs the addresses, which sit in the store's descriptor records)."""
    __slots__ = ("_xyz", "_perm", "_tab", "a", "b", "k", "n", "TX0", "TY0", "inside", "W", "slot", "slabs")

    def __init__(self, xyz_all, perm_all, tab_all, a, b, k, TX0, TY0, inside, W, slot, slabs=()):
        self._xyz, self._perm, self._tab, self.a, self.b, self.k = xyz_all, perm_all, tab_all, a, b, k
        self.n, self.TX0, self.TY0, self.inside, self.W, self.slot = b - a, TX0, TY0, inside, W, slot
        self.slabs = slabs   # ids of the store's slabs the frame's block was carved from

    xyz = property(lambda self: self._xyz[self.a:self.b])
    perm = property(lambda self: self._perm[self.a:self.b])
    tab = property(lambda self: self._tab[self.k])

    @property
    def n_inside(self) -> int:
        if not isinstance(self.inside, int):
            pin, k, ev = self.inside
            ev.synchronize()
            self.inside = int(pin[k])
        return self.inside

    @property
    def nbytes(self) -> int:
        return self.n * 16 + self._tab.shape[1] * 4

    def original_order(self) -> torch.Tensor:
        """(n,3) raw points in file order (for the stacked fallback path)."""
        xyz = self.xyz
        out = torch.empty_like(xyz)
        out[self.perm.long()] = xyz
        return out


@dataclass
class StoredFrame:
    xyz: torch.Tensor        # (n,3) f32 tile-sorted raw points
    perm: torch.Tensor       # (n,) int32 (u32 bits): sorted -> original index
    tab: torch.Tensor        # (NTF*NTF+1,) int32 prefix offsets
    n: int
    TX0: int
    TY0: int
    inside: object           # points inside the table: an int, or (pinned int32 tensor, index, event) of an async sort
    W: np.ndarray            # (4,4) f64 raw frame -> world metres
    slot: int = -1

    @property
    def n_inside(self) -> int:
        """points inside the table (== n unless the frame has outliers)"""
        if not isinstance(self.inside, int):
            pin, k, ev = self.inside
            ev.synchronize()
            self.inside = int(pin[k])
        return self.inside

    @property
    def nbytes(self) -> int:
        return self.xyz.numel() * 4 + self.perm.numel() * 4 + self.tab.numel() * 4

    def original_order(self) -> torch.Tensor:
        """(n,3) raw points in file order (for the stacked fallback path)."""
        out = torch.empty_like(self.xyz)
        out[self.perm.long()] = self.xyz
        return out


SPLIT_BLOCK = "split"   # block_tables: the scans are worth the block path, in two halves


class FrameStore:
    """LRU store of tile-sorted frames on one device, for one ``max_neighbor_dist``."""

    def __init__(self, device, radius: float, capacity_bytes: float = 64 * 2 ** 30, ctx=None):
        self.device = torch.device(device)
        self.radius = float(radius)
        # lattice cell edge r (1 + 2^-8): the block path (modest_pp_score_block) uses the lattice as a spatial filter for
        # distances evaluated in every scan's own frame and needs more slack than r 2^-10 (the single-scan kernels
        # build their own grid per scan and only use the tile ORDER of a frame)
        self.cell = self.radius * (1.0 + 1.0 / 256.0)
        self.cap = int(capacity_bytes)
        self.bytes = 0
        self.frames: "OrderedDict[Hashable, StoredFrame]" = OrderedDict()
        self.anchor: Optional[np.ndarray] = None
        self.ntf = int(load().modest_frame_table_tiles())
        self.hits = self.misses = 0
        self.ctx = ctx
        self.lock = threading.RLock()   # an ingest thread inserts while the compute loop describes scans
        import collections
        self._inflight = collections.deque()   # (event, job table, raw inputs) of asynchronous sorts
        self._pin, self._pin_pos = None, 0
        # device memory of block insertions comes from slabs taken from the driver in large pieces: hipMalloc costs
        # milliseconds per call (and a lock shared by every process on the GPU); PyTorch's caching allocator grows
        # in 20 MB segments, i.e. one driver call every two or three scans once its cache is used up (measured:
        # the PP CLI dropped from 810 to 290 scans/s at that point)
        self._slab, self._slab_off, self.slab_bytes = None, 0, 256 << 20
        # Capacity is counted in what the store really holds: whole slabs -- a slab goes back to the allocator only when its
        # last frame is gone -- plus the tensors of frames inserted one by one (`bytes` stays the payload of the resident
        # frames, for reports).  A slab is allocated on the ingest stream and read by kernels of other streams: every stream
        # that reads frames is recorded on every slab (note_reader), so a slab handed back is not reused before the work
        # queued on those streams at that time has finished.
        self._slabs: dict = {}    # id -> [tensor, bytes, resident frames carved from it]
        self._slab_id = -1        # the slab being carved
        self._pinned: set = set()  # slabs an insertion in progress has carved from (their frames are not counted yet)
        self._loose = 0
        self._readers: dict = {}
        # slot tables: the static part of every frame's descriptor, gathered per scan by fancy indexing
        self._rec = np.zeros(1024, dtype=PP_FRAME)
        self._W = np.zeros((1024, 4, 4))
        self._lat = np.zeros((1024, 8))            # the lattice map every frame was sorted with (block path)
        self._perm = np.zeros(1024, dtype=np.uint64)   # device address of the frame's perm array
        self._clean = np.zeros(1024, dtype=bool)   # no point outside the table
        self._checked = np.zeros(1024, dtype=bool)   # ... _clean is known (asynchronous sorts report it later)
        self._key_of: dict = {}                    # slot -> key
        lim = (C.c_int32 * 3)()
        check(load().modest_pp_block_limits(C.byref(lim, 0), C.byref(lim, 4), C.byref(lim, 8)), "modest_pp_block_limits")
        self.block_window, self.block_max_scans = int(lim[0]), int(lim[1])
        self._free: List[int] = list(range(1023, -1, -1))
        self._slot_index = np.full(1 << 16, -1, dtype=np.int64)   # integer key -> slot (-1: not resident)

    # ------------------------------------------------------------------ lattice
    def lattice_rows(self, M44: np.ndarray) -> np.ndarray:
        """Rows x, y of (1/c) * (M - anchor) for a 4x4 map into world metres: 8 float64."""
        M = np.asarray(M44, dtype=np.float64)
        rows = M[:2, :].copy()
        rows[:, 3] -= self.anchor[:2]
        return np.ascontiguousarray(rows / self.cell).reshape(8)

    def _table_origin(self, W44: np.ndarray) -> Tuple[int, int]:
        o = (np.asarray(W44, dtype=np.float64)[:2, 3] - self.anchor[:2]) / self.cell
        return (int(np.floor(o[0] / 8.0)) - self.ntf // 2, int(np.floor(o[1] / 8.0)) - self.ntf // 2)

    def _ctx(self, ctx=None):
        return ctx if ctx is not None else (self.ctx if self.ctx is not None
                                            else _lib.default_context(self.device.index or 0))

    def _take_slot(self) -> int:
        if not self._free:
            old = self._rec.shape[0]
            self._rec = np.concatenate([self._rec, np.zeros(old, dtype=PP_FRAME)])
            self._W = np.concatenate([self._W, np.zeros((old, 4, 4))])
            self._lat = np.concatenate([self._lat, np.zeros((old, 8))])
            self._perm = np.concatenate([self._perm, np.zeros(old, dtype=np.uint64)])
            self._clean = np.concatenate([self._clean, np.zeros(old, dtype=bool)])
            self._checked = np.concatenate([self._checked, np.zeros(old, dtype=bool)])
            self._free = list(range(2 * old - 1, old - 1, -1))
        return self._free.pop()

    # ------------------------------------------------------------------ insertion
    def insert_many(self, items: Sequence[Tuple[Hashable, torch.Tensor, np.ndarray]], ctx=None,
                    protect: Optional[Sequence[Hashable]] = None, blocking: bool = True) -> None:
        """items: (key, raw (n,3|4) f32 device tensor, W (4,4) f64 raw->world).  One launch.
        `protect`: keys that must stay resident whatever the capacity says -- the frames of the scan
        that is being prepared (its resident history frames have not been touched yet when the missing
        ones are inserted; evicting them would also hand their slots, which the scan's descriptor table
        names, to other frames).  A scan whose own working set exceeds the capacity simply overshoots it.
        `blocking=False`: the sort is only enqueued on the current stream (modest_frame_sort_async): the frames
        are usable by work that is ordered behind it; the raw inputs must stay alive until then (the store
        keeps a reference for the last few batches)."""
        seen, todo = set(), []
        with self.lock:
            for it in items:
                if it[0] not in self.frames and it[0] not in seen:
                    seen.add(it[0])
                    todo.append(it)
            if todo and self.anchor is None:
                self.anchor = np.floor(np.asarray(todo[0][2], dtype=np.float64)[:3, 3])
        if not todo:
            return
        lib = load()
        jobs = (SortJob * len(todo))()
        made = []
        # one allocation per output kind for the whole batch (frames are views: three allocator calls instead
        # of three per frame; a block lives as long as any of its frames)
        tw = self.ntf * self.ntf + 1
        offs = np.cumsum([0] + [int(raw.shape[0]) for _, raw, _ in todo])
        xyz_all = torch.empty((int(offs[-1]), 3), dtype=torch.float32, device=self.device)
        perm_all = torch.empty((int(offs[-1]),), dtype=torch.int32, device=self.device)
        tab_all = torch.empty((len(todo), tw), dtype=torch.int32, device=self.device)
        for k, (key, raw, W) in enumerate(todo):
            assert raw.is_cuda and raw.dtype == torch.float32 and raw.is_contiguous() and raw.ndim == 2
            n = int(raw.shape[0])
            xyz, perm, tab = xyz_all[offs[k]:offs[k + 1]], perm_all[offs[k]:offs[k + 1]], tab_all[k]
            TX0, TY0 = self._table_origin(W)
            j = jobs[k]
            j.raw_dev, j.n, j.stride, j.TX0, j.TY0 = raw.data_ptr(), n, int(raw.shape[1]), TX0, TY0
            j.W[:] = list(self.lattice_rows(W))
            j.xyz_dev, j.perm_dev, j.tab_dev = xyz.data_ptr(), perm.data_ptr(), tab.data_ptr()
            made.append((key, xyz, perm, tab, n, TX0, TY0, np.asarray(W, dtype=np.float64).copy(), np.array(j.W[:])))
        if blocking:
            inside = (C.c_int32 * len(todo))()
            check(lib.modest_frame_sort(self._ctx(ctx).handle, jobs, len(todo), inside,
                                        torch.cuda.current_stream().cuda_stream), "modest_frame_sort")
            inside_of = [int(v) for v in inside]
        else:
            # the kernel writes every frame's inside-count into pinned host memory: one pool per store (a pinned
            # allocation per batch is a driver call that takes a lock shared by every process on the GPU)
            if self._pin is None or self._pin_pos + len(todo) > self._pin.shape[0]:
                self._pin = torch.empty((max(1 << 18, len(todo)),), dtype=torch.int32, pin_memory=True)
                self._pin_pos = 0
            pin = self._pin[self._pin_pos:self._pin_pos + len(todo)]
            self._pin_pos += len(todo)
            jdev = torch.empty((len(todo) * 128,), dtype=torch.uint8, device=self.device)
            check(lib.modest_frame_sort_async(self._ctx(ctx).handle, jobs, len(todo), jdev.data_ptr(), pin.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream), "modest_frame_sort_async")
            ev = torch.cuda.Event()
            ev.record()
            inside_of = [(pin, k, ev) for k in range(len(todo))]
            # raw inputs and the job table live until the launch has run: keep the last batches, drop finished ones
            self._inflight.append((ev, jdev, [raw for _, raw, _ in todo]))
            while len(self._inflight) > 2 and self._inflight[0][0].query():
                self._inflight.popleft()
        with self.lock:
            for k, (key, xyz, perm, tab, n, TX0, TY0, W, lat) in enumerate(made):
                sf = StoredFrame(xyz, perm, tab, n, TX0, TY0, inside_of[k], W, self._take_slot())
                r = self._rec[sf.slot]
                r["xyz_dev"], r["tab_dev"], r["n"], r["TX0"], r["TY0"] = xyz.data_ptr(), tab.data_ptr(), n, TX0, TY0
                self._W[sf.slot] = W
                self._lat[sf.slot] = lat
                self._perm[sf.slot] = perm.data_ptr()
                self._key_of[sf.slot] = key
                self._checked[sf.slot] = isinstance(sf.inside, int)
                self._clean[sf.slot] = (sf.inside == n) if isinstance(sf.inside, int) else True   # (async: checked on demand)
                self.frames[key] = sf
                self.bytes += sf.nbytes
                self._loose += sf.nbytes
                if isinstance(key, (int, np.integer)) and key >= 0:
                    if key >= self._slot_index.shape[0]:
                        grown = np.full(max(2 * self._slot_index.shape[0], int(key) + 1), -1, dtype=np.int64)
                        grown[: self._slot_index.shape[0]] = self._slot_index
                        self._slot_index = grown
                    self._slot_index[key] = sf.slot
            if self.footprint() > self.cap:
                keep = set(protect) if protect is not None else set()
                keep.update(k for k, *_ in made)
                self._evict(keep)

    def reserve(self, nbytes: int) -> None:
        """take `nbytes` of device memory for future block insertions in ONE driver call"""
        nbytes = int(min(max(nbytes, 0), self.cap))
        if nbytes > 0 and (self._slab is None or self._slab.shape[0] - self._slab_off < nbytes):
            self._slab, self._slab_off = self._new_slab(nbytes), 0

    def _new_slab(self, nbytes: int) -> torch.Tensor:
        t = torch.empty((int(nbytes),), dtype=torch.uint8, device=self.device)
        with self.lock:   # (note_reader inserts into _readers and walks _slabs under the same lock: no reader is missed)
            for st in self._readers.values():
                t.record_stream(st)
            old = self._slabs.get(self._slab_id)
            if old is not None and old[2] == 0 and self._slab_id not in self._pinned:
                del self._slabs[self._slab_id]   # nothing resident (or being inserted) was carved from the slab that is replaced
            self._slab_id += 1
            self._slabs[self._slab_id] = [t, int(nbytes), 0]
        return t

    def _release(self, old) -> None:
        """a frame left the store (call under the lock): its slabs go back to the allocator with their last frame"""
        if not isinstance(old, BlockFrame):
            self._loose -= old.nbytes
            return
        for sid in old.slabs:
            ent = self._slabs.get(sid)
            if ent is not None:
                ent[2] -= 1
                if ent[2] <= 0 and sid != self._slab_id:
                    del self._slabs[sid]

    def note_reader(self, stream=None) -> None:
        """`stream` (default: the current one) reads frames of the store"""
        st = torch.cuda.current_stream(self.device) if stream is None else stream
        if st.cuda_stream not in self._readers:
            with self.lock:
                self._readers[st.cuda_stream] = st
                for ent in self._slabs.values():
                    ent[0].record_stream(st)

    def footprint(self) -> int:
        """device bytes held: whole slabs + the tensors of frames inserted one by one"""
        return self._loose + sum(ent[1] for ent in self._slabs.values())

    def _carve(self, nbytes: int) -> torch.Tensor:
        nbytes = (int(nbytes) + 255) & ~255
        if self._slab is None or self._slab.shape[0] - self._slab_off < nbytes:
            self._slab, self._slab_off = self._new_slab(max(self.slab_bytes, nbytes)), 0
        out = self._slab[self._slab_off:self._slab_off + nbytes]
        self._slab_off += nbytes
        return out

    def insert_block(self, keys: Sequence[int], raw_block: torch.Tensor, offs: np.ndarray, Ws: np.ndarray, ctx=None,
                     protect: Optional[Sequence[Hashable]] = None) -> None:
        """The ingest path: `keys` (none resident, all distinct) are the frames raw_block[offs[k]:offs[k+1]] of ONE
        (P,4) float32 device block, Ws (k,4,4) their raw->world matrices.  Everything per frame is vectorised
        (job table, descriptor records, table origins); the sort is enqueued, not awaited (insert_many has the
        general, blocking form)."""
        lib = load()
        nf = len(keys)
        if nf == 0:
            return
        Ws = np.ascontiguousarray(Ws, dtype=np.float64).reshape(nf, 4, 4)
        with self.lock:
            if self.anchor is None:
                self.anchor = np.floor(Ws[0, :3, 3])
        tw = self.ntf * self.ntf + 1
        P = int(offs[-1])
        slabs = self._pinned = set()
        xyz_all = self._carve(12 * P)[:12 * P].view(torch.float32).view(P, 3)
        slabs.add(self._slab_id)
        perm_all = self._carve(4 * P)[:4 * P].view(torch.int32)
        slabs.add(self._slab_id)
        tab_all = self._carve(4 * tw * nf)[:4 * tw * nf].view(torch.int32).view(nf, tw)
        slabs.add(self._slab_id)
        a, b = np.asarray(offs[:-1], dtype=np.int64), np.asarray(offs[1:], dtype=np.int64)
        o = (Ws[:, :2, 3] - self.anchor[:2]) / self.cell
        T0 = np.floor(o / 8.0).astype(np.int64) - self.ntf // 2
        rows = Ws[:, :2, :].copy()
        rows[:, :, 3] -= self.anchor[:2]
        jobs = np.zeros(nf, dtype=SORT_JOB)
        jobs["raw_dev"] = raw_block.data_ptr() + 16 * a
        jobs["n"], jobs["stride"] = b - a, 4
        jobs["TX0"], jobs["TY0"] = T0[:, 0], T0[:, 1]
        jobs["W"] = (rows / self.cell).reshape(nf, 8)
        jobs["xyz_dev"] = xyz_all.data_ptr() + 12 * a
        jobs["perm_dev"] = perm_all.data_ptr() + 4 * a
        jobs["tab_dev"] = tab_all.data_ptr() + 4 * tw * np.arange(nf, dtype=np.int64)
        if self._pin is None or self._pin_pos + nf > self._pin.shape[0]:
            self._pin = torch.empty((max(1 << 18, nf),), dtype=torch.int32, pin_memory=True)
            self._pin_pos = 0
        pin = self._pin[self._pin_pos:self._pin_pos + nf]
        self._pin_pos += nf
        jdev = self._carve(nf * 128)
        check(lib.modest_frame_sort_async(self._ctx(ctx).handle, jobs.ctypes.data, nf, jdev.data_ptr(), pin.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream), "modest_frame_sort_async")
        ev = torch.cuda.Event()
        ev.record()
        self._inflight.append((ev, jdev, raw_block))
        while len(self._inflight) > 2 and self._inflight[0][0].query():
            self._inflight.popleft()
        with self.lock:
            slots = np.array([self._take_slot() for _ in range(nf)], dtype=np.int64)
            rec = self._rec
            rec["xyz_dev"][slots], rec["tab_dev"][slots] = jobs["xyz_dev"], jobs["tab_dev"]
            rec["n"][slots], rec["TX0"][slots], rec["TY0"][slots] = jobs["n"], jobs["TX0"], jobs["TY0"]
            self._W[slots] = Ws
            self._lat[slots] = jobs["W"]
            self._perm[slots] = jobs["perm_dev"]
            self._checked[slots] = False
            for k, key in enumerate(keys):
                self._key_of[int(slots[k])] = key
            self._clean[slots] = True   # (async: inside-counts are read on demand)
            kmax = int(max(keys))
            if kmax >= self._slot_index.shape[0]:
                grown = np.full(max(2 * self._slot_index.shape[0], kmax + 1), -1, dtype=np.int64)
                grown[: self._slot_index.shape[0]] = self._slot_index
                self._slot_index = grown
            self._slot_index[np.asarray(keys, dtype=np.int64)] = slots
            fr = self.frames
            slab_ids = tuple(sorted(slabs))
            for k, key in enumerate(keys):
                fr[key] = BlockFrame(xyz_all, perm_all, tab_all, int(a[k]), int(b[k]), k, int(T0[k, 0]), int(T0[k, 1]),
                                     (pin, k, ev), Ws[k], int(slots[k]), slab_ids)
            for sid in slab_ids:
                self._slabs[sid][2] += nf
            self._pinned = set()
            self.bytes += 16 * P + 4 * tw * nf
            if self.footprint() > self.cap:
                self._evict(set(protect) | set(keys) if protect is not None else set(keys))

    def drop(self, keys) -> None:
        """forget frames (their slab memory is reused only with the whole slab; descriptor slots are recycled)"""
        with self.lock:
            for key in keys:
                old = self.frames.pop(key, None)
                if old is None:
                    continue
                self.bytes -= old.nbytes
                self._release(old)
                self._free.append(old.slot)
                if isinstance(key, (int, np.integer)) and 0 <= key < self._slot_index.shape[0]:
                    self._slot_index[key] = -1

    def _evict(self, keep) -> None:
        for key in [k for k in self.frames if k not in keep]:   # LRU order, oldest first (call under the lock)
            if self.footprint() <= self.cap:
                break
            old = self.frames.pop(key)
            self.bytes -= old.nbytes
            self._release(old)
            self._free.append(old.slot)
            if isinstance(key, (int, np.integer)) and 0 <= key < self._slot_index.shape[0]:
                self._slot_index[key] = -1

    def insert(self, key, raw: torch.Tensor, W: np.ndarray) -> StoredFrame:
        self.insert_many([(key, raw, W)])
        return self.frames[key]

    def get(self, key) -> Optional[StoredFrame]:
        with self.lock:
            f = self.frames.get(key)
            if f is not None:
                self.frames.move_to_end(key)
                self.hits += 1
            else:
                self.misses += 1
            return f

    def __contains__(self, key) -> bool:
        return key in self.frames

    def missing(self, keys) -> list:
        """the distinct keys that are not resident, in first-occurrence order"""
        with self.lock:
            if len(keys) and isinstance(keys[0], (int, np.integer)):
                k = np.asarray(keys, dtype=np.int64)
                if k.min() >= 0:
                    res = np.ones(len(k), dtype=bool)
                    inr = k < self._slot_index.shape[0]
                    res[inr] = self._slot_index[k[inr]] < 0
                    if not res.any():
                        return []
                    return list(dict.fromkeys(int(x) for x in k[res]))
            return [k for k in dict.fromkeys(keys) if k not in self.frames]

    def touch(self, keys) -> None:
        """LRU order + hit statistics of a scan's frames, one lock round trip"""
        with self.lock:
            fr = self.frames
            for k in keys:
                if k in fr:
                    fr.move_to_end(k)
                    self.hits += 1
                else:
                    self.misses += 1

    # ------------------------------------------------------------------ the PP stage of one scan
    def consistent(self, slots: np.ndarray, rels: np.ndarray, A44: np.ndarray, tol: float = 1e-4,
                   reach: float = 160.0) -> bool:
        """|A @ rel_f @ p - W_f @ p| < tol for every raw point p within `reach` of the sensor."""
        A = np.asarray(A44, dtype=np.float64)
        D = A[None] @ rels.astype(np.float64) - self._W[slots]
        dev = np.abs(D[:, :2, :3]).sum(axis=2) * reach + np.abs(D[:, :2, 3])
        return bool(np.all(np.isfinite(dev)) and dev.max() < tol)

    def slots_of(self, keys) -> np.ndarray:
        """slot of every key (call under the lock); small non-negative integer keys (file ids) go through an
        index array instead of one dictionary look-up each"""
        if len(keys) and isinstance(keys[0], (int, np.integer)):
            k = np.asarray(keys, dtype=np.int64)
            if k.min() >= 0 and k.max() < self._slot_index.shape[0]:
                s = self._slot_index[k]
                if (s >= 0).all():
                    return s
        return np.fromiter((self.frames[k].slot for k in keys), dtype=np.int64, count=len(keys))

    def points_of(self, keys) -> int:
        with self.lock:
            return int(self._rec["n"][self.slots_of(list(keys))].sum()) if len(keys) else 0

    def describe(self, live_key, live_rel: np.ndarray, hist_keys: Sequence[Hashable], travs: Sequence[int],
                 rels: np.ndarray, remove_center: bool = False):
        """Descriptor table of a scan: (live record (1,), history records (F,), slots (F+1,))."""
        with self.lock:
            slots = self.slots_of(hist_keys) if len(hist_keys) else np.zeros(0, dtype=np.int64)
            arr = self._rec[slots] if len(slots) else np.zeros(1, dtype=PP_FRAME)
            lslot = self.frames[live_key].slot
            lv = self._rec[[lslot]]
        if len(slots):
            arr["trav"] = np.asarray(travs, dtype=np.int32)
            arr["flags"] = REMOVE_CENTER if remove_center else 0
            arr["rel"] = np.asarray(rels, dtype=np.float32).reshape(len(slots), 4, 4)[:, :3, :].reshape(len(slots), 12)
        lv["rel"] = np.asarray(live_rel, dtype=np.float32).reshape(4, 4)[:3, :].reshape(1, 12)
        return lv, arr, np.concatenate([slots, [lslot]])

    def describe_many(self, live_keys, live_rels, hist_keys_list, travs_list, rels_list, remove_center: bool = False):
        """describe() for several scans with ONE gather from the slot tables (a block of 16 scans x 360 frames: 0.95 -> 0.25 ms):
        [(live record, history records, slots)] -- the history records of the scans are views of one array."""
        lens = [len(h) for h in hist_keys_list]
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        flat = [k for h in hist_keys_list for k in h]
        with self.lock:
            slots = self.slots_of(flat) if flat else np.zeros(0, dtype=np.int64)
            arr = self._rec[slots] if len(slots) else np.zeros(0, dtype=PP_FRAME)
            lslots = np.array([self.frames[k].slot for k in live_keys], dtype=np.int64)
            lvs = self._rec[lslots]
        if len(slots):
            arr["trav"] = np.concatenate([np.asarray(t, dtype=np.int32) for t in travs_list])
            arr["flags"] = REMOVE_CENTER if remove_center else 0
            arr["rel"] = np.concatenate([np.asarray(r, dtype=np.float32).reshape(-1, 4, 4)[:, :3, :].reshape(-1, 12)
                                         for r in rels_list if len(r)])
        lvs["rel"] = np.stack([np.asarray(r, dtype=np.float32).reshape(4, 4)[:3, :].reshape(12) for r in live_rels])
        out = []
        for i in range(len(live_keys)):
            a = arr[offs[i]:offs[i + 1]] if lens[i] else np.zeros(1, dtype=PP_FRAME)
            out.append((lvs[i:i + 1], a, np.concatenate([slots[offs[i]:offs[i + 1]], lslots[i:i + 1]])))
        return out

    def pp_score(self, live_key, live_rel: np.ndarray, hist: Sequence[Tuple[Hashable, int]], rels: np.ndarray,
                 A44: np.ndarray, n_trav: int, remove_center: bool = False, return_counts: bool = False,
                 out: Optional[torch.Tensor] = None, ctx=None, force_stacked: bool = False, desc=None):
        """PP score of one scan.  hist: (frame key, traversal index) per history frame; rels: (F,4,4)
        float32 relative poses (get_relative_pose); live_rel (4,4) float32; A44 (4,4) float64 common
        frame -> world.  Returns H (N,) float32 in the live frame's file order (and counts (N,T)).
        `desc` = a table from describe() reused across calls (the frames must still be resident)."""
        lib = load()
        self.note_reader()
        live = self.frames[live_key]
        F = len(hist)
        rels = np.ascontiguousarray(np.asarray(rels, dtype=np.float32).reshape(F, 4, 4))
        live_rel = np.ascontiguousarray(np.asarray(live_rel, dtype=np.float32).reshape(4, 4))
        N, T = live.n, int(n_trav)
        if desc is None:
            desc = self.describe(live_key, live_rel, [k for k, _ in hist], [t for _, t in hist], rels, remove_center)
        lv, arr, slots = desc
        # the streaming kernels read frames as plain point lists (tile order only helps their scatter
        # pass): outliers and a lattice that disagrees with the poses are harmless; > 64 traversals take
        # the stacked path
        ok = not force_stacked and T <= 64
        if not ok:
            frames = [self.frames[k] for k, _ in hist]
            return self._pp_score_stacked(live, live_rel, frames, [t for _, t in hist], rels, T, remove_center,
                                          return_counts, out, ctx)
        A8 = np.ascontiguousarray(self.lattice_rows(A44))
        H = out if out is not None else torch.empty((N,), dtype=torch.float32, device=self.device)
        counts = torch.empty((N, T), dtype=torch.int32, device=self.device) if return_counts else None
        check(lib.modest_pp_score_frames(self._ctx(ctx).handle, lv.ctypes.data, live.perm.data_ptr(), arr.ctypes.data,
                                         F, T, A8.ctypes.data, self.radius,
                                         counts.data_ptr() if counts is not None else None,
                                         H.data_ptr(), torch.cuda.current_stream().cuda_stream),
              "modest_pp_score_frames")
        return (H, counts) if return_counts else H

    # ------------------------------------------------------------------ several scans per call
    def _all_clean(self, slots: np.ndarray) -> bool:
        """no frame of `slots` has points outside its table (asynchronous sorts report that later: read here, once)"""
        todo = slots[~self._checked[slots]]
        for sl in np.unique(todo):
            f = self.frames.get(self._key_of.get(int(sl)))
            if f is not None and f.slot == sl:
                self._clean[sl] = f.n_inside == f.n
                self._checked[sl] = True
        return bool(self._clean[slots].all())

    def block_tables(self, descs, n_trav: int, force: Optional[bool] = None):
        """The tables of modest_pp_score_block for these scans, or None when the block path does not apply: more than
        64 traversals / scans, a frame with points outside its table, poses that disagree with the lattice by more
        than 1e-4 m, live scans further apart than the block window, mixed remove_center flags -- or (unless forced)
        too little sharing (fewer than 4 scans, union > 4 x a scan's frames): the block path bins the UNION of the scans' frames once, which pays when the scans are
        consecutive scans of a shard (35 of 36 frames per traversal shared, split_traintest.py:64,97).
        The tables themselves -- union of (frame, occurrence) entries ordered by first + last using scan, member tables, every
        check -- are built by the library (modest_pp_block_tables, csrc/block_tables.hip: one pass over the members; the numpy
        statement it replaces took 0.9 ms for 7 scans and 3.7 ms for 32, tests/block_tables_numpy.py keeps it as the test's reference).
        What the rule's numbers rest on:
          whole pipeline, 8 processes: Lyft shape, 36 frames per traversal, 16 scans: union 1.42 x a scan's frames, 137 against 191 us
          per scan; nuScenes shape, 16 frames per traversal: union 1.94 x, 281 against 263 us alone, but 3 400 against 2 790 scans/s;
          round 5 (tools/pp_block_probe.py --scans B): block 185 / 157 / 142 / 127 us per scan at B = 4 / 5 / 6 / 8 against 210 / 206 /
          206 / 197 for the chain, B = 3: 229 against 220; windows by the reference's rule (profiles/r05_sharing_sensitivity.json):
          union 0.94 ... 2.52 x -> 106 ... 151 us against 193-200; 16 -> 32 scans: 79 -> 69 us at 1.42 -> 1.86 x;
          round 6 (member packing, union ordered by first + last; --halves, 32 scans, union 3.0 ... 4.0 x): ONE block 101 / 105 / 117 /
          121 / 127 us per scan against 106 / 112 / 128 / 124 / 137 in two halves and 200 for the chain -- halves past 4 x only;
          the join keeps a scan's pose table in LDS up to 2 048 entries (pp_v4.hip: B4_POSE_LDS_MAX): beyond, two halves."""
        env = os.environ.get("MODEST_PP_BLOCK", "")
        if force is None:
            force = True if env == "1" else (False if env == "0" else None)
        if force is False:
            return None
        B = len(descs)
        Ts = np.full(B, int(n_trav), dtype=np.int32) if np.isscalar(n_trav) else np.ascontiguousarray(n_trav, dtype=np.int32)   # (per scan)
        T = int(Ts.max()) if B else 0
        if B > self.block_max_scans and force is None and T <= 64:
            return SPLIT_BLOCK
        if B < 1 or B > self.block_max_scans or T > 64:
            return None
        lib = load()
        with self.lock:
            slots = [np.ascontiguousarray(sl, dtype=np.int64) for _, _, sl in descs]
            lens = np.array([len(sl) - 1 for sl in slots], dtype=np.int32)
            members = int(lens.sum())
            if members == 0:
                return None
            # (asynchronous sorts report points outside a table later: read here, once per slot)
            every = np.concatenate(slots)
            if not self._checked[every].all():
                self._all_clean(every)
            view = self._store_view()
            recs = [(lv if lv.flags.c_contiguous else np.ascontiguousarray(lv), arr if arr.flags.c_contiguous else np.ascontiguousarray(arr))
                    for lv, arr, _ in descs]   # (slices of describe_many's arrays: contiguous already)
            ptrs = (C.c_void_p * (3 * B))(*([lv.ctypes.data for lv, _ in recs] + [arr.ctypes.data for _, arr in recs]
                                           + [sl.ctypes.data for sl in slots]))
            base = C.addressof(ptrs)
            fr = np.zeros(min(members, (1 << 16) - 1), dtype=BLOCK_FRAME)
            sc = np.zeros(B, dtype=BLOCK_SCAN)
            ms_all, mt_all = np.empty(members, dtype=np.int32), np.empty(members, dtype=np.int32)
            mr_all = np.empty((members, 12), dtype=np.float32)
            n_fr = C.c_int32(0)
            rc = lib.modest_pp_block_tables(C.byref(view), B, base, base + 8 * B, base + 16 * B, lens.ctypes.data, Ts.ctypes.data,
                                            1 if force is None else 0, fr.ctypes.data, len(fr), C.byref(n_fr), sc.ctypes.data,
                                            ms_all.ctypes.data, mt_all.ctypes.data, mr_all.ctypes.data)
        if rc == 2:
            return SPLIT_BLOCK
        if rc == 1:
            return None
        if rc != 0:
            raise ValueError(f"modest_pp_block_tables: bad arguments (rc={rc})")
        return fr[:n_fr.value], sc, [ms_all, mt_all, mr_all]

    def _store_view(self):
        """modest_pp_store_view of the slot tables (the arrays are re-read at every call: they are replaced when the store grows)"""
        v = StoreView()
        v.records, v.W, v.lat = self._rec.ctypes.data, self._W.ctypes.data, self._lat.ctypes.data
        v.perm_dev, v.clean, v.n_slots = self._perm.ctypes.data, self._clean.ctypes.data, int(self._rec.shape[0])   # (numpy bool: one byte, 0 / 1)
        v.radius, v.cell, v.window_span = float(self.radius), float(self.cell), int(self.block_window - self.ntf - 2)
        return v

    def pp_score_batch(self, live_keys, descs, n_trav, outs=None, return_counts: bool = False, ctx=None,
                       block: Optional[bool] = None, _may_split: bool = True, _cs=None):
        """PP scores of several scans in ONE call.  live_keys: the live frame of every scan; descs: their tables from
        describe() (the frames must be resident).  Scans that share most of their history frames (consecutive scans of
        a shard) go through modest_pp_score_block -- the union of their frames is binned once; others through
        modest_pp_score_frames_batch (one chain of launches, every scan on its own).  `block`: True / False force the
        choice (MODEST_PP_BLOCK=1 / 0 in the environment does the same).  Returns [H] (and [counts]); results are
        those of separate pp_score calls, bit for bit.
        n_trav: one number, or one per scan -- the reference accepts a traversal per scan (split_traintest.py:17,79,111), so the
        scans of a sequence differ in T; a block takes them as they come (modest_pp_score_block_mixed), the per-scan chain runs
        once per run of equal T."""
        lib = load()
        self.note_reader()
        B = len(descs)
        Ts = [int(n_trav)] * B if np.isscalar(n_trav) else [int(t) for t in n_trav]
        if len(Ts) != B:
            raise ValueError("n_trav: one number or one per scan")
        if B and max(Ts) > 64:
            raise ValueError("the batched path takes at most 64 traversals")
        Hs, cs = [], []
        for i, (lv, arr, slots) in enumerate(descs):
            N = int(lv["n"][0])
            H = outs[i] if outs is not None and outs[i] is not None else torch.empty((N,), dtype=torch.float32, device=self.device)
            Hs.append(H)
            cs.append(None if not return_counts else _cs[i] if _cs is not None else torch.empty((N, Ts[i]), dtype=torch.int32, device=self.device))
        tabs = self.block_tables(descs, Ts, force=block)
        if tabs is SPLIT_BLOCK and not _may_split:
            tabs = None   # (ONE level of halving: a half the rule would halve again takes the per-scan chain in one call)
        if tabs is SPLIT_BLOCK:   # (the automatic rule: too little sharing for ONE block -- two calls of half the scans)
            h = B // 2
            lo = self._pp_score_batch_c(live_keys[:h], descs[:h], Ts[:h], Hs[:h], cs[:h], ctx, block, B > self.block_max_scans)
            hi = self._pp_score_batch_c(live_keys[h:], descs[h:], Ts[h:], Hs[h:], cs[h:], ctx, block, B > self.block_max_scans)
            return (Hs, cs) if return_counts else Hs
        if tabs is not None:
            fr, sc, keep = tabs
            sc["H_dev"] = [H.data_ptr() for H in Hs]
            sc["counts_dev"] = [c.data_ptr() if c is not None else 0 for c in cs]
            tarr = np.asarray(Ts, dtype=np.int32)
            check(lib.modest_pp_score_block_mixed(self._ctx(ctx).handle, fr.ctypes.data, len(fr), sc.ctypes.data, B, tarr.ctypes.data,
                                                  self.radius, self.cell, torch.cuda.current_stream().cuda_stream),
                  "modest_pp_score_block_mixed")
            self.block_calls = getattr(self, "block_calls", 0) + 1
            return (Hs, cs) if return_counts else Hs
        if len(set(Ts)) > 1:   # the chain of launches takes one T: one call per run of equal T
            i = 0
            while i < B:
                j = i
                while j < B and Ts[j] == Ts[i]:
                    j += 1
                self._pp_score_batch_c(live_keys[i:j], descs[i:j], Ts[i:j], Hs[i:j], cs[i:j], ctx, False, False)
                i = j
            return (Hs, cs) if return_counts else Hs
        T = Ts[0] if B else 0
        livep, permp, framep = np.zeros(B, dtype=np.uint64), np.zeros(B, dtype=np.uint64), np.zeros(B, dtype=np.uint64)
        Hp, cp = np.zeros(B, dtype=np.uint64), np.zeros(B, dtype=np.uint64)
        nfr = np.zeros(B, dtype=np.int32)
        keep = []
        for i, (lv, arr, slots) in enumerate(descs):
            lv, arr = np.ascontiguousarray(lv), np.ascontiguousarray(arr)
            keep.append((lv, arr))
            livep[i], framep[i], nfr[i] = lv.ctypes.data, arr.ctypes.data, len(slots) - 1
            permp[i] = self.frames[live_keys[i]].perm.data_ptr()
            Hp[i] = Hs[i].data_ptr()
            cp[i] = cs[i].data_ptr() if cs[i] is not None else 0
        self.chain_calls = getattr(self, "chain_calls", 0) + 1
        check(lib.modest_pp_score_frames_batch(self._ctx(ctx).handle, B, livep.ctypes.data, permp.ctypes.data,
                                               framep.ctypes.data, nfr.ctypes.data, T, self.radius,
                                               cp.ctypes.data if return_counts else None, Hp.ctypes.data,
                                               torch.cuda.current_stream().cuda_stream),
              "modest_pp_score_frames_batch")
        return (Hs, cs) if return_counts else Hs

    def _pp_score_batch_c(self, live_keys, descs, Ts, Hs, cs, ctx, block, may_split):
        """pp_score_batch into given outputs (count tensors included)"""
        want = any(c is not None for c in cs)
        return self.pp_score_batch(live_keys, descs, Ts, outs=Hs, return_counts=want, ctx=ctx, block=block, _may_split=may_split,
                                   _cs=cs if want else None)

    def _pp_score_stacked(self, live, live_rel, frames, travs, rels, T, remove_center, return_counts, out, ctx):
        """Stacked path (V3 kernels on a transformed copy) for scans the frame path does not take:
        frames with outliers, more than 64 traversals, a lattice that disagrees with the poses."""
        parts = [[] for _ in range(T)]
        for f, t, rel in zip(frames, travs, rels):
            parts[t].append(ops.transform_points(f.xyz, rel, remove_center=remove_center, ctx=ctx))
        flat = [p for tp in parts for p in tp]
        hist = torch.cat(flat) if flat else torch.empty((0, 3), dtype=torch.float32, device=self.device)
        offsets = np.cumsum([0] + [sum(int(p.shape[0]) for p in tp) for tp in parts]).astype(np.int64)
        lv = ops.transform_points(live.original_order(), live_rel, ctx=ctx)
        return ops.pp_score(lv, hist, offsets, self.radius, ctx=ctx, return_counts=return_counts, out=out)
