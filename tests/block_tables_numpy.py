"""The numpy statement of FrameStore.block_tables as it stood until round 6 (the product builds the tables in the library now:
modest_pp_block_tables, modest_amd/csrc/block_tables.hip).  Test infrastructure: tests/test_block_tables_cpu.py compares the two on
random scans -- same union, same order, same member tables, same refusals."""
import os
from typing import Optional

import numpy as np

from modest_amd.frame_store import BLOCK_FRAME, BLOCK_SCAN, SPLIT_BLOCK


def block_tables_numpy(self, descs, n_trav, force: Optional[bool] = None):
    """The tables of modest_pp_score_block for these scans, or None when the block path does not apply: more than
    64 traversals / scans, a frame with points outside its table, poses that disagree with the lattice by more
    than 1e-4 m, live scans further apart than the block window, mixed remove_center flags -- or (unless forced)
    too little sharing (fewer than 4 scans, union > 4 x a scan's frames): the block path bins the UNION of the scans' frames once, which pays when the scans are
    consecutive scans of a shard (35 of 36 frames per traversal shared, split_traintest.py:64,97)."""
    env = os.environ.get("MODEST_PP_BLOCK", "")
    if force is None:
        force = True if env == "1" else (False if env == "0" else None)
    if force is False:
        return None
    B = len(descs)
    Ts = np.full(B, int(n_trav), dtype=np.int64) if np.isscalar(n_trav) else np.asarray(n_trav, dtype=np.int64)   # (per scan)
    T = int(Ts.max()) if B else 0
    if B > self.block_max_scans and force is None and T <= 64:
        return SPLIT_BLOCK
    if B < 1 or B > self.block_max_scans or T > 64:
        return None
    with self.lock:   # (everything below is vectorised over the scans: ~0.4 ms for 16 scans x 360 frames, 1.5 ms scan by scan)
        lens = np.array([len(sl) - 1 for _, _, sl in descs], dtype=np.int64)
        lslots = np.array([int(sl[-1]) for _, _, sl in descs], dtype=np.int64)
        members = int(lens.sum())
        if members == 0:
            return None
        allh = np.concatenate([np.asarray(sl[:-1], dtype=np.int64) for _, _, sl in descs])
        sid = np.repeat(np.arange(B, dtype=np.int64), lens)   # the scan of every member, ascending
        # A scan may list a frame more than once (split_traintest.py:86-101: two distance thresholds select the same
        # pose), and the reference stacks it as often as it is listed (pre_compute_pp_score.py:132-150).  The union is
        # therefore one of (frame, occurrence): the k-th listing of a frame inside a scan is union entry (frame, k), the
        # scatter writes that frame's points k times, and every member keeps a pose entry of its own.
        srt = np.argsort(sid * (int(allh.max()) + 1) + allh, kind="stable")
        ks = (sid * (int(allh.max()) + 1) + allh)[srt]
        run0 = np.flatnonzero(np.concatenate([[True], ks[1:] != ks[:-1]]))
        occ = np.empty(members, dtype=np.int64)
        occ[srt] = np.arange(members) - np.repeat(run0, np.diff(np.concatenate([run0, [members]])))
        K = int(occ.max()) + 1
        ukey = allh * K + occ
        us, first_idx = np.unique(ukey, return_index=True)
        if force is None:
            # measured (bench.py, whole pipeline, 8 processes): the block pays when there are enough scans to share the
            # binning (>= 8) and the windows are not too short -- Lyft shape, 36 frames per traversal, 16 scans: union
            # 1.42 x a scan's frames, 137 against 191 us per scan alone on the GPU; nuScenes shape, 16 frames per
            # traversal: union 1.94 x, 281 against 263 us alone, but 3 400 against 2 790 scans/s in the pipeline (one
            # sequence of large launches instead of sixteen chains of small ones next to seven other processes)
            # round 5 (wave-independent join, tools/pp_block_probe.py --scans B): block 185 / 157 / 142 / 127 us per scan at
            # B = 4 / 5 / 6 / 8 against 210 / 206 / 206 / 197 for the chain; B = 3: 229 against 220
            per_scan = members / B
            # ... and on windows chosen by the reference's rule (synth.make_shard_matched; profiles/r05_sharing_sensitivity.json):
            # union 0.94 / 1.89 / 1.92 / 2.07 / 2.52 x a scan's frames -> 106 / 120 / 130 / 123 / 151 us against 193-200
            if B < 4 or members < 12 * int(Ts.sum()):   # (fewer than 12 entries per traversal and scan)
                return None
            # Large blocks amortise the union's binning further (16 -> 32 Lyft-shape scans: 79 -> 69 us per scan at a union of
            # 1.42 -> 1.86 x a scan's entries; nuScenes shape, 16 frames per traversal: 111 -> 101 us at 1.94 -> 2.94 x; windows by the
            # reference's rule: 111 -> 117 at 2.1 -> 2.6 x) -- ONE block beats two of half the scans wherever the block path pays
            # at all (measured: tools/pp_block_probe.py --auto).
            # round 6 (member packing, union ordered by first + last; tools/pp_block_probe.py --halves, 32 scans, union 3.0 ... 4.0 x):
            # ONE block 101 / 105 / 117 / 121 / 127 us per scan against 106 / 112 / 128 / 124 / 137 in two halves and 200 for the chain --
            # halves are tried past 4 x only
            if len(us) > 4.0 * per_scan:
                return SPLIT_BLOCK if B >= 8 else None
            # the join keeps a scan's pose table (49 B per union entry) in LDS up to 2 048 entries (pp_v4.hip: B4_POSE_LDS_MAX);
            # beyond that it reads the poses from memory (measured with a table of 1 024: 32 scans with little sharing, 1 113-1 138
            # entries, 156 us per scan against 127 in two halves)
            if len(us) > 2048 and B >= 8:
                return SPLIT_BLOCK
        if len(us) >= (1 << 16) or not self._all_clean(np.concatenate([np.unique(allh), lslots])):
            return None
        # the lattice is a conservative filter only while two points within r of each other (pose error < 1e-4 m checked
        # below, float32 evaluation < 4e-5 m, per point) stay within one cell: cell - r >= 2 (1e-4 + 4e-5)
        if self.cell - self.radius < 2.0 * (1e-4 + 4e-5):
            return None
        lrec = self._rec[lslots]
        span = self.block_window - self.ntf - 2   # (the library pads the window by one tile on every side)
        if (int(lrec["TX0"].max()) - int(lrec["TX0"].min()) > span
                or int(lrec["TY0"].max()) - int(lrec["TY0"].min()) > span):
            # (the live scans of the block lie further apart than the block window: 32 scans at more than 11 m/s -- half the
            # scans span half the distance)
            return SPLIT_BLOCK if (force is None and B >= 8) else None
        with_hist = [(arr, int(n)) for (_, arr, _), n in zip(descs, lens) if n]
        flags = np.unique(np.concatenate([arr["flags"][:n] for arr, n in with_hist]))
        if len(flags) > 1:
            return None
        mr_all = np.ascontiguousarray(np.concatenate([arr["rel"][:n] for arr, n in with_hist]), dtype=np.float32).reshape(members, 12)
        mt_all = np.ascontiguousarray(np.concatenate([arr["trav"][:n] for arr, n in with_hist]), dtype=np.int32)
        live_rel = np.ascontiguousarray(np.stack([lv["rel"][0] for lv, _, _ in descs]), dtype=np.float32).reshape(B, 12)
        # every pose against the lattice (FrameStore.consistent, all scans at once): A_scan = W_live inv(rel_live)
        has = lens > 0
        R = np.zeros((B, 4, 4))
        R[:, :3, :] = live_rel.reshape(B, 3, 4)
        R[:, 3, 3] = 1.0
        A = np.zeros((B, 4, 4))
        A[has] = self._W[lslots[has]] @ np.linalg.inv(R[has])
        # (rows x, y of A_scan @ rel_f - W_f only -- the lattice is two-dimensional --, and rel's last row is (0, 0, 0, 1): a
        # (2,3) @ (3,4) product per member instead of a (4,4) @ (4,4) one: 5.1 -> 3.4 ms of host time per block of 32 x 360 members)
        A2 = A[:, :2, :]
        D = np.matmul(A2[sid][:, :, :3], mr_all.reshape(members, 3, 4).astype(np.float64))
        D[:, :, 3] += A2[sid][:, :, 3]
        D -= self._W[allh][:, :2, :]
        dev = np.abs(D[:, :, :3]).sum(axis=2) * 160.0 + np.abs(D[:, :, 3])
        if not (np.all(np.isfinite(dev)) and dev.max() < 1e-4):
            return None
        # the block's frame table in the order (first scan that uses the frame, last scan that uses it): the frames of every
        # scan of a sliding window are then one contiguous range of the table, and the join skips -- run by run of the
        # cell-sorted store -- the records of the frames a scan does not use (modest_hip.h)
        first = sid[first_idx]
        _, last_idx = np.unique(ukey[::-1], return_index=True)
        last = sid[::-1][last_idx]
        # (round 6: by first + last -- the middle of the interval of scans that use the entry -- then first.  On sliding windows it is the
        # same order as (first, last); on windows chosen by the reference's rule, where an entry's users are not an interval of consecutive
        # scans and traversals come and go, a scan's slot range shrinks from 2.23 to 2.04 x its own entries (1.47 -> 1.37 without absent
        # traversals): fewer foreign records for the join to load, transform and mask)
        us = us[np.lexsort((first, first + last))]
        pos = np.empty(int(us.max()) + 1, dtype=np.int32)
        pos[us] = np.arange(len(us), dtype=np.int32)
        fr = np.zeros(len(us), dtype=BLOCK_FRAME)
        ur = self._rec[us // K]
        for k in ("xyz_dev", "tab_dev", "n", "TX0", "TY0"):
            fr[k] = ur[k]
        fr["flags"] = int(flags[0]) if len(flags) else 0
        fr["lat"] = self._lat[us // K]
        sc = np.zeros(B, dtype=BLOCK_SCAN)
        for k in ("xyz_dev", "tab_dev", "n", "TX0", "TY0"):
            sc[k] = lrec[k]
        sc["perm_dev"] = self._perm[lslots]
        sc["lat"] = self._lat[lslots]
        sc["rel"] = live_rel
        ms_all = np.ascontiguousarray(pos[ukey])   # (int32) the members of all scans, scan after scan: distinct inside a scan
        offs = (np.cumsum(lens) - lens).astype(np.uint64)
        sc["n_members"] = lens
        sc["member_slot"] = np.uint64(ms_all.ctypes.data) + np.uint64(4) * offs
        sc["member_trav"] = np.uint64(mt_all.ctypes.data) + np.uint64(4) * offs
        sc["member_rel"] = np.uint64(mr_all.ctypes.data) + np.uint64(48) * offs
        keep = [ms_all, mt_all, mr_all]
    return fr, sc, keep

