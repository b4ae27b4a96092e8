"""Host logic of the block path without a GPU: FrameStore.block_tables on a store whose slot tables are filled by hand
(no device memory is touched: the tables only carry addresses).  What is checked: the block's frame table is ordered by
(first, last) scan that uses a frame, so that every scan's members are ONE range of slots; member slots / traversals /
poses arrive per scan as the C ABI expects them; the rule that decides between the block and the per-scan chain; poses
that disagree with the lattice and frames with points outside their table are refused."""
import ctypes as C
import threading

import numpy as np
import pytest

from modest_amd import frame_store as fs


def _store(n_slots, T, L):
    """a FrameStore shell: slot s = frame (t, j) -> s = t * L + j, live frames behind; identity lattice, W = translation"""
    st = object.__new__(fs.FrameStore)
    st.lock = threading.RLock()
    st.block_window, st.block_max_scans, st.ntf = 160, 64, 128
    st.radius, st.cell = 0.3, 0.3 * (1.0 + 1.0 / 256.0)
    st._rec = np.zeros(n_slots, dtype=fs.PP_FRAME)
    st._rec["xyz_dev"] = 0x10000 + 4096 * np.arange(n_slots)
    st._rec["tab_dev"] = 0x9000000 + 4096 * np.arange(n_slots)
    st._rec["n"] = 1000
    st._W = np.tile(np.eye(4), (n_slots, 1, 1))
    st._lat = np.zeros((n_slots, 8))
    st._perm = (0x5000000 + 4096 * np.arange(n_slots)).astype(np.uint64)
    st._clean = np.ones(n_slots, dtype=bool)
    st._checked = np.ones(n_slots, dtype=bool)
    st._key_of, st.frames = {}, {}
    return st


def _descs(st, B, T, F, L, shift=1.0, bad_pose=None):
    """B consecutive scans: scan i looks at frames i .. i+F-1 of every traversal; frame (t, j) sits at x = j * shift"""
    live0 = T * L
    for t in range(T):
        for j in range(L):
            st._W[t * L + j, 0, 3] = j * shift
    out = []
    for i in range(B):
        ls = live0 + i
        st._W[ls, 0, 3] = (i + F) * shift
        slots = np.array([t * L + (i + j) for t in range(T) for j in range(F)] + [ls], dtype=np.int64)
        arr = st._rec[slots[:-1]].copy()
        arr["trav"] = np.repeat(np.arange(T), F)
        rel = np.linalg.inv(st._W[ls])[None] @ st._W[slots[:-1]]          # frame -> live frame
        if bad_pose is not None and i == bad_pose:
            rel[3, 0, 3] += 0.01
        arr["rel"] = rel[:, :3, :].reshape(-1, 12).astype(np.float32)
        lv = st._rec[[ls]].copy()
        lv["rel"] = np.eye(4)[:3].reshape(1, 12).astype(np.float32)        # the common frame IS the live frame
        out.append((lv, arr, slots))
    return out


def _read(ptr, n, ctype):
    return np.ctypeslib.as_array((ctype * n).from_address(int(ptr))).copy()


@pytest.mark.parametrize("B,T,F", [(16, 10, 36), (8, 2, 24), (16, 20, 16)])
def test_block_tables_order_members_and_layout(B, T, F):
    L = F + B - 1
    st = _store(T * L + B, T, L)
    descs = _descs(st, B, T, F, L)
    fr, sc, keep = st.block_tables(descs, T)
    assert len(fr) == T * L and len(sc) == B
    pos_of = {int(a): k for k, a in enumerate(fr["xyz_dev"])}              # table position of every frame (by its address)
    for i, (lv, arr, slots) in enumerate(descs):
        n = int(sc["n_members"][i])
        assert n == T * F
        ms = _read(sc["member_slot"][i], n, C.c_int32)
        mt = _read(sc["member_trav"][i], n, C.c_int32)
        mr = _read(sc["member_rel"][i], 12 * n, C.c_float).reshape(n, 12)
        assert np.array_equal(ms, [pos_of[int(a)] for a in arr["xyz_dev"]])   # the member table names the scan's own frames
        assert np.array_equal(mt, arr["trav"]) and np.array_equal(mr, arr["rel"])
        # ... and they are ONE range of the block's table, holes only for frames another scan of the block uses
        lo, hi = int(ms.min()), int(ms.max())
        inside = set(range(lo, hi + 1)) - set(ms.tolist())
        used_by = {}
        for q, (_, a2, _) in enumerate(descs):
            for a in a2["xyz_dev"]:
                used_by.setdefault(pos_of[int(a)], set()).add(q)
        assert all(i not in used_by[p] for p in inside)
        assert hi - lo + 1 <= T * L
        assert sc["xyz_dev"][i] == lv["xyz_dev"][0] and sc["perm_dev"][i] == st._perm[int(slots[-1])]
    # (first, last) order: the first scan that uses a frame never decreases along the table
    first = np.array([min(q for q in range(B) if int(a) in set(descs[q][1]["xyz_dev"].tolist())) for a in fr["xyz_dev"]])
    assert np.all(np.diff(first) >= 0)


def test_block_tables_rule_and_refusals():
    T, F, B = 4, 36, 16
    L = F + B - 1
    st = _store(T * L + B, T, L)
    descs = _descs(st, B, T, F, L)
    assert st.block_tables(descs, T) is not None
    assert st.block_tables(descs[:7], T) is not None                   # (round 5: the block pays from four scans on)
    assert st.block_tables(descs[:3], T) is None                       # fewer than four scans: the chain (unless forced)
    assert st.block_tables(descs[:3], T, force=True) is not None
    assert st.block_tables(descs, T, force=False) is None
    # windows shorter than 12 frames per traversal, or scans that share too little: the chain
    st2 = _store(T * 27 + 16, T, 27)
    assert st2.block_tables(_descs(st2, 16, T, 8, 27), T) is None
    # a pose that disagrees with the lattice by a centimetre, a frame with points outside its table: refused even when forced
    st3 = _store(T * L + B, T, L)
    assert st3.block_tables(_descs(st3, B, T, F, L, bad_pose=5), T, force=True) is None
    st4 = _store(T * L + B, T, L)
    d4 = _descs(st4, B, T, F, L)
    st4._clean[3] = False
    assert st4.block_tables(d4, T, force=True) is None
    # live scans further apart than the block window
    st5 = _store(T * L + B, T, L)
    d5 = _descs(st5, B, T, F, L)
    st5._rec["TX0"][T * L + 9] = 400
    d5[9][0]["TX0"] = 400
    assert st5.block_tables(d5, T, force=True) is None


def test_block_tables_repeated_frames_get_a_union_entry_per_occurrence():
    """A scan that lists a frame k times (pre_compute_pp_score.py:132-150 stacks it k times; split_traintest.py:86-101 produces
    such lists) gets k union entries with the same buffers; its member list names every union slot at most once, so that
    every occurrence keeps its own pose entry in modest_pp_score_block."""
    T, F, B = 4, 36, 16
    L = F + B - 1
    st = _store(T * L + B, T, L)
    descs = _descs(st, B, T, F, L)
    # scan 3: member 1 := member 0 (twice in a row); scan 5: one frame three times and the window's first frame again at its end
    for i, edits in ((3, {1: 0}), (5, {40: 38, 39: 38, F - 1: 0})):
        lv, arr, slots = descs[i]
        arr, slots = arr.copy(), slots.copy()
        for dst, src in edits.items():
            arr[dst] = arr[src]
            slots[dst] = slots[src]
        descs[i] = (lv, arr, slots)
    fr, sc, keep = st.block_tables(descs, T)
    distinct = len(np.unique(np.concatenate([sl[:-1] for _, _, sl in descs])))
    assert len(fr) == distinct + 1 + 2 + 1          # one extra entry per extra occurrence (frame 38 of scan 5: two extras)
    for i, (lv, arr, slots) in enumerate(descs):
        n = int(sc["n_members"][i])
        assert n == T * F                                               # a repeated frame stays a member
        ms = _read(sc["member_slot"][i], n, C.c_int32)
        assert len(np.unique(ms)) == n                                  # ... with a union slot of its own
        assert np.array_equal(fr["xyz_dev"][ms], arr["xyz_dev"])        # ... that carries the frame's buffers
        assert np.array_equal(_read(sc["member_rel"][i], 12 * n, C.c_float).reshape(n, 12), arr["rel"])
    # a lattice cell with less slack than pose error + float32 rounding needs: the block path is refused (radius < 0.072 m)
    st.radius, st.cell = 0.05, 0.05 * (1.0 + 1.0 / 256.0)
    assert st.block_tables(descs, T, force=True) is None


def test_large_blocks_split_when_the_union_is_large():
    """32 scans of Lyft shape (36 frames per traversal: union 1.86 x a scan's entries), of nuScenes shape (16 frames: 2.94 x) and with
    12 frames per traversal (3.58 x) stay ONE block (round 6: one block beat two halves up to 4 x); 48 scans with 12 frames per
    traversal (4.9 x: past the rule's 4 x) are tried in two halves (2.9 x) before the per-scan chain -- unless the caller forces one
    block."""
    T = 4
    for B, F, expect_split in ((32, 36, False), (32, 16, False), (32, 12, False), (48, 12, True)):
        L = F + B - 1
        st = _store(T * L + B, T, L)
        descs = _descs(st, B, T, F, L)
        got = st.block_tables(descs, T)
        assert (got is fs.SPLIT_BLOCK) == expect_split, (B, F, got if isinstance(got, str) else type(got))
        forced = st.block_tables(descs, T, force=True)
        assert forced is not None and forced is not fs.SPLIT_BLOCK and len(forced[0]) == T * L
        half = st.block_tables(descs[:B // 2], T)   # each half then goes as a block of its own
        assert half is not None and half is not fs.SPLIT_BLOCK and len(half[1]) == B // 2


def test_block_tables_with_a_traversal_count_per_scan():
    """Scans of one block may have different numbers of traversals (the reference accepts a traversal per scan,
    data_preprocessing/lyft/split_traintest.py:17,79,111): the rule counts entries against the SUM of the scans' T, the tables carry every
    scan's own member list, and a traversal index beyond a scan's own T never appears in it."""
    T, F, L, B = 5, 12, 40, 8
    st = _store(T * L + B, T, L)
    descs = _descs(st, B, T, F, L)
    Ts = [5, 5, 4, 4, 3, 3, 5, 2]
    cut = []
    for (lv, arr, slots), t in zip(descs, Ts):   # scan i keeps its first T_i traversals
        keep = arr["trav"] < t
        cut.append((lv, arr[keep], np.concatenate([slots[:-1][keep], slots[-1:]])))
    tabs = st.block_tables(cut, Ts)
    assert tabs is not None and tabs is not fs.SPLIT_BLOCK
    fr, sc, keep = tabs
    for i, t in enumerate(Ts):
        n = int(sc["n_members"][i])
        assert n == t * F
        tr = _read(sc["member_trav"][i], n, C.c_int32)
        assert tr.min() == 0 and tr.max() == t - 1
        ms = _read(sc["member_slot"][i], n, C.c_int32)
        assert len(set(ms.tolist())) == n and ms.min() >= 0 and ms.max() < len(fr)
    # the rule: fewer than 12 entries per traversal AND scan on average -> the per-scan chain (unless forced)
    short = [(lv, arr[arr["trav"] < 2][:10], np.concatenate([slots[:-1][arr["trav"] < 2][:10], slots[-1:]])) for lv, arr, slots in descs]
    assert st.block_tables(short, [2] * B) is None
    assert st.block_tables(short, [2] * B, force=True) is not None


def _same_tables(a, b):
    if a is None or b is None or isinstance(a, str) or isinstance(b, str):
        return a is b or a == b
    (fa, sa, ka), (fb, sb, kb) = a, b
    if len(fa) != len(fb) or fa.tobytes() != fb.tobytes():
        return False
    if any(not np.array_equal(sa[k], sb[k]) for k in ("xyz_dev", "perm_dev", "tab_dev", "n", "TX0", "TY0", "n_members", "lat", "rel")):
        return False
    return all(np.array_equal(x, y) for x, y in zip(ka, kb))   # member slots, traversals, poses (scan after scan)


@pytest.mark.parametrize("direct_max", [None, "0"])
def test_library_tables_equal_the_numpy_statement_on_random_blocks(direct_max, monkeypatch):
    """modest_pp_block_tables (csrc/block_tables.hip) against the numpy statement it replaced (tests/block_tables_numpy.py): the same
    union in the same order, the same member tables, the same answer where the block path is refused or halved -- on sliding windows,
    windows with gaps and reversals (an entry's users are then no interval of scans), repeated frames, a traversal count per scan,
    scans without history, bad poses, unclean frames and far-apart live scans, with the rule and forced."""
    from block_tables_numpy import block_tables_numpy
    if direct_max is not None:   # the library's path for very large stores (members sorted by key instead of a key-indexed table)
        monkeypatch.setenv("MODEST_BLOCK_TABLES_DIRECT_MAX", direct_max)
    rng = np.random.default_rng(11)
    outcomes = {"block": 0, "chain": 0, "split": 0}
    for case in range(60):
        T, F = int(rng.integers(2, 7)), int(rng.choice([6, 12, 16, 36]))
        B = int(rng.choice([3, 4, 7, 8, 16, 32, 48]))
        if case % 6 == 5:   # many scans over short windows: the union passes 4 x a scan's entries (two halves)
            B, F = int(rng.choice([48, 64])), 12
        L = F + B + 3
        st = _store(T * L + B, T, L)
        descs = _descs(st, B, T, F, L, bad_pose=(int(rng.integers(B)) if case % 13 == 5 else None))
        Ts = [T] * B
        kind = case % 6
        out = []
        for i, (lv, arr, slots) in enumerate(descs):
            arr, slots = arr.copy(), slots.copy()
            if kind == 1:      # some scans look at a window of their own (no sharing with the neighbours), in a shuffled order
                if rng.random() < 0.4:
                    sh = int(rng.integers(0, L - F - i))
                    slots[:-1] = np.array([t * L + (sh + j) for t in range(T) for j in range(F)])
                    rel = np.linalg.inv(st._W[slots[-1]])[None] @ st._W[slots[:-1]]
                    arr["rel"] = rel[:, :3, :].reshape(-1, 12).astype(np.float32)
                p = rng.permutation(len(arr))
                arr, slots = arr[p], np.concatenate([slots[:-1][p], slots[-1:]])
            elif kind == 2:    # repeated frames (split_traintest.py:86-101 lists a frame once per threshold that selects it)
                for _ in range(int(rng.integers(0, 4))):
                    d, s_ = rng.integers(0, len(arr), size=2)
                    arr[d], slots[d] = arr[s_], slots[s_]
            elif kind == 3:    # a traversal count per scan
                t = int(rng.integers(1, T + 1))
                keep = arr["trav"] < t
                arr, slots, Ts[i] = arr[keep], np.concatenate([slots[:-1][keep], slots[-1:]]), t
            elif kind == 4 and rng.random() < 0.2:   # a scan without history
                arr, slots = np.zeros(1, dtype=fs.PP_FRAME), slots[-1:]
            out.append((lv, arr, slots))
        if case % 17 == 3:
            st._clean[int(rng.integers(T * L))] = False
        if case % 19 == 7:
            st._rec["TX0"][T * L + B // 2] = 400
        for force in (None, True):
            a, b = st.block_tables(out, Ts, force=force), block_tables_numpy(st, out, Ts, force=force)
            assert _same_tables(a, b), (case, kind, B, T, F, force, type(a), type(b))
            outcomes["chain" if a is None else ("split" if a is fs.SPLIT_BLOCK else "block")] += 1
    assert min(outcomes.values()) >= 5, outcomes   # every answer occurs
