"""CPU: the C-ABI library loads and exports every symbol include/modest_hip.h declares (no
compute calls without a GPU); host-side logic (config composer, RANSAC control helpers)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "modest_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(modest_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from modest_amd import _lib, build
    build.build(verbose=False)
    lib = _lib.load()
    names = _header_symbols()
    assert len(names) >= 24
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/modest_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names          # the ctypes table mirrors the header one to one
    assert lib.modest_version() >= 100
    assert isinstance(lib.modest_device_count(), int)


def test_hot_kernels_use_no_scratch_memory():
    """A gfx950 kernel that touches scratch memory (spills, indexed per-thread arrays) pays ~25 us
    at dispatch; the build records hipcc's per-kernel resource remarks, every kernel on the
    default path must report ScratchSize 0."""
    import json
    from modest_amd import build
    build.build(verbose=False)
    res = json.load(open(os.path.join(os.path.dirname(build.LIB), "kernel_resources.json")))
    assert len(res) > 40
    # not on the default path: one-lane-per-angle fits (variance_to_edge, clusters > 90 k points), profiling builds
    # (b4_join<.., true>: MODEST_PP4_DBG=512).  b4_join (round 5: every wavefront on its own, ~110 registers) spills nothing.
    allowed = ("variance_kernel", "closeness_kernel", "pp3_joinILb1", "b4_joinILb1ELb1")
    joins = {k: v["ScratchSize [bytes/lane]"] for k, v in res.items() if "b4_joinILb" in k and "ELb0" in k}
    assert len(joins) == 2 and max(joins.values()) == 0, joins
    bad = {k: v["ScratchSize [bytes/lane]"] for k, v in res.items()
           if v.get("ScratchSize [bytes/lane]", 0) > 0 and not any(a in k for a in allowed)}
    assert not bad, bad


def test_no_cpu_fallback_without_device():
    from modest_amd import _lib
    lib = _lib.load()
    if lib.modest_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.ModestHipError, match="no HIP device"):
        _lib.Context(0)
    # argument errors are reported, never exit()
    h = ctypes.c_void_p()
    assert lib.modest_ctx_create(0, None) != 0
    assert b"NULL" in lib.modest_last_error()


def test_config_composer_matches_reference_defaults():
    from modest_amd import config
    c = config.compose("generate_mask", ["data_root=/d"])
    assert c.plane_estimate.range == [[-70, 70], [-20, 20]] and c.plane_estimate.max_hs == -1.5
    assert c.graph.n_neighbors == 70 and c.graph.radius == 2.0 and c.clustering.DBSCAN.eps == 0.1
    assert dict(**c.filtering)["min_percentile_pp_score"] == 0.7
    assert c.calib_path == "/d/calib" and c.data_paths.seg_save_dst.endswith("lyft_seg_pp_score_fw70_2m_r0.3/")
    n = config.compose("pp_score", ["data_root=/d", "data_paths=nusc.yaml", "nusc=True", "total_part=8", "part=3"])
    assert n.nusc is True and n.total_part == 8 and n.part == 3 and "nuscenes" in n.data_paths.track_path
    assert n.max_neighbor_dist == 0.3 and n.limit_traversals == -1 and n.ephe_type == "entropy"
    lab = config.compose("generate_label_files", ["data_root=/d", "image_shape=[900,1600]"])
    assert lab.image_shape == [900, 1600] and lab.nms.threshold == 0.1 and lab.fov_only is True
    with pytest.raises(config.MissingMandatoryValue):
        config.compose("pp_score").data_root
    with pytest.raises(KeyError):
        config.compose("pp_score", ["data_root=/d", "no_such_key=1"])
    assert config.compose("pp_score", ["data_root=/d", "+extra=5"]).extra == 5
    assert "max_neighbor_dist" in config.to_yaml(n)


def test_ransac_control_helpers():
    from sklearn.linear_model import LinearRegression
    from sklearn.linear_model._ransac import _dynamic_max_trials
    from modest_amd.utils import ransac
    for args in ((5000, 10000, 3, 0.99), (10, 10000, 3, 0.99), (10000, 10000, 3, 0.99), (1, 7, 3, 0.5)):
        assert ransac.dynamic_max_trials(*args) == _dynamic_max_trials(*args)
    rng = np.random.default_rng(0)
    p = (rng.standard_normal((50, 3, 3)) * [10, 10, 0.05] + [0, 0, -1.7]).astype(np.float32)
    m = ransac.planes_through_triplets(p)
    for b in range(50):
        reg = LinearRegression().fit(p[b, :, :2], p[b, :, 2])
        np.testing.assert_allclose(m[b, :2], reg.coef_, rtol=2e-3, atol=2e-5)
        np.testing.assert_allclose(m[b, 2], reg.intercept_, rtol=2e-3, atol=2e-4)
        assert np.max(np.abs(p[b, :, 0] * m[b, 0] + p[b, :, 1] * m[b, 1] + m[b, 2] - p[b, :, 2])) < 1e-4
    assert ransac.r2_from_sums(10, 0.0, 5.0, 10.0) == 1.0
    assert ransac.r2_from_sums(4, 1.0, 4.0, 8.0) == pytest.approx(1.0 - 1.0 / 4.0)


def test_vectorised_triplet_stream_equals_sklearn():
    from sklearn.utils.random import sample_without_replacement
    from modest_amd.utils.ransac import draw_triplets
    for n_pop in (7, 120, 301, 9973, 17001):
        for seed in (0, 5):
            a, b = np.random.RandomState(seed), np.random.RandomState(seed)
            ref = np.stack([sample_without_replacement(n_pop, 3, random_state=a) for _ in range(60)])
            trip, consumed = draw_triplets(b, n_pop, 60)
            assert np.array_equal(trip, ref), n_pop
            # the caller advances its stream by the executed trials only: emulate 37 executed
            c = np.random.RandomState(seed)
            for _ in range(37):
                sample_without_replacement(n_pop, 3, random_state=c)
            if consumed is not None:
                b.randint(n_pop, size=int(consumed[36]))
                assert np.array_equal(b.get_state()[1], c.get_state()[1]) and b.get_state()[2] == c.get_state()[2]


def test_sample_without_replacement_statement_equals_sklearn():
    """modest_amd.utils.ransac no longer imports sklearn: its statement of sample_without_replacement(method="auto") --
    permutation for 0.01 < ratio < 0.99, tracking selection below, reservoir sampling at ratio >= 0.99 -- makes sklearn's
    draws and leaves the generator in sklearn's state, for every population size up to the library's own method."""
    from sklearn.utils.random import sample_without_replacement as ref_swr
    from modest_amd.utils.ransac import check_random_state, sample_without_replacement
    for n_pop in list(range(3, 40)) + [100, 299, 300, 301, 302, 500, 5000]:
        a, b = np.random.RandomState(n_pop), np.random.RandomState(n_pop)
        for _ in range(25):
            assert np.array_equal(ref_swr(n_pop, 3, random_state=a), sample_without_replacement(n_pop, 3, b)), n_pop
        sa, sb = a.get_state(), b.get_state()
        assert np.array_equal(sa[1], sb[1]) and sa[2:] == sb[2:], n_pop
    np.random.seed(7)
    g = check_random_state(None)
    assert g is np.random.mtrand._rand and check_random_state(g) is g and isinstance(check_random_state(3), np.random.RandomState)
    import modest_amd.utils.ransac as mr
    src = open(mr.__file__).read()
    assert "import sklearn" not in src and "from sklearn" not in src


def test_library_mt19937_triplets_equal_sklearn():
    """The generator half of modest_ransac_plane (host code of the library): triplets AND the generator
    state after them equal sklearn's sample_without_replacement on numpy's legacy RandomState, incl. a
    stream that crosses several 624-word refills, a fresh generator (pos = 624) and one whose Gaussian
    cache is occupied (left untouched)."""
    from sklearn.utils.random import sample_without_replacement
    from modest_amd import ops
    for n_pop in (301, 1024, 1025, 9973, 17001, 2_000_000):
        for seed in (0, 5, 123456789):
            a, b = np.random.RandomState(seed), np.random.RandomState(seed)
            if seed == 5:
                a.standard_normal(3), b.standard_normal(3)
            for chunk in (1, 48, 500):
                ref = np.stack([sample_without_replacement(n_pop, 3, random_state=a) for _ in range(chunk)])
                got = ops.mt19937_triplets(b, n_pop, chunk)
                assert np.array_equal(got, ref), (n_pop, seed, chunk)
                sa, sb = a.get_state(), b.get_state()
                assert np.array_equal(sa[1], sb[1]) and sa[2:] == sb[2:]
            assert a.randint(1 << 30) == b.randint(1 << 30)


def test_percentile_lerp_equals_numpy():
    from modest_amd.utils.clustering_utils import percentile_from_order_stats
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 5, 10, 11, 37, 100, 1001, 20011):
        for q in (20, 10, 50, 73.5, 100):
            x = rng.uniform(0, 1, n).astype(np.float32)
            if n > 4:
                x[:3] = x[3]                    # duplicates
            s = np.sort(x)
            q32 = np.true_divide(q, np.float32(100))
            vi = np.float32(n - 1) * q32        # the float32 arithmetic cluster_stats.hip performs
            fl = np.floor(vi)
            prev = int(fl)
            nxt = prev + 1
            if vi >= n - 1:
                prev = nxt = n - 1
            nxt = min(nxt, n - 1)
            got = percentile_from_order_stats(s[prev], s[nxt], vi - fl)
            ref = np.percentile(x, q)
            assert ref.dtype == np.float32 and np.float32(got) == ref, (n, q)


def test_label_helpers_equal_numpy_formulation():
    """compact_labels == searchsorted(unique(l), l); members_by_label == flatnonzero per label."""
    from modest_amd.utils.clustering_utils import compact_labels, members_by_label
    rng = np.random.default_rng(0)
    for trial in range(20):
        n = int(rng.integers(1, 5000))
        lab = rng.integers(-1, 40, n) * (rng.uniform(size=n) < 0.3) - (rng.uniform(size=n) < 0.2)
        lab = np.maximum(lab, -1).astype(np.int64)
        ref = np.searchsorted(np.unique(lab), lab).astype(lab.dtype)
        got = compact_labels(lab)
        assert got.dtype == ref.dtype and np.array_equal(got, ref)
        c = compact_labels(np.maximum(lab, 0))       # 0 = background, 1..C
        n_lab = int(c.max())
        mem = members_by_label(c, n_lab)
        assert len(mem) == n_lab
        for i, m in enumerate(mem, start=1):
            assert np.array_equal(m, np.flatnonzero(c == i))
    assert compact_labels(np.zeros(0, dtype=np.int64)).shape == (0,)


def test_min_area_rectangles_equal_reference_golden(golden_dir):
    """fit_method='min_zx_area_fit': the host caliper search over Qhull's hull order reproduces the
    reference's rectangles (fixtures generated by tools/make_golden_fits.py from the reference)."""
    import os
    from modest_amd.utils import pointcloud_utils as pu
    g = np.load(os.path.join(golden_dir, "mask_stage.npz"))
    f = np.load(os.path.join(golden_dir, "fit_variants.npz"))
    off, pts = g["cl_offsets"], g["cl_pts"]
    clusters = [pts[off[k]:off[k + 1]] for k in range(len(off) - 1)]
    got = pu.min_area_rectangles(clusters)
    assert len(got) == len(f["minarea"])
    for k, (corners, angle, area) in enumerate(got):
        assert np.array_equal(np.concatenate([corners.reshape(-1), [angle, area]]), f["minarea"][k]), k


def test_stage_params_follow_config_mutation():
    """The native mask stage's parameter block is rebuilt when the config node is mutated in place
    (it used to be cached on the node's identity: the native and the host path then disagreed)."""
    from modest_amd import config, generate_mask
    cfg = config.compose("generate_mask")
    P1 = generate_mask._stage_params(cfg)
    assert P1 is generate_mask._stage_params(cfg)            # unchanged values: cached
    eps0, r0 = P1.eps, P1.radius
    cfg.clustering.DBSCAN.eps = eps0 * 2
    cfg.graph.radius = r0 + 1.0
    P2 = generate_mask._stage_params(cfg)
    assert P2.eps == eps0 * 2 and P2.radius == r0 + 1.0
    cfg.plane_estimate.max_hs = -1.3
    assert generate_mask._stage_params(cfg).max_hs1 == np.float32(-1.3)   # (a float32 field)


def test_frame_store_eviction_protects_the_scan(monkeypatch):
    """insert_many never evicts the frames named by the scan that is being prepared (CPU: the eviction
    logic only, the sort call is stubbed)."""
    import collections
    import numpy as np
    from modest_amd import frame_store as fs

    st = fs.FrameStore.__new__(fs.FrameStore)
    st.frames, st.bytes, st.cap, st._free = collections.OrderedDict(), 0, 250, []

    class F:
        def __init__(self, slot):
            self.nbytes, self.slot = 100, slot
    for k in range(3):
        st.frames[k] = F(k)
        st.bytes += 100
    # re-create the eviction step of insert_many: frame 9 was just made, the scan names 0, 1 and 9
    made, protect = [(9,)], [0, 1, 9]
    st.frames[9] = F(9)
    st.bytes += 100
    keep = set(protect)
    keep.update(k for k, *_ in made)
    for key in [k for k in st.frames if k not in keep]:
        if st.bytes <= st.cap:
            break
        old = st.frames.pop(key)
        st.bytes -= old.nbytes
        st._free.append(old.slot)
    assert list(st.frames) == [0, 1, 9] and st._free == [2]      # 2 went, although 0 and 1 were older


def test_batched_relative_poses_equal_per_frame_calls(golden_dir):
    """relative_poses (one gesv per nesting level with all frames' columns as right-hand sides) is bit-identical
    to get_relative_pose called per frame (pre_compute_pp_score.py:27-28; what the reference's loop does), also
    on the reference-generated pose fixture, for both KITTI2NU matrices."""
    from scipy.spatial.transform import Rotation as R
    from modest_amd import pre_compute_pp_score as pcs
    rng = np.random.default_rng(9)

    def pose(scale):
        t = np.eye(4)
        t[:3, :3] = R.from_euler("xyz", np.r_[rng.uniform(-0.05, 0.05, 2), rng.uniform(-3.1, 3.1)]).as_matrix()
        t[:3, 3] = rng.uniform(-scale, scale, 3)
        return t.astype(np.float32)
    for K in (pcs._KITTI2NU_lyft, pcs._KITTI2NU_nusc):
        for trial in range(20):
            fe, fl = pose(2000.0), pose(2.0)
            qs = [(pose(2000.0), pose(2.0)) for _ in range(int(rng.integers(1, 400)))]
            W = np.stack([qe @ ql @ K for qe, ql in qs])
            got = pcs.relative_poses(fl, fe, W, K)
            ref = np.stack([pcs.get_relative_pose(fl, fe, ql, qe, K) for qe, ql in qs])
            assert got.dtype == np.float32 and np.array_equal(got, ref), trial
    g = np.load(os.path.join(golden_dir, "pose.npz"))
    for i in range(len(g["fixed_ego"])):
        W = (g["query_ego"][i] @ g["query_l2e"][i] @ pcs._KITTI2NU_lyft)[None]
        assert np.array_equal(pcs.relative_poses(g["fixed_l2e"][i], g["fixed_ego"][i], W, pcs._KITTI2NU_lyft)[0],
                              pcs.get_relative_pose(g["fixed_l2e"][i], g["fixed_ego"][i], g["query_l2e"][i], g["query_ego"][i],
                                                    pcs._KITTI2NU_lyft))
    # WorldTable gathers what the per-frame dictionary held
    wt = pcs.WorldTable({5: np.eye(4) * 2, 9: np.eye(4) * 3}).freeze()
    assert np.array_equal(wt.stack([9, 5, 9]), np.stack([np.eye(4) * 3, np.eye(4) * 2, np.eye(4) * 3]))


def test_last_block_reductions_drain_their_stores_before_the_ticket(tmp_path):
    """common.h: the partial results of the last-block reductions (score_kernel, refit_kernel, closeness_tree_kernel,
    lowest_kernel and their chained forms) are ordered before the ticket by an inline `s_waitcnt vmcnt(0)`, not by the HIP
    memory model (rounds 1-2 shipped without it: 0.08 % wrong planes / boxes under load).  The build checks the machine
    code of every such kernel (build.ticket_drain_violations); here: the shipped assembly passes, and the check FAILS
    when modest_drain_stores() is removed from any one of its four call sites."""
    import re
    from modest_amd import build
    build.build(verbose=False)
    for name in build.DRAIN_SOURCES:
        asm = (build.OBJDIR / (name.replace(".hip", "") + ".gfx950.s")).read_text()
        assert build.ticket_drain_violations(asm) == []
        assert len(re.findall(r"ASMSTART\s+s_waitcnt vmcnt\(0\)", asm)) >= 2, name   # the drains are there at all
    sites = 0
    for name in build.DRAIN_SOURCES:
        src = (build.CSRC / name).read_text()
        calls = [m.start() for m in re.finditer(r"modest_drain_stores\(\);", src)]
        for k, pos in enumerate(calls):
            broken = src[:pos] + "/* removed */" + src[pos + len("modest_drain_stores();"):]
            f = tmp_path / f"{k}_{name}"
            f.write_text(broken)
            bad = build.ticket_drain_violations(build.device_asm(f, tmp_path / f"{k}_{name}.s"))
            assert bad, (name, k)
            sites += 1
    assert sites == 4


def test_relative_poses_block_is_the_per_scan_solve():
    """the batch's pose solves run a few scans per thread: per scan they ARE relative_poses (get_relative_pose,
    pre_compute_pp_score.py:27-28, batched over the frames) -- same values bit for bit, ragged frame counts included"""
    from modest_amd import synth
    from modest_amd.pre_compute_pp_score import _KITTI2NU_lyft as K, get_relative_pose, relative_poses, relative_poses_block
    rng = np.random.default_rng(3)
    egos = [synth._pose_matrix(rng.uniform(0, 900), rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(-.05, .05),
                               rng.uniform(-.05, .05)).astype(np.float32) for _ in range(7)]
    l2es = [synth.default_l2e() for _ in range(7)]
    stacks = [np.stack([synth._pose_matrix(rng.uniform(0, 900), rng.uniform(-3, 3), rng.uniform(-3, 3)) @ synth.default_l2e() @ K
                        for _ in range(f)]) for f in (361, 361, 5, 1, 361, 40, 361)]
    got = relative_poses_block(l2es, egos, stacks, K, threads=3)
    for i in range(7):
        assert got[i].dtype == np.float32 and np.array_equal(got[i], relative_poses(l2es[i], egos[i], stacks[i], K))
    # ... and relative_poses is the reference's per-frame call
    E, L = np.linalg.inv(stacks[0][3] @ np.linalg.inv(K) @ np.linalg.inv(l2es[0])), l2es[0]   # any ego pose with W = E @ L @ K
    one = get_relative_pose(l2es[0], egos[0], L, np.linalg.inv(E), K)
    assert one.shape == (4, 4) and one.dtype == np.float32


def test_fused_cli_composes_the_three_stage_configs():
    """modest_amd.seed_labels: `key=value` goes to every stage config that has the key (data_root, the data_paths group, total_part,
    workers ...), `pp.` / `mask.` / `labels.` prefixes address one stage; a key no stage has is an error, as Hydra's would be; what the
    work split reads (workers, device, total_part, part) comes from the PP stage's config; the worker hand-off is a plain container."""
    from modest_amd import seed_labels as sl
    c = sl.compose_all(["data_root=/d", "data_paths=nusc.yaml", "nusc=True", "workers=3", "total_part=2", "part=1", "mask.mask_batch=8",
                        "mask.plane_estimate.max_hs=-1.3", "labels.nms.threshold=0.2", "labels.image_shape=[900,1600]"])
    assert c.workers == 3 and c.get("total_part") == 2 and c.get("part") == 1 and c.get("nonsense", 7) == 7
    assert c.pp.nusc is True and c.pp.total_part == 2 and c.mask.total_part == 2 and c.labels.total_part == 2
    assert c.mask.mask_batch == 8 and c.mask.plane_estimate.max_hs == -1.3 and c.labels.nms.threshold == 0.2
    assert list(c.labels.image_shape) == [900, 1600] and c.mask.graph.n_neighbors == 70
    assert "nuscenes" in c.pp.data_paths.track_path and c.mask.ptc_path == "/d/velodyne" and c.labels.calib_path == "/d/calib"
    d = c.to_container()
    assert set(d) >= {"pp", "mask", "labels", "workers", "device", "total_part", "part"} and d["mask"]["data_root"] == "/d"
    with pytest.raises(KeyError):
        sl.compose_all(["data_root=/d", "no_such_key=1"])
    with pytest.raises(KeyError):
        sl.compose_all(["data_root=/d", "mask.no_such_key=1"])


def test_host_read_files_packs_files_back_to_back(tmp_path):
    """modest_host_read_files (the group read of FrameLoader; load_velo_scan's np.fromfile per frame, pointcloud_utils.py:22-25):
    sizes, order, the too-small-buffer answer, empty files and a missing one."""
    import ctypes as C
    from modest_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(3)
    blobs = [rng.integers(0, 256, size=n, dtype=np.uint8) for n in (480016, 0, 16, 1 << 20, 33, 7 * 16)]
    paths = []
    for k, b in enumerate(blobs):
        p = tmp_path / f"{k:06d}.bin"
        b.tofile(p)
        paths.append(str(p).encode())
    n = len(paths)
    arr = (C.c_char_p * n)(*paths)
    sizes = np.zeros(n, dtype=np.uint64)
    total = sum(len(b) for b in blobs)
    for threads in (1, 3, 16):
        sizes[:] = 0
        assert lib.modest_host_read_files(arr, n, None, 0, sizes.ctypes.data, threads) == total   # size query
        assert sizes.tolist() == [len(b) for b in blobs]
        small = np.full(total - 1, 7, dtype=np.uint8)
        assert lib.modest_host_read_files(arr, n, small.ctypes.data, small.size, sizes.ctypes.data, threads) == total
        assert (small == 7).all()   # nothing written
        dst = np.zeros(total + 64, dtype=np.uint8)
        assert lib.modest_host_read_files(arr, n, dst.ctypes.data, dst.size, sizes.ctypes.data, threads) == 0
        assert np.array_equal(dst[:total], np.concatenate(blobs)) and not dst[total:].any()
    bad = (C.c_char_p * 3)(paths[0], str(tmp_path / "missing.bin").encode(), paths[2])
    dst = np.zeros(total, dtype=np.uint8)
    assert lib.modest_host_read_files(bad, 3, dst.ctypes.data, dst.size, sizes.ctypes.data, 2) == -(1 + 2)
    assert lib.modest_host_read_files(bad, 0, None, 0, None, 2) == 0
    assert lib.modest_host_read_files(None, 2, None, 0, None, 2) == -1
