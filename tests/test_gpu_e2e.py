"""GPU: the three CLIs end to end on the reference-generated KITTI tree; every
output file is compared with what the reference's own mains wrote."""
import os
import pickle

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_cli_end_to_end(gpu, golden_dir, tmp_path):
    from modest_amd import config, gen_label_files, generate_mask, pre_compute_pp_score
    from tests.golden_tree import unpack_tree
    g, train, paths = unpack_tree(golden_dir, str(tmp_path))
    out = str(tmp_path / "out")
    origin = int(g["origin"])
    ov = [f"data_root={train}"] + [f"data_paths.{k}={v}" for k, v in paths.items()] + [
        f"data_paths.pp_score_path={out}/pp", f"data_paths.seg_save_dst={out}/seg",
        f"data_paths.bbox_info_save_dst={out}/bbox", f"data_paths.label_file_save_dst={out}/labels"]
    tot = pre_compute_pp_score.main(config.compose("pp_score", ov))
    assert tot["scans"] == 1
    pp = np.load(f"{out}/pp/{origin:06d}.npy")
    assert pp.dtype == np.float32 and pp.shape == g["pp"].shape
    assert np.max(np.abs(pp.astype(np.float64) - g["pp"].astype(np.float64))) <= 1e-6
    # rerun skips (the reference's skip test is broken, ours is not)
    assert pre_compute_pp_score.main(config.compose("pp_score", ov))["scans"] == 0
    # stage 2 consumes the REFERENCE's pp file so that stage parity is isolated
    np.save(f"{out}/pp/{origin:06d}.npy", g["pp"])
    generate_mask.main(config.compose("generate_mask", ov + [f"ransac_seed={int(g['seed']) - origin}"]))
    seg = np.load(f"{out}/seg/{origin:06d}.npy")
    assert seg.dtype == np.int64
    assert os.path.exists(f"{out}/seg/configs.yaml") and os.path.exists(f"{out}/bbox/configs.yaml")
    objs = pickle.load(open(f"{out}/bbox/{origin:06d}.pkl", "rb"))
    assert np.array_equal(seg, g["seg"])
    got = np.array([[*o.t, o.l, o.w, o.h, o.ry, o.volume] for o in objs]).reshape(-1, 8)
    np.testing.assert_allclose(got, g["objs"], rtol=1e-9, atol=1e-12)
    gen_label_files.main(config.compose("generate_label_files", ov))
    txt = open(f"{out}/labels/{origin:06d}.txt").read()
    assert txt == str(g["label_txt"])


def test_cli_end_to_end_nuscenes_config(gpu, golden_dir, tmp_path):
    """BASELINE config 5 in small: nusc=True (remove_center on history frames, rot-z pi/2),
    plane_estimate.max_hs=-1.3, image_shape=[900,1600]; outputs vs the reference's own mains."""
    from modest_amd import config, gen_label_files, generate_mask, pre_compute_pp_score
    from tests.golden_tree import unpack_tree
    g, train, paths = unpack_tree(golden_dir, str(tmp_path), "e2e_tree_nusc.npz")
    out = str(tmp_path / "out")
    origin = int(g["origin"])
    ov = [f"data_root={train}"] + [f"data_paths.{k}={v}" for k, v in paths.items()] + [
        f"data_paths.pp_score_path={out}/pp", f"data_paths.seg_save_dst={out}/seg",
        f"data_paths.bbox_info_save_dst={out}/bbox", f"data_paths.label_file_save_dst={out}/labels"]
    assert pre_compute_pp_score.main(config.compose("pp_score", ov + ["nusc=True"]))["scans"] == 1
    pp = np.load(f"{out}/pp/{origin:06d}.npy")
    assert pp.dtype == np.float32 and pp.shape == g["pp"].shape
    assert np.max(np.abs(pp.astype(np.float64) - g["pp"].astype(np.float64))) <= 1e-6
    np.save(f"{out}/pp/{origin:06d}.npy", g["pp"])
    generate_mask.main(config.compose("generate_mask", ov + ["plane_estimate.max_hs=-1.3",
                                                             f"ransac_seed={int(g['seed']) - origin}"]))
    assert np.array_equal(np.load(f"{out}/seg/{origin:06d}.npy"), g["seg"])
    objs = pickle.load(open(f"{out}/bbox/{origin:06d}.pkl", "rb"))
    got = np.array([[*o.t, o.l, o.w, o.h, o.ry, o.volume] for o in objs]).reshape(-1, 8)
    np.testing.assert_allclose(got, g["objs"], rtol=1e-9, atol=1e-12)
    gen_label_files.main(config.compose("generate_label_files", ov + ["image_shape=[900,1600]"]))
    assert open(f"{out}/labels/{origin:06d}.txt").read() == str(g["label_txt"])


def test_full_pipeline_is_deterministic_and_thread_safe(gpu):
    """The whole seed-label pipeline of one synthetic scan (BASELINE config-3 shape, smaller sizes):
    identical PP scores, labels, boxes and label text when repeated on one context and when run from
    four threads at once (one HIP stream + modest_ctx each, as bench.py's in-process mode does) --
    atomics-based kernels (compaction look-back, union-find, cursor scatter) must not leak their
    scheduling into the results."""
    import os
    import tempfile
    import threading

    import torch
    from modest_amd import _lib, config, ops, synth
    from modest_amd.gen_label_files import gen_label_scan
    from modest_amd.generate_mask import generate_mask_scan
    from modest_amd.utils import kitti_util

    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "c.txt"), "w").write(synth.CALIB_TXT)
        calib = kitti_util.Calibration(os.path.join(d, "c.txt"))
    margs = config.compose("generate_mask", ["data_root=/unused"])
    largs = config.compose("generate_label_files", ["data_root=/unused"])
    sc = synth.make_scan(77, n_live=20_000, n_trav=6, n_frames=12)
    live_raw = torch.from_numpy(sc.live_raw).to(gpu)
    live_xyz = torch.from_numpy(sc.live_xyz).to(gpu)
    hist = torch.from_numpy(np.concatenate(sc.hist)).to(gpu)
    offsets = np.cumsum([0] + [len(h) for h in sc.hist]).astype(np.int64)

    def run(ctx):
        H = ops.pp_score(live_xyz, hist, offsets, 0.3, ctx=ctx)
        pp_host = H.cpu().numpy()
        labels, objs, _ = generate_mask_scan(sc.live_raw, pp_host, calib, margs, random_state=np.random.RandomState(5),
                                             ptc_dev=live_raw, pp_dev=H)
        text, kept = gen_label_scan(objs, calib, largs)
        boxes = np.array([[*o.t, o.l, o.w, o.h, o.ry] for o in objs], dtype=np.float64).reshape(-1, 7)
        return pp_host, labels, boxes, text

    ctx0 = _lib.Context(torch.cuda.current_device())
    ref = run(ctx0)
    assert ref[1].max() >= 3 and len(ref[2]) >= 3, "the scan must produce a few clusters and boxes"
    for _ in range(2):
        got = run(ctx0)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
        assert np.array_equal(got[2], ref[2]) and got[3] == ref[3]

    results, errs = [None] * 4, []

    def worker(k):
        try:
            torch.cuda.set_device(gpu)
            with torch.cuda.stream(torch.cuda.Stream(device=gpu)):
                ctx = _lib.Context(torch.cuda.current_device())
                results[k] = [run(ctx) for _ in range(3)]
        except BaseException as e:   # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for per_thread in results:
        for got in per_thread:
            assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
            assert np.array_equal(got[2], ref[2]) and got[3] == ref[3]


@pytest.mark.parametrize("case", ["dense", "sparse", "no_objects", "low_pp"])
def test_scan_pipeline_equals_oracle_on_synthetic_scans(gpu, case):
    """generate_mask_scan + gen_label_scan of whole synthetic scans against the oracle's restatement
    of the reference loop (same RandomState for the two RANSAC fits): final labels equal, boxes within
    1e-9, label text identical -- on a scan with many objects, a sparse one, one whose points are all
    ground (no cluster survives) and one whose PP scores are high everywhere (every cluster is
    filtered as persistent)."""
    import os
    import tempfile
    import torch
    from modest_amd import config, ops, synth
    from modest_amd.gen_label_files import gen_label_scan
    from modest_amd.generate_mask import generate_mask_scan
    from modest_amd.utils import kitti_util
    from oracle import labels as ol
    from oracle import mask as om

    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "c.txt"), "w").write(synth.CALIB_TXT)
        calib = kitti_util.Calibration(os.path.join(d, "c.txt"))
    margs = config.compose("generate_mask", ["data_root=/unused"])
    largs = config.compose("generate_label_files", ["data_root=/unused"])
    n_live = {"dense": 24_000, "sparse": 3_000, "no_objects": 12_000, "low_pp": 12_000}[case]
    sc = synth.make_scan(31 + len(case), n_live=n_live, n_trav=4, n_frames=8)
    raw = sc.live_raw.copy()
    if case == "no_objects":
        raw = raw[raw[:, 2] < -1.45]                   # ground returns only
    live_xyz = np.ascontiguousarray(raw[:, :3])
    hist = np.concatenate(sc.hist)
    offsets = np.cumsum([0] + [len(h) for h in sc.hist]).astype(np.int64)
    H = ops.pp_score(torch.from_numpy(live_xyz).to(gpu), torch.from_numpy(hist).to(gpu), offsets, 0.3)
    if case == "low_pp":
        H = torch.full_like(H, 0.95)                   # persistent everywhere: min_percentile_pp_score filters all
    pp = H.cpu().numpy()
    labels, objs, _ = generate_mask_scan(raw, pp, calib, margs, random_state=np.random.RandomState(9),
                                         ptc_dev=torch.from_numpy(raw).to(gpu), pp_dev=H)
    text, _ = gen_label_scan(objs, calib, largs)
    ref = om.generate_mask_scan(raw, pp, calib, random_state=np.random.RandomState(9), n_jobs=1)
    ref_text, _ = ol.gen_label_scan(ref["objs"], calib)
    assert np.array_equal(labels, ref["labels"])
    assert len(objs) == len(ref["objs"])
    if case in ("no_objects", "low_pp"):
        assert len(objs) == 0 and labels.max() == 0
    for o, r in zip(objs, ref["objs"]):
        np.testing.assert_allclose([*o.t, o.l, o.w, o.h, o.ry, o.volume], [*r.t, r.l, r.w, r.h, r.ry, r.volume],
                                   rtol=1e-9, atol=1e-9)
    assert text == ref_text


def test_library_stage_driver_equals_python_statement(gpu):
    """The WHOLE scan in the library -- modest_mask_stage (both RANSAC fits, mask, graph + DBSCAN, cluster
    statistics, validity rules, relabelling), modest_scan_boxes (members, rect points, box fit, get_obj, volume
    gate, final labels), modest_objs_iou + modest_label_lines (NMS walk, FOV filter, label text) -- against the
    Python statement of the same steps: labels, boxes,
    label text AND the generator state after the scan (the CLI keeps one stream per scan; the reference
    its global one) -- Lyft and nuScenes-style configurations, several scans and seeds, the global
    generator, and an input the library hands back (a tiny scan)."""
    import os
    import tempfile
    import torch
    from modest_amd import config, generate_mask as gm, gen_label_files as glf, synth
    from modest_amd.gen_label_files import gen_label_scan
    from modest_amd.utils import kitti_util
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "c.txt"), "w").write(synth.CALIB_TXT)
        calib = kitti_util.Calibration(os.path.join(d, "c.txt"))
    largs = config.compose("generate_label_files", ["data_root=/unused"])
    cfgs = [config.compose("generate_mask", ["data_root=/unused"]),
            config.compose("generate_mask", ["data_root=/unused", "plane_estimate.max_hs=-1.3", "filtering.percentile=10",
                                             "graph.n_neighbors=40", "limit_range=[[0,70],[-40,40]]"])]
    rng = np.random.default_rng(0)
    n_boxes = 0
    for k in range(6):
        sc = synth.make_scan(100 + k, n_live=int(rng.choice([4000, 12000, 30000])), n_trav=2, n_frames=1)
        raw = sc.live_raw
        pp = np.clip(0.45 + 0.5 * np.sin(raw[:, 0] * 0.3) + rng.normal(0, 0.05, len(raw)), 0, 1).astype(np.float32)
        dev, ppd = torch.from_numpy(raw).to(gpu), torch.from_numpy(pp).to(gpu)
        margs = cfgs[k % 2]
        out = []
        for native in (True, False):
            gm.NATIVE_STAGE = gm.NATIVE_BOXES = glf.NATIVE_LABELS = native
            try:
                if k == 5:    # the reference's own mode: numpy's global generator
                    np.random.seed(77)
                    rs = None
                else:
                    rs = np.random.RandomState(40 + k)
                labels, objs, info = gm.generate_mask_scan(raw, pp, calib, margs, random_state=rs, ptc_dev=dev, pp_dev=ppd)
                text, _ = gen_label_scan(objs, calib, largs)
                st = (np.random.mtrand._rand if rs is None else rs).get_state()
                out.append((labels, [(*o.t, o.l, o.w, o.h, o.ry, o.volume) for o in objs], text, info["plane"], st))
            finally:
                gm.NATIVE_STAGE = gm.NATIVE_BOXES = glf.NATIVE_LABELS = True
        a, b = out
        assert np.array_equal(a[0], b[0]) and a[1] == b[1] and a[2] == b[2], k
        assert np.array_equal(a[3], b[3])
        assert np.array_equal(a[4][1], b[4][1]) and a[4][2:] == b[4][2:], k
        n_boxes += len(a[1])
    assert n_boxes >= 10
    # a scan with fewer ground candidates than sklearn's tracking selection needs: handed back, same result
    sc = synth.make_scan(7, n_live=700, n_trav=2, n_frames=1)
    raw, pp = sc.live_raw, np.full(len(sc.live_raw), 0.2, dtype=np.float32)
    res = []
    for native in (True, False):
        gm.NATIVE_STAGE = native
        try:
            res.append(gm.generate_mask_scan(raw, pp, calib, cfgs[0], random_state=np.random.RandomState(1))[0])
        except ValueError as e:
            res.append(str(e))
        finally:
            gm.NATIVE_STAGE = True
    assert (isinstance(res[0], str) and res[0] == res[1]) or np.array_equal(res[0], res[1])


def test_bench_contract_json_line(gpu):
    """`python bench.py` prints exactly one JSON line with the driver's contract fields, the roofline
    object of the PP stage and the cpu_baseline object (small sizes, two helper processes)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "16", "--warmup", "2", "--procs", "2", "--scans", "2",
           "--cpu-scans", "1", "--cpu-best-effort", "2", "--cli-scans", "3", "--cli-workers", "2", "--n-live", "8000", "--traversals", "3",
           "--frames", "6"]   # (two CLI workers for three scans: six would only start interpreters)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 16 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["unit"] == "scans/s" and d["scaling"] == "weak" and d["data"] == "synthetic" and d["value"] > 0
    assert abs(d["ms_per_step"] * d["value"] - 1000.0) < 1e-6 * 1000.0
    assert "workload" in d["config"] and "model" not in d["config"]
    # `roofline`: the isolated PP measurement on reference-rule windows with traversals that enter and leave (what a real
    # valid_idx_info.pkl holds); `roofline_best_case`: the same on the windows of frames i..i+F-1 the timed regions run on
    rr = d["roofline"]
    assert rr["bound"] == "hbm" and rr["unit"] == "GB/s" and rr["peak"] == 8000.0 and "kernel" in rr
    assert rr["achieved"] > 0 and abs(rr["frac"] - rr["achieved"] / rr["peak"]) < 1e-12
    assert abs(rr["achieved"] - rr["algorithmic_bytes_per_launch"] / (rr["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * rr["achieved"]
    assert rr["sharing"]["distinct_traversal_counts"] >= 1 and "split_traintest.py" in rr["sharing"]["rule"]
    rf = d["roofline_best_case"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert rf["achieved"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["in_pipeline"]["scans_timed"] == 16 and rf["isolated"]["kernel_ms"] > 0
    # two resident scans (consecutive scans of one shard) -> two scans per PP call: bytes per call and per scan must say so
    assert rf["scans_per_launch"] == 2 and rf["algorithmic_bytes_per_launch"] == 2 * rf["algorithmic_bytes_per_scan"]
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * rf["achieved"]
    assert rf["block_path"] is False   # fewer than 6 scans per call: every scan streams its own frames
    assert d["parity"]["pp_scans_checked"] == 1 and d["parity"]["pp_scans_per_call"] == 2   # one CPU sample scan
    assert d["config"]["host_processes_per_gpu"] == 2 and d["config"]["rccl_world_size"] == 1
    # 8 steps per helper: the pool was used whole; fewer than 24 per helper: the steady-state figure is added
    assert d["steady_state"]["host_processes_per_gpu"] == 2 and d["steady_state"]["value"] > 0
    assert d["config"]["history_input"].startswith("frame store")
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    assert cb["best_effort"]["value"] > 0
    assert d["parity"]["pp_counts_equal"] is True and d["parity"]["labels_equal"] is True
    assert d["parity"]["label_text_equal"] is True
    assert d["cli"]["label_files"] == 3 and d["cli"]["pp_scans_per_s"] > 0 and d["cli"]["mask_scans_per_s"] > 0


def _pp_cli(tmp_path, golden_dir, tag, extra):
    from modest_amd import config, pre_compute_pp_score
    from tests.golden_tree import unpack_tree
    root = tmp_path / tag
    g, train, paths = unpack_tree(golden_dir, str(root), name="pp_branches.npz")
    out = str(root / "out")
    ov = [f"data_root={train}"] + [f"data_paths.{k}={v}" for k, v in paths.items()] + [f"data_paths.pp_score_path={out}/pp"]
    tot = pre_compute_pp_score.main(config.compose("pp_score", ov + extra))
    return g, out, tot


def test_pp_cli_non_default_branches(gpu, golden_dir, tmp_path):
    """SURVEY 8f-3, the PP CLI's own branches against what the REFERENCE's pre_compute_pp_score.main wrote on
    the same tree (tools/make_golden_pp_branches.py: three history traversals, two live scans):
    limit_traversals (:181-186), add_random_noise (:175-179; numpy's global generator, consumed scan after
    scan, left in the same state), the two dump options (:152-155, :168-171) and skip_ephe (:172-173)."""
    import pickle
    g, out, tot = _pp_cli(tmp_path, golden_dir, "default", [])
    origins = [int(x) for x in g["origins"]]
    assert tot["scans"] == len(origins) == 2
    for o in origins:
        assert np.max(np.abs(np.load(f"{out}/pp/{o:06d}.npy").astype(np.float64) - g[f"pp_default_{o}"])) <= 1e-6
    # limit_traversals=2 of the three history traversals
    g, out, tot = _pp_cli(tmp_path, golden_dir, "limit2", ["limit_traversals=2"])
    for o in origins:
        pp = np.load(f"{out}/pp/{o:06d}.npy")
        assert pp.dtype == np.float32 and np.max(np.abs(pp.astype(np.float64) - g[f"pp_limit2_{o}"])) <= 1e-6
        assert np.max(np.abs(pp - g[f"pp_default_{o}"])) > 1e-3          # the branch does something
    # add_random_noise: seeded as the reference run was; the draws of scan 2 follow those of scan 1
    np.random.seed(int(g["noise_seed"]))
    g, out, tot = _pp_cli(tmp_path, golden_dir, "noise", ["add_random_noise=0.05"])
    for o in origins:
        assert np.max(np.abs(np.load(f"{out}/pp/{o:06d}.npy").astype(np.float64) - g[f"pp_noise_{o}"])) <= 1e-6
    assert np.random.uniform() == float(g["noise_next_draw"])
    # dumps + skip_ephe: relative pose of the live scan and the stacked, transformed history per traversal
    lid, tm = str(tmp_path / "lid"), str(tmp_path / "tm")
    g, out, tot = _pp_cli(tmp_path, golden_dir, "dumps",
                          [f"data_paths.load_precomputed_lidars={lid}", f"data_paths.load_save_precomputed_trans_mat={tm}",
                           "skip_ephe=True"])
    assert tot["scans"] == 0 and not os.listdir(f"{out}/pp")
    for o in origins:
        t = np.load(f"{tm}/{o:06d}.npy")
        assert t.dtype == np.float32 and np.array_equal(t, g[f"trans_{o}"])
        comb = pickle.load(open(f"{lid}/{o:06d}.pkl", "rb"))
        assert sorted(comb) == [int(k) for k in g[f"lidar_keys_{o}"]]
        for k in comb:
            assert comb[k].dtype == np.float32 and np.array_equal(comb[k], g[f"lidar_{o}_{k}"]), (o, k)


def test_native_box_and_label_stage_many_scans(gpu):
    """modest_scan_boxes / modest_objs_iou / modest_label_lines against the Python statement (get_objs, the
    volume gate, relabel_after_drop, objs_nms, is_within_fov, objs2label) on 40 scans with the mask stage
    shared: boxes field by field, final labels, the boxes NMS + FOV keep, label text; row mode == object mode."""
    import os
    import tempfile
    import torch
    from modest_amd import config, generate_mask as gm, gen_label_files as glf, synth
    from modest_amd.utils import kitti_util
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "c.txt"), "w").write(synth.CALIB_TXT)
        calib = kitti_util.Calibration(os.path.join(d, "c.txt"))
    largs = [config.compose("generate_label_files", ["data_root=/unused"]),
             config.compose("generate_label_files", ["data_root=/unused", "image_shape=[900,1600]", "nms.threshold=0.3"]),
             config.compose("generate_label_files", ["data_root=/unused", "fov_only=False", "nms.enable=False"])]
    margs = config.compose("generate_mask", ["data_root=/unused"])
    rng = np.random.default_rng(3)
    n_boxes = n_lines = 0
    worst = 0.0
    for k in range(40):
        sc = synth.make_scan(300 + k, n_live=int(rng.choice([8000, 20000, 30000])), n_trav=2, n_frames=1)
        raw = np.ascontiguousarray(sc.live_raw)
        pp = np.clip(0.45 + 0.5 * np.sin(raw[:, 0] * 0.3 + k) + rng.normal(0, 0.05, len(raw)), 0, 1).astype(np.float32)
        dev, ppd = torch.from_numpy(raw).to(gpu), torch.from_numpy(pp).to(gpu)
        res = []
        for native in (True, False):
            gm.NATIVE_BOXES = glf.NATIVE_LABELS = native
            try:
                labels, objs, _ = gm.generate_mask_scan(raw, pp, calib, margs, random_state=np.random.RandomState(k),
                                                        ptc_dev=dev, pp_dev=ppd)
                text, kept = glf.gen_label_scan(objs, calib, largs[k % 3])
                res.append((labels, np.array([[*o.t, o.l, o.w, o.h, o.ry, o.volume] for o in objs]).reshape(-1, 8), text,
                            np.array([[*o.t, o.l] for o in kept]).reshape(-1, 4)))
            finally:
                gm.NATIVE_BOXES = glf.NATIVE_LABELS = True
        a, b = res
        assert np.array_equal(a[0], b[0]) and a[1].shape == b[1].shape, k
        if a[1].size:
            worst = max(worst, float(np.max(np.abs(a[1] - b[1]) / np.maximum(np.abs(b[1]), 1e-6))))
        assert np.allclose(a[1], b[1], rtol=1e-12, atol=1e-13), (k, np.abs(a[1] - b[1]).max())
        assert a[2] == b[2] and np.allclose(a[3], b[3], rtol=1e-12, atol=1e-13), k
        # row mode (what an in-memory pipeline uses) gives the same text
        labels_r, rows, _ = gm.generate_mask_scan(raw, pp, calib, margs, random_state=np.random.RandomState(k),
                                                   ptc_dev=dev, pp_dev=ppd, as_rows=True)
        assert isinstance(rows, np.ndarray) and np.array_equal(rows, a[1]) and np.array_equal(labels_r, a[0])
        assert glf.gen_label_scan(rows, calib, largs[k % 3])[0] == a[2]
        n_boxes += len(a[1])
        n_lines += len(a[2].splitlines())
    print("boxes", n_boxes, "label lines", n_lines, "largest relative difference", worst)
    assert n_boxes >= 200 and n_lines >= 40


def test_results_do_not_depend_on_load(gpu):
    """Determinism soak (tools/soak_mask.py): 6 processes repeat the same scans through the PP stage and the
    mask / box / label stages and compare every result with their first.  Round 3 found 11 differing results
    in 14 400 runs here -- the "last block finishes the reduction" kernels took their ticket before the other
    wavefronts' partial results had left the CU -- and fixed it (csrc/common.h: modest_drain_stores); with the
    bug this test fails with probability ~0.9."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_mask.py"), "6", "200"], cwd=root,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "mismatches: 0" in r.stdout, r.stdout[-2000:]


def test_mask_stage_chain_equals_separate_calls(gpu, tmp_path):
    """modest_mask_stage_batch: the mask stage of several scans as one chain of launches (mask / graph / DBSCAN block and
    cluster statistics: the scan is a second grid dimension) gives the labels, planes, boxes and generator states of
    separate calls -- scans of different size, a scan whose candidate set is too small (handed back to the host
    statement inside a chain), a chain of seven."""
    import torch
    from modest_amd import config, generate_mask as gm, synth
    from modest_amd.utils import kitti_util
    open(tmp_path / "c.txt", "w").write(synth.CALIB_TXT)
    calib = kitti_util.Calibration(str(tmp_path / "c.txt"))
    args = config.compose("generate_mask", ["data_root=/unused"])
    scans = []
    for k, n_live in enumerate([30000, 12000, 21000, 400, 30000, 8000, 17000]):
        s = synth.make_scan(70 + k, n_live=n_live, n_trav=2, n_frames=1)
        raw = np.ascontiguousarray(s.live_raw)
        rng = np.random.default_rng(k)
        pp = np.clip(0.5 + 0.5 * np.sin(raw[:, 0] * 0.3) + rng.normal(0, 0.03, len(raw)), 0, 1).astype(np.float32)
        scans.append((raw, pp, torch.from_numpy(raw).to(gpu), torch.from_numpy(pp).to(gpu)))
    single, states = [], []
    for k, (raw, pp, rd, pd) in enumerate(scans):
        rs = np.random.RandomState(100 + k)
        single.append(gm.generate_mask_scan(raw, pp, calib, args, random_state=rs, ptc_dev=rd, pp_dev=pd, as_rows=True))
        states.append(rs.get_state())
    for group in ([0, 1, 2, 3, 4, 5, 6], [4, 0], [1, 2, 5]):
        rss = [np.random.RandomState(100 + k) for k in group]
        chain = gm.generate_mask_chain([dict(ptc=scans[k][0], pp_score=scans[k][1], random_state=rs, ptc_dev=scans[k][2],
                                             pp_dev=scans[k][3]) for k, rs in zip(group, rss)], calib, args, as_rows=True)
        for k, rs, (labels, rows, info) in zip(group, rss, chain):
            assert np.array_equal(labels, single[k][0]), k
            assert np.array_equal(rows, single[k][1]), k
            assert np.array_equal(info["plane"], single[k][2]["plane"]) and info["n_kept"] == single[k][2]["n_kept"], k
            a, b = rs.get_state(), states[k]
            assert a[2] == b[2] and np.array_equal(a[1], b[1]), k
    assert any(len(r[1]) > 0 for r in single)


def test_one_call_chain_equals_the_separate_chain_calls(gpu, tmp_path, monkeypatch):
    """modest_seed_chain (mask stage + box tail + IoU matrices of the kept boxes of a chain behind ONE library call)
    against the three separate chain calls it replaces: final labels, box rows, planes, generator states, the IoU matrices the label
    stage's NMS reads and the label text -- with a scan the mask stage hands back inside the chain (a candidate set of <= 300 points)
    and, with room for two boxes only, scans whose box tail goes back to the separate call."""
    import torch
    from modest_amd import config, generate_mask as gm, ops, synth
    from modest_amd.gen_label_files import gen_label_chain
    from modest_amd.utils import kitti_util
    open(tmp_path / "c.txt", "w").write(synth.CALIB_TXT)
    calib = kitti_util.Calibration(str(tmp_path / "c.txt"))
    margs = config.compose("generate_mask", ["data_root=/unused"])
    largs = config.compose("generate_label_files", ["data_root=/unused"])
    scans = []
    for k, n_live in enumerate([30000, 12000, 400, 21000, 8000]):
        s = synth.make_scan(170 + k, n_live=n_live, n_trav=2, n_frames=1)
        raw = np.ascontiguousarray(s.live_raw)
        rng = np.random.default_rng(k)
        pp = np.clip(0.5 + 0.5 * np.sin(raw[:, 0] * 0.3) + rng.normal(0, 0.03, len(raw)), 0, 1).astype(np.float32)
        scans.append((raw, pp, torch.from_numpy(raw).to(gpu), torch.from_numpy(pp).to(gpu)))

    def run(one_call):
        monkeypatch.setattr(gm, "ONE_CALL_CHAIN", one_call)
        rss = [np.random.RandomState(300 + k) for k in range(len(scans))]
        res = gm.generate_mask_chain([dict(ptc=r, pp_score=p, random_state=rs, ptc_dev=rd, pp_dev=pd) for (r, p, rd, pd), rs in zip(scans, rss)],
                                     calib, margs, as_rows=True, with_iou=True)
        lab = gen_label_chain([r[1] for r in res], calib, largs, ious=[r[3] for r in res])
        return res, lab, [rs.get_state() for rs in rss]

    ref, ref_lab, ref_st = run(False)
    assert all(r[3] is None for r in ref) and any(len(r[1]) > 2 for r in ref)
    for cap in (ops.SEED_MAX_BOXES, 2):
        monkeypatch.setattr(ops, "SEED_MAX_BOXES", cap)
        got, got_lab, got_st = run(True)
        n_iou = 0
        for k, (a, b) in enumerate(zip(ref, got)):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (cap, k)
            assert np.array_equal(a[2]["plane"], b[2]["plane"]) and a[2]["n_kept"] == b[2]["n_kept"], (cap, k)
            assert ref_st[k][2] == got_st[k][2] and np.array_equal(ref_st[k][1], got_st[k][1]), (cap, k)
            assert ref_lab[k][0] == got_lab[k][0] and np.array_equal(ref_lab[k][1], got_lab[k][1]), (cap, k)
            if b[3] is not None and len(b[1]):
                assert np.array_equal(b[3], ops.objs_iou(b[1])), (cap, k)   # the matrices equal modest_objs_iou's, bit for bit
                n_iou += 1
        assert n_iou >= (2 if cap > 2 else 0)


def test_label_chain_equals_separate_calls(gpu, tmp_path):
    """gen_label_chain: the IoU matrices of several scans' box sets from one launch (modest_objs_iou_batch); label text
    and kept boxes equal gen_label_scan's, incl. an empty box set inside the chain."""
    from modest_amd import config, synth
    from modest_amd.gen_label_files import gen_label_chain, gen_label_scan
    from modest_amd.utils import kitti_util
    open(tmp_path / "c.txt", "w").write(synth.CALIB_TXT)
    calib = kitti_util.Calibration(str(tmp_path / "c.txt"))
    args = config.compose("generate_label_files", ["data_root=/unused"])
    rng = np.random.default_rng(5)
    sets = []
    for k in (13, 0, 40, 1, 7):
        t = np.c_[rng.uniform(5, 40, k), rng.uniform(-1, 1, k), rng.uniform(-15, 15, k)]
        sets.append(np.c_[t, rng.uniform(1, 5, k), rng.uniform(1, 2.5, k), rng.uniform(1, 2, k), rng.uniform(-3.1, 3.1, k),
                          rng.uniform(2, 30, k)].astype(np.float64).reshape(-1, 8))
    sets[2][5] = sets[2][4]   # a duplicate box: ties in the self-IoU order
    single = [gen_label_scan(r, calib, args) for r in sets]
    chain = gen_label_chain(sets, calib, args)
    for (t0, k0), (t1, k1) in zip(single, chain):
        assert t0 == t1 and np.array_equal(k0, k1)


def test_mask_stage_chain_with_a_scan_that_keeps_too_few_rows(gpu, tmp_path):
    """A chain in which one scan keeps fewer rows than the graph has neighbours (sklearn's kneighbors raises, the library
    hands the scan back, the host statement raises): the chain raises like the separate call, and the contexts it used
    (their persistent cell counters hold that scan's counts) serve the next chain correctly."""
    import torch
    from modest_amd import config, generate_mask as gm, synth
    from modest_amd._lib import ModestHipError
    from modest_amd.utils import kitti_util
    open(tmp_path / "c.txt", "w").write(synth.CALIB_TXT)
    calib = kitti_util.Calibration(str(tmp_path / "c.txt"))
    args = config.compose("generate_mask", ["data_root=/unused"])
    rng = np.random.default_rng(3)
    # a flat ground (both fits find it, the mask removes it) and 40 points above it: 40 kept rows < 70 neighbours
    g = np.c_[rng.uniform(-40, 40, 6000), rng.uniform(-15, 15, 6000), rng.normal(-1.7, 0.01, 6000), rng.uniform(0, 1, 6000)]
    up = np.c_[rng.uniform(5, 8, 40), rng.uniform(-2, 2, 40), rng.uniform(-0.5, 0.5, 40), rng.uniform(0, 1, 40)]
    few = np.ascontiguousarray(np.r_[g, up].astype(np.float32))
    scans = [few]
    for k in range(3):
        scans.append(np.ascontiguousarray(synth.make_scan(90 + k, n_live=15000, n_trav=2, n_frames=1).live_raw))
    pps = [np.clip(0.5 + 0.5 * np.sin(r[:, 0] * 0.3), 0, 1).astype(np.float32) for r in scans]
    dev = [(torch.from_numpy(r).to(gpu), torch.from_numpy(p).to(gpu)) for r, p in zip(scans, pps)]

    def items(idx):
        return [dict(ptc=scans[k], pp_score=pps[k], random_state=np.random.RandomState(7 + k), ptc_dev=dev[k][0], pp_dev=dev[k][1])
                for k in idx]

    with pytest.raises((ModestHipError, ValueError)):
        gm.generate_mask_scan(scans[0], pps[0], calib, args, random_state=np.random.RandomState(7), ptc_dev=dev[0][0], pp_dev=dev[0][1])
    single = [gm.generate_mask_scan(scans[k], pps[k], calib, args, random_state=np.random.RandomState(7 + k), ptc_dev=dev[k][0],
                                    pp_dev=dev[k][1], as_rows=True) for k in (1, 2, 3)]
    with pytest.raises((ModestHipError, ValueError)):
        gm.generate_mask_chain(items([1, 0, 2, 3]), calib, args, as_rows=True)
    for _ in range(2):   # the same contexts again, twice: counters left by the failed chain must not leak into these
        chain = gm.generate_mask_chain(items([1, 2, 3]), calib, args, as_rows=True)
        for (l0, r0, i0), (l1, r1, i1) in zip(single, chain):
            assert np.array_equal(l0, l1) and np.array_equal(r0, r1) and np.array_equal(i0["plane"], i1["plane"])


def test_mask_stage_chains_from_concurrent_threads(gpu, tmp_path):
    """Three host threads (a stream and their own contexts each) run chains of scans at the same time: every result equals
    the single-threaded, scan-by-scan one (launch capture, device tables and staging slots are per thread / per context)."""
    import threading
    import torch
    from modest_amd import config, generate_mask as gm, synth
    from modest_amd.gen_label_files import gen_label_chain, gen_label_scan
    from modest_amd.utils import kitti_util
    open(tmp_path / "c.txt", "w").write(synth.CALIB_TXT)
    calib = kitti_util.Calibration(str(tmp_path / "c.txt"))
    args = config.compose("generate_mask", ["data_root=/unused"])
    largs = config.compose("generate_label_files", ["data_root=/unused"])
    scans = []
    for k in range(4):
        raw = np.ascontiguousarray(synth.make_scan(120 + k, n_live=20000, n_trav=2, n_frames=1).live_raw)
        pp = np.clip(0.5 + 0.5 * np.sin(raw[:, 0] * 0.3), 0, 1).astype(np.float32)
        scans.append((raw, pp, torch.from_numpy(raw).to(gpu), torch.from_numpy(pp).to(gpu)))
    single = []
    for k, (raw, pp, rd, pd) in enumerate(scans):
        labels, rows, _ = gm.generate_mask_scan(raw, pp, calib, args, random_state=np.random.RandomState(k), ptc_dev=rd, pp_dev=pd,
                                                as_rows=True)
        single.append((labels, rows, gen_label_scan(rows, calib, largs)[0]))
    bad, errs = [], []

    def work(t):
        try:
            torch.cuda.set_device(gpu)
            with torch.cuda.stream(torch.cuda.Stream(device=gpu)):
                for rep in range(6):
                    order = [(t + rep + q) % 4 for q in range(4)]
                    res = gm.generate_mask_chain([dict(ptc=scans[k][0], pp_score=scans[k][1], random_state=np.random.RandomState(k),
                                                       ptc_dev=scans[k][2], pp_dev=scans[k][3]) for k in order], calib, args, as_rows=True)
                    lab = gen_label_chain([r[1] for r in res], calib, largs)
                    for k, (labels, rows, _), (text, _) in zip(order, res, lab):
                        if not (np.array_equal(labels, single[k][0]) and np.array_equal(rows, single[k][1]) and text == single[k][2]):
                            bad.append((t, rep, k))
        except Exception as e:   # surfaced below
            errs.append(e)

    th = [threading.Thread(target=work, args=(t,)) for t in range(3)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs[0]
    assert not bad, bad[:5]
