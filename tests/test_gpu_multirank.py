"""GPU: the N > 1 path of the three CLIs on the HIP path without a multi-GPU node (BASELINE
config 4 in small): WORLD_SIZE = 2 with both ranks computing on GPU 0 (MODEST_DIST_BACKEND=gloo: RCCL refuses
two ranks on one device; the collectives here are a barrier and a counter all-reduce) over a 200-scan
synthetic KITTI tree.  The union of the output files must be byte-identical to a one-rank run, to the
reference's manual total_part=2 runs (pre_compute_pp_score.py:114-116) and to a run with the dynamic work
queue; the per-rank scan counts and busy times are written to gpurun_out/multirank_report.json."""
import filecmp
import json
import os
import re
import signal
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SCANS = 200
CLIS = ("pre_compute_pp_score", "generate_mask", "gen_label_files")
CFG = {"pre_compute_pp_score": "pp", "generate_mask": "seg", "gen_label_files": "labels"}


def _overrides(train, paths, out):
    return [f"data_root={train}"] + \
           [f"data_paths.{k}={v}" for k, v in paths.items()] + \
           [f"data_paths.pp_score_path={out}/pp", f"data_paths.seg_save_dst={out}/seg",
            f"data_paths.bbox_info_save_dst={out}/bbox", f"data_paths.label_file_save_dst={out}/labels"]


def _run(module, ov, ranks, port, extra_env=None):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4", **(extra_env or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    if ranks == 1:
        cmd = [sys.executable, "-m", f"modest_amd.{module}"] + ov
    else:
        env["MODEST_DIST_BACKEND"] = "gloo"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
               "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", f"modest_amd.{module}"] + ov
    _, err = _exec(cmd, env, int(os.environ.get("MODEST_TEST_CLI_TIMEOUT", "600")), module)
    return err


class _Done:
    def __init__(self, returncode, stdout, stderr):
        self.returncode, self.stdout, self.stderr = returncode, stdout, stderr


def _exec(cmd, env, timeout, what="run"):
    """subprocess.run with a post-mortem: a run that hangs dumps the stacks of every thread of every process of its session
    (faulthandler on SIGABRT) into the assertion message before it is killed; a non-zero exit code fails with the tail of stderr."""
    env = dict(env, PYTHONFAULTHANDLER="1")
    p = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGABRT)
        try:
            out, err = p.communicate(timeout=30)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)
            out, err = p.communicate()
        raise AssertionError((what, "timed out after %d s" % timeout, err[-12000:]))
    assert p.returncode == 0, (what, err[-3000:])
    return out, err


def _same_tree(a, b):
    for sub in ("pp", "seg", "bbox", "labels"):
        fa = sorted(f for f in os.listdir(os.path.join(a, sub)) if f != "configs.yaml")
        fb = sorted(f for f in os.listdir(os.path.join(b, sub)) if f != "configs.yaml")
        assert fa == fb and len(fa) == N_SCANS, (sub, len(fa), len(fb))
        match, mismatch, err = filecmp.cmpfiles(os.path.join(a, sub), os.path.join(b, sub), fa, shallow=False)
        assert not mismatch and not err, (sub, mismatch[:5], err[:5])


def test_three_clis_two_ranks_one_gpu(gpu, tmp_path):
    from modest_amd import synth
    root, meta = str(tmp_path / "data"), str(tmp_path / "meta")
    paths = synth.write_kitti_tree(root, meta, n_seq=3, n_frames=N_SCANS + 6, n_pts=3000, origins=tuple(range(N_SCANS)),
                                   hist_frames=6, max_range=60.0)
    train = os.path.join(root, "training")
    outs = {k: str(tmp_path / k) for k in ("one", "two", "parts", "queue")}
    report = {}
    for m in CLIS:                                                     # one rank
        _run(m, _overrides(train, paths, outs["one"]), 1, 0)
    for i, m in enumerate(CLIS):                                       # two ranks, static split
        err = _run(m, _overrides(train, paths, outs["two"]), 2, 29541 + i)
        mm = re.search(r"ranks: scans (\d+)\.\.(\d+), busy ([\d.]+)\.\.([\d.]+) s \(imbalance ([\d.]+) %\)", err)
        assert mm, err[-1500:]
        assert int(mm.group(1)) + int(mm.group(2)) == N_SCANS and int(mm.group(1)) == N_SCANS // 2
        report[m] = dict(rank_scans=[int(mm.group(1)), int(mm.group(2))], busy_seconds=[float(mm.group(3)), float(mm.group(4))],
                         imbalance_percent=float(mm.group(5)))
    for m in CLIS:                                                     # the reference's manual split
        for part in range(2):
            _run(m, _overrides(train, paths, outs["parts"]) + ["total_part=2", f"part={part}"], 1, 0)
    for i, m in enumerate(CLIS):                                       # two ranks, dynamic queue
        err = _run(m, _overrides(train, paths, outs["queue"]) + ["work_queue=dynamic", "queue_chunk=16"], 2, 29551 + i)
        mm = re.search(r"ranks: scans (\d+)\.\.(\d+), busy ([\d.]+)\.\.([\d.]+) s \(imbalance ([\d.]+) %\)", err)
        assert mm and int(mm.group(1)) + int(mm.group(2)) == N_SCANS, err[-1500:]
        report[m + "_dynamic"] = dict(rank_scans=[int(mm.group(1)), int(mm.group(2))],
                                      busy_seconds=[float(mm.group(3)), float(mm.group(4))], imbalance_percent=float(mm.group(5)))
    for k in ("two", "parts", "queue"):
        _same_tree(outs["one"], outs[k])
    # scans per chain of launches of the mask stage (mask_batch, default 4) do not change a label or box file
    for nb in (1, 5):
        o = str(tmp_path / f"mbatch{nb}")
        ov = _overrides(train, paths, o)
        ov = [x if not x.startswith("data_paths.pp_score_path=") else f"data_paths.pp_score_path={outs['one']}/pp" for x in ov]
        _run(CLIS[1], ov + [f"mask_batch={nb}"], 1, 0)
        for sub in ("seg", "bbox"):
            fa = sorted(f for f in os.listdir(os.path.join(o, sub)) if f != "configs.yaml")
            assert len(fa) == N_SCANS
            match, mismatch, err = filecmp.cmpfiles(os.path.join(outs["one"], sub), os.path.join(o, sub), fa, shallow=False)
            assert not mismatch and not err, (nb, sub, mismatch[:5], err[:5])
    # scans per chain of launches (pp_batch, default 4) do not change a score file
    for nb in (1, 7):
        o = str(tmp_path / f"batch{nb}")
        _run(CLIS[0], _overrides(train, paths, o) + [f"pp_batch={nb}"], 1, 0)
        fa = sorted(f for f in os.listdir(os.path.join(o, "pp")) if f != "configs.yaml")
        assert len(fa) == N_SCANS
        match, mismatch, err = filecmp.cmpfiles(os.path.join(outs["one"], "pp"), os.path.join(o, "pp"), fa, shallow=False)
        assert not mismatch and not err, (nb, mismatch[:5], err[:5])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "multirank_report.json"), "w") as f:
        json.dump(dict(scans=N_SCANS, world_size=2, note="both ranks on GPU 0, gloo collectives", per_cli=report), f, indent=1)


def test_rccl_process_group_at_world_size_one(gpu, tmp_path):
    """The RCCL branches of dist.init / barrier(device_ids) / reduce_counters (CUDA tensors) on the hardware there is: the PP
    CLI under `torch.distributed.run --nproc-per-node 1` with MODEST_DIST_FORCE=1 initialises an `nccl` process group
    of one rank, passes the barrier, counts its ranks with an all-reduce of ones -- and writes the files of a plain run
    (the sharding the group serves: pre_compute_pp_score.py:114-116)."""
    from modest_amd import synth
    root, meta = str(tmp_path / "data"), str(tmp_path / "meta")
    n = 12
    paths = synth.write_kitti_tree(root, meta, n_seq=3, n_frames=n + 5, n_pts=3000, origins=tuple(range(n)), hist_frames=5,
                                   max_range=60.0)
    train = os.path.join(root, "training")
    plain, forced = str(tmp_path / "plain"), str(tmp_path / "forced")
    _run("pre_compute_pp_score", _overrides(train, paths, plain), 1, 0)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4", MODEST_DIST_FORCE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MODEST_DIST_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29533", "-m", "modest_amd.pre_compute_pp_score"] + _overrides(train, paths, forced)
    r = _Done(0, *_exec(cmd, env, 900))
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stderr.splitlines() if l.startswith("[dist] ")]
    assert line, r.stderr[-2000:]
    got = dict(kv.split("=") for kv in line[0][len("[dist] "):].split())
    assert got == {"backend": "nccl", "world_size": "1", "ranks_seen": "1", "barrier": "ok"}, got
    fa = sorted(os.listdir(os.path.join(plain, "pp")))
    assert fa == sorted(os.listdir(os.path.join(forced, "pp"))) and len(fa) == n
    match, mismatch, err = filecmp.cmpfiles(os.path.join(plain, "pp"), os.path.join(forced, "pp"), fa, shallow=False)
    assert not mismatch and not err


def test_bench_two_ranks_two_helpers_one_gpu(gpu):
    """bench.py's multi-rank path without a multi-GPU node: `--gpus 2 --procs 2` under its own
    torch.distributed.run launch, both ranks and their four helper processes on GPU 0 (MODEST_DIST_BACKEND=gloo).  One JSON
    line from rank 0, both ranks counted by an all-reduce, whole-job throughput over the max of the ranks' clocks; the line
    carries every helper's start-up seconds and peak resident set (what an 8 x 8 run multiplies by 64: DESIGN section 6)."""
    import time
    env = dict(os.environ, MODEST_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MODEST_MIN_STEPS_PER_HELPER="4")   # (8 steps: two helpers per rank)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--procs", "2", "--steps", "8", "--warmup", "2", "--scans", "8",
           "--shard-scans", "8", "--n-live", "3000", "--frames", "4", "--traversals", "3", "--cli-scans", "0", "--cpu-scans", "0"]
    t0 = time.time()
    r = _Done(0, *_exec(cmd, env, 600))
    wall = time.time() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    cfg = d["config"]
    assert d["n_gpus"] == 2 and cfg["world_size"] == 2 and cfg["ranks_seen"] == 2 and cfg["process_group_backend"] == "gloo" and cfg["rccl_world_size"] is None
    assert d["scaling"] == "weak" and d["steps"] == 8 and d["value"] > 0
    assert abs(d["value"] - 2 * 8 / (d["ms_per_step"] * 8e-3)) < 1e-6 * d["value"]   # whole job: both ranks' scans over the slower rank's clock
    assert cfg["host_processes_per_gpu"] == 2 and len(cfg["startup"]["helper_seconds"]) == 2
    assert max(cfg["startup"]["helper_peak_rss_mb"]) < 16 * 1024
    assert wall < 300, wall
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_two_ranks_one_gpu.json"), "w") as f:
        json.dump(dict(wall_seconds=wall, line=d), f, indent=1)


def test_three_clis_eight_ranks_one_gpu(gpu, tmp_path):
    """BASELINE config 4's launch shape without the node: WORLD_SIZE = 8, every rank on GPU 0 (gloo: a barrier and
    a counter all-reduce are the only collectives of the path), a 64-scan tree.  The union of the eight ranks' files == the reference's
    eight manual `total_part=8 part=i` runs (pre_compute_pp_score.py:114-116, generate_mask.py:35-37, gen_label_files.py:36-38) == one
    process, byte for byte, for all three CLIs."""
    from modest_amd import synth
    n = 64
    root, meta = str(tmp_path / "data"), str(tmp_path / "meta")
    paths = synth.write_kitti_tree(root, meta, n_seq=3, n_frames=n + 6, n_pts=3000, origins=tuple(range(n)), hist_frames=6, max_range=60.0)
    train = os.path.join(root, "training")
    outs = {k: str(tmp_path / k) for k in ("one", "eight", "parts")}
    report = {}
    for m in CLIS:
        _run(m, _overrides(train, paths, outs["one"]), 1, 0)
    for i, m in enumerate(CLIS):
        err = _run(m, _overrides(train, paths, outs["eight"]), 8, 29561 + i)
        mm = re.search(r"ranks: scans (\d+)\.\.(\d+), busy ([\d.]+)\.\.([\d.]+) s \(imbalance ([\d.]+) %\)", err)
        assert mm, err[-1500:]
        assert int(mm.group(1)) == n // 8 and int(mm.group(2)) == n // 8
        report[m] = dict(rank_scans_min_max=[int(mm.group(1)), int(mm.group(2))], busy_seconds_min_max=[float(mm.group(3)), float(mm.group(4))])
    env = dict(os.environ, OMP_NUM_THREADS="4")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    for m in CLIS:   # the reference's manual split: eight independent processes per CLI (a stage needs the previous stage's files)
        procs = [subprocess.Popen([sys.executable, "-m", f"modest_amd.{m}"] + _overrides(train, paths, outs["parts"]) + ["total_part=8", f"part={part}"],
                                  env=env, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True) for part in range(8)]
        for pr in procs:
            _, e = pr.communicate(timeout=900)
            assert pr.returncode == 0, (m, e[-2000:])
    for k in ("eight", "parts"):
        for sub in ("pp", "seg", "bbox", "labels"):
            fa = sorted(f for f in os.listdir(os.path.join(outs["one"], sub)) if f != "configs.yaml")
            fb = sorted(f for f in os.listdir(os.path.join(outs[k], sub)) if f != "configs.yaml")
            assert fa == fb and len(fa) == n, (k, sub, len(fa), len(fb))
            match, mismatch, err = filecmp.cmpfiles(os.path.join(outs["one"], sub), os.path.join(outs[k], sub), fa, shallow=False)
            assert not mismatch and not err, (k, sub, mismatch[:5], err[:5])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "multirank8_report.json"), "w") as f:
        json.dump(dict(scans=n, world_size=8, note="eight ranks on GPU 0, gloo collectives", per_cli=report), f, indent=1)


def test_bench_eight_ranks_one_gpu(gpu):
    """bench.py's N = 8 launch path (`--gpus 8 --procs 1`: what the driver's SCALE run starts, one process per rank) on one GPU under
    gloo: ONE JSON line from rank 0, eight ranks counted by the all-reduce, whole-job throughput = 8 ranks' scans over the slowest rank's
    clock, and every rank's host cost (busy threads, resident memory) in the line -- DESIGN section 6 multiplies it out."""
    env = dict(os.environ, MODEST_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--procs", "1", "--steps", "8", "--warmup", "2", "--scans", "8",
           "--shard-scans", "8", "--n-live", "3000", "--frames", "4", "--traversals", "3", "--cli-scans", "0", "--cpu-scans", "0", "--sharing", "best"]
    r = _Done(0, *_exec(cmd, env, 900))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    cfg = d["config"]
    assert d["n_gpus"] == 8 and cfg["world_size"] == 8 and cfg["ranks_seen"] == 8 and cfg["process_group_backend"] == "gloo"
    assert abs(d["value"] - 8 * 8 / (d["ms_per_step"] * 8e-3)) < 1e-6 * d["value"]
    hb = cfg["host_budget_per_rank"]
    assert len(hb) == 8 and sorted(h["rank"] for h in hb) == list(range(8))
    assert all(h["rss_mb"] > 100 and h["busy_threads"] >= 0 for h in hb)
    with open(os.path.join(ROOT, "gpurun_out", "bench_eight_ranks_one_gpu.json"), "w") as f:
        json.dump(d, f, indent=1)


def test_pp_cli_on_a_tree_whose_traversal_count_changes(gpu, tmp_path):
    """The reference accepts a traversal PER SCAN (data_preprocessing/lyft/split_traintest.py:17,79; two needed, :111): the number of
    traversals changes along an idx list.  The PP CLI takes such scans in blocks as they come (pp_batch=32: modest_pp_score_block_mixed,
    no flush where T changes): its score files == the per-scan path's (pp_batch=1) byte for byte, and == the oracle's entropy
    (pre_compute_pp_score.py:68-75: ln T of the scan's own T) on scans with different T."""
    import pickle
    import numpy as np
    from modest_amd import synth
    from oracle import pp_score as opp
    n = 48
    pres = synth.presence_ramp(n, 7, seed=4)
    root, meta = str(tmp_path / "data"), str(tmp_path / "meta")
    paths = synth.write_kitti_tree(root, meta, n_seq=8, n_frames=n + 16, n_pts=3000, origins=tuple(range(n)), hist_frames=16,
                                   max_range=60.0, presence=pres)
    valid = pickle.load(open(paths["idx_info"], "rb"))
    Ts = [len(valid[k][2]) for k in sorted(valid)]
    assert len(set(Ts)) >= 3, Ts
    train = os.path.join(root, "training")
    blk, one = str(tmp_path / "blk"), str(tmp_path / "one")
    err = _run(CLIS[0], _overrides(train, paths, blk) + ["pp_batch=32"], 1, 0, extra_env={"MODEST_PP_TRACE_PATHS": "1"})
    _run(CLIS[0], _overrides(train, paths, one) + ["pp_batch=1"], 1, 0)
    fa = sorted(os.listdir(os.path.join(blk, "pp")))
    assert fa == sorted(os.listdir(os.path.join(one, "pp"))) and len(fa) == n
    match, mismatch, e2 = filecmp.cmpfiles(os.path.join(blk, "pp"), os.path.join(one, "pp"), fa, shallow=False)
    assert not mismatch and not e2, mismatch[:5]
    mm = re.search(r"pp paths: block calls (\d+), chain calls (\d+)", err)
    # the block path took every batch (4 + 32 + 12 scans; a batch the sharing rule halves counts twice): no flush where T changes --
    # flushing there left batches of fewer than four scans to the per-scan chain
    assert mm and 3 <= int(mm.group(1)) <= 5 and int(mm.group(2)) == 0, err[-1500:]


def test_fused_cli_writes_the_three_clis_files(gpu, tmp_path):
    """`python -m modest_amd.seed_labels` (PP -> mask -> labels per batch in one process, the PP score staying on the device, stages 2 + 3
    of a batch under the next batch's PP kernels) writes every file the reference's three CLIs write (README.md:52-70), byte for byte: one
    process, `workers=2`, and a second run over a half-finished tree (scans whose files all exist are skipped, the others completed)."""
    from modest_amd import synth
    n = 44
    root, meta = str(tmp_path / "data"), str(tmp_path / "meta")
    paths = synth.write_kitti_tree(root, meta, n_seq=4, n_frames=n + 12, n_pts=4000, origins=tuple(range(n)), hist_frames=12, max_range=60.0)
    train = os.path.join(root, "training")
    ref, fused, fw = str(tmp_path / "ref"), str(tmp_path / "fused"), str(tmp_path / "fusedw")
    for m in CLIS:
        _run(m, _overrides(train, paths, ref), 1, 0)
    _run("seed_labels", _overrides(train, paths, fused) + ["mask.mask_batch=8"], 1, 0)
    _run("seed_labels", _overrides(train, paths, fw) + ["workers=2"], 1, 0)

    def same(a, b):
        for sub in ("pp", "seg", "bbox", "labels"):
            fa = sorted(os.listdir(os.path.join(a, sub)))
            fb = sorted(os.listdir(os.path.join(b, sub)))
            assert fa == fb and len([f for f in fa if f != "configs.yaml"]) == n, (sub, len(fa), len(fb))
            match, mismatch, err = filecmp.cmpfiles(os.path.join(a, sub), os.path.join(b, sub), [f for f in fa if f != "configs.yaml"], shallow=False)
            assert not mismatch and not err, (sub, mismatch[:5], err[:5])
        import yaml
        for sub in ("seg", "bbox"):   # the config dump the mask stage leaves behind (generate_mask.py:38-46): the mask stage's own config
            ca, cb = yaml.safe_load(open(os.path.join(a, sub, "configs.yaml"))), yaml.safe_load(open(os.path.join(b, sub, "configs.yaml")))
            assert set(ca) == set(cb) and ca["graph"] == cb["graph"] and ca["filtering"] == cb["filtering"], sub

    same(ref, fused)
    same(ref, fw)
    # resume: drop a third of the label files and one score file; the second run writes exactly those scans again
    gone = sorted(f for f in os.listdir(os.path.join(fused, "labels")))[::3]
    for f in gone:
        os.remove(os.path.join(fused, "labels", f))
    os.remove(os.path.join(fused, "pp", "000005.npy"))
    stamp = {f: os.path.getmtime(os.path.join(fused, "seg", f)) for f in os.listdir(os.path.join(fused, "seg"))}
    _run("seed_labels", _overrides(train, paths, fused), 1, 0)
    same(ref, fused)
    redone = {f for f, t in stamp.items() if os.path.getmtime(os.path.join(fused, "seg", f)) != t}
    assert redone == {f.replace(".txt", ".npy") for f in gone} | {"000005.npy"}, sorted(redone)[:8]
