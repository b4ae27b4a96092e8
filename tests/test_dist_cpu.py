"""CPU, world_size 2, gloo: the N>1 path of the CLIs (scan sharding, barrier,
counter all-reduce) without a GPU."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_semantics_single_process():
    from modest_amd import dist
    idx = np.arange(50)
    assert np.array_equal(dist.shard(idx, 1, 0, rank=0, ws=1), idx)
    parts = [dist.shard(idx, 4, p, rank=0, ws=1) for p in range(4)]
    assert np.array_equal(np.concatenate(parts), idx)                 # the reference's total_part/part split
    assert all(np.array_equal(a, b) for a, b in zip(parts, np.array_split(idx, 4)))
    both = [dist.shard(idx, 2, p, rank=r, ws=3) for p in range(2) for r in range(3)]
    assert np.array_equal(np.sort(np.concatenate(both)), idx)         # nested split stays a partition


def test_two_ranks_gloo(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    outs = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(2)]
    idx = np.arange(100, 137)
    assert [o["ws"] for o in outs] == [2, 2]
    for k in range(2):
        assert outs[k]["shard"] == [int(x) for x in np.array_split(idx, 2)[k]]
        assert outs[k]["shard2"] == [int(x) for x in np.array_split(np.array_split(idx, 3)[1], 2)[k]]
        assert outs[k]["tot"]["scans"] == 37 and outs[k]["tot"]["hist_points"] == 3000
        assert outs[k]["tot"]["max_seconds"] == 1.5


def test_worker_split_env(monkeypatch):
    """workers=N: a worker process takes a contiguous piece of its parent rank's shard; the pieces of
    all workers of all ranks partition the list in order."""
    from modest_amd import dist
    idx = np.arange(200, 263)
    pieces = []
    for r in range(2):
        for w in range(3):
            monkeypatch.setenv("MODEST_PARENT_RANK", str(r))
            monkeypatch.setenv("MODEST_PARENT_WS", "2")
            monkeypatch.setenv("MODEST_WORKER", f"{w}/3")
            pieces.append(dist.shard(idx, 1, 0))
            # what the CLIs do inside a worker: they pass the (0, 1) their own dist.init() returned
            assert np.array_equal(dist.shard(idx, 1, 0, rank=0, ws=1), pieces[-1])
    assert np.array_equal(np.concatenate(pieces), idx)
    monkeypatch.delenv("MODEST_WORKER")
    monkeypatch.delenv("MODEST_PARENT_RANK")
    monkeypatch.delenv("MODEST_PARENT_WS")
    assert np.array_equal(dist.shard(idx, 1, 0), idx)


def test_dynamic_work_queue_two_ranks_gloo(tmp_path):
    """SURVEY 8e's chunked dynamic queue: the ranks' chunks partition the list (whole chunks of consecutive
    scans), the slower rank ends up with fewer scans, reduce_counters carries min_/max_ spreads."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29537",
           os.path.join(ROOT, "tests", "_dist_queue_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    outs = [json.load(open(tmp_path / f"q{k}.json")) for k in range(2)]
    idx = np.arange(1000, 1093)
    assert sorted(outs[0]["got"] + outs[1]["got"]) == list(idx)
    for o in outs:                                            # whole chunks of 5 consecutive scans
        g = np.array(o["got"])
        assert np.all((g[::5] - 1000) % 5 == 0) and np.all(np.diff(g)[np.arange(len(g) - 1) % 5 != 4] == 1)
    assert len(outs[0]["got"]) < len(outs[1]["got"])          # rank 0 slept 4x longer per scan
    assert sorted(outs[0]["got2"] + outs[1]["got2"]) == list(np.array_split(idx, 3)[1])
    for k in range(2):
        assert outs[k]["got3"] == [int(x) for x in np.array_split(idx, 2)[k]]
        t = outs[k]["tot"]
        assert t["scans"] == 93 and t["min_rank_scans"] == len(outs[0]["got"]) and t["max_rank_scans"] == len(outs[1]["got"])
        assert 0.0 <= t["imbalance"] < 0.5 and t["min_busy_seconds"] <= t["max_busy_seconds"]
