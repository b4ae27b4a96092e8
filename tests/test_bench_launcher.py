"""CPU: bench.py's launcher logic (the driver calls `python bench.py --gpus N` with no launcher)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_helper_count_keeps_a_minimum_of_steps_per_helper():
    import bench
    assert bench.MIN_STEPS_PER_HELPER >= 2
    n20 = bench.helper_count(8, 20)                # the driver's 20-step run: the whole pool, 2-3 steps each
    assert n20 == min(8, 20 // bench.MIN_STEPS_PER_HELPER)
    assert bench.helper_count(8, 560) == 8
    assert bench.helper_count(8, 3) == max(1, 3 // bench.MIN_STEPS_PER_HELPER)
    assert bench.helper_count(1, 560) == 1
    assert sum(bench._split(20, n20)) == 20 and min(bench._split(20, n20)) >= bench.MIN_STEPS_PER_HELPER


def test_launch_command_is_one_rank_per_gpu_on_localhost():
    import bench
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "20", "--warmup", "5"])
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]
    assert os.path.basename(cmd[-7]) == "bench.py"


def test_world_size_mismatch_fails_loudly():
    """Under a launcher that started the wrong number of ranks the bench refuses to run (round 1: it
    silently measured one GPU)."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0
    assert "--gpus 2" in r.stderr and "WORLD_SIZE=1" in r.stderr


def test_more_gpus_than_devices_fails_loudly():
    """No launcher and fewer devices than --gpus: an error, not a one-GPU run."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], env=env, capture_output=True,
                       text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and "HIP device(s) visible" in r.stderr
