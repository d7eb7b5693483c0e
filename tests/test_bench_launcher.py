"""`bench.py --gpus N` must start N ranks itself when it is not already under torchrun (the driver calls it
both ways).  CPU check with gloo: the launcher's child ranks rendezvous, all-gather ragged shards in order and
rank 0 reports the world size RCCL/gloo saw."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--launcher-selftest", *extra], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout          # exactly ONE json line, from rank 0
    return json.loads(lines[0])


def test_gpus_flag_starts_that_many_ranks():
    one = _run("--gpus", "1")
    assert one["n_gpus"] == 1 and one["order_ok"]
    two = _run("--gpus", "2")
    assert two["n_gpus"] == 2 and two["requested_gpus"] == 2 and two["order_ok"] and two["frames"] == 7


def test_under_torchrun_env_it_does_not_relaunch():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29555")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--launcher-selftest", "--gpus", "8"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    assert json.loads(p.stdout.strip().splitlines()[-1])["n_gpus"] == 1     # the environment, not --gpus, is the truth


def test_force_dist_runs_the_collective_branch_at_world_one():
    """`--force-dist`: one rank under a child torchrun, process group initialised, the all-gather executed at world 1
    (what the GPU box runs with RCCL: tests/test_gpu_dist_world1.py)."""
    one = _run("--gpus", "1", "--force-dist")
    assert one["n_gpus"] == 1 and one["order_ok"] and one["backend"] == "gloo"
