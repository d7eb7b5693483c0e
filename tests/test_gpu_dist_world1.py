"""The RCCL branch of bench.py on ONE GPU: `--force-dist` starts one rank under a child torch.distributed.run (a fresh
process: nothing that has touched the GPU is re-executed), which sets HSA_ENABLE_IPC_MODE_LEGACY=0 before its first GPU
call, runs init_process_group("nccl", device_id=...), the all_gather_into_tensor of the restored clip at world size 1
and rank 0's download of the gathered sequence - and checks that sequence against the local shard bit for bit."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_force_dist_nccl_world_one(gpu_device):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--gpus", "1", "--steps", "1", "--warmup", "1",
           "--frames", "3", "--height", "256", "--width", "384", "--batch", "3", "--no-extras", "--no-cpu-baseline",
           "--no-kernel-timing"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    d = line["dist"]
    assert d["world"] == 1 and d["forced_at_world_1"] and d["backend"].startswith("RCCL")
    assert d["gathered_sequence_holds_rank0_shard_bit_for_bit"] is True
    assert line["n_gpus"] == 1 and line["hbm_resident"]["same_output_as_host_path"] is True
    assert line["value"] > 0
