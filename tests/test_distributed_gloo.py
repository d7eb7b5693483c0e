"""world_size-2 (and 3) gloo tests of the frame-sharded N>1 path on CPU: shard ranges follow the
chunk_for_devices rule, the single all-gather reassembles frames in order (ragged shards too), and
the result equals the single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from elvis_amd.distributed import all_gather_frames, restore_clip_sharded, shard_sizes
        from elvis_amd.sharding import rank_frame_range
        rng = np.random.default_rng(5)
        frames = [rng.integers(0, 256, size=(6, 10, 3), dtype=np.uint8) for _ in range(total)]
        maps = rng.integers(0, 3, size=(total, 3, 5))
        seen = []

        def restore_shard(fr, mp_, first):
            seen.append((first, len(fr)))
            # depends on the GLOBAL frame index, like the sampler noise
            return [np.clip(f.astype(np.int32) + (first + i) % 7, 0, 255).astype(np.uint8) for i, f in enumerate(fr)]

        out = restore_clip_sharded(frames, maps, restore_shard)
        s, e = rank_frame_range(total, world, rank)
        ok = len(out) == total and seen == ([(s, e - s)] if e > s else [])
        ref = [np.clip(f.astype(np.int32) + i % 7, 0, 255).astype(np.uint8) for i, f in enumerate(frames)]
        ok = ok and all(np.array_equal(a, b) for a, b in zip(out, ref))
        # raw gather with the wrong shard size must raise
        bad = False
        try:
            all_gather_frames(torch.zeros((shard_sizes(total, world)[rank] + 1, 2, 2, 3), dtype=torch.uint8), total)
        except ValueError:
            bad = True
        q.put((rank, bool(ok and bad)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 7), (2, 8), (3, 7), (2, 1)])
def test_sharded_restore_gloo(world, total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True) for r in range(world)]
