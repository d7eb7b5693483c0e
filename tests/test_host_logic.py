"""The product's host-side mirror of the reference interface (elvis_amd.sharding / recompose /
tiler gate) against the golden vectors of the reference's own code.  No GPU needed."""
import os

import numpy as np
import pytest
import torch

import elvis_amd as E


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_split_combine(golden_dir):
    g = _load(golden_dir, "blocks.npz")
    for i in range(5):
        img, b = g[f"img{i}"], int(g[f"b{i}"])
        blk = E.split_image_into_blocks(img, b)
        assert np.array_equal(blk, g[f"blocks{i}"])
        assert np.array_equal(E.combine_blocks_into_image(blk), img)
    with pytest.raises(ValueError, match="divisible by block_size"):
        E.split_image_into_blocks(np.zeros((10, 16, 3), np.uint8), 8)


def test_chunk_for_devices(golden_dir):
    g = _load(golden_dir, "chunks.npz")
    table = g["chunk_table"]
    for total, nd, mcs in sorted({(int(r[0]), int(r[1]), int(r[2])) for r in table}):
        specs = E.chunk_for_devices(total, [torch.device("cpu")] * nd, mcs)
        rows = [(total, nd, mcs, s.start, s.end, s.chunk_id) for s in specs] or [(total, nd, mcs, -1, -1, -1)]
        want = [tuple(int(v) for v in r) for r in table if (int(r[0]), int(r[1]), int(r[2])) == (total, nd, mcs)]
        assert rows == want
    # 30 frames / 8 devices -> 4,4,4,4,4,4,3,3 ; 240 / 8 -> 30 each
    assert [s.end - s.start for s in E.chunk_for_devices(30, [torch.device("cpu")] * 8)] == [4, 4, 4, 4, 4, 4, 3, 3]
    assert [s.end - s.start for s in E.chunk_for_devices(240, [torch.device("cpu")] * 8)] == [30] * 8
    assert E.chunk_for_devices(0, [torch.device("cpu")]) == [] and E.chunk_for_devices(5, []) == []


def test_rank_frame_range_matches_chunk_rule():
    for total in (0, 1, 7, 30, 240, 241):
        for ws in (1, 2, 3, 4, 8):
            specs = {s.chunk_id: (s.start, s.end) for s in E.chunk_for_devices(total, [torch.device("cpu")] * ws)}
            for r in range(ws):
                s, e = E.rank_frame_range(total, ws, r)
                assert (s, e) == specs.get(r, (s, s))
            assert sum(E.rank_frame_range(total, ws, r)[1] - E.rank_frame_range(total, ws, r)[0] for r in range(ws)) == total


def test_parallel_process_frames(golden_dir):
    g = _load(golden_dir, "chunks.npz")
    frames = [f for f in g["ppf_in"]]
    seen = []

    def proc(fr, dev):
        seen.append((len(fr), str(dev)))
        return [f + 100 for f in fr]

    cpu = torch.device("cpu")
    assert np.array_equal(np.stack(E.parallel_process_frames(proc, frames, [cpu] * 3)), g["ppf_auto"])
    assert np.array_equal(np.stack(E.parallel_process_frames(proc, frames, [cpu] * 2, chunk_size=4)), g["ppf_fixed4"])
    assert np.array_equal(np.stack(E.parallel_process_frames(proc, frames, [cpu])), g["ppf_single"])
    assert E.parallel_process_frames(proc, [], [cpu]) == []
    # order is preserved even when later chunks finish first
    import time

    def slow_first(fr, dev):
        if fr[0][0, 0, 0] == 0:
            time.sleep(0.05)
        return fr

    out = E.parallel_process_frames(slow_first, frames, [cpu] * 4)
    assert [int(f[0, 0, 0]) for f in out] == list(range(11))


def test_resolve_device_list_cpu_host(golden_dir):
    if torch.cuda.is_available():
        pytest.skip("CPU-host fixture")
    g = _load(golden_dir, "devices_cpu.npz")
    assert [str(d) for d in E.resolve_device_list(None)] == list(g["default"])
    assert [str(d) for d in E.resolve_device_list(["cpu", "cpu", torch.device("cpu")])] == list(g["cpu_dup"])
    for spec, raises in zip(([0], ["cuda"], ["cuda:1"]), g["cuda_specs_raise"]):
        if raises:
            with pytest.raises(ValueError):
                E.resolve_device_list(spec)
    with pytest.raises(ValueError):
        E.resolve_device_list(None, allow_cpu_fallback=False)


def test_adaptive_gate(golden_dir):
    g = _load(golden_dir, "gate.npz")
    maps = g["maps"]
    marker = lambda frames, **kw: [f + 1 for f in frames]
    fr = [np.zeros((4, 4, 3), np.uint8)]
    for row in g["decisions"]:
        tc, thr, want = tuple(int(v) for v in row[:6]), row[6] / 10.0, int(row[7])
        out = E.adaptive_restore(marker, fr, degradation_maps=maps, block_size=16, tile_coords=tc, threshold=thr)
        assert int(out[0][0, 0, 0]) == want
    assert int(E.adaptive_restore(marker, fr, degradation_maps=None)[0][0, 0, 0]) == 1


def test_tiler_passthrough_without_gpu():
    """No tiling and no chunking -> the wrapper just calls restore_fn (utils.py:206-207), so it
    works without a GPU; anything that needs the accumulate kernels must fail loudly instead of
    silently falling back to a CPU path."""
    frames = [np.zeros((8, 8, 3), np.uint8)] * 2
    out = E.resource_aware_restore(lambda frames, device=None, **kw: [f + 5 for f in frames], frames,
                                   tile_size=64, halo=4, chunk_size=8, device="cpu")
    assert int(out[0][0, 0, 0]) == 5
    assert E.resource_aware_restore(lambda **kw: 1 / 0, []) == []
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            E.resource_aware_restore(lambda frames, device=None, **kw: frames, frames, tile_size=4, halo=2,
                                     chunk_size=0, device="cuda")
        with pytest.raises(RuntimeError):
            E.restore_frames_sinsr(frames, np.zeros((2, 1, 1), np.int32), 8, "cuda:0")
        with pytest.raises(RuntimeError):
            E.get_sinsr_model("cpu")
