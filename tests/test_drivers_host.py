"""Directory drivers, frame IO and side-channel formats - host logic,
no GPU: the restorer is replaced by picklable stand-ins from tests/_hooks.py."""
import os

import numpy as np
import pytest
import torch

import _hooks
from elvis_amd import drivers, frameio, tiler


def _write_clip(d, n, h=16, w=24, seed=0):
    rng = np.random.default_rng(seed)
    frames = [rng.integers(0, 250, size=(h, w, 3), dtype=np.uint8) for _ in range(n)]
    for i, f in enumerate(frames):
        frameio.save_frame(f, os.path.join(d, f"{i + 1:05d}.png"))
    return frames


def test_frame_png_round_trip_is_bgr(tmp_path):
    f = np.zeros((4, 6, 3), np.uint8)
    f[..., 0], f[..., 2] = 10, 200                     # B = 10, R = 200
    p = tmp_path / "a" / "x.png"
    frameio.save_frame(f, p)                           # creates the directory (elvis.py:133)
    from PIL import Image
    assert Image.open(p).getpixel((0, 0)) == (200, 0, 10)   # stored RGB
    assert np.array_equal(frameio.load_frame(p), f)
    with pytest.raises(IOError):
        frameio.load_frame(tmp_path / "missing.png")
    with pytest.raises(IOError):
        frameio.save_frame(np.zeros((4, 4), np.uint8), tmp_path / "gray.png")
    (tmp_path / "a" / "note.txt").write_text("x")
    assert [q.name for q in frameio.get_frame_paths(tmp_path / "a")] == ["x.png"]
    assert frameio.get_frame_paths(tmp_path / "nope") == []
    frameio.clear_directory(tmp_path / "a")
    assert os.listdir(tmp_path / "a") == ["note.txt"]


def test_strength_map_codec_matches_reference_fixture(golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, "strength_maps.npz"))
    # the reference's own encoder output (wire bytes) decodes to its own decoder's result
    wire = tmp_path / "ref.npz"
    wire.write_bytes(g["wire_bytes"].tobytes())
    assert np.array_equal(frameio.load_strength_maps(wire), g["maps_decoded"])
    # and this encoder's file gives the same array back, uint8 on the wire (elvis.py:2253-2254)
    mine = tmp_path / "mine.npz"
    frameio.encode_strength_maps_to_npz(list(g["maps_in"]), mine)
    dec = frameio.decode_strength_maps_from_npz(mine)
    assert dec.dtype == np.uint8 and np.array_equal(dec, g["maps_decoded"])
    with pytest.raises(FileNotFoundError):
        frameio.decode_strength_maps_from_npz(tmp_path / "absent.npz")


def test_block_mask_packbits_round_trip(tmp_path):
    rng = np.random.default_rng(3)
    for shape in ((5, 3, 7), (1, 1, 1), (2, 9, 4)):            # bit counts that are not multiples of 8
        m = (rng.random(shape) < 0.4).astype(np.uint8)
        p = tmp_path / f"m{len(shape)}_{shape[1]}.npz"
        frameio.save_block_masks(m, p)
        f = np.load(p)
        assert np.array_equal(f["packed"], np.packbits(m)) and tuple(f["shape"]) == shape   # elvis.py:4412-4418
        assert np.array_equal(frameio.load_block_masks(p), m)


def test_extract_tile_with_halo(golden_dir):
    g = np.load(os.path.join(golden_dir, "halo.npz"))
    for k, q in enumerate(g["queries"]):
        y, x, th, tw, halo = [int(v) for v in q[:5]]
        tile, bounds = tiler.extract_tile_with_halo(g["frame"], y, x, th, tw, halo)
        assert np.array_equal(tile, g[f"tile{k}"]) and tuple(bounds) == tuple(int(v) for v in q[5:9])
        tile[...] = 0                                   # a copy, not a view (utils.py:1240)
    assert g["frame"].any()


def test_downsample_driver_files_and_errors(tmp_path):
    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    frames = _write_clip(src, 5)
    maps = np.zeros((5, 2, 3), np.uint8)
    maps[[0, 3]] = 1
    (dst / "stale").mkdir(parents=True)
    frameio.save_frame(frames[0], dst / "99999.png")          # cleared before writing (elvis.py:2716)
    drivers.restore_downsampled_with_sinsr(str(src), str(dst), maps, 8, devices=["cpu"], tile=0, model_name="ignored",
                                           _shard_fn=_hooks.plus_device_tag)
    assert sorted(p.name for p in frameio.get_frame_paths(dst)) == [f"{i + 1:05d}.png" for i in range(5)]
    for i, f in enumerate(frames):
        want = _hooks.plus_device_tag([f], maps[i:i + 1], 8, None, i)[0]
        assert np.array_equal(frameio.load_frame(dst / f"{i + 1:05d}.png"), want)
        assert np.array_equal(frameio.load_frame(src / f"{i + 1:05d}.png"), f)      # inputs untouched
    with pytest.raises(ValueError, match="does not match frame count"):
        drivers.restore_downsampled_with_sinsr(str(src), str(dst), maps[:4], 8, devices=["cpu"], _shard_fn=_hooks.plus_device_tag)
    with pytest.raises(ValueError, match="No frames found"):
        drivers.restore_downsampled_with_sinsr(str(tmp_path / "empty"), str(dst), maps, 8, devices=["cpu"],
                                               _shard_fn=_hooks.plus_device_tag)
    if not torch.cuda.is_available():                  # the product path has no CPU fallback
        with pytest.raises((RuntimeError, ValueError)):
            drivers.restore_downsampled_with_sinsr(str(src), str(dst), maps, 8)


def test_drivers_two_worker_processes(tmp_path):
    """Two devices -> two spawned workers (elvis.py:3124-3158); chunks by the chunk_for_devices rule; the frame
    index each worker sees is global; a failing worker surfaces as RuntimeError with its device, range and traceback."""
    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    frames = _write_clip(src, 5, seed=1)
    maps = np.ones((5, 2, 3), np.uint8)
    two = [torch.device("cpu"), torch.device("meta")]   # two distinct device entries on a GPU-less host
    drivers.restore_downsampled_with_sinsr(str(src), str(dst), maps, 8, devices=two, _shard_fn=_hooks.plus_device_tag)
    for i, f in enumerate(frames):
        assert np.array_equal(frameio.load_frame(dst / f"{i + 1:05d}.png"), _hooks.plus_device_tag([f], maps[i:i + 1], 8, None, i)[0])
    # a failing worker names its device, its frame range and the exception
    with pytest.raises(RuntimeError, match=r"(?s)shard on (cpu|meta), frames \[\d, \d\).*RuntimeError: boom") as ei:
        drivers.restore_downsampled_with_sinsr(str(src), str(dst), maps, 8, devices=two, _shard_fn=_hooks.failing)
    assert "frames [0, 3)" in str(ei.value) and "frames [3, 5)" in str(ei.value)


def test_in_place_drivers(tmp_path):
    d = tmp_path / "frames"
    d.mkdir()
    frames = _write_clip(d, 4, seed=2)
    maps = np.zeros((4, 2, 3), np.int32)
    drivers.restore_blur_adaptive(str(d), maps, 8, devices=["cpu"], _shard_fn=_hooks.failing)   # all-zero map: untouched, model never runs
    assert all(np.array_equal(frameio.load_frame(d / f"{i + 1:05d}.png"), f) for i, f in enumerate(frames))
    maps[1] = 2
    drivers.restore_blur_adaptive(str(d), maps, 8, cfg=3.0, seed=7, devices=["cpu"], batch_size=2, _shard_fn=_hooks.plus_device_tag)
    for i, f in enumerate(frames):
        assert np.array_equal(frameio.load_frame(d / f"{i + 1:05d}.png"), _hooks.plus_device_tag([f], maps[i:i + 1], 8, None, i)[0])
    with pytest.raises(ValueError, match="batch_size"):
        drivers.restore_blur_adaptive(str(d), maps, 8, batch_size=0, _shard_fn=_hooks.plus_device_tag)
    with pytest.raises(ValueError):
        drivers.restore_blur_adaptive(str(d), maps[:2], 8, devices=["cpu"], _shard_fn=_hooks.plus_device_tag)


def test_dct_driver_halo_is_independent_of_worker_count(tmp_path):
    """The temporal restorer reads decoded halo frames: one worker and two workers must give the same files."""
    outs = []
    for devs in (["cpu"], [torch.device("cpu"), torch.device("meta")]):
        d = tmp_path / f"f{len(devs)}"
        d.mkdir()
        _write_clip(d, 6, seed=5)
        drivers.restore_dct_adaptive(str(d), np.ones((6, 2, 3), np.uint8), 8, devices=devs, temporal_radius=1,
                                     _shard_fn=_hooks.temporal_window_max)
        assert sorted(os.listdir(d)) == [f"{i + 1:05d}.png" for i in range(6)]      # scratch directory removed
        outs.append(frameio.load_frames(d))
    assert all(np.array_equal(a, b) for a, b in zip(*outs))
