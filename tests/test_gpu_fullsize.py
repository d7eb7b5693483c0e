"""Full-size (1080p, the BASELINE.json configs) checks through size-independent properties: the oracle
would take minutes per frame at this size, so parity proper lives in the small-size tests and these
assert what must hold at ANY size - untouched level-0 blocks, invariance to batching / clip splitting
(sampler noise keyed on the global frame index), run-to-run bit reproducibility, and agreement of the
restored region with the small-tile computation where the network is local (recompose only)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

H, W, B = 1080, 1920, 8


def _clip(n, seed, max_level):
    from elvis_amd import synth
    clean, degraded, levels = synth.make_downsample_case(n, H, W, B, max_level=max_level)
    rng = np.random.default_rng(seed)
    levels = levels.copy()
    levels[:, : levels.shape[1] // 3] = 0            # a band of untouched blocks in every frame
    levels[n - 1] = 0                                  # and one frame with nothing to restore
    return [np.ascontiguousarray(f) for f in degraded], levels.astype(np.int32), rng


def _keep_mask(level_map):
    return np.repeat(np.repeat(level_map == 0, B, 0), B, 1)


def test_sinsr_1080p_properties(gpu_device):
    from elvis_amd import restore
    frames, maps, _ = _clip(3, 1, 3)
    out = restore.restore_frames_sinsr(frames, maps, B, gpu_device, schedule="single4x")
    assert len(out) == 3 and all(o.shape == (H, W, 3) and o.dtype == np.uint8 for o in out)
    # level-0 blocks are the decoded input, bit for bit (elvis.py:2584-2595 paste rule)
    assert np.array_equal(out[2], frames[2])
    for i in range(2):
        k = _keep_mask(maps[i])
        assert np.array_equal(out[i][k], frames[i][k])
        assert not np.array_equal(out[i][~k], frames[i][~k])
    # bit-reproducible, and independent of how the clip is split across calls / ranks
    again = restore.restore_frames_sinsr(frames, maps, B, gpu_device, schedule="single4x")
    a = restore.restore_frames_sinsr(frames[:1], maps[:1], B, gpu_device, first_frame_index=0, schedule="single4x")
    b = restore.restore_frames_sinsr(frames[1:], maps[1:], B, gpu_device, first_frame_index=1, schedule="single4x")
    for x, y, z in zip(out, again, a + b):
        assert np.array_equal(x, y) and np.array_equal(x, z)
    # ... and of the frames-per-invocation batching inside a call
    model = restore.get_sinsr_model(torch.device(gpu_device))
    fd = restore.frames_to_device(frames, model.device)
    md = restore.maps_to_device(maps, 3, model.device)
    o1 = restore.restore_clip_single4x_device(model, fd, md, B, [0, 1, 2], batch=1)
    o3 = restore.restore_clip_single4x_device(model, fd, md, B, [0, 1, 2], batch=3)
    assert torch.equal(o1, o3) and np.array_equal(o3[0].cpu().numpy(), out[0])


@pytest.mark.parametrize("mode", ["x3", "dec_f16"])
def test_sinsr_1080p_in_tolerance_modes(gpu_device, mode):
    """The precision modes that meet the tolerance (bench.py `in_tolerance`), at full size: the paste rule, run-to-run bit
    reproducibility and invariance to the frames-per-invocation batching - the planar compensated kernels fix their
    GroupNorm-statistics tile shape by the per-image shape only, like the f16 ones."""
    from elvis_amd import restore
    frames, maps, _ = _clip(3, 3, 3)
    model = restore.get_sinsr_model(torch.device(gpu_device), precision=mode)
    fd = restore.frames_to_device(frames, model.device)
    md = restore.maps_to_device(maps, 3, model.device)
    o1 = restore.restore_clip_single4x_device(model, fd, md, B, [5, 6, 7], batch=1)
    o2 = restore.restore_clip_single4x_device(model, fd, md, B, [5, 6, 7], batch=2)
    o2b = restore.restore_clip_single4x_device(model, fd, md, B, [5, 6, 7], batch=2)
    assert torch.equal(o1, o2) and torch.equal(o2, o2b)
    out = o1.cpu().numpy()
    assert np.array_equal(out[2], frames[2])
    for i in range(2):
        k = _keep_mask(maps[i])
        assert np.array_equal(out[i][k], frames[i][k]) and not np.array_equal(out[i][~k], frames[i][~k])
    # the modes agree with the default f16 mode except where f16 flips a VQ code (a few per cent of the restored pixels)
    f16 = restore.restore_clip_single4x_device(restore.get_sinsr_model(torch.device(gpu_device)), fd, md, B, [5, 6, 7], batch=2)
    diff = (o1.to(torch.int16) - f16.to(torch.int16)).abs()
    assert (diff > 1).float().mean().item() < 0.05


@pytest.mark.parametrize("slot", ["blur", "dct"])
def test_round_slots_1080p_properties(gpu_device, slot):
    import elvis_amd as E
    frames, maps, _ = _clip(3, 2, 2)
    fn = E.restore_frames_blur if slot == "blur" else E.restore_frames_dct
    out = fn(frames, maps, B, gpu_device)
    assert np.array_equal(out[2], frames[2])
    for i in range(2):
        k = _keep_mask(maps[i])
        assert np.array_equal(out[i][k], frames[i][k])
        assert not np.array_equal(out[i][~k], frames[i][~k])
    again = fn(frames, maps, B, gpu_device)
    for x, y in zip(out, again):
        assert np.array_equal(x, y)
    if slot == "blur":   # per-frame model: batching must not matter
        other = E.restore_frames_blur(frames, maps, B, gpu_device, batch_size=1)
        for x, y in zip(out, other):
            assert np.array_equal(x, y)
