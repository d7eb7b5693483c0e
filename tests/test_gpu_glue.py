"""GPU parity tests (bit-exact) of the uint8 block-map glue kernels and the on-device recompose
drivers against the numpy oracle and the committed golden fixtures."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("shape,block", [((3, 64, 96, 3), 8), ((1, 24, 40, 3), 4), ((2, 1080, 1920, 3), 8),
                                         ((2, 30, 50, 3), 8), ((1, 17, 13, 1), 4)])
def test_recompose_matches_oracle(gpu_device, shape, block):
    from elvis_amd import ops
    from oracle import glue_ref
    rng = np.random.default_rng(1)
    n, h, w, c = shape
    a = rng.integers(0, 256, size=shape, dtype=np.uint8)
    b = rng.integers(0, 256, size=shape, dtype=np.uint8)
    by, bx = h // block, w // block
    m = rng.integers(0, 4, size=(n, by, bx)).astype(np.int32)
    new_map = torch.empty((n, by, bx), dtype=torch.int32, device=gpu_device)
    out = ops.recompose_u8(_dev(a, gpu_device), _dev(b, gpu_device), _dev(m, gpu_device), block, 1,
                           map_out=new_map, clamp_to=1).cpu().numpy()
    for i in range(n):
        ref = glue_ref.recompose_select(a[i], b[i], m[i] <= 1, block)
        assert np.array_equal(out[i], ref)
    assert np.array_equal(new_map.cpu().numpy(), np.where(m <= 1, m, 1))


def test_recompose_idempotent_when_map_zero(gpu_device):
    from elvis_amd import ops
    rng = np.random.default_rng(2)
    a = rng.integers(0, 256, size=(1, 64, 64, 3), dtype=np.uint8)
    b = rng.integers(0, 256, size=(1, 64, 64, 3), dtype=np.uint8)
    m = torch.zeros((1, 8, 8), dtype=torch.int32, device=gpu_device)
    out = ops.recompose_u8(_dev(a, gpu_device), _dev(b, gpu_device), m, 8, 0)
    assert np.array_equal(out.cpu().numpy(), a)


@pytest.mark.parametrize("factor", [2, 4, 8, 3])
@pytest.mark.parametrize("rounding", [0, 1])
def test_area_downscale(gpu_device, factor, rounding):
    from elvis_amd import ops
    from oracle import glue_ref
    rng = np.random.default_rng(3)
    h, w = 24 * factor, 20 * factor
    x = rng.integers(0, 256, size=(2, h, w, 3), dtype=np.uint8)
    out = ops.area_downscale_u8(_dev(x, gpu_device), factor, rounding).cpu().numpy()
    for i in range(2):
        ref = glue_ref.area_downscale_u8(x[i], factor, "cv2" if rounding == 0 else "half_up")
        assert np.array_equal(out[i], ref)


def test_area_downscale_1080p_div4(gpu_device):
    from elvis_amd import ops
    from oracle import glue_ref
    rng = np.random.default_rng(4)
    x = rng.integers(0, 256, size=(1, 1080, 1920, 3), dtype=np.uint8)
    out = ops.area_downscale_u8(_dev(x, gpu_device), 4).cpu().numpy()
    assert np.array_equal(out[0], glue_ref.area_downscale_u8(x[0], 4))
    with pytest.raises(ValueError):
        ops.area_downscale_u8(_dev(x[:, :1079], gpu_device), 4)


@pytest.mark.parametrize("alpha", [1.0, 0.35])
def test_blend(gpu_device, alpha):
    from elvis_amd import ops
    from oracle import glue_ref
    rng = np.random.default_rng(5)
    o = rng.integers(0, 256, size=(2, 40, 56, 3), dtype=np.uint8)
    r = rng.integers(0, 256, size=(2, 40, 56, 3), dtype=np.uint8)
    m = (rng.random((2, 5, 7)) < 0.5).astype(np.int32) * rng.integers(1, 4, size=(2, 5, 7)).astype(np.int32)
    out = ops.blend_u8(_dev(o, gpu_device), _dev(r, gpu_device), _dev(m, gpu_device), 8, alpha).cpu().numpy()
    for i in range(2):
        assert np.array_equal(out[i], glue_ref.blend_by_map(o[i], r[i], m[i], 8, alpha))


def test_tiler_matches_reference_golden(gpu_device, golden_dir):
    """resource_aware_restore on the GPU accumulate/normalise kernels vs the outputs of the
    reference's own utils.resource_aware_restore (tests/golden/tiler.npz) - bit exact."""
    from elvis_amd import tiler
    g = np.load(os.path.join(golden_dir, "tiler.npz"))

    def ident(frames, device=None, **kw):
        return [f.copy() for f in frames]

    def affine(frames, device=None, **kw):
        return [np.clip(f.astype(np.float32) * 0.5 + 7.0, 0, 255).astype(np.uint8) for f in frames]

    def coord(frames, device=None, tile_coords=None, **kw):
        t0, t1, y0, y1, x0, x1 = tile_coords if tile_coords else (0, 0, 0, 0, 0, 0)
        add = (y0 * 3 + x0 * 5 + t0 * 11) % 37
        return [np.clip(f.astype(np.int32) + add, 0, 255).astype(np.uint8) for f in frames]

    fns = {"ident": ident, "affine": affine, "coord": coord}
    for k in range(int(g["count"])):
        n, h, w, tile, halo, chunk, ov = [int(v) for v in g[f"c{k}_cfg"]]
        frames = [f for f in g[f"c{k}_in"]]
        fn = fns[str(g[f"c{k}_fn"])]
        if int(g[f"c{k}_raised"]):
            with pytest.raises(ValueError):
                tiler.resource_aware_restore(fn, frames, tile_size=tile, halo=halo, chunk_size=chunk,
                                             chunk_overlap=ov, device="cuda:0")
            continue
        out = tiler.resource_aware_restore(fn, frames, tile_size=tile, halo=halo, chunk_size=chunk, chunk_overlap=ov,
                                           device="cuda:0")
        assert np.array_equal(np.stack(out), g[f"c{k}_out"]), f"case {k}"


def test_select_levels(gpu_device):
    from elvis_amd import recompose
    from oracle import glue_ref
    rng = np.random.default_rng(6)
    frames = [rng.integers(0, 256, size=(36, 52, 3), dtype=np.uint8) for _ in range(3)]
    maps = [rng.integers(0, 3, size=(4, 6)) * 2 for _ in range(3)]   # levels {0,2,4}, floored grid (b=8)

    def fn(frames, degradation_level=0, **kw):
        return [np.clip(f.astype(np.int32) + 10 * int(degradation_level), 0, 255).astype(np.uint8) for f in frames]

    got = recompose.restore_video_adaptively(fn, frames, maps, block_size=8, device="cuda:0")
    ref = glue_ref.restore_video_adaptively(fn, frames, maps, block_size=8)
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)


def test_upscale_adaptive_and_rounds_drivers(gpu_device):
    """a3 / a6 control flow on device vs the oracle with a deterministic stand-in model
    (nearest 2x / 4x upsample, +3 restorer) - isolates the recompose logic from the network."""
    from elvis_amd import recompose, ops
    from oracle import glue_ref
    rng = np.random.default_rng(7)
    h, w, b = 64, 96, 8
    frames = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for _ in range(3)]
    levels = rng.integers(0, 4, size=(3, h // b, w // b)).astype(np.int32)
    levels[1] = np.minimum(levels[1], 2)

    for scale in (2, 4):
        def up_np(img, s=scale):
            return np.repeat(np.repeat(img, s, axis=0), s, axis=1)

        def up_dev(t, s=scale):
            return t.repeat_interleave(s, dim=1).repeat_interleave(s, dim=2).contiguous()

        for i in range(3):
            fd = recompose.frames_to_device([frames[i]], gpu_device)
            md = recompose.maps_to_device(levels[i], 1, gpu_device)
            got = recompose.upscale_adaptive_device(fd, md, b, up_dev, sr_scale=scale).cpu().numpy()[0]
            ref = glue_ref.upscale_adaptive(frames[i], levels[i], b, up_np, step=scale)
            assert np.array_equal(got, ref), (scale, i)

    maps = rng.integers(0, 4, size=(3, h // b, w // b)).astype(np.int32)
    maps[2] = 0
    fd = recompose.frames_to_device(frames, gpu_device)
    md = recompose.maps_to_device(maps, 3, gpu_device)
    got = recompose.rounds_recompose_device(fd, md, b, lambda t: (t.to(torch.int32) + 3).clamp(0, 255).to(torch.uint8),
                                            batch_size=2).cpu().numpy()
    ref = glue_ref.rounds_recompose(frames, maps, b, lambda fs: [np.clip(f.astype(np.int32) + 3, 0, 255).astype(np.uint8) for f in fs], 2)
    for i in range(3):
        assert np.array_equal(got[i], ref[i])


def test_sse_psnr(gpu_device, golden_dir):
    from elvis_amd import ops
    g = np.load(os.path.join(golden_dir, "psnr.npz"))
    a, b, m = g["a"], g["b"], g["mask"]
    sse, cnt = ops.sse_u8(_dev(a, gpu_device), _dev(b, gpu_device))
    mse = (sse.double() / cnt.double()).cpu().numpy()
    assert np.allclose(mse, g["mse_full"], rtol=1e-6)
    sse, cnt = ops.sse_u8(_dev(a, gpu_device), _dev(b, gpu_device), _dev(m.astype(np.uint8), gpu_device))
    cntn = cnt.cpu().numpy()
    msem = np.where(cntn > 0, sse.cpu().numpy() / np.maximum(cntn, 1), 0.0)
    assert np.allclose(msem, g["mse_masked"], rtol=1e-6)


def test_error_conventions(gpu_device):
    from elvis_amd import ops, recompose, restore
    with pytest.raises(ValueError):
        recompose.split_image_into_blocks(np.zeros((10, 16, 3), np.uint8), 8)
    a = torch.zeros((1, 16, 16, 3), dtype=torch.uint8, device=gpu_device)
    with pytest.raises(ValueError):
        ops.recompose_u8(a, a, torch.zeros((1, 2, 2), dtype=torch.int32, device=gpu_device), 0, 0)
    with pytest.raises(RuntimeError):
        restore.get_sinsr_model("cpu")
