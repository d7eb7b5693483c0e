"""GPU parity of the DCT-slot (DCNv2 restorer) and Blur-slot (Swin deblur) paths against the
PyTorch-CPU oracle (oracle/restorers_ref.py), and of their frame-level drivers against the oracle's
round-loop recompose.  fp32 bar 1e-3 max-abs on the [0,1] image; f16 documented looser bars."""
import dataclasses

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _act(x_nchw, dtype, dev):
    from elvis_amd import ops
    n, c, h, w = x_nchw.shape
    a = ops.new_act(n, h, w, c, dtype, dev, zero=True)
    a.t[..., :c] = x_nchw.permute(0, 2, 3, 1).to(dev, dtype)
    return a


@pytest.mark.parametrize("dtype,tol,off_scale", [(torch.float32, 1e-4, 2.0), (torch.float16, 2e-2, 2.0),
                                                 (torch.float16, 2e-2, 0.5),    # every offset <= 6 px: the tiled kernel's unchecked sample loop
                                                 (torch.float16, 2e-2, -1.0)])  # small in the top rows only: both loops in one launch
def test_dcnv2_kernel(gpu_device, dtype, tol, off_scale):
    from elvis_amd import ops
    from oracle import restorers_ref as R
    g = torch.Generator().manual_seed(21)
    n, c, h, w, G, co = 2, 7, 19, 23, 7, 64
    x = torch.rand(n, c, h, w, generator=g)
    off = torch.randn(n, 18 * G, h, w, generator=g) * abs(off_scale)   # 2.0: up to several pixels, leaves the image at borders
    if off_scale < 0:
        off *= 3.0
        off[:, :, :9] *= 0.1
    mlog = torch.randn(n, 9 * G, h, w, generator=g)
    wt = torch.randn(co, c, 3, 3, generator=g) / 8
    b = torch.randn(co, generator=g) * 0.1
    om = torch.cat([off, mlog], 1)
    xa, oma = _act(x, dtype, gpu_device), _act(om, dtype, gpu_device)
    y = ops.dcnv2(xa, oma, wt.to(gpu_device, dtype).contiguous(), b.to(gpu_device), G, co, mask_sigmoid=True, act=3)
    q = lambda t: t.to(dtype).float()
    ref = torch.relu(R.dcnv2(q(x), q(off), torch.sigmoid(q(mlog)), q(wt), b, G))
    got = y.t[..., :co].float().cpu().permute(0, 3, 1, 2)
    assert (got - ref).abs().max().item() < tol
    # groups < channels (2 channels per deformable group)
    x2 = torch.rand(1, 8, 12, 16, generator=g)
    om2 = torch.cat([torch.randn(1, 18 * 4, 12, 16, generator=g), torch.randn(1, 9 * 4, 12, 16, generator=g)], 1)
    w2 = torch.randn(16, 8, 3, 3, generator=g) / 8
    y2 = ops.dcnv2(_act(x2, torch.float32, gpu_device), _act(om2, torch.float32, gpu_device),
                   w2.to(gpu_device).contiguous(), None, 4, 16, mask_sigmoid=True)
    ref2 = R.dcnv2(x2, om2[:, :72], torch.sigmoid(om2[:, 72:]), w2, torch.zeros(16), 4)
    assert (y2.t[..., :16].cpu().permute(0, 3, 1, 2) - ref2).abs().max().item() < 1e-4


def test_dcnv2_tile_kernel_eight_groups(gpu_device):
    """cin = 8, deformable_groups = 8, f16: the tiled kernel's other instantiation.  Its offset / mask row is 216 halfs
    (27 x 16-byte pieces), with the masks of groups 5-7 in the last three pieces (a fixed 24-piece load
    read an offset dword in their place)."""
    from elvis_amd import ops
    from oracle import restorers_ref as R
    g = torch.Generator().manual_seed(22)
    n, c, h, w, G, co = 1, 8, 21, 37, 8, 48
    x = torch.rand(n, c, h, w, generator=g)
    off = torch.randn(n, 18 * G, h, w, generator=g) * 2.0
    mlog = torch.randn(n, 9 * G, h, w, generator=g) * 2.0           # spread masks: a wrong mask moves the output by O(0.1)
    wt = torch.randn(co, c, 3, 3, generator=g) / 8
    b = torch.randn(co, generator=g) * 0.1
    dt = torch.float16
    xa, oma = _act(x, dt, gpu_device), _act(torch.cat([off, mlog], 1), dt, gpu_device)
    y = ops.dcnv2(xa, oma, wt.to(gpu_device, dt).contiguous(), b.to(gpu_device), G, co, mask_sigmoid=True, act=0)
    q = lambda t: t.to(dt).float()
    ref = R.dcnv2(q(x), q(off), torch.sigmoid(q(mlog)), q(wt), b, G)
    got = y.t[..., :co].float().cpu().permute(0, 3, 1, 2)
    assert (got - ref).abs().max().item() < 2e-2


@pytest.mark.parametrize("dtype,lsb", [(torch.float32, 1), (torch.float16, 2)])   # f16 measured: 1 LSB (bench line slots.dct.parity)
def test_dcn_restorer_vs_oracle(gpu_device, dtype, lsb):
    from elvis_amd.restorers import DCNRestorer
    from elvis_amd.weights import DCNRestorerConfig, make_dcn_weights
    from oracle import restorers_ref as R
    cfg = DCNRestorerConfig()
    sd = make_dcn_weights(cfg, 0)
    rng = np.random.default_rng(3)
    frames = torch.from_numpy(rng.integers(0, 256, size=(5, 32, 48, 3), dtype=np.uint8))
    model = DCNRestorer(cfg, sd, gpu_device, dtype)
    got = model.restore(frames.to(gpu_device), chunk=2).cpu().numpy().astype(int)
    ref, _ = R.dcn_restore_frames(sd, cfg, frames)
    diff = np.abs(got - ref.numpy().astype(int))
    assert diff.max() <= lsb
    if dtype == torch.float32:
        assert (diff > 0).mean() < 1e-3      # rounding ties only


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.float16, 4e-3)])   # f16 measured: 1.9e-3 (bench line slots.blur.parity)
def test_swin_deblur_vs_oracle(gpu_device, dtype, tol):
    from elvis_amd.restorers import SwinDeblur
    from elvis_amd.weights import SwinDeblurConfig, make_deblur_weights
    from oracle import restorers_ref as R
    cfg = SwinDeblurConfig()
    sd = make_deblur_weights(cfg, 0)
    rng = np.random.default_rng(4)
    frames = torch.from_numpy(rng.integers(0, 256, size=(2, 40, 72, 3), dtype=np.uint8))   # padded to 64 x 96 inside
    model = SwinDeblur(cfg, sd, gpu_device, dtype)
    u8, f32 = model.restore(frames.to(gpu_device), swap_rb=False, want_f32=True)
    ref_u8, ref = R.deblur_restore_frames(sd, cfg, frames)
    assert (f32.cpu() - ref).abs().max().item() < tol
    if dtype == torch.float32:
        assert np.abs(u8.cpu().numpy().astype(int) - ref_u8.numpy().astype(int)).max() <= 1


def test_blur_and_dct_drivers_vs_oracle_recompose(gpu_device):
    """Frame-level drivers (BGR surface) == oracle round loop with the oracle models (fp32)."""
    import elvis_amd as E
    from elvis_amd.weights import (DCNRestorerConfig, SwinDeblurConfig, make_dcn_weights, make_deblur_weights)
    from oracle import glue_ref, restorers_ref as R
    rng = np.random.default_rng(5)
    frames = [rng.integers(0, 256, size=(32, 64, 3), dtype=np.uint8) for _ in range(3)]
    maps = rng.integers(0, 3, size=(3, 4, 8)).astype(np.int32)
    maps[1] = 0
    bcfg, bsd = SwinDeblurConfig(), make_deblur_weights(SwinDeblurConfig(), 0)

    def oracle_blur(batch):   # BGR in/out around an RGB model
        t = torch.from_numpy(np.stack([f[:, :, ::-1] for f in batch]).copy())
        out, _ = R.deblur_restore_frames(bsd, bcfg, t)
        return [np.ascontiguousarray(o[:, :, ::-1]) for o in out.numpy()]

    got = E.restore_frames_blur(frames, maps, 8, gpu_device, batch_size=2, fp32=True, state_dict=bsd)
    ref = glue_ref.rounds_recompose(frames, maps, 8, oracle_blur, batch_size=2)
    for a, b in zip(got, ref):
        d = np.abs(a.astype(int) - b.astype(int))
        assert d.max() <= 2 and (d > 0).mean() < 0.02     # 2 rounds of <=1 LSB rounding ties
    assert np.array_equal(got[1], frames[1])              # map == 0: untouched
    keep = np.repeat(np.repeat(maps[0] == 0, 8, 0), 8, 1)
    assert np.array_equal(got[0][keep], frames[0][keep])

    dcfg, dsd = DCNRestorerConfig(), make_dcn_weights(DCNRestorerConfig(), 0)
    got = E.restore_frames_dct(frames, maps, 8, gpu_device, fp32=True, state_dict=dsd)
    rest, _ = R.dcn_restore_frames(dsd, dcfg, torch.from_numpy(np.stack(frames)))
    for i in range(3):
        ref_i = glue_ref.recompose_select(frames[i], rest[i].numpy(), maps[i] <= 0, 8)
        d = np.abs(got[i].astype(int) - ref_i.astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3
