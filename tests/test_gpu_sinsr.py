"""End-to-end GPU parity of the SinSR-style 4x path against the PyTorch-CPU oracle on identical
degraded inputs, weights (seed 0) and sampler noise (seed 42).

Bars (BASELINE.json): fp32 mode <= 1e-3 max-abs on the pre-quantisation image in [0,1], PSNR
within 0.01 dB; f16 mode is reported against looser documented bars.  The VQ nearest-code step is
discontinuous, so the strict bar is asserted on the continuous path (quantize=False) and stage-wise
(z0 before the lookup; decoder fed identical codes), and the quantised end-to-end run is held to a
code-agreement + PSNR bar."""
import dataclasses

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(cfg, h, w, dtype, dev, seed=0):
    from elvis_amd.sinsr import SinSRModel
    from elvis_amd.weights import frame_noise, make_sinsr_weights
    from oracle import sinsr_ref as R
    sd = make_sinsr_weights(cfg, seed)
    model = SinSRModel(cfg, sd, dev, dtype, fuse_gn=False)
    rng = np.random.default_rng(20260501)
    # smooth-ish synthetic LR content
    base = rng.random((h // 4 + 2, w // 4 + 2, 3)).astype(np.float32)
    lr = np.kron(base, np.ones((4, 4, 1), np.float32))[:h, :w]
    lr = np.clip(lr + rng.normal(0, 0.03, lr.shape), 0, 1)
    lr_u8 = torch.from_numpy(np.round(lr * 255).astype(np.uint8))
    hp, wp = R.padded_latent_shape(cfg, h, w)
    noise = frame_noise(cfg, 42, 0, hp, wp)
    return sd, model, lr_u8, noise


def _psnr(a, b):
    mse = np.mean((a.astype(np.float32) - b.astype(np.float32)) ** 2)
    return float("inf") if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


@pytest.mark.parametrize("hw", [(24, 40), (64, 64)])
def test_tiny_fp32_continuous_strict(gpu_device, hw):
    from elvis_amd.weights import tiny_config
    from oracle import sinsr_ref as R
    cfg = dataclasses.replace(tiny_config(), quantize=False)
    sd, model, lr, noise = _setup(cfg, hw[0], hw[1], torch.float32, gpu_device)
    stages = {}
    u8, f32 = model.forward(lr[None].to(gpu_device), noise.to(gpu_device), want_f32=True, stages=stages)
    ref, rst = R.sinsr_forward(sd, cfg, lr, noise, return_stages=True)
    for name in ("y_up", "z_y", "z0", "dec"):
        a = stages[name]
        got = a.t[0, :, :, :a.c].float().cpu().permute(2, 0, 1)
        assert (got - rst[name][0]).abs().max().item() < 1e-3, name
    assert (f32[0].cpu() - ref).abs().max().item() < 1e-3
    ref_u8 = R.to_u8(ref).numpy()
    got_u8 = u8[0].cpu().numpy()
    assert np.abs(got_u8.astype(int) - ref_u8.astype(int)).max() <= 1
    assert _psnr(got_u8, ref_u8) > 60.0   # >> the 0.01 dB agreement bar


def test_tiny_fp32_quantized(gpu_device):
    from elvis_amd.weights import tiny_config
    from oracle import sinsr_ref as R
    cfg = tiny_config()
    sd, model, lr, noise = _setup(cfg, 32, 48, torch.float32, gpu_device)
    stages = {}
    u8, f32 = model.forward(lr[None].to(gpu_device), noise.to(gpu_device), want_f32=True, stages=stages)
    ref, rst = R.sinsr_forward(sd, cfg, lr, noise, return_stages=True)
    z0 = stages["z0"]
    got_z0 = z0.t[0, :, :, :3].float().cpu().permute(2, 0, 1)
    assert (got_z0 - rst["z0"][0]).abs().max().item() < 1e-3
    # decoder parity when fed the oracle's z0 (identical codes by construction)
    from elvis_amd import ops
    za = ops.new_act(1, 32, 48, 3, torch.float32, gpu_device, zero=True)
    za.t[..., :3] = rst["z0"][0].permute(1, 2, 0).to(gpu_device)
    dec, idx = model.decode(za, want_idx=True)
    _, ref_idx = R.vq_quantize(sd, rst["z0"])
    assert torch.equal(idx.cpu().long(), ref_idx)
    got = dec.t[0, :, :, :3].cpu().permute(2, 0, 1)
    assert (got - rst["dec"][0]).abs().max().item() < 1e-3
    # end to end with the lookup in the loop: codes may flip where z0 sits on a cell boundary
    agree = np.mean(np.abs(u8[0].cpu().numpy().astype(int) - R.to_u8(ref).numpy().astype(int)) <= 1)
    assert agree > 0.995


def test_tiny_f16_fast_mode(gpu_device):
    from elvis_amd.weights import tiny_config
    from oracle import sinsr_ref as R
    cfg = dataclasses.replace(tiny_config(), quantize=False)
    sd, model, lr, noise = _setup(cfg, 32, 48, torch.float16, gpu_device)
    u8, f32 = model.forward(lr[None].to(gpu_device), noise.to(gpu_device), want_f32=True)
    ref = R.sinsr_forward(sd, cfg, lr, noise)
    err = (f32[0].cpu() - ref).abs().max().item()
    print("f16 max-abs", err)
    assert err < 3e-2                      # documented f16 bar (fp32 bar is 1e-3)
    assert _psnr(u8[0].cpu().numpy(), R.to_u8(ref).numpy()) > 40.0


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.float16, 3e-2)])
def test_fused_default_path_vs_oracle(gpu_device, dtype, tol):
    """The DEFAULT product configuration (GroupNorm statistics from the producing conv's epilogue,
    normalise+SiLU in the consuming conv's LDS prologue) against the CPU oracle."""
    from elvis_amd.sinsr import SinSRModel
    from elvis_amd.weights import tiny_config
    from oracle import sinsr_ref as R
    cfg = dataclasses.replace(tiny_config(), quantize=False)
    sd, _, lr, noise = _setup(cfg, 40, 72, dtype, gpu_device)
    model = SinSRModel(cfg, sd, gpu_device, dtype)
    assert model.fuse_gn
    f32 = model.forward(lr[None].to(gpu_device), noise.to(gpu_device), want_f32=True)[1]
    ref = R.sinsr_forward(sd, cfg, lr, noise)
    assert (f32[0].cpu() - ref).abs().max().item() < tol


def test_fused_gn_prologue_matches_unfused(gpu_device):
    from elvis_amd.sinsr import SinSRModel
    from elvis_amd.weights import tiny_config
    cfg = dataclasses.replace(tiny_config(), quantize=False)
    sd, model, lr, noise = _setup(cfg, 24, 40, torch.float32, gpu_device)
    fused = SinSRModel(cfg, sd, gpu_device, torch.float32, fuse_gn=True)
    a = model.forward(lr[None].to(gpu_device), noise.to(gpu_device), want_f32=True)[1]
    b = fused.forward(lr[None].to(gpu_device), noise.to(gpu_device), want_f32=True)[1]
    assert (a - b).abs().max().item() < 1e-4


def test_full_config_256_fp32(gpu_device):
    """BASELINE config 1: one 256x256 frame (64x64 LR), full-width SinSR config, fp32 mode."""
    from elvis_amd.weights import SinSRConfig
    from oracle import sinsr_ref as R
    cfg = dataclasses.replace(SinSRConfig(), quantize=False)
    sd, model, lr, noise = _setup(cfg, 64, 64, torch.float32, gpu_device)
    u8, f32 = model.forward(lr[None].to(gpu_device), noise.to(gpu_device), want_f32=True)
    torch.set_num_threads(max(1, torch.get_num_threads()))
    ref = R.sinsr_forward(sd, cfg, lr, noise)
    err = (f32[0].cpu() - ref).abs().max().item()
    print("full-config fp32 max-abs", err)
    assert err < 1e-3
    assert np.abs(u8[0].cpu().numpy().astype(int) - R.to_u8(ref).numpy().astype(int)).max() <= 1


def test_restore_frames_sinsr_surface(gpu_device):
    """P2 drop-in: restore_frames_sinsr keeps level-0 blocks bit-identical to the decoded input and
    is independent of how frames are chunked (noise keyed on the global frame index)."""
    from elvis_amd import restore
    from elvis_amd.weights import tiny_config
    cfg = tiny_config()
    rng = np.random.default_rng(11)
    frames = [rng.integers(0, 256, size=(64, 96, 3), dtype=np.uint8) for _ in range(3)]
    maps = rng.integers(0, 3, size=(3, 8, 12)).astype(np.int32)
    maps[2] = 0
    for sched in ("staged", "single4x"):
        o = restore.restore_frames_sinsr(frames, maps, 8, gpu_device, cfg=cfg, schedule=sched)
        assert np.array_equal(o[2], frames[2])
        k0 = np.repeat(np.repeat(maps[0] == 0, 8, 0), 8, 1)
        assert np.array_equal(o[0][k0], frames[0][k0]) and not np.array_equal(o[0][~k0], frames[0][~k0])
    out = restore.restore_frames_sinsr(frames, maps, 8, gpu_device, cfg=cfg)
    assert len(out) == 3 and out[0].shape == (64, 96, 3) and out[0].dtype == np.uint8
    assert np.array_equal(out[2], frames[2])
    keep = np.repeat(np.repeat(maps[0] == 0, 8, 0), 8, 1)
    assert np.array_equal(out[0][keep], frames[0][keep])
    # chunk independence
    a = restore.restore_frames_sinsr(frames[:1], maps[:1], 8, gpu_device, cfg=cfg, first_frame_index=0)
    b = restore.restore_frames_sinsr(frames[1:], maps[1:], 8, gpu_device, cfg=cfg, first_frame_index=1)
    for x, y in zip(a + b, out):
        assert np.array_equal(x, y)


# ---------------------------------------------------------------------------------------------------------
# Full-width config in the BENCHMARKED mode (f16 storage + f16 MFMA operands, fused GroupNorm, the VQ lookup
# on) and in the cheapest mode that meets the north star's 1e-3 bar.  What the f16 mode can and cannot meet is
# measured, not assumed: tools/precision_study.py emulates every rounding on the CPU (f16 operands ALONE give
# 1.6e-3, f16 storage alone 1.7e-3, each section alone 0.4 - 1.5e-3), tools/precision_gpu.py measures the modes
# on the device.  The bounds below are the measured values with ~1.5x head-room, so a precision
# regression of the fast path fails here.
def _psnr_delta_vs_fixed_target(got_u8, ref_u8, lr):
    """North star: "PSNR within 0.01 dB".  PSNR of the device's and the oracle's frame against ONE fixed target
    (the LR tile, nearest x4 - any target does, it only has to be the same for both); returns |difference| in dB."""
    target = np.kron(lr.numpy().astype(np.float32), np.ones((4, 4, 1), np.float32))
    return abs(_psnr(got_u8, target) - _psnr(ref_u8, target))


def _full_width_case(dev, quantize):
    from elvis_amd.synth import synth_clip
    from elvis_amd.weights import SinSRConfig, frame_noise, make_sinsr_weights
    from oracle import sinsr_ref as R
    cfg = dataclasses.replace(SinSRConfig(), quantize=quantize)
    sd = make_sinsr_weights(cfg, 0)
    lr = torch.from_numpy(synth_clip(20260501, 1, 64, 64)[0])      # the parity tile bench.py reports
    noise = frame_noise(cfg, 42, 0, 64, 64)
    ref, stages = R.sinsr_forward(sd, cfg, lr, noise, return_stages=True)
    return cfg, sd, lr, noise, ref, stages


def test_full_width_f16_benchmarked_mode_continuous(gpu_device):
    from elvis_amd.sinsr import SinSRModel
    from oracle import sinsr_ref as R
    cfg, sd, lr, noise, ref, _ = _full_width_case(gpu_device, quantize=False)
    model = SinSRModel(cfg, sd, gpu_device, torch.float16)           # fuse_gn=True: the default product path
    u8, f32 = model.forward(lr[None].to(gpu_device), noise.to(gpu_device), want_f32=True)
    d = (f32[0].cpu() - ref).abs()
    print(f"full-width f16: max-abs {d.max().item():.3e} rms {d.pow(2).mean().sqrt().item():.3e}")
    assert d.max().item() <= 3.5e-3          # measured 2.27e-3; the 1e-3 bar is out of reach for f16 operands
    assert d.pow(2).mean().sqrt().item() <= 5e-4      # measured 3.3e-4
    ref_u8, got_u8 = R.to_u8(ref).numpy(), u8[0].cpu().numpy()
    assert np.abs(got_u8.astype(int) - ref_u8.astype(int)).max() <= 1
    assert _psnr(got_u8, ref_u8) >= 58.0     # measured 60.0 dB
    assert _psnr_delta_vs_fixed_target(got_u8, ref_u8, lr) <= 0.01    # the north star's PSNR criterion holds even in f16
    again = model.forward(lr[None].to(gpu_device), noise.to(gpu_device))
    assert torch.equal(again, u8)            # bit-reproducible


def test_full_width_f16_benchmarked_mode_with_vq_lookup(gpu_device):
    """quantize=True (what bench.py times): the nearest-code lookup is discontinuous, so a latent within f16
    rounding of a Voronoi boundary picks the neighbouring code.  Asserted: how many codes agree with the fp32
    oracle's, and the PSNR of the frame that results."""
    from elvis_amd.sinsr import SinSRModel
    from oracle import sinsr_ref as R
    cfg, sd, lr, noise, ref, stages = _full_width_case(gpu_device, quantize=True)
    model = SinSRModel(cfg, sd, gpu_device, torch.float16)
    st = {}
    u8 = model.forward(lr[None].to(gpu_device), noise.to(gpu_device), stages=st)
    _, ref_idx = R.vq_quantize(sd, stages["z0"])
    _, idx = model.decode(st["z0"], want_idx=True)
    agree = (idx.cpu().long() == ref_idx).float().mean().item()
    psnr = _psnr(u8[0].cpu().numpy(), R.to_u8(ref).numpy())
    print(f"full-width f16 + VQ: code agreement {agree:.4f}, PSNR vs oracle {psnr:.2f} dB")
    assert agree >= 0.97
    assert psnr >= 44.0                       # measured 47.3 dB


def test_full_width_compensated_mode_is_fp32_grade(gpu_device):
    """precision="x3": fp32 tensors everywhere, every conv's products on the f16 matrix pipe with the operands
    split hi + lo (ELVIS_F32X3).  Same bar as the exact fp32 mode's full-width test."""
    from elvis_amd.sinsr import SinSRModel
    from oracle import sinsr_ref as R
    cfg, sd, lr, noise, ref, _ = _full_width_case(gpu_device, quantize=False)
    model = SinSRModel(cfg, sd, gpu_device, torch.float16, precision="x3")
    assert model.dtype == torch.float32 and model.x3
    u8, f32 = model.forward(lr[None].to(gpu_device), noise.to(gpu_device), want_f32=True)
    err = (f32[0].cpu() - ref).abs().max().item()
    print("full-width x3 max-abs", err)
    assert err <= 1e-4
    got_u8, ref_u8 = u8[0].cpu().numpy(), R.to_u8(ref).numpy()
    assert np.abs(got_u8.astype(int) - ref_u8.astype(int)).max() <= 1
    assert _psnr_delta_vs_fixed_target(got_u8, ref_u8, lr) <= 0.01


def test_full_width_mixed_mode_meets_the_1e3_bar(gpu_device):
    """precision="mixed": the 1080p-resolution decoder level in f16, everything else on fp32 tensors with the
    compensated f16 MFMA ("mixed_exact" keeps the fp32 MFMA for those sections and lands on the same error).  On THIS
    tile it is inside the north star's tolerance; over other tiles it ranges 0.7-1.05e-3 (tools/mixed_margin.py), so
    the mode that is safely inside the bar is "x3" (test above)."""
    from elvis_amd.sinsr import SinSRModel
    from oracle import sinsr_ref as R
    cfg, sd, lr, noise, ref, _ = _full_width_case(gpu_device, quantize=False)
    model = SinSRModel(cfg, sd, gpu_device, torch.float16, precision="mixed")
    u8, f32 = model.forward(lr[None].to(gpu_device), noise.to(gpu_device), want_f32=True)
    err = (f32[0].cpu() - ref).abs().max().item()
    print("full-width mixed max-abs", err)
    assert err <= 1e-3                        # measured 7.9e-4
    ref_u8, got_u8 = R.to_u8(ref).numpy(), u8[0].cpu().numpy()
    assert np.abs(got_u8.astype(int) - ref_u8.astype(int)).max() <= 1
    assert _psnr(got_u8, ref_u8) >= 62.0


def test_full_width_decoder_f16_mode_has_no_code_flips(gpu_device):
    """precision="dec_f16": encoder + Swin-UNet on fp32 tensors with the compensated f16 MFMA (the latent that reaches the
    nearest-code lookup carries ~1e-5 of error), everything behind the lookup in f16.  With quantize=True - the
    benchmarked configuration - every code must equal the fp32 oracle's, so the u8 frame is within 1 LSB of the CPU
    path's (plain f16 flips up to 3 % of the codes: a 27-LSB local difference).  The f32 max-abs of the
    decoder's f16 arithmetic alone (~1.5-2e-3) is printed and bounded, not hidden."""
    from elvis_amd.sinsr import SinSRModel
    from oracle import sinsr_ref as R
    cfg, sd, lr, noise, ref, stages = _full_width_case(gpu_device, quantize=True)
    model = SinSRModel(cfg, sd, gpu_device, torch.float16, precision="dec_f16")
    assert model.sec_dtype["unet"] == torch.float32 and model.sec_dtype["dec0"] == torch.float16 and model.x3
    st = {}
    u8, f32 = model.forward(lr[None].to(gpu_device), noise.to(gpu_device), want_f32=True, stages=st)
    z0 = st["z0"].t[0, :, :, :3].float().cpu().permute(2, 0, 1)
    print("dec_f16: latent max-abs", (z0 - stages["z0"][0]).abs().max().item())
    _, ref_idx = R.vq_quantize(sd, stages["z0"])
    _, idx = model.decode(st["z0"], want_idx=True)
    assert torch.equal(idx.cpu().long(), ref_idx)                     # 100 % code agreement
    err = (f32[0].cpu() - ref).abs().max().item()
    got_u8, ref_u8 = u8[0].cpu().numpy(), R.to_u8(ref).numpy()
    print(f"dec_f16 + VQ: f32 max-abs {err:.3e}, u8 max {np.abs(got_u8.astype(int) - ref_u8.astype(int)).max()}, "
          f"PSNR vs oracle {_psnr(got_u8, ref_u8):.2f} dB")
    assert np.abs(got_u8.astype(int) - ref_u8.astype(int)).max() <= 1
    assert err <= 3e-3                                                # the decoder's f16 arithmetic (measured ~1.6e-3)
    assert _psnr(got_u8, ref_u8) >= 58.0
    assert _psnr_delta_vs_fixed_target(got_u8, ref_u8, lr) <= 0.01
    assert torch.equal(model.forward(lr[None].to(gpu_device), noise.to(gpu_device)), u8)


def test_weights_file_round_trip(gpu_device, tmp_path):
    """Weights from a file (the reference resolves a checkpoint path, elvis.py:2445-2464): the upstream-layout state
    dict written by torch.save and read back with torch.load(weights_only=True) - and through safetensors - drives
    the device graph to the same output, bit for bit, as the seeded weights it was made from."""
    from safetensors.torch import load_file, save_file
    from elvis_amd.sinsr import SinSRModel
    from elvis_amd.weights import frame_noise, make_sinsr_weights, tiny_config
    cfg = tiny_config()
    sd = make_sinsr_weights(cfg, 0)
    pt, st = tmp_path / "sinsr.pt", tmp_path / "sinsr.safetensors"
    torch.save(sd, pt)
    save_file({k: v.contiguous() for k, v in sd.items()}, str(st))
    lr = torch.from_numpy(np.random.default_rng(5).integers(0, 256, size=(1, 24, 40, 3), dtype=np.uint8)).to(gpu_device)
    hp, wp = SinSRModel(cfg, sd, gpu_device).padded_latent_shape(24, 40)
    noise = frame_noise(cfg, 42, 0, hp, wp).to(gpu_device)
    want = SinSRModel(cfg, None, gpu_device, weight_seed=0).forward(lr, noise)
    for loaded in (torch.load(pt, weights_only=True), load_file(str(st))):
        assert set(loaded) == set(sd)
        assert torch.equal(SinSRModel(cfg, loaded, gpu_device).forward(lr, noise), want)
