#!/usr/bin/env python3
"""GPU companion of tools/precision_study.py (test tooling): max-abs of the pre-quantisation output against
the fp32 CPU oracle on the 256x256 parity tile (full-width config, continuous path) and 1080p frames/s,
for the uniform modes and a list of mixed modes.

    python tools/precision_gpu.py [--modes f16 mixed:dec0 f32] [--frames 2]"""
import argparse, dataclasses, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from elvis_amd.sinsr import SinSRModel
from elvis_amd.synth import synth_clip
from elvis_amd.weights import SinSRConfig, frame_noise, make_sinsr_weights
from oracle import sinsr_ref as R

ap = argparse.ArgumentParser()
ap.add_argument("--modes", nargs="+", default=["f16", "mixed:dec0", "mixed:enc0+dec0", "mixed:dec1+dec0",
                                               "mixed:enc0+enc1+dec0", "mixed:enc0+enc1+dec1+dec0", "f32"])
ap.add_argument("--frames", type=int, default=2)
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = dataclasses.replace(SinSRConfig(), quantize=False)
sd = make_sinsr_weights(cfg, 0)
torch.set_num_threads(int(os.environ.get("ELVIS_CPU_THREADS", "16")))
lr = torch.from_numpy(synth_clip(20260501, 1, 64, 64)[0])
noise = frame_noise(cfg, 42, 0, 64, 64)
ref = R.sinsr_forward(sd, cfg, lr, noise)
lr_big = (torch.rand(a.frames, 270, 480, 3, device=dev) * 255).to(torch.uint8)
print(f"{'mode':32s} max_abs_f32   rms        u8_max  1080p frames/s", flush=True)
for mode in a.modes:
    dt = torch.float32 if mode == "f32" else torch.float16
    m = SinSRModel(cfg, sd, dev, dt, precision=mode if (mode.startswith("mixed") or mode == "x3") else None)
    u8, f32 = m.forward(lr[None].to(dev), noise.to(dev), want_f32=True)
    d = (f32[0].cpu() - ref).abs()
    du8 = np.abs(u8[0].cpu().numpy().astype(int) - R.to_u8(ref).numpy().astype(int)).max()
    nz = m.make_noise(42, list(range(a.frames)), 270, 480)
    m.forward(lr_big[:1], nz[:1]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(0, a.frames, 1 if mode != "f16" else a.frames):
        k = 1 if mode != "f16" else a.frames
        m.forward(lr_big[i:i + k], nz[i:i + k])
    torch.cuda.synchronize()
    fps = a.frames / (time.perf_counter() - t0)
    print(f"{mode:32s} {d.max().item():.3e}   {d.pow(2).mean().sqrt().item():.3e}  {du8:3d}     {fps:6.2f}", flush=True)
    del m
    torch.cuda.empty_cache()
