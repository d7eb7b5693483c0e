#!/usr/bin/env python3
"""How much head-room do the mixed modes have under the 1e-3 bar on OTHER tiles than the bench's?  (test tooling)
    python tools/mixed_margin.py [n_tiles]"""
import dataclasses, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from elvis_amd.sinsr import SinSRModel
from elvis_amd.synth import synth_clip
from elvis_amd.weights import SinSRConfig, frame_noise, make_sinsr_weights
from oracle import sinsr_ref as R

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
cfg = dataclasses.replace(SinSRConfig(), quantize=False)
sd = make_sinsr_weights(cfg, 0)
torch.set_num_threads(16)
modes = ["x3", "mixed:dec0", "mixed:enc0+dec0", "mixed:enc0+enc1+dec0"]
models = {m: SinSRModel(cfg, sd, dev, torch.float16, precision=m) for m in modes}
worst = {m: 0.0 for m in modes}
for t in range(n):
    lr = torch.from_numpy(synth_clip(1000 + 17 * t, 1, 64, 64)[0])
    noise = frame_noise(cfg, 100 + t, t, 64, 64)
    ref = R.sinsr_forward(sd, cfg, lr, noise)
    row = []
    for m in modes:
        _, f32 = models[m].forward(lr[None].to(dev), noise.to(dev), want_f32=True)
        e = (f32[0].cpu() - ref).abs().max().item()
        worst[m] = max(worst[m], e)
        row.append(f"{e:.3e}")
    print(f"tile {t}: " + "  ".join(row), flush=True)
print("worst:  " + "  ".join(f"{m} {worst[m]:.3e}" for m in modes))
