#!/usr/bin/env python3
"""Kernel-time breakdown of one SinSR invocation in a given precision mode (HIP events are not used: run it under
rocprofv3 --kernel-trace --stats).   python tools/mode_profile.py mixed 6"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from elvis_amd import restore, synth

mode, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
model = restore.get_sinsr_model(dev, precision=None if mode == "f16" else mode)
lr = torch.from_numpy(synth.synth_clip(7, 2, 270, 480)).to(dev).repeat((n + 1) // 2, 1, 1, 1)[:n].contiguous()
noise = model.make_noise(42, list(range(n)), 270, 480)
model.forward(lr, noise)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
model.forward(lr, noise)
torch.cuda.synchronize()
print(f"{mode}: {n / (time.perf_counter() - t0):.2f} frames/s (network only)")
