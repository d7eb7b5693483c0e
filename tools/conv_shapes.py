#!/usr/bin/env python3
"""Per-shape table of the conv launches of one 15-frame 1080p SinSR invocation: calls, time, TFLOP/s and (for the
1x1 / HBM-bound ones) the effective bytes/s over in + out + weights.   python tools/conv_shapes.py [frames]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from elvis_amd import ops, restore, synth

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 15
model = restore.get_sinsr_model(dev)
lr = torch.from_numpy(synth.synth_clip(7, 2, 270, 480)).to(dev)
lr = lr.repeat((n + 1) // 2, 1, 1, 1)[:n].contiguous()
noise = model.make_noise(42, list(range(n)), 270, 480)
model.forward(lr, noise)
torch.cuda.synchronize()
prof, shapes = [], []
ops.CONV_PROFILER, ops.CONV_SHAPES = prof, shapes
model.forward(lr, noise)
torch.cuda.synchronize()
ops.CONV_PROFILER = ops.CONV_SHAPES = None
agg = collections.OrderedDict()
for (name, flops, e0, e1), sh in zip(prof, shapes):
    key = (sh, name)
    a = agg.setdefault(key, [0, 0.0, flops])
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
total = sum(a[1] for a in agg.values())
print(f"total conv time {total:.1f} ms for {n} frames")
for (sh, name), (cnt, ms, flops) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    nn, h, w, cin, cout, ks, stride, pro, res = sh
    ho, wo = h // stride, w // stride
    byts = 2.0 * nn * (h * w * cin + ho * wo * cout * (2 if res else 1)) + 2.0 * ks * ks * cin * cout
    print(f"{ms:8.2f} ms {100 * ms / total:5.1f}%  x{cnt:3d}  {ms / cnt * 1e3:8.1f} us  {flops * cnt / ms / 1e9:7.1f} TF/s  {byts * cnt / ms / 1e6:7.1f} GB/s  "
          f"n{nn} {h}x{w} {cin}->{cout} k{ks} s{stride} pro{int(pro)} res{int(res)}  {name.split('<')[1][:-1] if '<' in name else name}")
