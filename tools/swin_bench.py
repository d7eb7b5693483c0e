#!/usr/bin/env python3
"""Micro-benchmark of the fused Swin kernels (csrc/swin.hip) on the shapes of the two models.
    python tools/swin_bench.py            # ELVIS_SWIN_STAGGER=0/1 forces the stagger off / on"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elvis_amd import ops

SHAPES = [  # name, C, hidden, n, h, w
    ("sinsr192_320x512", 192, 768, 6, 320, 512),
    ("sinsr192_160x256", 192, 768, 6, 160, 256),
    ("blur64_1080p", 64, 128, 2, 1088, 1920),
    ("blur128_540p", 128, 256, 2, 544, 960),
    ("blur256_270p", 256, 512, 2, 272, 480),
]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for name, c, hid, n, h, w in SHAPES:
    nw, nb = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1
    w1, b1 = torch.randn(hid, c, generator=g) / math.sqrt(c), torch.randn(hid, generator=g) * 0.1
    w2, b2 = torch.randn(c, hid, generator=g) / math.sqrt(hid), torch.randn(c, generator=g) * 0.1
    wp, bp = torch.randn(c, c, generator=g) / math.sqrt(c), torch.randn(c, generator=g) * 0.1
    wq, bq = torch.randn(3 * c, c, generator=g) / math.sqrt(c), torch.randn(3 * c, generator=g) * 0.1
    x = ops.Act(torch.randn((n, h, w, c), device=dev, dtype=torch.float16), c)
    a = ops.Act(torch.randn((n, h, w, c), device=dev, dtype=torch.float16), c)
    mods = {"mlp": (ops.SwinFused(nw, nb, w1, b1, w2, b2, device=dev), (x,)),
            "proj_mlp": (ops.SwinFused(nw, nb, w1, b1, w2, b2, proj_w=wp, proj_b=bp, device=dev), (a, x)),
            "ln_qkv": (ops.SwinFused(nw, nb, wq, bq, device=dev), (x,))}
    for kind, (m, args) in mods.items():
        m(*args)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); m(*args); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = sorted(ts)[2]
        tok = n * h * w
        fl = (4.0 * c * hid + (2.0 * c * c if kind == "proj_mlp" else 0)) * tok if kind != "ln_qkv" else 2.0 * c * 3 * c * tok
        by = (3 if kind == "mlp" else 3 if kind == "proj_mlp" else 4) * c * 2.0 * tok
        print(f"{name:18s} {kind:9s} {t*1e3:8.1f} us  {fl/t/1e9:7.1f} TFLOP/s  {by/t/1e6:7.1f} GB/s algorithmic", flush=True)

# window attention on the same token images (qkv = 3C channels): ELVIS_ATTN_LDS=1 selects round 2's kernel
for name, c, hid, n, h, w in SHAPES:
    heads = c // 32
    qkv = ops.Act(torch.randn((n, h, w, 3 * c), device=dev, dtype=torch.float16), 3 * c)
    table = torch.randn((225, heads), device=dev) * 0.5
    for shift in (0, 4):
        ops.window_attention(qkv, heads, 32, 8, shift, table, 32 ** -0.5)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.window_attention(qkv, heads, 32, 8, shift, table, 32 ** -0.5); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = sorted(ts)[2]
        by = 4 * c * 2.0 * n * h * w
        print(f"{name:18s} attention shift {shift}  {t*1e3:8.1f} us  {by/t/1e6:7.1f} GB/s algorithmic", flush=True)
