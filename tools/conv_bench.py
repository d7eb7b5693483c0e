#!/usr/bin/env python3
"""Micro-benchmark of elvis_conv2d on the layer shapes that dominate the 1080p SinSR frame.
    python tools/conv_bench.py [--dtype f16|f32] [--iters N] [--only NAME] [--prologue] [--stats]
Random (gaussian) data, HIP-event timing on the launch stream, interleaved rounds."""
import argparse, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elvis_amd import ops

SHAPES = {  # name: (cin, cin2, cout, h, w, k, upsample)
    "dec128_1080p": (128, 0, 128, 1080, 1920, 3, False),
    "dec256_540p": (256, 0, 256, 540, 960, 3, False),
    "dec512_270p": (512, 0, 512, 270, 480, 3, False),
    "dec_up512_540p": (512, 0, 512, 270, 480, 3, True),
    "dec_up256_1080p": (256, 0, 256, 540, 960, 3, True),
    "dec256to128_1080p": (256, 0, 128, 1080, 1920, 3, False),
    "unet160_320x512": (160, 0, 160, 320, 512, 3, False),
    "unet_cat640_80x128": (320, 320, 320, 80, 128, 3, False),
    "lin192x768": (192, 0, 768, 320, 512, 1, False),
    "lin192x576": (192, 0, 576, 320, 512, 1, False),
    "lin768x192": (768, 0, 192, 320, 512, 1, False),
    "lin192x192": (192, 0, 192, 320, 512, 1, False),
    # sub-pixel form of "nearest 2x + 3x3" (ops.PackedUpConv: four 2x2 parity convs, kernel <half,128,256,8,false,2,false>)
    "sub_up256_1080p": (256, 0, 256, 540, 960, 3, "sub"),
    "sub_up512_540p": (512, 0, 512, 270, 480, 3, "sub"),
    "unet320_160x256": (320, 0, 320, 160, 256, 3, False),   # 64-channel tile, 8-row form with --prologue
}

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", default=None)
    ap.add_argument("--prologue", action="store_true")
    ap.add_argument("--stats", action="store_true")
    ap.add_argument("--res", action="store_true", help="add a residual input")
    ap.add_argument("--n", type=int, default=1, help="batch")
    ap.add_argument("--x3", action="store_true", help="fp32 tensors, error-compensated f16 MFMA (with --dtype f32); also reports the error vs the exact fp32 kernel")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = torch.float16 if a.dtype == "f16" else torch.float32
    g = torch.Generator().manual_seed(0)
    for name, (c1, c2, co, h, w, k, ups) in SHAPES.items():
        if a.only and a.only not in name:
            continue
        wt = torch.randn(co, c1 + c2, k, k, generator=g) / math.sqrt((c1 + c2) * k * k)
        bias_h = torch.randn(co, generator=g)
        if ups == "sub":
            up = ops.PackedUpConv(wt, bias_h, dt, dev, c1)
            x = ops.Act(torch.randn((a.n, h, w, c1), device=dev, dtype=dt), c1)
            up(x, want_stats=a.stats)
            torch.cuda.synchronize()
            ts = []
            for _ in range(a.iters):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                up(x, want_stats=a.stats)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            fl = 2.0 * 16 * c1 * co * h * w * a.n     # executed: 16 taps per low-res pixel
            t = sorted(ts)[len(ts) // 2]
            print(f"{name:22s} {a.dtype} sub-pixel st={int(a.stats)}  {t:8.3f} ms  {fl / t / 1e9:8.1f} TFLOP/s executed  ({fl / 1e9:.0f} GFLOP)", flush=True)
            continue
        conv = ops.PackedConv(wt, bias_h, dt, dev, c1, c2, x3=a.x3)
        x = ops.Act(torch.randn((a.n, h, w, c1), device=dev, dtype=dt), c1)
        x2 = ops.Act(torch.randn((a.n, h, w, c2), device=dev, dtype=dt), c2) if c2 else None
        pro = None
        if a.prologue:
            pro = (torch.rand((a.n, c1 + c2), device=dev) + 0.5, torch.randn((a.n, c1 + c2), device=dev) * 0.1)
        ho, wo = (2 * h, 2 * w) if ups else (h, w)
        res = ops.Act(torch.randn((a.n, ho, wo, co), device=dev, dtype=dt), co) if a.res else None
        y = conv(x, x2, upsample=ups, prologue=pro, want_stats=a.stats, residual=res)
        torch.cuda.synchronize()
        if a.x3:
            ref = ops.PackedConv(wt, bias_h, dt, dev, c1, c2, x3=False)(x, x2, upsample=ups, prologue=pro, want_stats=a.stats, residual=res)
            err = (y.t.float() - ref.t.float()).abs().max().item()
            print(f"  x3 vs exact fp32 kernel: max abs diff {err:.3e} (output rms {ref.t.float().pow(2).mean().sqrt().item():.3f})")
            del ref
        ts = []
        for _ in range(a.iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            conv(x, x2, upsample=ups, prologue=pro, want_stats=a.stats, out=y, residual=res)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        fl = 2.0 * k * k * (c1 + c2) * co * ho * wo * a.n
        t = sorted(ts)[len(ts) // 2]
        if os.environ.get("ELVIS_STAMP") and y.stats is not None:
            torch.cuda.synchronize()
            st = y.stats.view(-1, co, 2)[:, :3, :].reshape(-1, 6).double()
            m = st[:, :4].mean(0).tolist(); md = st[:, :4].median(0).values.tolist()
            print(f"  stamps (cycles, mean/median over {st.shape[0]} WGs): prologue {m[0]:.0f}/{md[0]:.0f}  loop {m[1]:.0f}/{md[1]:.0f}  "
                  f"epilogue {m[2]:.0f}/{md[2]:.0f}  store-drain {m[3]:.0f}/{md[3]:.0f}")
        print(f"{name:22s} {a.dtype} pro={int(a.prologue)} st={int(a.stats)} res={int(a.res)}  {t:8.3f} ms  {fl / t / 1e9:8.1f} TFLOP/s  ({fl / 1e9:.0f} GFLOP)", flush=True)

if __name__ == "__main__":
    main()
