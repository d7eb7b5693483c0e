#!/usr/bin/env python3
"""Where does dcnv2_tile_kernel's time go?  Times elvis_dcnv2 at 1080p (6 colour planes = 2 frames) on the DCT slot's own
offset / mask tensor, and on copies with the offsets scaled down, and prints how often a sample leaves the kernel's LDS
window (8-pixel halo): a sample outside it takes the kernel's bounds-checked global-read path for the whole wave.
    python tools/dcn_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elvis_amd import ops
from elvis_amd.restorers import DCNRestorer, RELU

def main():
    dev = torch.device("cuda:0")
    m = DCNRestorer(device=dev)
    cfg = m.cfg
    g = torch.Generator().manual_seed(0)
    frames = torch.randint(0, 256, (4, 1080, 1920, 3), dtype=torch.uint8, generator=g).to(dev)
    planes = ops.temporal_stack(frames, 1, 2, cfg.t // 2, torch.float16)
    c1 = m.c1(planes, act=RELU); d1 = m.d1(c1, stride=2, act=RELU); d2 = m.d2(d1, act=RELU)
    u1 = m.u1(d2, act=RELU); f = m.f(u1, c1, act=RELU); om = m.om(f)
    G = cfg.t
    off = om.t[..., :18 * G].float()
    print(f"offsets: mean {off.mean().item():+.3f} std {off.std().item():.3f} max|.| {off.abs().max().item():.2f}  "
          f"frac |d| > 6: {(off.abs() > 6).float().mean().item():.2e}  > 7: {(off.abs() > 7).float().mean().item():.2e}")
    per_px = (off.abs() > 6.5).any(dim=-1).float().mean().item()
    print(f"pixels with at least one sample near / outside the window edge: {per_px:.3f}")
    def timeit(omx, label):
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.dcnv2(planes, omx, m.dcn_w, m.dcn_b, G, cfg.feat, mask_sigmoid=True, act=RELU)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = sorted(ts)[2]
        by = (27 * G + planes.c + cfg.feat) * 2 * planes.n * planes.h * planes.w
        print(f"{label:28s} {t:7.3f} ms  {by / t / 1e6:7.1f} GB/s algorithmic")
    timeit(om, "om as computed")
    for s in (0.5, 0.1, 0.0):
        o2 = ops.Act(om.t.clone(), om.c)
        o2.t[..., :18 * G] *= s
        timeit(o2, f"offsets x {s}")
    o3 = ops.Act(om.t.clone(), om.c)
    o3.t[..., :18 * G] = torch.randn_like(o3.t[..., :18 * G]) * 3.0
    timeit(o3, "offsets ~ N(0, 3^2)")

if __name__ == "__main__":
    main()
