#!/usr/bin/env python3
"""Timing of the DCT-slot and Blur-slot client paths at 1080p (BASELINE configs 3 and 4): frames
resident in HBM -> restored frames in HBM, including the block-map recompose.
    python tools/slot_bench.py [--frames 8] [--mode f16|f32]"""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elvis_amd import ops, synth
from elvis_amd.recompose import rounds_recompose_device
from elvis_amd.restorers import DCNRestorer, SwinDeblur

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--mode", default="f16")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = torch.float16 if a.mode == "f16" else torch.float32
    F, H, W, B = a.frames, 1080, 1920, 8
    clip = synth.synth_clip(synth.CLIP_SEED, min(F, 4), H, W)
    clip = np.concatenate([clip] * ((F + 3) // 4))[:F]
    frames_d = torch.from_numpy(clip).to(dev)
    lv = synth.synth_level_maps(synth.MAP_SEED, F, H // B, W // B)
    dct_map = torch.from_numpy(lv.astype(np.int32)).to(dev)
    blur_map = torch.from_numpy(np.minimum(lv, 1).astype(np.int32)).to(dev)      # rounds = 1 mapping

    dcn = DCNRestorer(device=dev, dtype=dt)
    def dct_step():
        return ops.recompose_u8(frames_d, dcn.restore(frames_d, chunk=2), dct_map, B, 0)
    deb = SwinDeblur(device=dev, dtype=dt)
    def blur_step():
        return rounds_recompose_device(frames_d, blur_map, B, lambda d: deb.restore(d, swap_rb=True), batch_size=2)
    for name, fn in (("ELVIS v2 DCT (DCNv2 restorer)", dct_step), ("ELVIS v2 Blur (Swin deblur, 1 round)", blur_step)):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt_s = time.perf_counter() - t0
        print(f"{name:40s} {a.mode}  {F} x 1080p frames  {dt_s*1e3:8.1f} ms  {F/dt_s:7.2f} frames/s", flush=True)

if __name__ == "__main__":
    main()
