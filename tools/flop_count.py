#!/usr/bin/env python3
"""Whole-model FLOP total of the SinSR 4x path at a given LR size, counted by torch's FlopCounterMode on the CPU
oracle (oracle/sinsr_ref.py) - SURVEY.md 8(d).  Runs on meta tensors (no arithmetic), so 1080p takes seconds.
    python tools/flop_count.py [--h 270 --w 480] [--out profiles/r02_flops.json]
Test tooling: imports oracle/, never imported by the product."""
import argparse, dataclasses, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils.flop_counter import FlopCounterMode
from elvis_amd.weights import SinSRConfig, make_sinsr_weights
from oracle import sinsr_ref as R

ap = argparse.ArgumentParser()
ap.add_argument("--h", type=int, default=270)
ap.add_argument("--w", type=int, default=480)
ap.add_argument("--out", default=None)
a = ap.parse_args()
cfg = dataclasses.replace(SinSRConfig(), quantize=False)     # the VQ lookup is a nearest-code search, not a matmul
sd = make_sinsr_weights(cfg, 0)
dev = "meta"
try:
    sd_m = {k: v.to(dev) for k, v in sd.items()}
    lr = torch.zeros((a.h, a.w, 3), dtype=torch.uint8, device=dev)
    hp, wp = R.padded_latent_shape(cfg, a.h, a.w)
    noise = torch.zeros((1, cfg.latent_ch, hp, wp), device=dev)
    with torch.device(dev), FlopCounterMode(display=False) as fc:   # tensors the oracle creates land on meta too
        R.sinsr_forward(sd_m, cfg, lr, noise)
    how = "meta tensors"
except Exception as exc:   # an op without a meta kernel: count on a real (small) tile and scale is NOT done silently
    raise SystemExit(f"FlopCounterMode on meta tensors failed ({type(exc).__name__}: {exc}); run with a small --h/--w on real tensors instead")
total = fc.get_total_flops()
by_op = {str(k): int(v) for k, v in sorted(fc.get_flop_counts()["Global"].items(), key=lambda kv: -kv[1])}
rec = {"lr_h": a.h, "lr_w": a.w, "out_h": 4 * a.h, "out_w": 4 * a.w, "counted_on": how, "total_flops": int(total),
       "tflop_per_frame": total / 1e12, "by_op": by_op,
       "note": "FlopCounterMode counts matmul/conv/attention FLOPs (2 per MAC) of the oracle's forward; the bench's "
               "tflop_per_frame sums 2*k*k*Cin*Cout*Ho*Wo over the device's conv launches (window attention excluded)"}
print(json.dumps(rec, indent=1))
if a.out:
    json.dump(rec, open(a.out, "w"), indent=1)
