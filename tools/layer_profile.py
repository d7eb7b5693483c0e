"""Per-shape conv time table for one SinSR invocation (HIP events per launch).
   python tools/layer_profile.py [--batch 6] [--h 1080 --w 1920]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from elvis_amd import ops
from elvis_amd.restore import get_sinsr_model, sr4x_device

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=6)
ap.add_argument("--h", type=int, default=1080)
ap.add_argument("--w", type=int, default=1920)
ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()
dev = torch.device("cuda:0")
model = get_sinsr_model(dev)
lr = (torch.rand(a.batch, a.h // 4, a.w // 4, 3, device=dev) * 255).to(torch.uint8)
run = lambda: sr4x_device(model, lr, list(range(a.batch)), batch=a.batch)
run(); torch.cuda.synchronize()
agg = {}
order = []
for _ in range(a.reps):
    prof, shapes = [], []
    ops.CONV_PROFILER, ops.CONV_SHAPES = prof, shapes
    run(); torch.cuda.synchronize()
    ops.CONV_PROFILER = ops.CONV_SHAPES = None
    for (name, fl, e0, e1), sh in zip(prof, shapes):
        k = (name.replace("conv3x3_halo_kernel", "halo").replace("conv_igemm_kernel", "igemm"), sh)
        if k not in agg:
            agg[k] = [0.0, 0.0, 0]; order.append(k)
        agg[k][0] += fl; agg[k][1] += e0.elapsed_time(e1); agg[k][2] += 1
tot = sum(v[1] for v in agg.values())
print(f"total conv ms per invocation {tot / a.reps:.2f}  ({tot / a.reps / a.batch:.2f} ms/frame)")
for k in sorted(agg, key=lambda k: -agg[k][1]):
    fl, ms, cnt = agg[k]
    print(f"{k[0]:34s} n{k[1][0]} {k[1][1]:4d}x{k[1][2]:<4d} {k[1][3]:4d}->{k[1][4]:<4d} k{k[1][5]} s{k[1][6]} pro={int(k[1][7])} res={int(k[1][8])}"
          f"  x{cnt // a.reps:<3d} {ms / a.reps:8.2f} ms {100 * ms / tot:5.1f}%  {fl / ms / 1e9:7.1f} TF")
