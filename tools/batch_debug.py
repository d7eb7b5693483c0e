"""Debug aid: is the network's result independent of the batch size?  Records every UNet module's
output for a 2-frame invocation and for the same frames run one at a time, reports the first mismatch."""
import dataclasses, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from elvis_amd import restore, synth
from elvis_amd.weights import SinSRConfig

dev = torch.device("cuda:0")
h, w = int(sys.argv[1]), int(sys.argv[2])
cfg = dataclasses.replace(SinSRConfig(), quantize=False)
model = restore.get_sinsr_model(dev, cfg=cfg)
rec = []


def wrap(lst):
    for j, (kind, m) in enumerate(lst):
        def f(*a, _m=m, _k=kind, **kw):
            out = _m(*a, **kw)
            rec.append((_k, type(_m).__name__, out.t.float().clone()))
            return out
        lst[j] = (kind, f)


for mods in model.u_input: wrap(mods)
wrap(model.u_middle)
for mods in model.u_output: wrap(mods)
lr = torch.from_numpy(synth.synth_clip(7, 2, h, w)).to(dev)
noise = model.make_noise(42, [0, 1], h, w)
model.forward(lr, noise)
r2 = list(rec); rec.clear()
model.forward(lr[:1].contiguous(), noise[:1].contiguous())
r1 = list(rec)
for i, ((k, nm, a), (_, _, b)) in enumerate(zip(r1, r2)):
    d = float((a[0] - b[0]).abs().max())
    print(i, k, nm, tuple(a.shape), "max diff", d)
    if d > 0 and i > 3:
        break
