import sys, torch
sys.path.insert(0, "/root/repo")
from elvis_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
z = ops.Act(torch.randn((6, 270, 480, 8), generator=g).to(dev, torch.float16), 3)
cb = torch.randn((8192, 3), generator=g).to(dev)
ops.vq_nearest(z, cb); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): ops.vq_nearest(z, cb)
e1.record(); torch.cuda.synchronize()
print("vq 6x270x480 ms", e0.elapsed_time(e1) / 5)
