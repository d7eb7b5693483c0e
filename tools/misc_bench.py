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

# window attention at the UNet's top level (6 heads x 32, 320x512, 6 frames)
qkv = ops.Act(torch.randn((6, 320, 512, 576), generator=g).to(dev, torch.float16), 576)
table = (torch.randn((225, 6), generator=g) * 0.5).to(dev)
for shift in (0, 4):
    ops.window_attention(qkv, 6, 32, 8, shift, table, 32 ** -0.5); torch.cuda.synchronize()
    e0.record()
    for _ in range(5): ops.window_attention(qkv, 6, 32, 8, shift, table, 32 ** -0.5)
    e1.record(); torch.cuda.synchronize()
    print(f"attention 6x320x512 shift={shift} ms", e0.elapsed_time(e1) / 5)
