import time, torch, sys
sys.path.insert(0,'.')
from elvis_amd import ops
dev=torch.device("cuda:0")
n=15
a=(torch.rand(n,1080,1920,3,device=dev)*255).to(torch.uint8); b=(torch.rand(n,1080,1920,3,device=dev)*255).to(torch.uint8)
m=torch.randint(0,4,(n,135,240),device=dev,dtype=torch.int32)
out=torch.empty_like(a)
for blk,mm in ((8,m),):
    ops.recompose_u8(a,b,mm,blk,0,out=out); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.recompose_u8(a,b,mm,blk,0,out=out)
    e1.record(); torch.cuda.synchronize()
    t=e0.elapsed_time(e1)/10
    alg=n*(3*1080*1920*3+135*240*4)
    print(f"recompose block {blk}: {t*1e3:.1f} us per {n} frames, algorithmic {alg/t/1e6:.0f} GB/s = {alg/t/1e6/8000:.3f} of 8 TB/s")
