python -m pytest tests/test_gpu_degrade.py tests/test_cabi.py -x -q 2>&1 | tail -5
python - <<'PY'
import time, numpy as np, torch, sys
sys.path.insert(0, '.')
from elvis_amd import degrade, synth
dev = torch.device("cuda:0")
clip = synth.synth_clip(7, 2, 1080, 1920)
fd = torch.from_numpy(np.concatenate([clip] * 15)).to(dev)
lv = torch.from_numpy(np.concatenate([synth.synth_level_maps(8, 2, 135, 240).astype(np.int32)] * 15)).to(dev)
for name, fn in (("downsample", lambda: degrade.degrade_downsample_device(fd, lv, 8)), ("gaussian", lambda: degrade.degrade_gaussian_device(fd, lv, 8)), ("dct", lambda: degrade.degrade_dct_device(fd, lv))):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name:12s} 30 x 1080p: {dt*1e3:7.2f} ms  {30/dt:8.1f} frames/s  {2*fd.numel()/dt/1e9:7.1f} GB/s (read+write)")
PY
