#!/bin/bash
# DCT slot with and without the weight-stationary conv kernel (conv_ws.inc), and its stagger policies; run on the GPU box.
python -m pytest tests/test_gpu_kernels.py -x -q -k "weight_stationary" 2>&1 | tail -1
ELVIS_WS_STAG=1 python -m pytest tests/test_gpu_kernels.py -x -q -k "weight_stationary" 2>&1 | tail -1
python -m pytest tests/test_gpu_restorers.py tests/test_gpu_surfaces.py -x -q 2>&1 | tail -1
run() {
  python bench.py --slot dct --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1));
ks=[d['roofline']]+d.get('other_kernels',[])
[print('    ',k['kernel'], round(k['frac'],3), 'ms', round(k['avg_launch_ms'],3), 'share', round(k['time_share'],3)) for k in ks]"
}
for r in 1 2; do
  ELVIS_NO_WS=1 run "no_ws      "
  ELVIS_WS_STAG=0 run "ws lockstep"
  ELVIS_WS_STAG=1 run "ws stagger "
  run "ws default "
done
