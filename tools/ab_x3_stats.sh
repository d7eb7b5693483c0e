python -m pytest tests/test_gpu_kernels.py -x -q -k "compensated or x3" 2>&1 | tail -1
for r in 1 2; do
  python tools/conv_bench.py --dtype f32 --x3 --n 6 --prologue --stats --res --only dec 2>&1 | grep TFLOP | sed 's/^/new /' | head -3
  ELVIS_AMD_LIB=$PWD/elvis_amd/lib/variants/x3dpp.so python tools/conv_bench.py --dtype f32 --x3 --n 6 --prologue --stats --res --only dec 2>&1 | grep TFLOP | sed 's/^/dpp /' | head -3
done
for r in 1 2; do
  python tools/mode_profile.py x3 2>&1 | tail -1 | sed 's/^/new /'
  ELVIS_AMD_LIB=$PWD/elvis_amd/lib/variants/x3dpp.so python tools/mode_profile.py x3 2>&1 | tail -1 | sed 's/^/dpp /'
done
