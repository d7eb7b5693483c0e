python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -4
for r in 1 2; do for v in 1 0; do for sh in dec128_1080p dec256_540p dec512_270p; do for f in "--prologue --stats --res" "--prologue --stats"; do
ELVIS_PERSIST=$v python tools/conv_bench.py --only $sh $f --n 15 --iters 5 2>&1 | grep -v amdgpu.ids | sed "s/^/persist=$v /"; done; done; done; done
