python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sinsr.py -m gpu -x -q 2>&1 | tail -4
for r in 1 2; do for sh in dec128_1080p dec256_540p dec512_270p; do for f in "--prologue --stats --res" "--prologue --stats"; do
python tools/conv_bench.py --only $sh $f --n 15 --iters 5 2>&1 | grep -v amdgpu.ids | sed "s/^/asm   /"
ELVIS_AMD_LIB=elvis_amd/lib/variants/noasm.so python tools/conv_bench.py --only $sh $f --n 15 --iters 5 2>&1 | grep -v amdgpu.ids | sed "s/^/noasm /"; done; done; done
