for sh in dec128_1080p dec512_270p; do
for f in "--prologue --stats --res" "--prologue --stats" "--stats" ""; do python tools/conv_bench.py --only $sh $f --n 15 --iters 5 2>&1 | grep -v amdgpu.ids; done
ELVIS_STAMP=1 ELVIS_AMD_LIB=elvis_amd/lib/variants/stamp.so python tools/conv_bench.py --only $sh --prologue --stats --res --n 15 --iters 3 2>&1 | grep -v amdgpu.ids
ELVIS_STAMP=1 ELVIS_AMD_LIB=elvis_amd/lib/variants/stamp.so python tools/conv_bench.py --only $sh --prologue --stats --n 15 --iters 3 2>&1 | grep -v amdgpu.ids
done
