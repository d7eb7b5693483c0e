#!/bin/bash
# Same-box A/B of two builds of libelvis_amd.so on the hot conv shapes and the headline step (run on the GPU box):
#   tools/ab_libs.sh NAME [OUTDIR]      NAME = elvis_amd/lib/variants/NAME.so (tools/build_variant.py)
# Order A B A B so that clock / temperature drift shows up as a difference between equal arms.
set -e
V=elvis_amd/lib/variants/$1.so
O=${2:-gpurun_out/ab_$1}
mkdir -p $O
for r in 1 2; do
  python tools/conv_bench.py --n 15 --prologue --stats --res --only dec > $O/base_res_$r.txt 2>&1
  ELVIS_AMD_LIB=$V python tools/conv_bench.py --n 15 --prologue --stats --res --only dec > $O/var_res_$r.txt 2>&1
  python tools/conv_bench.py --n 15 --prologue --stats --only dec > $O/base_nores_$r.txt 2>&1
  ELVIS_AMD_LIB=$V python tools/conv_bench.py --n 15 --prologue --stats --only dec > $O/var_nores_$r.txt 2>&1
done
for r in 1 2; do
  python bench.py --steps 3 --warmup 1 --no-extras --no-slots --no-cpu-baseline > $O/bench_base_$r.json 2> $O/bench_base_$r.err
  ELVIS_AMD_LIB=$V python bench.py --steps 3 --warmup 1 --no-extras --no-slots --no-cpu-baseline > $O/bench_var_$r.json 2> $O/bench_var_$r.err
done
grep -h "dec128\|dec256_\|dec512" $O/base_res_*.txt $O/var_res_*.txt $O/base_nores_*.txt $O/var_nores_*.txt
python - <<P
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["value"],3), round(d["roofline"]["frac"],4))
P
