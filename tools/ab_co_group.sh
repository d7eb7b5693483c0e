set -e
O=gpurun_out/ab_cg; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -x -q -k "conv" > $O/tests.log 2>&1; tail -2 $O/tests.log
for r in 1 2; do
 for g in 0 2 1; do
  ELVIS_CO_GROUP=$g python tools/conv_bench.py --n 15 --prologue --stats --res --only 512 > $O/f16_g${g}_$r.txt 2>&1
  ELVIS_CO_GROUP=$g python tools/conv_bench.py --n 15 --prologue --stats --only unet > $O/unet_g${g}_$r.txt 2>&1
 done
done
for g in 0 2; do ELVIS_CO_GROUP=$g python tools/conv_bench.py --n 6 --dtype f32 --x3 --prologue --stats --res --only dec512 > $O/x3_g${g}.txt 2>&1; done
grep -h "TFLOP" $O/f16_g0_*.txt | sed 's/^/g0 /'; grep -h "TFLOP" $O/f16_g2_*.txt | sed 's/^/g2 /'; grep -h "TFLOP" $O/f16_g1_*.txt | sed 's/^/g1 /'
grep -h "TFLOP" $O/unet_g0_1.txt | sed 's/^/g0 /'; grep -h "TFLOP" $O/unet_g2_1.txt | sed 's/^/g2 /'
grep -h "TFLOP" $O/x3_g0.txt | sed 's/^/x3 g0 /'; grep -h "TFLOP" $O/x3_g2.txt | sed 's/^/x3 g2 /'
export TMPDIR=/tmp
for g in 0 2; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/fetch_g$g -o p -- python3 tools/conv_bench.py --n 15 --prologue --stats --res --only dec512_270p --iters 2 > /dev/null 2> $O/fetch_g$g.log
  python3 - $O/fetch_g$g $g <<'P'
import csv,glob,sys,os
v=[float(r["Counter_Value"]) for f in glob.glob(os.path.join(sys.argv[1],"**","*counter_collection.csv"),recursive=True) for r in csv.DictReader(open(f)) if r["Counter_Name"]=="FETCH_SIZE" and "conv3x3_halo" in r["Kernel_Name"]]
print("co_group",sys.argv[2],"dec512 res FETCH (x2) GB per launch:", [round(2*1024*x/1e9,2) for x in v])
P
done
for r in 1 2; do for g in 0 2; do ELVIS_CO_GROUP=$g python bench.py --steps 3 --warmup 1 --no-extras --no-slots --no-cpu-baseline > $O/bench_g${g}_$r.json 2> $O/bench_g${g}_$r.err; done; done
python - <<P
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["value"],3), round(d["roofline"]["frac"],4))
P
