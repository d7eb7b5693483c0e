#!/bin/bash
# HBM read traffic (rocprofv3 --pmc FETCH_SIZE, doubled as the guide prescribes for gfx950) of the hot conv shapes, one
# shape per profiled process, against the algorithmic read bytes (input + residual, 15 frames).  Run on the GPU box:
#   bash tools/conv_traffic.sh [OUTDIR]
export TMPDIR=/tmp
O=${1:-gpurun_out/conv_traffic}
mkdir -p $O
for shape in dec128_1080p dec256_540p dec512_270p; do
  for res in 1 0; do
    R=""; [ $res = 1 ] && R="--res"
    for ctr in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --kernel-trace --pmc $ctr -f csv -d $O/${shape}_r${res}_$ctr -o p -- python3 tools/conv_bench.py --n 15 --prologue --stats $R --only $shape --iters 2 > $O/${shape}_r${res}_$ctr.txt 2> $O/${shape}_r${res}_$ctr.log
    done
  done
done
python3 - "$O" <<'P'
import csv, glob, sys, os
O = sys.argv[1]
alg = {"dec128_1080p": 1080*1920*128*2*15, "dec256_540p": 540*960*256*2*15, "dec512_270p": 270*480*512*2*15}
for shape in alg:
    for res in (1, 0):
        row = [shape, f"res={res}"]
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            fs = glob.glob(os.path.join(O, f"{shape}_r{res}_{ctr}", "**", "*counter_collection.csv"), recursive=True)
            vals = []
            for f in fs:
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] == ctr and "conv3x3_halo" in r["Kernel_Name"]:
                        vals.append(float(r["Counter_Value"]))
            if not vals:
                row.append(f"{ctr}: none"); continue
            b = 1024.0 * sorted(vals)[len(vals)//2] * (2.0 if ctr == "FETCH_SIZE" else 1.0)
            a = alg[shape] * ((1 + res) if ctr == "FETCH_SIZE" else 1)
            row.append(f"{ctr}: {b/1e9:.2f} GB per launch = {b/a:.2f} x algorithmic ({a/1e9:.2f} GB)")
        print("  ".join(row))
P
