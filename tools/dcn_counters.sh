#!/bin/bash
# SQ / TCP counters of the DCT slot's kernels (dcnv2_tile_kernel first), ON THE GPU BOX.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/dcn_pmc; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -f csv -d $OUT/sq -o p -- python3 bench.py --slot dct --frames 6 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/sq.log
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr -f csv -d $OUT/tcp -o p -- python3 bench.py --slot dct --frames 6 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/tcp.log
python3 - <<'PY'
import csv, collections, glob
for sub in ("sq", "tcp"):
    f = glob.glob(f"gpurun_out/dcn_pmc/{sub}/**/*counter_collection.csv", recursive=True)
    if not f: print("no csv for", sub); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:60]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, c in agg.items():
        if "dcnv2" in k or "halo_kernelIDF16_Li32" in k: print(sub, k, {a: int(b) for a, b in c.items()})
PY
tail -2 $OUT/sq.log $OUT/tcp.log
