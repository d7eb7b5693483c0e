#!/bin/bash
# LDS bank-conflict share of given conv_bench shapes with the product library (one rocprofv3 --pmc pass each), ON THE GPU BOX.
#   bash tools/lds_check.sh tag "--only sub_up256 --n 6 --stats" ["--only unet320 ..."]
set -u
TAG=$1; shift
export TMPDIR=/tmp
PMC="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"
i=0
for ARGS in "$@"; do
  OUT=gpurun_out/ldschk_${TAG}_$i; mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc $PMC -f csv -d $OUT -o p -- python3 tools/conv_bench.py $ARGS --iters 2 > $OUT/bench.txt 2> $OUT/log.txt || tail -3 $OUT/log.txt
  f=$(find $OUT -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 tools/pmc_counters.py $f $OUT/counters.json | head -2
  cat $OUT/bench.txt
  i=$((i+1))
done
