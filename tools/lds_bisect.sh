#!/bin/bash
# LDS bank-conflict bisect of one conv shape, run ON THE GPU BOX: the product library and the timing-only variants of
# tools/variants (no weight staging / no halo staging / no staging at all / no stores), one rocprofv3 --pmc pass each.
#   bash tools/lds_bisect.sh "--only sub_up256 --n 6 --stats" tag
set -u
ARGS=${1:---only sub_up256 --n 6}
TAG=${2:-sub}
OUT=gpurun_out/lds_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
PMC="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
for v in base now noh nostage nostore; do
  lib=elvis_amd/lib/variants/$v.so
  [ $v = base ] && lib=elvis_amd/lib/libelvis_amd.so
  [ -f $lib ] || { echo "missing $lib"; continue; }
  ELVIS_AMD_LIB=$lib rocprofv3 --kernel-trace --pmc $PMC -f csv -d $OUT/$v -o p -- python3 tools/conv_bench.py $ARGS --iters 2 > $OUT/$v.txt 2> $OUT/$v.log || { echo "$v failed"; tail -3 $OUT/$v.log; }
  f=$(find $OUT/$v -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 tools/pmc_counters.py $f $OUT/$v.json | head -4
done
