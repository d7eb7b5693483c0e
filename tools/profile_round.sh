#!/bin/bash
# Profiles of one round, run ON THE GPU BOX (gpurun): rocprofv3 kernel statistics of the default bench, of the in-tolerance
# precision modes and of the two other slots, and PMC passes (separate runs, never combined with a trace domain other than
# the kernel trace): HBM traffic (FETCH_SIZE / WRITE_SIZE) and the matrix-core / LDS counters of the SQ block.
#   bash tools/profile_round.sh r03      -> gpurun_out/prof_r03/...   (tools/profile_collect.py copies the summaries into profiles/)
set -u
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
echo "== kernel stats (default bench)"; rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o p -- $BENCH > $OUT/bench_under_stats.json 2> $OUT/stats.log
echo "== FETCH_SIZE";  rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/fetch -o p -- $BENCH > /dev/null 2> $OUT/fetch.log
echo "== WRITE_SIZE";  rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/write -o p -- $BENCH > /dev/null 2> $OUT/write.log
echo "== SQ counters"; rocprofv3 --kernel-trace --pmc $SQ -f csv -d $OUT/sq -o p -- $BENCH > /dev/null 2> $OUT/sq.log
echo "== precision modes"
for m in x3 dec_f16; do
  rocprofv3 --kernel-trace --stats -f csv -d $OUT/mode_$m -o p -- python3 tools/mode_profile.py $m > $OUT/mode_$m.txt 2> $OUT/mode_$m.log
  rocprofv3 --kernel-trace --pmc $SQ -f csv -d $OUT/mode_${m}_sq -o p -- python3 tools/mode_profile.py $m > /dev/null 2> $OUT/mode_${m}_sq.log
done
echo "== slots"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/dct -o p -- python3 bench.py --slot dct --frames 12 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/slot_dct.json 2> $OUT/dct.log
rocprofv3 --kernel-trace --stats -f csv -d $OUT/blur -o p -- python3 bench.py --slot blur --frames 12 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/slot_blur.json 2> $OUT/blur.log
find $OUT -name "*.csv" | head -60
for f in stats fetch write sq dct blur; do tail -2 $OUT/$f.log; done
