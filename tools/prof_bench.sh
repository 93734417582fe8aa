#!/bin/bash
# Profiles bench.py on the GPU box: kernel trace + stats, then PMC passes in separate runs
# (never together with a trace domain other than kernel-trace).  Usage: tools/prof_bench.sh <tag> [bench args]
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 60 --warmup 6 --no-cpu-baseline $@"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1 || { echo "trace run failed"; tail -5 $OUT/trace.log; exit 1; }
for PASS in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" "TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  NAME=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  timeout -k 10 120 rocprofv3 --pmc $PASS --output-format csv -d $OUT/pmc_$NAME -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_$NAME.log 2>&1 || { echo "pmc pass $PASS failed"; tail -3 $OUT/pmc_$NAME.log; }
done
python3 $ROOT/tools/rocprof_summary.py $OUT/trace $OUT/pmc_* > $OUT/summary.txt 2>&1
grep -h '"metric"' $OUT/trace.log | head -1 > $OUT/bench_line_under_profiler.json
# keep only the summaries (the raw CSVs of 60 dispatches are small, but drop the .db files)
find $OUT -name "*.db" -delete
grep -E "compose3|gather2?_kernel|scatter|stats_kernel|axpy" $OUT/summary.txt | head -60
