#!/bin/bash
# Profiles bench.py on the GPU box, ONE SAMPLING PATTERN PER RUN (scale = the BASELINE config, shift, rot), so that every
# row of the summary is one kernel on one pattern and its average duration reproduces the `roofline.frac` of that run's
# JSON line: a rocprofv3 kernel trace + stats per pattern, then FETCH_SIZE and WRITE_SIZE in separate --pmc passes per
# pattern (never together with a trace domain other than kernel-trace), then the instruction / cache passes on the
# headline pattern only.  No pre-warm and no secondary patterns inside the runs: every compose3 dispatch of a run belongs
# to its pattern.   Usage: tools/prof_bench.sh <tag> [extra bench args]
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 60 --warmup 6 --no-cpu-baseline --no-secondary --no-configs --no-strong --min-timed-ms 0 --prewarm-ms 0 $@"
for PAT in scale shift rot; do
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$PAT/trace -- python3 $ROOT/bench.py $ARGS --pattern $PAT > $OUT/$PAT.trace.log 2>&1 || { echo "trace run ($PAT) failed"; tail -5 $OUT/$PAT.trace.log; exit 1; }
  grep -h '"metric"' $OUT/$PAT.trace.log | head -1 > $OUT/$PAT.bench_line.json
  for PASS in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $PASS --output-format csv -d $OUT/$PAT/pmc_$PASS -- python3 $ROOT/bench.py $ARGS --pattern $PAT > $OUT/$PAT.pmc_$PASS.log 2>&1 || { echo "pmc pass $PASS ($PAT) failed"; tail -3 $OUT/$PAT.pmc_$PASS.log; }
  done
done
for PASS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" "TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  NAME=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $PASS --output-format csv -d $OUT/scale/pmc_$NAME -- python3 $ROOT/bench.py $ARGS --pattern scale > $OUT/scale.pmc_$NAME.log 2>&1 || { echo "pmc pass $PASS failed"; tail -3 $OUT/scale.pmc_$NAME.log; }
done
find $OUT -name "*.db" -delete
python3 $ROOT/tools/prof_bench_summary.py $OUT > $OUT/summary.txt 2> $OUT/summary.err
cat $OUT/summary.txt | head -40
