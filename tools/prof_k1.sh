#!/bin/bash
# kernel trace + PMC passes of tools/bench_k1.py: tools/prof_k1.sh <tag> <bench_k1 args>
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT/tools
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/bench_k1.py "$@" > $OUT/trace.log 2>&1
for PASS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TD_TD_BUSY TD_TC_STALL TA_TA_BUSY TA_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  NAME=$(echo $PASS | tr ' ' '_' | cut -c1-30)
  timeout -k 10 200 rocprofv3 --pmc $PASS --output-format csv -d $OUT/pmc_$NAME -- python3 $ROOT/tools/bench_k1.py "$@" > $OUT/pmc_$NAME.log 2>&1 || { echo "pmc pass $PASS failed"; tail -3 $OUT/pmc_$NAME.log; }
done
python3 $ROOT/tools/rocprof_summary.py $OUT/trace $OUT/pmc_* > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete
grep gather2 $OUT/summary.txt
