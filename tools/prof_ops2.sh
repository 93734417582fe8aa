#!/bin/bash
# kernel trace of a single ops-bench selection: tools/prof_ops2.sh <tag> <bench_ops args>
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/bench_ops.py "$@" > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
python3 $ROOT/tools/rocprof_summary.py $OUT/trace > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete
head -40 $OUT/summary.txt
