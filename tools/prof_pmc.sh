#!/bin/bash
# tools/prof_pmc.sh <tag> "<counters pass 1>" "<counters pass 2>" ... -- [bench args]
set -o pipefail
TAG=$1; shift
PASSES=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do PASSES+=("$1"); shift; done
[ "$1" == "--" ] && shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 40 --warmup 5 --no-cpu-baseline --no-configs --no-strong --min-timed-ms 0 $@"
i=0
for PASS in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $PASS --output-format csv -d $OUT/pmc_$i -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_$i.log 2>&1 || { echo "pmc pass $PASS failed"; tail -3 $OUT/pmc_$i.log; }
done
python3 $ROOT/tools/rocprof_summary.py $OUT/pmc_* > $OUT/summary_pmc.txt 2>&1
find $OUT -name "*.db" -delete
grep -E "compose3|gather_kernel|scatter|stats_kernel" $OUT/summary_pmc.txt | head -60
