#!/bin/bash
# kernel traces of tools/bench_ops.py, ONE SECTION PER RUN (config2, config3, config4, delaunay, config5, resize) and one row per
# (kernel, grid size), so that every entry of bench.py's "configs" can be recomputed from the rows of its section:
# tools/prof_ops.sh <tag> [sections...]
set -o pipefail
TAG=$1; shift
SECTIONS=${@:-config2 config3 config4 delaunay config5 resize}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/summary.txt
for SEC in $SECTIONS; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$SEC -- python3 $ROOT/tools/bench_ops.py --iters 8 --only $SEC > $OUT/$SEC.log 2>&1 || { echo "trace of $SEC failed"; tail -5 $OUT/$SEC.log; exit 1; }
  echo "#### section $SEC: the JSON lines of this (profiled) run, then its kernels by (kernel, grid)" >> $OUT/summary.txt
  grep -h '^{' $OUT/$SEC.log | cut -c1-400 >> $OUT/summary.txt
  python3 $ROOT/tools/rocprof_summary.py --by-grid $OUT/$SEC | grep -v "rocclr\|^== \|^-- " | head -60 >> $OUT/summary.txt
done
find $OUT -name "*.db" -delete
head -60 $OUT/summary.txt
