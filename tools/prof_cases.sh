#!/bin/bash
# per-case kernel breakdown of the Delaunay path: tools/prof_cases.sh <tag> [ops...]
TAG=$1; shift
OPS=${@:-speckle hole stripes}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for op in $OPS; do
  EXTRA=""; [ "$op" = config5 ] && EXTRA="--size 4320 7680"      # BASELINE config 5: the tiled Sintel field at 8K
  OFL_LIB=$ROOT/oflibnumpy_amd/libofl_hip_exp.so OFL_DL_DEBUG=1 PYTHONPATH=$ROOT/tools python3 $ROOT/tools/bench_invert.py --op $op $EXTRA --iters 5 2>&1 | tail -4 >> $ROOT/gpurun_out/${TAG}_time.log
  PYTHONPATH=$ROOT/tools timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_${TAG}_$op -- python3 $ROOT/tools/bench_invert.py --op $op $EXTRA --iters 5 > /dev/null 2>&1
  python3 $ROOT/tools/rocprof_summary.py $ROOT/gpurun_out/prof_${TAG}_$op > $ROOT/gpurun_out/prof_${TAG}_$op.txt
done
find $ROOT/gpurun_out -name "*.db" -delete
cat $ROOT/gpurun_out/${TAG}_time.log
