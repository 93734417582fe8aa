#!/bin/bash
# kernel breakdown of one band of the slab-wise scatter: tools/prof_slab.sh <tag> [op] [H W]
TAG=$1; OP=${2:-config5}; H=${3:-4320}; W=${4:-7680}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_${TAG} -- python3 $ROOT/tools/slab_check.py --op $OP --size $H $W --world 8 --band 3 --no-check --profile > $ROOT/gpurun_out/prof_${TAG}.log 2>&1
python3 $ROOT/tools/rocprof_summary.py $ROOT/gpurun_out/prof_${TAG} > $ROOT/gpurun_out/prof_${TAG}.txt
find $ROOT/gpurun_out -name "*.db" -delete
cat $ROOT/gpurun_out/prof_${TAG}.log | tail -2
head -50 $ROOT/gpurun_out/prof_${TAG}.txt
