#!/bin/bash
# the Delaunay-path cases once per library on the same box: tools/ab_cases.sh "<ops>" <lib.so | lean> ...
OPS=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "$@"; do
  if [ "$v" = lean ]; then unset OFL_LIB; else export OFL_LIB=$ROOT/oflibnumpy_amd/$v; fi
  for op in $OPS; do
    EXTRA=""; [ "$op" = config5 ] && EXTRA="--size 4320 7680"
    PYTHONPATH=$ROOT/tools timeout -k 10 200 python $ROOT/tools/bench_invert.py --op $op $EXTRA --iters 10 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['op'], d['device_ms'])"
  done
done
