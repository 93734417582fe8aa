#!/bin/bash
# A/B of a compose-kernel knob on the experiments build: tools/ab_c3.sh <ENVVAR> <values...>  (one bench.py run per value, same box;
# the value "lean" runs the shipped library instead)
VAR=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/r3
for v in "$@"; do
  if [ "$v" = lean ]; then LIB=""; else LIB="OFL_LIB=$ROOT/oflibnumpy_amd/libofl_hip_exp.so $VAR=$v"; fi
  env $LIB timeout -k 10 200 python $ROOT/bench.py --steps 40 --warmup 10 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$VAR=$v', d['value'], r['frac'], {k: p['frac'] for k, p in r.get('other_patterns', {}).items()})"
done
