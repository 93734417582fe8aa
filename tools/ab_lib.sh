#!/bin/bash
# bench.py once per library on the same box: tools/ab_lib.sh <lib.so | lean> ...   (optional NAME=VALUE arguments are exported first)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "$@"; do
  case "$v" in *=*) export "$v"; continue;; esac
  if [ "$v" = lean ]; then unset OFL_LIB; else export OFL_LIB=$ROOT/oflibnumpy_amd/$v; fi
  timeout -k 10 200 python $ROOT/bench.py --steps 40 --warmup 10 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', d['value'], r['frac'], {k: p['frac'] for k, p in r.get('other_patterns', {}).items()})"
done
