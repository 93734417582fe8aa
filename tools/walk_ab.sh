#!/bin/bash
# A/B of the certified-walk kernel's tilings (experiments build): time per call and HBM fetch per variant.  tools/walk_ab.sh <tag>
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/walkab_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT/tools
for LIB in libofl_hip_exp.so libofl_hip_w8.so; do
  [ -f $ROOT/oflibnumpy_amd/$LIB ] || continue
  for T in ${WALK_TILINGS:-0 5 9}; do
    export OFL_LIB=$ROOT/oflibnumpy_amd/$LIB OFL_WALK_TILING=$T
    for OP in invert invert_rot; do
      MS=$(python3 $ROOT/tools/bench_invert.py --op $OP --iters 30 2>/dev/null | grep -o '"device_ms": [0-9.]*' | head -1)
      timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f_${LIB}_${T}_$OP -- python3 $ROOT/tools/bench_invert.py --op $OP --iters 3 > /dev/null 2>&1
      F=$(python3 - "$OUT/f_${LIB}_${T}_$OP" <<'PY'
import csv, glob, sys
v = [float(r["Counter_Value"]) for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "scatter_walk_kernel" in r["Kernel_Name"]]
print("fetch_MB {:.1f}".format(2 * sum(v) / max(1, len(v)) * 1024 / 1e6) if v else "fetch_MB n/a")
PY
)
      echo "$LIB tiling=$T $OP: $MS $F"
    done
  done
done
find $OUT -name "*.db" -delete
