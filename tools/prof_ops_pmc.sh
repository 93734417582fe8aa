#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the non-headline kernels: tools/prof_ops_pmc.sh <tag>
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for PASS in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -k 10 150 rocprofv3 --pmc $PASS --output-format csv -d $OUT/pmc_$PASS -- python3 $ROOT/tools/bench_ops.py --iters 4 > $OUT/pmc_$PASS.log 2>&1 || { echo "pass $PASS failed"; tail -3 $OUT/pmc_$PASS.log; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        grid = r.get("Grid_Size", "")
        acc[(k, grid)][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("{:<34} {:>10} {:>6} {:>14} {:>14} {:>16}".format("kernel", "grid", "calls", "FETCH_KiB", "WRITE_KiB", "HBM_MB(2F+W)"))
for (k, grid), c in sorted(acc.items()):
    f = sum(c.get("FETCH_SIZE", [0])) / max(1, len(c.get("FETCH_SIZE", [1])))
    w = sum(c.get("WRITE_SIZE", [0])) / max(1, len(c.get("WRITE_SIZE", [1])))
    n = max(len(c.get("FETCH_SIZE", [])), len(c.get("WRITE_SIZE", [])))
    print("{:<34} {:>10} {:>6} {:>14.1f} {:>14.1f} {:>16.1f}".format(k[-34:], grid, n, f, w, (2 * f + w) * 1024 / 1e6))
PY
