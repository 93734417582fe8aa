#!/bin/bash
# PMC passes over tools/bench_ops.py --only config2 (the K1 kernels: float and 8-bit, alone and batched): tools/prof_k1_pmc.sh <tag>
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT/tools
for PASS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "TA_BUSY_avr TA_BUSY_max TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum"; do
  NAME=$(echo $PASS | tr ' ' '_' | cut -c1-30)
  timeout -k 10 200 rocprofv3 --pmc $PASS --output-format csv -d $OUT/pmc_$NAME -- python3 $ROOT/tools/bench_ops.py --only config2 --iters 4 > $OUT/pmc_$NAME.log 2>&1 || { echo "pmc pass $PASS failed"; tail -3 $OUT/pmc_$NAME.log; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if "gather2" not in k: continue
        acc[(k, r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, grid), c in sorted(acc.items()):
    print("==", k, "grid", grid)
    for name, v in sorted(c.items()):
        print("   {:<40} {:>16.1f}  ({} calls)".format(name, sum(v) / len(v), len(v)))
PY
find $OUT -name "*.db" -delete
