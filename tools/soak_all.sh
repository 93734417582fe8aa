#!/bin/bash
# Every soak for SECONDS each (default 120) with one seed base (default: the day of the year): tools/soak_all.sh [seconds] [seed]
# GPU box: all of them; without a GPU only the CPU soak of the geometry core runs.
SECS=${1:-120}; SEED=${2:-$(date +%j)}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/soak; mkdir -p $OUT
cd $ROOT
python tools/soak_core_cpu.py --seconds $SECS --seed $SEED > $OUT/core_cpu.json 2> $OUT/core_cpu.err
if python -c "import oflibnumpy_amd as of; of.native.ensure_device()" 2>/dev/null; then
  python tools/soak_scatter.py --seconds $SECS --seed $SEED > $OUT/scatter_grid.json 2> $OUT/scatter_grid.err
  python tools/soak_scatter.py --cluster --seconds $SECS --seed $SEED > $OUT/scatter_cluster.json 2> $OUT/scatter_cluster.err
  python tools/soak_scatter.py --mode query --seconds $SECS --seed $SEED > $OUT/scatter_query.json 2> $OUT/scatter_query.err
  python tools/soak_scatter.py --mode track --seconds $SECS --seed $SEED --max 90 130 > $OUT/track.json 2> $OUT/track.err
  python tools/soak_gather.py --seconds $SECS --seed $SEED > $OUT/gather.json 2> $OUT/gather.err
  python tools/soak_chains.py --seconds $SECS --seed $SEED > $OUT/chains.json 2> $OUT/chains.err
  python tools/soak_slab.py --seconds $SECS --seed $SEED > $OUT/slab.json 2> $OUT/slab.err
  python tools/soak_paths.py --seconds $SECS --seed $SEED > $OUT/paths.json 2> $OUT/paths.err
  python tools/soak_api.py --seconds $SECS --seed $SEED > $OUT/api.json 2> $OUT/api.err
fi
python - <<PY
import glob, json
bad = 0
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no result:", e); bad += 1; continue
    n = d.get("mismatching_nodes_or_cases", d.get("mismatching_pixels_or_cases", d.get("missing", d.get("validity_mismatches", 0))))
    print("{:<22} cases {:>7}  off {}".format(f.split("/")[-1][:-5], d.get("cases"), n))
    bad += 0 if f.endswith("paths.json") else (1 if n else 0)        # (the two paths differ on a few co-circular cells: printed, not counted)
raise SystemExit(1 if bad else 0)
PY
