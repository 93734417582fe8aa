mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -q -x > gpurun_out/r3/full2.log 2>&1; tail -2 gpurun_out/r3/full2.log
python tools/bench_ops.py --iters 20 > gpurun_out/r3/ops_bench_final.jsonl 2> gpurun_out/r3/ops_bench_final.err
bash tools/prof_cases.sh r3g object stripes speckle wobble hole config5 > gpurun_out/r3/prof_cases.log 2>&1
bash tools/prof_ops.sh r03_ops > gpurun_out/r3/prof_ops.log 2>&1
bash tools/prof_ops_pmc.sh r03_ops_pmc > gpurun_out/r3/prof_ops_pmc.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r3/bench_n1.json 2> gpurun_out/r3/bench_n1.err
python - <<'PY'
import json
for l in open('gpurun_out/r3/ops_bench_final.jsonl'):
    d=json.loads(l); print("%-78s %9.4f ms %6.3f" % (d['op'][:78], d['device_ms'], d['frac_of_8TBps']))
d=json.load(open('gpurun_out/r3/bench_n1.json')); print(d['value'], d['roofline']['frac'], d['roofline'].get('other_patterns'))
PY
