mkdir -p gpurun_out/r3
python tools/bench_ops.py --iters 20 > gpurun_out/r3/ops_bench.jsonl 2> gpurun_out/r3/ops_bench.err; tail -2 gpurun_out/r3/ops_bench.err
PYTHONPATH=tools python tools/bench_k1.py --iters 30 > gpurun_out/r3/k1_bench.jsonl 2> gpurun_out/r3/k1_bench.err; tail -2 gpurun_out/r3/k1_bench.err
python - <<'PY'
import json
for f in ('gpurun_out/r3/ops_bench.jsonl','gpurun_out/r3/k1_bench.jsonl'):
    for l in open(f):
        d=json.loads(l); print("%-78s %9.4f ms %6.3f  sets %d" % (d['op'][:78], d['device_ms'], d['frac_of_8TBps'], d.get('rotating_sets',1)))
PY
