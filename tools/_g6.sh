mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_gather.py tests/test_gpu_fullsize.py -m gpu -q -x > gpurun_out/r3/t6.log 2>&1; tail -3 gpurun_out/r3/t6.log
PYTHONPATH=tools python tools/bench_k1.py --iters 30 > gpurun_out/r3/k1_bench2.jsonl 2> gpurun_out/r3/k1_bench2.err; tail -2 gpurun_out/r3/k1_bench2.err
python - <<'PY'
import json
for l in open('gpurun_out/r3/k1_bench2.jsonl'):
    d=json.loads(l); print("%-60s %s %9.4f ms %6.3f" % (d['op'][:60], d['shape'][0], d['device_ms'], d['frac_of_8TBps']))
PY
