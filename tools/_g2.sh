mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_scatter_exact.py tests/test_gpu_scatter.py tests/test_gpu_chains.py -m gpu -q -x > gpurun_out/r3/t5.log 2>&1; tail -3 gpurun_out/r3/t5.log
for op in invert switch; do PYTHONPATH=tools python tools/bench_invert.py --op $op --iters 100 2>&1 | tail -1 | cut -c1-120; done
python tools/bench_ops.py --only config3 --iters 20 2>/dev/null | cut -c1-150
