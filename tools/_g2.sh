mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_scatter_exact.py tests/test_gpu_scatter.py tests/test_gpu_chains.py tests/test_host_api.py -m gpu -q -x > gpurun_out/r3/t5.log 2>&1; tail -3 gpurun_out/r3/t5.log
for op in invert switch; do PYTHONPATH=tools python tools/bench_invert.py --op $op --iters 50 2>&1 | tail -1; done
