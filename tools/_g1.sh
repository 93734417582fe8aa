mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_scatter_exact.py tests/test_gpu_scatter.py tests/test_gpu_fullsize.py -m gpu -q -x > gpurun_out/r3/t3.log 2>&1; tail -3 gpurun_out/r3/t3.log
for op in object stripes speckle hole config5; do
  EXTRA=""; [ "$op" = config5 ] && EXTRA="--size 4320 7680"
  PYTHONPATH=tools python tools/bench_invert.py --op $op $EXTRA --iters 8 2>&1 | tail -1 | cut -c1-110
done
PYTHONPATH=tools python tools/bench_mode1t.py 2>&1 | tail -1 | cut -c1-130
