mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_scatter_exact.py tests/test_gpu_scatter.py -m gpu -q -x > gpurun_out/r3/t3.log 2>&1; tail -3 gpurun_out/r3/t3.log
for op in object stripes speckle hole config5; do
  EXTRA=""; [ "$op" = config5 ] && EXTRA="--size 4320 7680"
  PYTHONPATH=tools python tools/bench_invert.py --op $op $EXTRA --iters 5 2>&1 | tail -1
done > gpurun_out/r3/cases3.log
cat gpurun_out/r3/cases3.log
echo "--- raster in bucket order (experiments build)"
OFL_LIB=$PWD/oflibnumpy_amd/libofl_hip_exp.so OFL_DL_RASTER_BUCKET=1 PYTHONPATH=tools python tools/bench_invert.py --op config5 --size 4320 7680 --iters 5 2>&1 | tail -1
OFL_LIB=$PWD/oflibnumpy_amd/libofl_hip_exp.so OFL_DL_RASTER_BUCKET=1 PYTHONPATH=tools python tools/bench_invert.py --op speckle --iters 5 2>&1 | tail -1
