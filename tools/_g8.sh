mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_scatter.py -m gpu -q -x -k "track" > gpurun_out/r3/t8.log 2>&1; tail -15 gpurun_out/r3/t8.log
