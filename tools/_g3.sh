for lib in libofl_hip.so libofl_w6.so libofl_w7.so libofl_w8.so; do
  echo $lib
  for op in invert; do OFL_LIB=$PWD/oflibnumpy_amd/$lib PYTHONPATH=tools python tools/bench_invert.py --op $op --iters 100 2>&1 | tail -1 | cut -c1-120; done
done
