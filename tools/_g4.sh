mkdir -p gpurun_out/r3
export OFL_LIB=$PWD/oflibnumpy_amd/libofl_hip_exp.so
for R in 1 2 4 8; do
  for P in scale shift rot; do
    OFL_C3_XCD_ROWS=$R python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary --pattern $P 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('R=$R $P', d['roofline']['kernel_ms'], d['roofline']['frac'])"
  done
done
