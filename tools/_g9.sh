mkdir -p gpurun_out/r3
bash tools/prof_cases.sh r3f object stripes speckle wobble hole config5 > gpurun_out/r3/prof_cases.log 2>&1
bash tools/prof_one.sh r03_walk --op invert --iters 12 > gpurun_out/r3/prof_walk.log 2>&1
python tools/bench_ops.py --iters 20 > gpurun_out/r3/ops_bench_final.jsonl 2> gpurun_out/r3/ops_bench_final.err
PYTHONPATH=tools python tools/bench_k1.py --iters 30 > gpurun_out/r3/k1_bench_final.jsonl 2> gpurun_out/r3/k1_bench_final.err
bash tools/prof_ops.sh r03_ops > gpurun_out/r3/prof_ops.log 2>&1
bash tools/prof_ops_pmc.sh r03_ops_pmc > gpurun_out/r3/prof_ops_pmc.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r3/bench_n1.json 2> gpurun_out/r3/bench_n1.err
python bench.py > gpurun_out/r3/bench_n1_200.json 2> gpurun_out/r3/bench_n1_200.err
python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r3/bench_gpus2.json 2> gpurun_out/r3/bench_gpus2.err
tail -c 400 gpurun_out/r3/bench_n1.json; echo; tail -c 300 gpurun_out/r3/bench_gpus2.json; cat gpurun_out/r3c_time.log 2>/dev/null | tail -3
