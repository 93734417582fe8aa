mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -q -x > gpurun_out/r3/full.log 2>&1; tail -3 gpurun_out/r3/full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
