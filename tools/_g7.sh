mkdir -p gpurun_out/r3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_m1t -- python3 $GRAFT_REPO_ROOT/tools/bench_mode1t.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/rocprof_summary.py gpurun_out/prof_m1t | head -40
find gpurun_out/prof_m1t -name "*.db" -delete
