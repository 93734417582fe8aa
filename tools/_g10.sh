mkdir -p gpurun_out/r3
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT/tools
for PASS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  NAME=$(echo $PASS | tr ' ' '_' | cut -c1-20)
  timeout -k 10 200 rocprofv3 --pmc $PASS --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c5pmc/pmc_$NAME -- python3 $GRAFT_REPO_ROOT/tools/bench_invert.py --op config5 --size 4320 7680 --iters 2 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python3 tools/rocprof_summary.py gpurun_out/prof_c5pmc/pmc_* | grep -E "near_kernel|raster_small|cell_kernel|star_fan" 
find gpurun_out/prof_c5pmc -name "*.db" -delete
