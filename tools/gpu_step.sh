#!/bin/bash
# one gpurun call of the round: tools/gpu_step.sh <tag> <step> [<step> ...]   (steps are joined with &&: a failed or killed step ends the call)
#   tests            the whole -m gpu suite                     -> gpurun_out/<tag>_tests.log
#   tests:<expr>     pytest -k <expr>
#   bench            python bench.py --steps 20 --warmup 5      -> gpurun_out/<tag>_bench.json (+ .err)
#   bench2           the same with --gpus 2 on the one GPU (rehearsal of the N > 1 path; ranks share the card)
#   cases[:ops]      per-kernel breakdown of the Delaunay-path cases (tools/prof_cases.sh)
#   ops[:only]       tools/bench_ops.py [--only ...]            -> gpurun_out/<tag>_ops.jsonl
#   soak:<seconds>   tools/soak_all.sh <seconds>                -> gpurun_out/<tag>_soak.jsonl
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
cd $ROOT
for STEP in "$@"; do
  echo "== step $STEP ($(date +%T))"
  case $STEP in
    tests)   timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1 || { tail -40 gpurun_out/${TAG}_tests.log; exit 1; }; tail -3 gpurun_out/${TAG}_tests.log ;;
    tests:*) timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "${STEP#tests:}" > gpurun_out/${TAG}_tests.log 2>&1 || { tail -40 gpurun_out/${TAG}_tests.log; exit 1; }; tail -3 gpurun_out/${TAG}_tests.log ;;
    bench)   timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { tail -20 gpurun_out/${TAG}_bench.err; exit 1; }; wc -c gpurun_out/${TAG}_bench.json; cut -c1-600 gpurun_out/${TAG}_bench.json ;;
    bench2)  timeout -k 10 600 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench2.json 2> gpurun_out/${TAG}_bench2.err || { tail -20 gpurun_out/${TAG}_bench2.err; exit 1; }; cut -c1-300 gpurun_out/${TAG}_bench2.json; grep -o '"strong".*' gpurun_out/${TAG}_bench2.json | cut -c1-900 ;;
    cases)   bash tools/prof_cases.sh $TAG object stripes speckle wobble hole config5 || exit 1 ;;
    cases:*) bash tools/prof_cases.sh $TAG $(echo "${STEP#cases:}" | tr ',' ' ') || exit 1 ;;
    ops)     timeout -k 10 600 python tools/bench_ops.py > gpurun_out/${TAG}_ops.jsonl 2> gpurun_out/${TAG}_ops.err || { tail -20 gpurun_out/${TAG}_ops.err; exit 1; }; cut -c1-200 gpurun_out/${TAG}_ops.jsonl ;;
    ops:*)   timeout -k 10 600 python tools/bench_ops.py --only "${STEP#ops:}" > gpurun_out/${TAG}_ops.jsonl 2> gpurun_out/${TAG}_ops.err || { tail -20 gpurun_out/${TAG}_ops.err; exit 1; }; cut -c1-200 gpurun_out/${TAG}_ops.jsonl ;;
    profbench) bash tools/prof_bench.sh $TAG || exit 1 ;;
    profops)   bash tools/prof_ops.sh $TAG || exit 1 ;;
    profops:*) bash tools/prof_ops.sh $TAG $(echo "${STEP#profops:}" | tr ',' ' ') || exit 1 ;;
    opspmc)    bash tools/prof_ops_pmc.sh $TAG || exit 1 ;;
    k1)        timeout -k 10 400 python tools/bench_k1.py > gpurun_out/${TAG}_k1.jsonl 2> gpurun_out/${TAG}_k1.err || { tail -5 gpurun_out/${TAG}_k1.err; exit 1; }; cut -c1-160 gpurun_out/${TAG}_k1.jsonl ;;
    walkab)    bash tools/walk_ab.sh $TAG || exit 1 ;;
    k1pmc)     bash tools/prof_k1_pmc.sh ${TAG}_k1pmc || exit 1 ;;
    walk)      bash tools/prof_one.sh ${TAG}_walk --op invert --iters 10 > gpurun_out/${TAG}_walk.log 2>&1 || { tail -5 gpurun_out/${TAG}_walk.log; exit 1; }; grep walk gpurun_out/prof_${TAG}_walk/summary.txt | head -30 ;;
    parity)    timeout -k 10 600 python tools/scatter_parity_table.py > gpurun_out/${TAG}_parity.txt 2> gpurun_out/${TAG}_parity.err || { tail -5 gpurun_out/${TAG}_parity.err; exit 1; }; tail -30 gpurun_out/${TAG}_parity.txt ;;
    soak:*)  timeout -k 10 1100 bash tools/soak_all.sh "${STEP#soak:}" > gpurun_out/${TAG}_soak.log 2>&1; RC=$?; cat gpurun_out/${TAG}_soak.log; cat gpurun_out/soak/*.json > gpurun_out/${TAG}_soak.jsonl; [ $RC = 0 ] || exit 1 ;;
    *) echo "unknown step $STEP"; exit 2 ;;
  esac || exit 1
done
echo "== done ($(date +%T))"
