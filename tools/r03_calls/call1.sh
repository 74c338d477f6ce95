#!/bin/bash
# round 3, GPU call 1: the whole -m gpu suite on the new "generated" semantics + the Oracle-R fixture test, the headline
# with the two cheap probe knobs, the DDD primitives microbenchmark, one full bench line.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/c1
rm -f gpurun_out/sweep.log
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/c1/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/c1/tests.log
tools/sweep.sh "base||" "line|-DKMC_LINE_PROBE=1|" "dedup|-DKMC_FLUSH_DEDUP=1|" "line_dedup|-DKMC_LINE_PROBE=1 -DKMC_FLUSH_DEDUP=1|" "base2||"
KMC_BENCH_TABLE=$((1<<29)) tools/sweep.sh "base_4g||" "line_4g|-DKMC_LINE_PROBE=1|" "line_dedup_4g|-DKMC_LINE_PROBE=1 -DKMC_FLUSH_DEDUP=1|"
cp gpurun_out/sweep.log gpurun_out/c1/sweep.log
timeout 300 tools/membench/ddd_prims 26 250 > gpurun_out/c1/ddd_prims.txt 2>&1
timeout 600 python bench.py > gpurun_out/c1/bench.json 2> gpurun_out/c1/bench.err
tail -3 gpurun_out/c1/tests.log; cat gpurun_out/c1/sweep.log; cat gpurun_out/c1/ddd_prims.txt; cut -c1-600 gpurun_out/c1/bench.json
