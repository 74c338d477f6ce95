#!/bin/bash
# full GPU verification of the current tree: pytest -m gpu, smoke, the bench line, then rocprofv3 stats + PMC (tools/profile.sh)
cd "$(dirname "$0")/.."
REPO=$PWD
OUT=gpurun_out/r02c19
mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $OUT/gpu_tests.txt
grep -n "passed\|failed\|error" $OUT/gpu_tests.txt | tail -3
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -n 1 $OUT/smoke.txt
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.json
timeout 600 bash tools/profile.sh r02f > $OUT/profile.log 2>&1; tail -n 5 $OUT/profile.log; cat gpurun_out/prof_r02f/passes.log
