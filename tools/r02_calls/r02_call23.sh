#!/bin/bash
# full GPU verification of the tree: pytest -m gpu, smoke, bench line, rocprofv3 stats + PMC (tools/profile.sh), third seed of the 6.45 G run
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02c23
mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $OUT/gpu_tests.txt
grep -n "passed\|failed\|rror" $OUT/gpu_tests.txt | tail -3
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -n 1 $OUT/smoke.txt
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.json
timeout 700 bash tools/profile.sh r02g > $OUT/profile.log 2>&1; tail -n 2 $OUT/profile.log; cat gpurun_out/prof_r02g/passes.log
rm -f gpurun_out/ladder.jsonl
timeout 120 python tools/run_ladder.py stretch_kip320_3_6_6_3_seed3 > $OUT/ladder.log 2>&1; cut -c1-400 $OUT/ladder.log
