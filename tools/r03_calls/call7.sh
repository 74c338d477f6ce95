#!/bin/bash
# round 3, GPU call 7: the whole -m gpu suite on the per-block tail, the deadlock-witness reduction against the old
# per-state atomicMax, wide slots, the 6.45 G-state stretch, a bench line
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/c7
rm -f gpurun_out/sweep.log
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/c7/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/c7/tests.log
tail -5 gpurun_out/c7/tests.log
tools/sweep.sh "base||" "base_again||"
KMC_BENCH_FP128=1 tools/sweep.sh "fp128_16g||"
cp gpurun_out/sweep.log gpurun_out/c7/sweep.log
KMC_NO_TORCH=1 timeout 600 python tools/fp128_stretch.py 0 0x5EED2 0xC0FFEE > gpurun_out/c7/fp128_stretch.jsonl 2> gpurun_out/c7/fp128_stretch.err
KMC_NARROW=1 KMC_NO_TORCH=1 timeout 600 python tools/fp128_stretch.py 0 >> gpurun_out/c7/fp128_stretch.jsonl 2>> gpurun_out/c7/fp128_stretch.err
cat gpurun_out/c7/fp128_stretch.jsonl; tail -3 gpurun_out/c7/fp128_stretch.err
timeout 600 python bench.py > gpurun_out/c7/bench.json 2> gpurun_out/c7/bench.err
cut -c1-300 gpurun_out/c7/bench.json
