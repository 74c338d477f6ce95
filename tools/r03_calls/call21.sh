#!/bin/bash
# round 3, GPU call 21 (= call 19 on the final device code): the committed evidence of the tree as it stands — rocprofv3 kernel stats + counters
# (tools/profile.sh), copied into profiles/ BEFORE the bench line is taken so that the line quotes them, then the bench line
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/c21; mkdir -p $O
timeout 900 tools/profile.sh r03 > $O/profile.log 2>&1; tail -3 $O/profile.log
cp gpurun_out/prof_r03/kernel_stats.csv profiles/r03_kernel_stats.csv
cp gpurun_out/prof_r03/pmc_summary.json profiles/r03_pmc_summary.json
cp gpurun_out/prof_r03/summary.json profiles/r03_summary.json
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
KMC_NO_TORCH=1 KMC_JIT_DEFINES="-DKMC_PROFILE=1" timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "per-wave|leaves" | tee $O/phases.txt
