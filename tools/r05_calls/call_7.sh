#!/bin/bash
# round 5, call 7: per-phase accounting (KMC_PROFILE, a tuning build: per-wave s_memtime ticks) of the seven-broker kernels on the
# final code — plain over 10 levels, orbit counting over 14
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r05_7; mkdir -p $O
export KMC_NO_TORCH=1 KMC_BENCH_TABLE=$((1<<31)) KMC_BENCH_FRONTIER=$((1<<29))
export KMC_JIT_DEFINES="-DKMC_TUNING=1 -DKMC_PROFILE=1"
timeout 300 python bench.py --workload Kip320,7,8,8,3 --level-budget 10 --no-cpu-baseline --steps 1 --warmup 1 2>&1 >/dev/null | grep "kmc\]" | tail -3
timeout 300 python bench.py --workload Kip320,7,8,8,3 --level-budget 14 --symmetry --no-cpu-baseline --steps 1 --warmup 1 2>&1 >/dev/null | grep "kmc\]" | tail -3
