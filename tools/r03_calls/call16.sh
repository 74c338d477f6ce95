#!/bin/bash
# round 3, GPU call 16: the step around the kernel — chained batches of 8 / 16 / 32 levels, the double-buffered seen-set
# (KMC_SPARE_TABLE) on and off — the ablation ladder of the kind-major kernel, and BASELINE config 5 over a level budget
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/c16; mkdir -p $O; rm -f gpurun_out/sweep.log gpurun_out/ablate.log
export KMC_NO_TORCH=1
for spare in 1 0; do for chain in 8 16 32; do
  KMC_SPARE_TABLE=$spare KMC_CHAIN_MAX=$chain tools/sweep.sh "spare${spare}_chain${chain}||"
done; done
KMC_SPARE_TABLE=1 KMC_CHAIN_MAX=16 tools/sweep.sh "spare1_chain16_again||"
cp gpurun_out/sweep.log $O/sweep.log
timeout 600 tools/ablate.sh > /dev/null 2>&1; cp gpurun_out/ablate.log $O/ablate.log; cat $O/ablate.log
KMC_BENCH_TABLE=$((1<<31)) KMC_BENCH_FRONTIER=$((1<<29)) timeout 600 python bench.py --workload Kip320,7,8,8,3 --level-budget 10 --no-cpu-baseline --steps 1 --warmup 1 > $O/config5.json 2> $O/config5.err
cut -c1-900 $O/config5.json; tail -2 $O/config5.err
