#!/bin/bash
# round 6, call 8: what would BASELINE config 5's k_expand gain if a flush never waited for its claims?  Shadow passes of a tuning
# build (every level runs first on a copy of the table with KMC_XFLAGS applied, then for real): 128 = the compare-and-swap is a
# plain store (no round trip for the claim), 128 + 2048 = ... and a chain ends at its first slot (no round trip at all after the
# deferred load): the upper bound of a fully asynchronous sink.  Then the per-phase ticks of the kernel as it is.
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_8; mkdir -p $O
export KMC_NO_TORCH=1 KMC_BENCH_TABLE=$(( (7<<30)/4 )) KMC_BENCH_FRONTIER=$((1<<29))
C5="python bench.py --workload Kip320,7,8,8,3 --level-budget 10 --no-cpu-baseline --no-orbit-counting --no-cold-start --no-baseline-configs --no-stretch --steps 1 --warmup 1"
export KMC_JIT_DEFINES="-DKMC_TUNING=1 -DKMC_MERGED_STEPS=1"
for x in 128 2176 128 2176; do
  echo "== shadow pass, KMC_XFLAGS=$x" | tee -a $O/shadow.txt
  KMC_SHADOW=1 KMC_XFLAGS=$x timeout 300 $C5 2>&1 >/dev/null | grep "kmc\]" | grep -v "spec" | tail -3 | tee -a $O/shadow.txt
done
export KMC_JIT_DEFINES="-DKMC_TUNING=1 -DKMC_MERGED_STEPS=1 -DKMC_PROFILE=1"
echo "== per-phase ticks" | tee -a $O/shadow.txt
timeout 300 $C5 2>&1 >/dev/null | grep "kmc\]" | grep -v "spec" | tail -3 | tee -a $O/shadow.txt
