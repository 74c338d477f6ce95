#!/bin/bash
# round 6, call 19: the last bound on BASELINE config 5's flush - a shadow pass in which the deferred probe's LOAD is not issued
# either (flag 4096: an experiment's three lines in kmc_expand_body's defer(), `pd_v = hash says duplicate ? fp : 0`, not kept in the
# source): with plain-store claims (128) and no walk (2048) the flush then touches the table without ever waiting for it.
cd "${GRAFT_REPO_ROOT:-.}"
export KMC_NO_TORCH=1 KMC_BENCH_TABLE=$(( (7<<30)/4 )) KMC_BENCH_FRONTIER=$((1<<29))
C5="python bench.py --workload Kip320,7,8,8,3 --level-budget 10 --no-cpu-baseline --no-orbit-counting --no-cold-start --no-baseline-configs --no-stretch --steps 1 --warmup 1"
export KMC_JIT_DEFINES="-DKMC_TUNING=1"
for x in 2176 6272 4224 2176 6272; do
  echo "== shadow pass, KMC_XFLAGS=$x (128 plain-store claims, 2048 no walk, 4096 no probe load)"
  KMC_SHADOW=1 KMC_XFLAGS=$x timeout 300 $C5 2>&1 >/dev/null | grep "kmc\]" | grep -v "spec" | tail -2
done
