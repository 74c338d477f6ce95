#!/bin/bash
# The symmetric (orbit-counting) headline kernel under the ablation switches of the plain one: dry passes, occupancy targets,
# phase split, table size.  Output: gpurun_out/sym_ablate.log
cd "$(dirname "$0")/.."
export KMC_NO_TORCH=1
out=gpurun_out/sym_ablate.log
: > $out
run() {  # label, env..., -- args
  echo "== $1" >> $out; shift
  env "$@" timeout 200 python tools/sym_headline.py 3 sym $TLOG 2>&1 | grep -E "ms_step|dry/shadow|per-wave|leaves" | tail -4 >> $out
}
run "baseline"  A=1
run "KMC_DRYRUN=1 (table untouched)" KMC_DRYRUN=1
run "KMC_DRYRUN=2 (read-only probes)" KMC_DRYRUN=2
run "KMC_DRYRUN=4 (+ no-op atomics)" KMC_DRYRUN=4
run "KMC_NO_CHAIN=1" KMC_NO_CHAIN=1
for w in 6 5 3; do run "KMC_MIN_WAVES=$w" KMC_JIT_DEFINES=-DKMC_MIN_WAVES=$w; done
run "KMC_PROFILE=1" KMC_JIT_DEFINES=-DKMC_PROFILE=1
TLOG=27 run "table 2^27" A=1
TLOG=26 run "table 2^26" A=1
TLOG=30 run "table 2^30" A=1
cat $out
