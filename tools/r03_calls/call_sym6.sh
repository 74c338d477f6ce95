#!/bin/bash
# round 3, symmetry call 6: the headline's orbit-counting kernel (stabiliser plane, sparse tiles) under occupancy targets / dry pass / phase split
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/s6
export KMC_NO_TORCH=1
out=gpurun_out/s6/ablate.log
: > $out
run() { echo "== $1" >> $out; shift; env "$@" timeout 200 python tools/sym_headline.py 3 sym 2>&1 | grep -E "ms_step|dry/shadow|per-wave|leaves" | tail -3 >> $out; }
run "default" A=1
run "KMC_DRYRUN=1" KMC_DRYRUN=1
run "KMC_DRYRUN=2" KMC_DRYRUN=2
for w in 6 5 4; do run "KMC_MIN_WAVES=$w" KMC_JIT_DEFINES=-DKMC_MIN_WAVES=$w; done
run "KMC_PROFILE=1" KMC_JIT_DEFINES=-DKMC_PROFILE=1
cat $out
