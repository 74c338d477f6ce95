#!/bin/bash
# round 3, GPU call 5: the per-block tail (one atomic instruction per block for all of a launch's counters) against the
# same kernel without the conservation counters
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/c5
rm -f gpurun_out/sweep.log
tools/sweep.sh "base||"
KMC_X_NO_CONSERVATION=1 tools/sweep.sh "no_counters|-DKMC_X_NO_COUNTERS=1|"
tools/sweep.sh "base_again||"
cp gpurun_out/sweep.log gpurun_out/c5/sweep.log
