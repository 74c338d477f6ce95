#!/bin/bash
# round 3, GPU call 4: where did 3 ms go?  Same box, same run: the headline with the conservation counters and / or the
# wide-slot branch compiled out (KMC_X_* are experiment-only switches).
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/c4
rm -f gpurun_out/sweep.log
tools/sweep.sh "base||" "no_counters|-DKMC_X_NO_COUNTERS=1|" "no_fp128|-DKMC_X_NO_FP128=1|" "neither|-DKMC_X_NO_COUNTERS=1 -DKMC_X_NO_FP128=1|" "base_again||"
cp gpurun_out/sweep.log gpurun_out/c4/sweep.log
