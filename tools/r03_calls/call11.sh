#!/bin/bash
# round 3, GPU call 11: the headline under the two arrangements of the state vector (kmc_layout.h): replica-major (automatic
# at these constants) against the tight packing of rounds 1-2, real and with the table untouched (KMC_DRYRUN=1)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/c11; mkdir -p $O; rm -f gpurun_out/sweep.log
export KMC_NO_TORCH=1
for lay in auto tight auto tight; do
  export KMC_LAYOUT=$lay
  tools/sweep.sh "layout_$lay||"
  KMC_DRYRUN=1 timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "dry/shadow" | tail -1 | sed "s/^/layout_$lay /" | tee -a gpurun_out/sweep.log
done
cp gpurun_out/sweep.log $O/sweep.log
