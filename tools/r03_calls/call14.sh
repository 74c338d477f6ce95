#!/bin/bash
# round 3, GPU call 14: kind-major walk with the replica-word selects made opaque (no scratch-indexed state words any more),
# 6 and 5 waves per SIMD, against the instance-major walk on the tight layout
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/c14; mkdir -p $O; rm -f gpurun_out/sweep.log
export KMC_NO_TORCH=1
run() {  # name, defines
  tools/sweep.sh "$1|$2|"
  KMC_JIT_DEFINES="$2" KMC_DRYRUN=1 timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "dry/shadow" | tail -1 | sed "s/^/$1 /" | tee -a gpurun_out/sweep.log
}
run km_w6 "-DKMC_MIN_WAVES=6"
run km_w5 "-DKMC_MIN_WAVES=5"
KMC_LAYOUT=tight run im_tight ""
run km_w5_again "-DKMC_MIN_WAVES=5"
run km_w6_again "-DKMC_MIN_WAVES=6"
cp gpurun_out/sweep.log $O/sweep.log
