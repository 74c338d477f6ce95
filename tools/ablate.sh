#!/bin/bash
cd "$(dirname "$0")/.."
export KMC_NO_TORCH=1
export KMC_JIT_DEFINES="-DKMC_ERRCHK_TILE=1 -DKMC_SETPRIO=1"
for spec in "KMC_DRYRUN=1" "KMC_DRYRUN=2" "KMC_DRYRUN=4" "KMC_SHADOW=1 KMC_XFLAGS=128" "KMC_SHADOW=1 KMC_XFLAGS=32" "KMC_SHADOW=1 KMC_XFLAGS=64" "KMC_SHADOW=1 KMC_XFLAGS=160"; do
  echo "== $spec" >> gpurun_out/ablate.log
  env $spec timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "dry/shadow" | tail -1 >> gpurun_out/ablate.log
done
cat gpurun_out/ablate.log
