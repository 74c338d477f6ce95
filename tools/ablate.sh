#!/bin/bash
# Ablation of k_expand on the headline workload: dry passes (no table / read-only probes / + no-op atomics)
# and shadow passes with parts switched off, all in one process so that they share a box.
# Output: gpurun_out/ablate.log (the round-1 result is profiles/r01_ablation.txt).
cd "$(dirname "$0")/.."
export KMC_NO_TORCH=1
for spec in "KMC_DRYRUN=1" "KMC_DRYRUN=2" "KMC_DRYRUN=4" "KMC_SHADOW=1 KMC_XFLAGS=128" "KMC_SHADOW=1 KMC_XFLAGS=32" "KMC_SHADOW=1 KMC_XFLAGS=64" "KMC_SHADOW=1 KMC_XFLAGS=160"; do
  echo "== $spec" >> gpurun_out/ablate.log
  env $spec timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "dry/shadow" | tail -1 >> gpurun_out/ablate.log
done
cat gpurun_out/ablate.log
