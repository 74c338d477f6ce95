#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02c3c
mkdir -p $OUT
export KMC_NO_TORCH=1
for d in "" "" "-O1" "-DKMC_MIN_WAVES=4" "-mllvm -amdgpu-spill-sgpr-to-vgpr=0" "-O2"; do
  KMC_JIT_DEFINES="$d" timeout 150 python tools/experimental/old_r01_repro.py Kip320 7 1 1 0 2>&1 | tail -2 | cut -c1-400 >> $OUT/old.txt
done
cat $OUT/old.txt
