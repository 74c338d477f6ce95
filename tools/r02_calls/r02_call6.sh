#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02c6
mkdir -p $OUT
export KMC_NO_TORCH=1
export KMC_CACHE_DIR=$PWD/kafka_specification_amd/kmc_cache_exp
for d in "-DKMC_MIN_WAVES=6" "-DKMC_MIN_WAVES=6" "-DKMC_MIN_WAVES=6 -DKMC_RING_FENCE=1" "-DKMC_MIN_WAVES=6 -O1" "-DKMC_MIN_WAVES=4"; do
  KMC_JIT_DEFINES="$d" timeout 120 python tools/experimental/wide_repro.py 2>&1 | tail -1 | cut -c1-500 >> $OUT/wide.txt
done
cat $OUT/wide.txt
