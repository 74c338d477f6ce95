#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02c13
mkdir -p $OUT
export KMC_CACHE_DIR=$PWD/kafka_specification_amd/kmc_cache_exp
rm -f gpurun_out/sweep.log
timeout 900 tools/sweep.sh "base||" "chunk8|-DKMC_TILE_CHUNK=8|" "chunk8_f256|-DKMC_TILE_CHUNK=8 -DKMC_LDS_FILTER=256|6" \
   "chunk4_f256|-DKMC_TILE_CHUNK=4 -DKMC_LDS_FILTER=256|6" "f256|-DKMC_LDS_FILTER=256|6" "chunk16_f256|-DKMC_TILE_CHUNK=16 -DKMC_LDS_FILTER=256|6" "base2||" > $OUT/sweep.txt 2>&1
cat $OUT/sweep.txt
export KMC_NO_TORCH=1
for d in "-DKMC_PROFILE=1 -DKMC_LDS_FILTER=256" "-DKMC_PROFILE=1 -DKMC_TILE_CHUNK=8 -DKMC_LDS_FILTER=256" "-DKMC_PROFILE=1 -DKMC_TILE_CHUNK=16 -DKMC_LDS_FILTER=256"; do
  echo "== $d" >> $OUT/filter.txt
  KMC_BLOCKS_PER_CU=6 KMC_NO_CHAIN=1 KMC_JIT_DEFINES="$d" timeout 120 python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "\[kmc\]" >> $OUT/filter.txt
done
cat $OUT/filter.txt
