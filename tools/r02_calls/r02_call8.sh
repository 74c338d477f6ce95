#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02c8
mkdir -p $OUT
export KMC_CACHE_DIR=$PWD/kafka_specification_amd/kmc_cache_exp
rm -f gpurun_out/sweep.log
timeout 600 tools/sweep.sh "base||" "contig|-DKMC_TILE_CONTIG=1|" "base2||" "contig2|-DKMC_TILE_CONTIG=1|" > $OUT/sweep.txt 2>&1
cat $OUT/sweep.txt
export KMC_NO_TORCH=1
for d in "-DKMC_PROFILE=1" "-DKMC_PROFILE=1 -DKMC_TILE_CONTIG=1"; do
  echo "== $d" >> $OUT/leaves.txt
  KMC_NO_CHAIN=1 KMC_JIT_DEFINES="$d" timeout 120 python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "\[kmc\]" >> $OUT/leaves.txt
done
cat $OUT/leaves.txt
