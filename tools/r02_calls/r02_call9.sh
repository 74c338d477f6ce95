#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02c9
mkdir -p $OUT
export KMC_CACHE_DIR=$PWD/kafka_specification_amd/kmc_cache_exp
rm -f gpurun_out/sweep.log
timeout 800 tools/sweep.sh "w6||" "w7|-DKMC_MIN_WAVES=7|" "w8|-DKMC_MIN_WAVES=8|" "w8_bpc7|-DKMC_MIN_WAVES=8|7" "w8_bpc6|-DKMC_MIN_WAVES=8|6" "w6_again||" "w8_again|-DKMC_MIN_WAVES=8|" > $OUT/sweep.txt 2>&1
cat $OUT/sweep.txt
