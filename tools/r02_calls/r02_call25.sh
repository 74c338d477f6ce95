#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02c25
mkdir -p $OUT
rm -f gpurun_out/sweep.log
bash tools/sweep.sh "base||" "setbits_auto|-DKMC_WALK_SETBITS=1|" "setbits_w6|-DKMC_WALK_SETBITS=1 -DKMC_MIN_WAVES=6|" "setbits_w5|-DKMC_WALK_SETBITS=1 -DKMC_MIN_WAVES=5|" "base_again||" "setbits_w6_again|-DKMC_WALK_SETBITS=1 -DKMC_MIN_WAVES=6|"
cp gpurun_out/sweep.log $OUT/sweep.log
