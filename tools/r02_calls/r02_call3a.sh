#!/bin/bash
# lost-successor hunt, bounded: round-1 code shape (volatile guard asm) at 80 VGPRs on the wide configurations
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02c3a
mkdir -p $OUT
export KMC_NO_TORCH=1
export KMC_CACHE_DIR=$PWD/kafka_specification_amd/kmc_cache_exp
for d in "-DKMC_MIN_WAVES=6 -DKMC_GUARD_VOLATILE=1" "-DKMC_MIN_WAVES=6 -DKMC_GUARD_VOLATILE=1 -DKMC_RING_FENCE=1" \
         "-DKMC_MIN_WAVES=6 -DKMC_GUARD_VOLATILE=1 -DKMC_ERRCHK_TILE=0 -DKMC_SETPRIO=0"; do
  for c in "Kip320 7 1 1 0" "Kip279 7 1 1 0"; do
    echo "---- $d :: $c" >> $OUT/lost.txt
    KMC_JIT_DEFINES="$d" timeout 90 python tests/diag_missing.py $c 2>&1 | head -c 6000 >> $OUT/lost.txt
    echo "[rc ${PIPESTATUS[0]}]" >> $OUT/lost.txt
  done
done
head -c 30000 $OUT/lost.txt
