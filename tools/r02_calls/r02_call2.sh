#!/bin/bash
# Round-2 GPU call 2: (a) headline with the integer-domain invariants (A/B against round 1's 36.7 ms on the same box class),
# (b) the whole -m gpu suite (native exchange under the C ABI, RCCL world-1 selftest, new sharded tests),
# (c) lost-successor hunt with the round-1 code shape (volatile guard asm) on the wide configurations.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02c2
mkdir -p $OUT
echo "== gpu suite" > $OUT/gpu_tests.txt
timeout 2400 python -m pytest tests -m gpu -x -q >> $OUT/gpu_tests.txt 2>&1
tail -5 $OUT/gpu_tests.txt
export KMC_NO_TORCH=1
rm -f gpurun_out/sweep.log
timeout 600 tools/sweep.sh "inv_int||" "inv_int_again||" > $OUT/sweep.txt 2>&1
cat $OUT/sweep.txt
export KMC_CACHE_DIR=$PWD/kafka_specification_amd/kmc_cache_exp
echo "== lost successors" > $OUT/lost.txt
for d in "-DKMC_MIN_WAVES=6" "-DKMC_MIN_WAVES=6 -DKMC_GUARD_VOLATILE=1" \
         "-DKMC_MIN_WAVES=6 -DKMC_GUARD_VOLATILE=1 -DKMC_ERRCHK_TILE=0 -DKMC_SETPRIO=0" \
         "-DKMC_MIN_WAVES=6 -DKMC_GUARD_VOLATILE=1 -DKMC_RING_FENCE=1"; do
  for c in "Kip320 7 1 1 0" "Kip279 7 1 1 0" "Kip320FirstTry 8 1 1 0" "KafkaTruncateToHighWatermark 6 1 1 1"; do
    echo "---- $d :: $c" >> $OUT/lost.txt
    KMC_JIT_DEFINES="$d" timeout 600 python tests/diag_missing.py $c >> $OUT/lost.txt 2>&1
  done
done
grep -c "missing" $OUT/lost.txt
grep -B2 -A8 "level .*missing" $OUT/lost.txt | head -60
