#!/bin/bash
# Round-2 GPU call 1: (a) A/B of the super-tile k_expand against the per-tile kernel on the headline,
# (b) parity suites with the super-tile kernel, (c) experiments on the lost-successor bug
# (Kip320, 7 replicas, 80-VGPR build).  Everything writes under gpurun_out/r02c1/.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02c1
mkdir -p $OUT
export KMC_NO_TORCH=1
echo "== sweep" > $OUT/sweep.txt
rm -f gpurun_out/sweep.log
timeout 900 tools/sweep.sh "base||" "st4|-DKMC_SUPERTILE=1|" "st2|-DKMC_SUPERTILE=1 -DKMC_ST_K=2|" \
    "st8|-DKMC_SUPERTILE=1 -DKMC_ST_K=8|" "base2||" "st4b|-DKMC_SUPERTILE=1|" >> $OUT/sweep.txt 2>&1
echo "== parity with the super-tile kernel" > $OUT/parity_st.txt
KMC_JIT_DEFINES="-DKMC_SUPERTILE=1" timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_async_isr.py -m gpu -x -q >> $OUT/parity_st.txt 2>&1
echo "== lost successors" > $OUT/lost.txt
for d in "-DKMC_MIN_WAVES=6" "-DKMC_MIN_WAVES=6" "-DKMC_MIN_WAVES=6 -DKMC_RING_FENCE=1" "-DKMC_MIN_WAVES=6 -O1" \
         "-DKMC_MIN_WAVES=6 -mllvm -amdgpu-spill-sgpr-to-vgpr=0" "-DKMC_MIN_WAVES=4"; do
  echo "---- $d" >> $OUT/lost.txt
  KMC_JIT_DEFINES="$d" timeout 600 python tests/diag_missing.py Kip320 7 1 1 0 >> $OUT/lost.txt 2>&1
done
tail -5 $OUT/sweep.txt $OUT/parity_st.txt
tail -40 $OUT/lost.txt
