#!/bin/bash
# Round-2 GPU call 10: final kernel (64-entry stager) — suite, rocprofv3 passes, counter calibration + randbench, ablation, bench
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02c10
mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.txt 2>&1
echo "pytest rc=$?" >> $OUT/gpu_tests.txt
tail -8 $OUT/gpu_tests.txt | cut -c1-200
rm -rf gpurun_out/prof_r02 gpurun_out/calib_r02
timeout 1000 tools/profile.sh r02 > $OUT/profile.log 2>&1; tail -2 $OUT/profile.log
timeout 500 tools/calibrate_fetch.sh r02 > $OUT/calib.txt 2>&1; tail -28 $OUT/calib.txt
rm -f gpurun_out/ablate.log; timeout 300 tools/ablate.sh > $OUT/ablate.txt 2>&1; tail -14 $OUT/ablate.txt
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; cut -c1-400 $OUT/bench.json
du -sh gpurun_out
