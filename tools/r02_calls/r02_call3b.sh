#!/bin/bash
# Round-2 GPU call 3: the whole -m gpu suite, the default bench line, rocprofv3 passes + counter calibration + ablation.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02c3b
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q --durations=25 > $OUT/gpu_tests.txt 2>&1
echo "pytest rc=$?" >> $OUT/gpu_tests.txt
tail -45 $OUT/gpu_tests.txt | cut -c1-250
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; cut -c1-1500 $OUT/bench.json
KMC_NO_CHAIN=1 KMC_NO_TORCH=1 timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | cut -c1-400 > $OUT/bench_nochain.json
cut -c1-300 $OUT/bench_nochain.json
sed -i 's/timeout 300 rocprofv3/timeout 150 rocprofv3/' tools/profile.sh
timeout 1000 tools/profile.sh r02 > $OUT/profile.log 2>&1; tail -3 $OUT/profile.log
timeout 400 tools/calibrate_fetch.sh r02 > $OUT/calib.txt 2>&1; tail -25 $OUT/calib.txt
rm -f gpurun_out/ablate.log; timeout 300 tools/ablate.sh > $OUT/ablate.txt 2>&1; tail -16 $OUT/ablate.txt
du -sh gpurun_out
