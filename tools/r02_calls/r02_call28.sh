#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02c28
mkdir -p $OUT
timeout 150 python tests/native_exchange_threads.py Kip320 3 2 2 1 2 TypeOk,WeakIsr,StrongIsr > $OUT/first.txt 2>&1; echo "first rc=$?"
tail -n 6 $OUT/first.txt | cut -c1-700
if grep -q '"matches_oracle": true' $OUT/first.txt; then
  timeout 400 python -m pytest tests/test_gpu_native_exchange_threads.py -m gpu -x -q > $OUT/tests.txt 2>&1; echo "pytest rc=$?"
  tail -n 15 $OUT/tests.txt | cut -c1-600
fi
