#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02c20
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_sharded_and_traces.py -m gpu -x -q -k "eight_logical or truncate_to_hw or sender_side" > $OUT/new_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $OUT/new_tests.txt
grep -n "passed\|failed\|rror" $OUT/new_tests.txt | tail -5
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; python3 -c "
import json; j=json.load(open('$OUT/bench.json')); r=j['roofline']
print(j['value'], j['ms_per_step'], r['kernel_seconds_per_step'], r['frac'], r['traffic'], r['line_granular_GBps'])
print(json.dumps(r['random_access'])); print(json.dumps(j['cpu_baseline']))"
