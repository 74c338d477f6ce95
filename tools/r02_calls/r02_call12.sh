#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02c12
mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.txt 2>&1
echo "pytest rc=$?" >> $OUT/gpu_tests.txt
tail -4 $OUT/gpu_tests.txt | cut -c1-200
rm -f gpurun_out/ladder.jsonl
timeout 300 python tools/run_ladder.py config0_idsequence config1_finite_replicated_log config2_headline config3_kip279_5brokers > $OUT/ladder.log 2>&1
python3 -c "
import json
for l in open('gpurun_out/ladder.jsonl'):
    d=json.loads(l); print(d['name'], d['verdict'], d['distinct'], round(d['seconds_total'],5), round(d['seconds_expand'],5))"
