#!/bin/bash
# Round-2 GPU call 7: final validation — the -m gpu suite, smoke(), the default bench line, the configuration ladder
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02c7
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/gpu_tests.txt 2>&1
echo "pytest rc=$?" >> $OUT/gpu_tests.txt
tail -22 $OUT/gpu_tests.txt | cut -c1-220
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; cut -c1-700 $OUT/bench.json; tail -c 900 $OUT/bench.json
rm -f gpurun_out/ladder.jsonl
timeout 900 python tools/run_ladder.py config0_idsequence config1_finite_replicated_log config2_headline config3_kip279_5brokers config4_kip320_7brokers_log8_levels stretch_kip320_3_6_6_3 > $OUT/ladder.log 2>&1
cp gpurun_out/ladder.jsonl $OUT/ladder.jsonl
cut -c1-420 $OUT/ladder.jsonl
