#!/bin/bash
# round 3, final call 4 (the device sources are final): a ladder of configurations with and without orbit counting, the
# profile passes over the plain headline and the orbit-counting search, the bench line quoting them, the orbit-counting suite
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/f4; mkdir -p $O
( export KMC_NO_TORCH=1
  : > $O/sym_ladder.jsonl
  timeout 120 python tools/sym_ab.py Kip320 3 6 6 2 3 30 >> $O/sym_ladder.jsonl 2>&1
  timeout 120 python tools/sym_ab.py Kip279 5 2 2 1 3 28 >> $O/sym_ladder.jsonl 2>&1
  timeout 120 python tools/sym_ab.py Kip320 5 1 1 1 3 26 >> $O/sym_ladder.jsonl 2>&1
  timeout 200 python tools/sym_ab.py KafkaTruncateToHighWatermark 6 1 1 1 3 29 >> $O/sym_ladder.jsonl 2>&1
  KMC_AB_FP128=1 timeout 300 python tools/sym_ab.py Kip320 3 6 6 3 2 33 >> $O/sym_ladder.jsonl 2>&1 )
grep -E "run\": [12]|equal" $O/sym_ladder.jsonl | cut -c1-260
bash tools/profile.sh r03h > $O/profile_plain.log 2>&1; tail -1 $O/profile_plain.log
PROFILE_BENCH_ARGS=--symmetry bash tools/profile.sh r03h_sym > $O/profile_sym.log 2>&1; tail -1 $O/profile_sym.log
cp gpurun_out/prof_r03h/pmc_summary.json profiles/r03_pmc_summary.json; cp gpurun_out/prof_r03h/summary.json profiles/r03_summary.json
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
KMC_BENCH_TABLE=$((1<<28)) KMC_BENCH_FRONTIER=$((1<<25)) timeout 600 python bench.py --workload Kip279,5,2,2,1 --no-cpu-baseline --steps 3 > $O/bench_config4.json 2> $O/bench_config4.err
timeout 600 python -m pytest tests/test_gpu_symmetry.py -q -n 4 > $O/tests_sym.log 2>&1; echo "sym tests rc=$?" >> $O/tests_sym.log; tail -3 $O/tests_sym.log
