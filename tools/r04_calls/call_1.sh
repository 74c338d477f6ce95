#!/bin/bash
# round 4, call 1: fresh rocprof evidence on the shipped code for every kernel a bench line quotes.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/r04_calls/call_1.sh'
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04_1; mkdir -p $O
export KMC_NO_TORCH=1
( export KMC_BENCH_TABLE=$((1<<31)) KMC_BENCH_FRONTIER=$((1<<29))
  for lv in 10 14 17; do
    timeout 300 python bench.py --workload Kip320,7,8,8,3 --level-budget $lv --symmetry --no-cpu-baseline --steps 1 --warmup 0 > $O/config5_sym_L$lv.json 2> $O/config5_sym_L$lv.err
  done
  cat $O/config5_sym_L10.json $O/config5_sym_L14.json $O/config5_sym_L17.json > $O/config5_orbit_counting.jsonl
  PROFILE_BENCH_ARGS="--workload Kip320,7,8,8,3 --level-budget 10" bash tools/profile.sh r04_config5 > $O/profile_config5.log 2>&1; tail -1 $O/profile_config5.log )
PROFILE_BENCH_ARGS=--symmetry bash tools/profile.sh r04_sym > $O/profile_sym.log 2>&1; tail -1 $O/profile_sym.log
PROFILE_BENCH_ARGS="--workload Kip279,5,2,2,1" bash tools/profile.sh r04_config4 > $O/profile_config4.log 2>&1; tail -1 $O/profile_config4.log
bash tools/profile.sh r04 > $O/profile_plain.log 2>&1; tail -1 $O/profile_plain.log
unset KMC_NO_TORCH
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
