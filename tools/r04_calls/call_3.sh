#!/bin/bash
# round 4, call 3: first GPU execution of this round's new tests, then fresh rocprofv3 evidence on the final device code
# (kmc_device.h changed: KMC_FLAG_INV_ONLY in every k_expand) for every kernel a bench line quotes.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/r04_calls/call_3.sh'
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04_3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_oracle_r_successors.py -q -x -n 4 > $O/t_successors.log 2>&1; tail -2 $O/t_successors.log
timeout 900 python -m pytest tests/test_gpu_symmetry.py -q -n 4 -k "logical_shards or level_budget_checks or refuses_traces or eight_logical" > $O/t_sym_shards.log 2>&1; tail -2 $O/t_sym_shards.log
timeout 900 python -m pytest tests/test_gpu_native_exchange_threads.py -q -n 2 -k "orbit" > $O/t_sym_threads.log 2>&1; tail -2 $O/t_sym_threads.log
timeout 900 python -m pytest tests/test_gpu_zz_beyond_the_exact_oracle.py -q -k "exact_orbit or first_violation or orbit_oracle" > $O/t_exact.log 2>&1; tail -2 $O/t_exact.log
export KMC_NO_TORCH=1
bash tools/profile.sh r04 > $O/profile_plain.log 2>&1; tail -1 $O/profile_plain.log
PROFILE_BENCH_ARGS=--symmetry bash tools/profile.sh r04_sym > $O/profile_sym.log 2>&1; tail -1 $O/profile_sym.log
PROFILE_BENCH_ARGS="--workload Kip279,5,2,2,1" bash tools/profile.sh r04_config4 > $O/profile_config4.log 2>&1; tail -1 $O/profile_config4.log
( export KMC_BENCH_TABLE=$((1<<31)) KMC_BENCH_FRONTIER=$((1<<29))
  PROFILE_BENCH_ARGS="--workload Kip320,7,8,8,3 --level-budget 10" bash tools/profile.sh r04_config5 > $O/profile_config5.log 2>&1; tail -1 $O/profile_config5.log
  for lv in 10 14 17; do
    timeout 300 python bench.py --workload Kip320,7,8,8,3 --level-budget $lv --symmetry --no-cpu-baseline --steps 1 --warmup 0 > $O/config5_sym_L$lv.json 2> $O/config5_sym_L$lv.err
  done
  cat $O/config5_sym_L10.json $O/config5_sym_L14.json $O/config5_sym_L17.json > $O/config5_orbit_counting.jsonl
  timeout 300 python bench.py --workload Kip320,7,8,8,3 --level-budget 10 --no-cpu-baseline --steps 3 --warmup 1 > $O/config5_level_budget.json 2> $O/config5_level_budget.err )
timeout 300 python bench.py --workload Kip279,5,2,2,1 --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_config4.json 2> $O/bench_config4.err
unset KMC_NO_TORCH
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
