#!/bin/bash
# round 4, call 4: the rocprofv3 evidence, once more — from this call on the bench (torch's HIP runtime) and a profile run (system
# ROCm, KMC_NO_TORCH=1) load the SAME cached code object (the cache key no longer holds the runtime's build number), so a
# summary's kernel_code_sha256 is the identity of what the bench line runs.
#   /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tools/r04_calls/call_4.sh'
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04_4; mkdir -p $O
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
ls kafka_specification_amd/kmc_cache | wc -l > $O/cache_files_after.txt   # (must be what build() left: nothing was compiled on the box)
