#!/bin/bash
# round 5, call 3: the evidence on the final tree — the whole -m gpu suite as the driver runs it, smoke, the default bench line
# (with its baseline_configs and cold_start legs), and rocprofv3 summaries for every kernel a line quotes (plain headline, orbit
# counting, BASELINE configs 4 and 5), config 5 under orbit counting over 10 / 14 / 17 levels.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/r05_calls/call_3.sh'
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05_3; mkdir -p $O
ls kafka_specification_amd/kmc_cache | wc -l > $O/cache_files_before.txt
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
export KMC_NO_TORCH=1
bash tools/profile.sh r05 > $O/profile_plain.log 2>&1; tail -1 $O/profile_plain.log
PROFILE_BENCH_ARGS=--symmetry bash tools/profile.sh r05_sym > $O/profile_sym.log 2>&1; tail -1 $O/profile_sym.log
PROFILE_BENCH_ARGS="--workload Kip279,5,2,2,1" bash tools/profile.sh r05_config4 > $O/profile_config4.log 2>&1; tail -1 $O/profile_config4.log
( export KMC_BENCH_TABLE=$((1<<31)) KMC_BENCH_FRONTIER=$((1<<29))
  PROFILE_BENCH_ARGS="--workload Kip320,7,8,8,3 --level-budget 10" bash tools/profile.sh r05_config5 > $O/profile_config5.log 2>&1; tail -1 $O/profile_config5.log
  for lv in 10 14 17; do
    timeout 300 python bench.py --workload Kip320,7,8,8,3 --level-budget $lv --symmetry --no-cpu-baseline --steps 1 --warmup 0 > $O/config5_sym_L$lv.json 2> $O/config5_sym_L$lv.err
  done
  cat $O/config5_sym_L10.json $O/config5_sym_L14.json $O/config5_sym_L17.json > $O/config5_orbit_counting.jsonl )
unset KMC_NO_TORCH
ls kafka_specification_amd/kmc_cache | wc -l > $O/cache_files_after.txt   # (must be what build() left: nothing was compiled on the box)
cat $O/cache_files_before.txt $O/cache_files_after.txt
