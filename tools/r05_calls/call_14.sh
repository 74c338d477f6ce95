#!/bin/bash
# round 5, call 14 (= call 10 on the tree with the deferred probe): the evidence on the LAST tree:
# the whole -m gpu suite as the driver runs it with the objects the box had to specialise itself listed (build() should leave
# none), smoke, rocprofv3 summaries for every kernel a line quotes, config 5 under orbit counting, and — last, so that it finds
# the summaries of its own machine code — the default bench line.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/r05_calls/call_14.sh'
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05_14; mkdir -p $O
ls kafka_specification_amd/kmc_cache | sort > $O/cache_before.txt
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
ls kafka_specification_amd/kmc_cache | sort > $O/cache_after.txt
echo "specialised on the box:"; comm -13 $O/cache_before.txt $O/cache_after.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
export KMC_NO_TORCH=1
bash tools/profile.sh r05 > $O/profile_plain.log 2>&1; tail -1 $O/profile_plain.log
PROFILE_BENCH_ARGS=--symmetry bash tools/profile.sh r05_sym > $O/profile_sym.log 2>&1; tail -1 $O/profile_sym.log
PROFILE_BENCH_ARGS="--workload Kip279,5,2,2,1" bash tools/profile.sh r05_config4 > $O/profile_config4.log 2>&1; tail -1 $O/profile_config4.log
( export KMC_BENCH_TABLE=$((1<<31)) KMC_BENCH_FRONTIER=$((1<<29))
  PROFILE_BENCH_ARGS="--workload Kip320,7,8,8,3 --level-budget 10" bash tools/profile.sh r05_config5 > $O/profile_config5.log 2>&1; tail -1 $O/profile_config5.log
  for lv in 10 14 17; do
    timeout 300 python bench.py --workload Kip320,7,8,8,3 --level-budget $lv --symmetry --no-cpu-baseline --steps 1 --warmup 0 > $O/config5_sym_L$lv.json 2> $O/config5_sym_L$lv.err
    timeout 300 python bench.py --workload Kip320,7,8,8,3 --level-budget $lv --symmetry --no-cpu-baseline --steps 2 --warmup 1 > $O/config5_sym_warm_L$lv.json 2> $O/config5_sym_warm_L$lv.err
  done
  cat $O/config5_sym_L10.json $O/config5_sym_L14.json $O/config5_sym_L17.json > $O/config5_orbit_counting.jsonl
  cat $O/config5_sym_warm_L10.json $O/config5_sym_warm_L14.json $O/config5_sym_warm_L17.json > $O/config5_orbit_counting_warm.jsonl )
unset KMC_NO_TORCH
# the bench reads profiles/r05_*pmc_summary.json: put this call's summaries where it looks, then run it as the driver does
for t in r05 r05_sym r05_config4 r05_config5; do for f in kernel_stats.csv summary.json pmc_summary.json; do cp gpurun_out/prof_$t/$f profiles/${t}_$f; done; done
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
