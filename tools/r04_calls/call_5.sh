#!/bin/bash
# round 4, call 5: the whole `-m gpu` suite as the driver runs it at round end, then the bench lines with the committed r04
# summaries in place (their roofline.traffic is quoted by machine-code identity).
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/r04_calls/call_5.sh'
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04_5; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -n 4 > $O/tests_gpu.log 2>&1; tail -4 $O/tests_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
timeout 300 python bench.py --workload Kip279,5,2,2,1 --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_config4.json 2> $O/bench_config4.err
KMC_BENCH_TABLE=$((1<<31)) KMC_BENCH_FRONTIER=$((1<<29)) timeout 300 python bench.py --workload Kip320,7,8,8,3 --level-budget 10 --no-cpu-baseline --steps 3 --warmup 1 > $O/config5_level_budget.json 2> $O/config5_level_budget.err
python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
ls kafka_specification_amd/kmc_cache | wc -l > $O/cache_files_after.txt
