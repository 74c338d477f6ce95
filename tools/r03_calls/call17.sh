#!/bin/bash
# round 3, GPU call 17: the grouped replica-major layout + kind-major walk on BASELINE configs 4 and 5 against the tight
# layout + instance-major walk (KMC_LAYOUT=tight), and the headline again (one replica per word, now through the generic
# accessors)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/c17; mkdir -p $O; rm -f gpurun_out/sweep.log
export KMC_NO_TORCH=1
tools/sweep.sh "headline||" "headline_again||"
KMC_LAYOUT=rmg tools/sweep.sh "headline_grouped||"
cp gpurun_out/sweep.log $O/sweep.log
for lay in auto tight; do
  KMC_LAYOUT=$lay KMC_BENCH_TABLE=$((1<<31)) KMC_BENCH_FRONTIER=$((1<<29)) timeout 900 python bench.py --workload Kip320,7,8,8,3 --level-budget 10 --no-cpu-baseline --steps 1 --warmup 1 > $O/config5_$lay.json 2> $O/config5_$lay.err
  python - <<PY
import json
try:
    j = json.load(open("$O/config5_$lay.json")); print("config5 $lay", j["ms_per_step"], j["value"]/1e9, j["config"]["distinct_states"], j["config"]["states_generated"], j["roofline"]["kernel_seconds_per_step"])
except Exception as e: print("config5 $lay FAILED", e)
PY
  KMC_LAYOUT=$lay KMC_BENCH_TABLE=$((1<<29)) timeout 900 python bench.py --workload Kip279,5,2,2,1 --no-cpu-baseline --steps 2 --warmup 1 > $O/config4_$lay.json 2> $O/config4_$lay.err
  python - <<PY
import json
try:
    j = json.load(open("$O/config4_$lay.json")); print("config4 $lay", j["ms_per_step"], j["value"]/1e9, j["config"]["distinct_states"], j["config"]["states_generated"], j["roofline"]["kernel_seconds_per_step"])
except Exception as e: print("config4 $lay FAILED", e)
PY
done
