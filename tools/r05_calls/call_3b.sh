#!/bin/bash
# round 5, call 3b: the default bench line again, now that profiles/r05_*pmc_summary.json exist (the line quotes `traffic` only
# from a summary measured on the machine code it runs); config 5 under orbit counting in the steady state (one warm-up step)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05_3b; mkdir -p $O
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
export KMC_NO_TORCH=1 KMC_BENCH_TABLE=$((1<<31)) KMC_BENCH_FRONTIER=$((1<<29))
for lv in 10 14 17; do
  timeout 300 python bench.py --workload Kip320,7,8,8,3 --level-budget $lv --symmetry --no-cpu-baseline --steps 2 --warmup 1 > $O/config5_sym_L$lv.json 2> $O/config5_sym_L$lv.err
done
cat $O/config5_sym_L10.json $O/config5_sym_L14.json $O/config5_sym_L17.json > $O/config5_orbit_counting_warm.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r05_3b/config5_orbit_counting_warm.jsonl"):
    j = json.loads(l); print(j["config"]["level_budget"], "ms/step %.1f kernel %.1f" % (j["ms_per_step"], 1e3 * j["roofline"]["kernel_seconds_per_step"]), j["config"]["distinct_states"], j["config"]["matches_oracle_golden"])
PY
