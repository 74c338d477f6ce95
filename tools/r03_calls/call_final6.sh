#!/bin/bash
# round 3, final call 6: orbit counting at seven replicas (BASELINE config 5 among them) — the orbit-counting suite, config 5
# over level budgets with and without it, then (the device sources changed: unrolled forms discarded beyond four replicas) the
# profile passes and the bench line once more
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/f6; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_symmetry.py -q -n 4 > $O/tests_sym.log 2>&1; echo "sym tests rc=$?" >> $O/tests_sym.log; tail -4 $O/tests_sym.log
( export KMC_NO_TORCH=1 KMC_BENCH_TABLE=$((1<<31)) KMC_BENCH_FRONTIER=$((1<<29))
  for lv in 10 14 17; do
    timeout 300 python bench.py --workload Kip320,7,8,8,3 --level-budget $lv --symmetry --no-cpu-baseline --steps 1 --warmup 0 > $O/config5_sym_L$lv.json 2> $O/config5_sym_L$lv.err
    python - $O/config5_sym_L$lv.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d["config"]
    print({k:d[k] for k in ("value","ms_per_step")}, c["distinct_states"], c["states_generated"], c["depth"], c["verdict"], d["roofline"]["kernel_seconds_per_step"])
except Exception as e: print("no line:", e)
PY
  done )
bash tools/profile.sh r03i > $O/profile_plain.log 2>&1; tail -1 $O/profile_plain.log
PROFILE_BENCH_ARGS=--symmetry bash tools/profile.sh r03i_sym > $O/profile_sym.log 2>&1; tail -1 $O/profile_sym.log
cp gpurun_out/prof_r03i/pmc_summary.json profiles/r03_pmc_summary.json; cp gpurun_out/prof_r03i/summary.json profiles/r03_summary.json
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
