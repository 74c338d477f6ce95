#!/bin/bash
# round 5, call 1b: where does orbit counting at seven replicas go wrong — full leaves, or the mode split?
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r05_1b; mkdir -p $O
export KMC_NO_TORCH=1
for m in Kip320 Kip279; do
  echo "== $m 7/1/1/0 full leaves"; timeout 120 python tools/sym_ab.py $m 7 1 1 0 2 24 2>&1 | tail -5
  echo "== $m 7/1/1/0 mode split only"; KMC_JIT_DEFINES=-DKMC_FULL_LEAVES_MIN_INSTANCES=1000000 timeout 120 python tools/sym_ab.py $m 7 1 1 0 2 24 2>&1 | tail -5
done
export KMC_BENCH_TABLE=$((1<<31)) KMC_BENCH_FRONTIER=$((1<<29))
C5="--workload Kip320,7,8,8,3 --level-budget 10 --no-cpu-baseline --steps 3 --warmup 1 --symmetry"
KMC_JIT_DEFINES=-DKMC_FULL_LEAVES_MIN_INSTANCES=1000000 timeout 300 python bench.py $C5 > $O/c5_sym_mode_split_only.json 2> $O/c5_sym_mode_split_only.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r05_1b/c5_sym_mode_split_only.json").read().strip().splitlines()[-1])
print("c5 sym mode split only: ms/step %.2f kernel %.2f golden %s generated %d" % (j["ms_per_step"], 1e3 * j["roofline"]["kernel_seconds_per_step"], j["config"]["matches_oracle_golden"], j["config"]["states_generated"]))
PY
unset KMC_NO_TORCH KMC_BENCH_TABLE KMC_BENCH_FRONTIER
timeout 600 python -m pytest tests/test_gpu_sharded_and_traces.py -x -q -k "test_loopback_shards_match_oracle or test_level_limit_still_checks or test_baseline_config5_seven" 2>&1 | tail -3
