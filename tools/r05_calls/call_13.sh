#!/bin/bash
# round 5, call 13: the deferred probe (the search's flush issues its batch's first probe and completes the batch at the next
# flush: wide states, W >= KMC_DEFER_MIN_WORDS) — forced onto small configurations against the oracle, then config 5
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r05_13; mkdir -p $O
export KMC_JIT_DEFINES=-DKMC_DEFER_MIN_WORDS=1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_symmetry.py -x -q -k "reports_the_plain_counts or counterexample_trace_is_a_real or violations_under_continue" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_sharded_and_traces.py -x -q -k "counterexample_trace_is_a_real_behaviour or checkpoint_and_recover or overfull or frontier_and_table" 2>&1 | tail -3
unset KMC_JIT_DEFINES
export KMC_NO_TORCH=1 KMC_BENCH_TABLE=$((1<<31)) KMC_BENCH_FRONTIER=$((1<<29))
for rep in a b; do
  C5="--workload Kip320,7,8,8,3 --level-budget 10 --no-cpu-baseline --steps 5 --warmup 1"
  timeout 300 python bench.py $C5 > $O/c5_defer_$rep.json 2> $O/c5_defer_$rep.err
  KMC_JIT_DEFINES=-DKMC_DEFER_MIN_WORDS=100 timeout 300 python bench.py $C5 > $O/c5_nodefer_$rep.json 2> $O/c5_nodefer_$rep.err
  timeout 300 python bench.py --workload Kip320,7,8,8,3 --level-budget 14 --symmetry --no-cpu-baseline --steps 2 --warmup 1 > $O/c5_sym_L14_defer_$rep.json 2> $O/c5_sym_L14_defer_$rep.err
  KMC_JIT_DEFINES=-DKMC_DEFER_MIN_WORDS=100 timeout 300 python bench.py --workload Kip320,7,8,8,3 --level-budget 14 --symmetry --no-cpu-baseline --steps 2 --warmup 1 > $O/c5_sym_L14_nodefer_$rep.json 2> $O/c5_sym_L14_nodefer_$rep.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r05_13/*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    print(os.path.basename(f), "ms/step %.2f kernel %.2f golden %s" % (j["ms_per_step"], 1e3 * j["roofline"]["kernel_seconds_per_step"], j["config"]["matches_oracle_golden"]))
PY
