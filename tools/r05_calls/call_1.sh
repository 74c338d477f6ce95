#!/bin/bash
# round 5, call 1: A/B of the mode split (k_expand one kernel per mode, KmcArgsLocal) and of FULL leaves at seven brokers against
# round 4's kernels ON THE SAME BOX.  ab_old/ = the tree at e9f48f4 (machine code of round 4's four profiled objects) with its
# own libkmc.so and cache; the new tree's objects are prebuilt (tools/precompile_some.py).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/r05_calls/call_1.sh'
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r05_1; mkdir -p $O
export KMC_NO_TORCH=1
B="--no-cpu-baseline --no-cold-start --steps 10 --warmup 2"
one() {  # tag dir extra-env...
  tag=$1; dir=$2; shift 2
  ( cd $dir; env "$@" timeout 300 python bench.py $B > $O/$tag.json 2> $O/$tag.err; cut -c1-200 $O/$tag.json )
}
if [ -d ab_old ]; then one old_a ab_old; fi
one new_a .
if [ -d ab_old ]; then one old_b ab_old; fi
one new_b .
# config 5 alone: mode split without full leaves, with full leaves, old
C5="--workload Kip320,7,8,8,3 --level-budget 10 --no-cpu-baseline --steps 5 --warmup 1"
export KMC_BENCH_TABLE=$((1<<31)) KMC_BENCH_FRONTIER=$((1<<29))
( cd ab_old && timeout 300 python bench.py $C5 > $O/c5_old.json 2> $O/c5_old.err )
timeout 300 python bench.py $C5 > $O/c5_full_leaves.json 2> $O/c5_full_leaves.err
KMC_JIT_DEFINES=-DKMC_FULL_LEAVES_MIN_INSTANCES=1000000 timeout 300 python bench.py $C5 > $O/c5_mode_split_only.json 2> $O/c5_mode_split_only.err
timeout 300 python bench.py $C5 --symmetry > $O/c5_sym_full_leaves.json 2> $O/c5_sym_full_leaves.err
( cd ab_old && timeout 300 python bench.py $C5 --symmetry > $O/c5_sym_old.json 2> $O/c5_sym_old.err )
unset KMC_BENCH_TABLE KMC_BENCH_FRONTIER
python - <<'PY'
import json, glob, os
O = os.environ.get("O", "gpurun_out/r05_1")
for f in sorted(glob.glob("gpurun_out/r05_1/*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    r, c = j.get("roofline", {}), j.get("config", {})
    line = [os.path.basename(f), "ms/step %.2f" % j["ms_per_step"], "kernel ms %.2f" % (1e3 * r.get("kernel_seconds_per_step", 0)),
            "golden", c.get("matches_oracle_golden"), "distinct", c.get("distinct_states")]
    oc = j.get("orbit_counting")
    if oc: line += ["| sym ms/step %.2f kernel %.2f same %s" % (oc["ms_per_step"], 1e3 * oc.get("kernel_seconds_per_step", 0), oc.get("every_count_equals_the_plain_run"))]
    for k, v in (j.get("baseline_configs") or {}).items():
        line += ["| %s ms/step %.2f kernel %.2f golden %s %s" % (k[:7], v.get("ms_per_step", -1), 1e3 * v.get("kernel_seconds_per_step", 0), v.get("matches_oracle_golden"), v.get("error", ""))]
    print(*line)
PY
# correctness of the three kernels where the new paths run: ENUM + k_inv state by state against the executed reference at the
# headline's and config 5's constants (plain and orbit counting), whole level sets at seven and eight replicas (full leaves),
# SHARDED with the meta plane at 7 / 8 / 8 / 3 on eight logical shards, the small loopback shards, the last-frontier check
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python -m pytest tests/test_gpu_oracle_r_successors.py -x -q -k "Kip320-3-6-6-2 or Kip320-7-8-8-3" > $O/pytest_successors.log 2>&1; tail -3 $O/pytest_successors.log
timeout 600 python -m pytest tests/test_gpu_zzz_oracle_r_wide.py -x -q -k "7/ or 8/" > $O/pytest_wide.log 2>&1; tail -3 $O/pytest_wide.log
timeout 600 python -m pytest tests/test_gpu_sharded_and_traces.py -x -q -k "test_loopback_shards_match_oracle or test_level_limit_still_checks or test_baseline_config5_seven" > $O/pytest_sharded.log 2>&1; tail -3 $O/pytest_sharded.log
