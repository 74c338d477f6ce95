#!/bin/bash
# round 5, call 2: full leaves with the cross-lane pulls made opaque (call 1b: the compiler had sunk a ds_bpermute into a
# divergent branch) — counts and times at seven replicas, plain and orbit counting, 10 and 14 levels; the cold start's breakdown
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r05_2; mkdir -p $O
export KMC_NO_TORCH=1
for m in Kip320 Kip279; do
  echo "== $m 7/1/1/0 full leaves"; timeout 120 python tools/sym_ab.py $m 7 1 1 0 2 24 2>&1 | tail -5
done
export KMC_BENCH_TABLE=$((1<<31)) KMC_BENCH_FRONTIER=$((1<<29))
for lv in 10 14; do
  C5="--workload Kip320,7,8,8,3 --level-budget $lv --no-cpu-baseline --steps 3 --warmup 1"
  timeout 300 python bench.py $C5 --symmetry > $O/c5_sym_L${lv}_full_leaves.json 2> $O/c5_sym_L${lv}_full_leaves.err
  KMC_JIT_DEFINES=-DKMC_FULL_LEAVES_MIN_INSTANCES=1000000 timeout 300 python bench.py $C5 --symmetry > $O/c5_sym_L${lv}_mode_split_only.json 2> $O/c5_sym_L${lv}_mode_split_only.err
  ( cd ab_old && timeout 300 python bench.py $C5 --symmetry > $O/c5_sym_L${lv}_old.json 2> $O/c5_sym_L${lv}_old.err )
done
C5="--workload Kip320,7,8,8,3 --level-budget 10 --no-cpu-baseline --steps 5 --warmup 1"
timeout 300 python bench.py $C5 > $O/c5_full_leaves.json 2> $O/c5_full_leaves.err
KMC_JIT_DEFINES=-DKMC_FULL_LEAVES_MIN_INSTANCES=1000000 timeout 300 python bench.py $C5 > $O/c5_mode_split_only.json 2> $O/c5_mode_split_only.err
unset KMC_BENCH_TABLE KMC_BENCH_FRONTIER
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r05_2/*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    r, c = j.get("roofline", {}), j.get("config", {})
    print(os.path.basename(f), "ms/step %.2f" % j["ms_per_step"], "kernel ms %.2f" % (1e3 * r.get("kernel_seconds_per_step", 0)),
          "golden", c.get("matches_oracle_golden"), "distinct", c.get("distinct_states"), "generated", c.get("states_generated"))
PY
# what a CLI user waits for, and where it goes (kmc_timing)
for extra in "" "-notrace"; do
  for i in 1 2; do
    /usr/bin/time -f "wall %e s" kafka_specification_amd/tlc models/Kip320.tla -table $((1<<30)) -frontier $((1<<26)) -v $extra 2>&1 | grep -E "Wall time|wall |distinct states found" | tail -3
  done
done
# occupancy of the mode-split kernels: the search's kernel now fits 64 VGPRs (8 waves per SIMD) with 5 spilled
for w in 4 5 6 7 8; do
  D="-DKMC_MIN_WAVES=$w"; [ $w = 6 ] && D=""
  KMC_JIT_DEFINES=$D timeout 200 python bench.py --no-cpu-baseline --no-cold-start --no-baseline-configs --steps 10 --warmup 2 > $O/head_w$w.json 2> $O/head_w$w.err
  python - $w <<'PY'
import json, sys
j = json.loads(open("gpurun_out/r05_2/head_w%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
oc = j.get("orbit_counting", {})
print("waves", sys.argv[1], "plain ms/step %.2f kernel %.2f golden %s | sym ms/step %.2f kernel %.2f same %s" % (
    j["ms_per_step"], 1e3 * j["roofline"]["kernel_seconds_per_step"], j["config"]["matches_oracle_golden"],
    oc.get("ms_per_step", -1), 1e3 * oc.get("kernel_seconds_per_step", 0), oc.get("every_count_equals_the_plain_run")))
PY
done
