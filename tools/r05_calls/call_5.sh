#!/bin/bash
# round 5, call 5: the representative at five to seven replicas chosen by RANK + one run-time permutation (KmcSymm::canon_sorted)
# instead of the odd-even transposition network of masked exchanges
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r05_5; mkdir -p $O
export KMC_NO_TORCH=1
for m in Kip320 Kip279; do
  echo "== $m 7/1/1/0"; timeout 120 python tools/sym_ab.py $m 7 1 1 0 2 24 2>&1 | tail -5
done
timeout 200 python bench.py --workload Kip279,5,2,2,1 --symmetry --no-cpu-baseline --steps 5 --warmup 1 > $O/c4_sym.json 2> $O/c4_sym.err
export KMC_BENCH_TABLE=$((1<<31)) KMC_BENCH_FRONTIER=$((1<<29))
for lv in 10 14 17; do
  timeout 300 python bench.py --workload Kip320,7,8,8,3 --level-budget $lv --symmetry --no-cpu-baseline --steps 2 --warmup 1 > $O/c5_sym_L$lv.json 2> $O/c5_sym_L$lv.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r05_5/*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    r, c = j.get("roofline", {}), j.get("config", {})
    print(os.path.basename(f), "ms/step %.2f" % j["ms_per_step"], "kernel ms %.2f" % (1e3 * r.get("kernel_seconds_per_step", 0)),
          "golden", c.get("matches_oracle_golden"), "distinct", c.get("distinct_states"), "generated", c.get("states_generated"))
PY
