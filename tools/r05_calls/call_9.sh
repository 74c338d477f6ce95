#!/bin/bash
# round 5, call 9: tied neighbours checked by table reads only (KmcSymm::trade_is_identity) and the sorted form from four replicas on
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r05_9; mkdir -p $O
export KMC_NO_TORCH=1
for w in "Kip320 7 1 1 0" "Kip279 7 1 1 0" "Kip320 4 2 2 1"; do
  echo "== $w"; timeout 120 python tools/sym_ab.py $w 3 24 2>&1 | grep -E '"symmetry": true|counts' | tail -3
done
echo "== Kip101 4 2 1 2"; timeout 120 python tools/sym_ab.py Kip101 4 2 1 2 3 26 2>&1 | grep -E '"symmetry": true|counts' | tail -3
timeout 200 python bench.py --workload Kip279,5,2,2,1 --symmetry --no-cpu-baseline --steps 5 --warmup 1 > $O/c4_sym.json 2> $O/c4_sym.err
export KMC_BENCH_TABLE=$((1<<31)) KMC_BENCH_FRONTIER=$((1<<29))
for rep in a b; do for lv in 14 17; do
  timeout 300 python bench.py --workload Kip320,7,8,8,3 --level-budget $lv --symmetry --no-cpu-baseline --steps 2 --warmup 1 > $O/c5_sym_L${lv}_$rep.json 2> $O/c5_sym_L${lv}_$rep.err
done; done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r05_9/*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), "ms/step %.2f kernel %.2f golden %s" % (j["ms_per_step"], 1e3 * j["roofline"]["kernel_seconds_per_step"], j["config"]["matches_oracle_golden"]))
PY
