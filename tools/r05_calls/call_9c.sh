#!/bin/bash
# round 5, call 9c: orbit counting at seven replicas with the invalid lanes of a partial flush zeroed before the representative
# search (their garbage used to send whole waves through the 5,039-step walk) — repeated runs in one process, then config 5
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r05_9c; mkdir -p $O
export KMC_NO_TORCH=1
for i in 1 2; do for w in "Kip279 7 1 1 0" "Kip320 7 1 1 0"; do
  echo "== $w"; timeout 120 python tools/sym_ab.py $w 4 24 2>&1 | grep -E '"symmetry": true|counts' | tail -4 | cut -c1-125
done; done
timeout 200 python bench.py --workload Kip279,5,2,2,1 --symmetry --no-cpu-baseline --steps 5 --warmup 1 > $O/c4_sym.json 2> $O/c4_sym.err
export KMC_BENCH_TABLE=$((1<<31)) KMC_BENCH_FRONTIER=$((1<<29))
for rep in a b; do for lv in 10 14 17; do
  timeout 300 python bench.py --workload Kip320,7,8,8,3 --level-budget $lv --symmetry --no-cpu-baseline --steps 2 --warmup 1 > $O/c5_sym_L${lv}_$rep.json 2> $O/c5_sym_L${lv}_$rep.err
done; done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r05_9c/*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), "ms/step %.2f kernel %.2f golden %s" % (j["ms_per_step"], 1e3 * j["roofline"]["kernel_seconds_per_step"], j["config"]["matches_oracle_golden"]))
PY
