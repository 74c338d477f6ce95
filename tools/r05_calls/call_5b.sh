#!/bin/bash
# round 5, call 5b: same-box A/B of the sorted image at seven brokers — rank + one run-time permutation (this tree) against the
# odd-even transposition network of masked exchanges (ab_net/: this tree with kmc_symm.h of the commit before)
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r05_5b; mkdir -p $O
export KMC_NO_TORCH=1 KMC_BENCH_TABLE=$((1<<31)) KMC_BENCH_FRONTIER=$((1<<29))
for rep in a b; do
  for lv in 14 17; do
    C5="--workload Kip320,7,8,8,3 --level-budget $lv --symmetry --no-cpu-baseline --steps 2 --warmup 1"
    timeout 300 python bench.py $C5 > $O/rank_L${lv}_$rep.json 2> $O/rank_L${lv}_$rep.err
    ( cd ab_net && timeout 300 python bench.py $C5 > $O/network_L${lv}_$rep.json 2> $O/network_L${lv}_$rep.err )
  done
done
unset KMC_BENCH_TABLE KMC_BENCH_FRONTIER
timeout 200 python bench.py --workload Kip279,5,2,2,1 --symmetry --no-cpu-baseline --steps 5 --warmup 1 > $O/rank_c4.json 2> $O/rank_c4.err
( cd ab_net && timeout 200 python bench.py --workload Kip279,5,2,2,1 --symmetry --no-cpu-baseline --steps 5 --warmup 1 > $O/network_c4.json 2> $O/network_c4.err )
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r05_5b/*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    r, c = j.get("roofline", {}), j.get("config", {})
    print(os.path.basename(f), "ms/step %.2f" % j["ms_per_step"], "kernel ms %.2f" % (1e3 * r.get("kernel_seconds_per_step", 0)), "golden", c.get("matches_oracle_golden"))
PY
