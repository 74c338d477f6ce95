#!/bin/bash
# round 5, call 8: the representative at three and four replicas among the SORTED images (rank + one run-time permutation:
# -DKMC_SYMM_UNROLLED_MAX=2 moves the device's threshold; counts do not depend on the choice of representative) against the
# smallest of the N! unrolled images
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r05_8; mkdir -p $O
export KMC_NO_TORCH=1
for rep in a b; do
  timeout 200 python bench.py --symmetry --no-cpu-baseline --steps 10 --warmup 2 > $O/head_sym_unrolled_$rep.json 2> $O/head_sym_unrolled_$rep.err
  KMC_JIT_DEFINES=-DKMC_SYMM_UNROLLED_MAX=2 timeout 200 python bench.py --symmetry --no-cpu-baseline --steps 10 --warmup 2 > $O/head_sym_sorted_$rep.json 2> $O/head_sym_sorted_$rep.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r05_8/*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), "ms/step %.2f kernel %.2f golden %s" % (j["ms_per_step"], 1e3 * j["roofline"]["kernel_seconds_per_step"], j["config"]["matches_oracle_golden"]))
PY
for w in "Kip320 4 2 2 1" "Kip101 4 2 1 2"; do
  echo "== $w unrolled"; timeout 120 python tools/sym_ab.py $w 4 24 2>&1 | grep -E '"symmetry": true|counts' | tail -3
  echo "== $w sorted"; KMC_JIT_DEFINES=-DKMC_SYMM_UNROLLED_MAX=2 timeout 120 python tools/sym_ab.py $w 4 24 2>&1 | grep -E '"symmetry": true|counts' | tail -3
done
