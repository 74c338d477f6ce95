#!/bin/bash
# round 5, call 9b: Kip279 / Kip320 7/1/1/0 under orbit counting, four runs each, twice — the 1 ms / 23 ms searches (call_9c.sh: fixed)
cd "${GRAFT_REPO_ROOT:-.}"
export KMC_NO_TORCH=1
for i in 1 2; do for w in "Kip279 7 1 1 0" "Kip320 7 1 1 0"; do
  echo "== $w"; timeout 120 python tools/sym_ab.py $w 4 24 2>&1 | grep -E '"symmetry": true' | tail -3 | cut -c1-120
done; done
