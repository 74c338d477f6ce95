#!/bin/bash
# round 6, call 13: WHICH of a handle's buffers carries the headline's mode?  One handle; before every search ONE buffer (the
# seen-set / frontier 0 / frontier 1 / the control blocks) is moved to another place in the HBM (KMC_DEBUG_REALLOC, an experiment's
# hook in kmc_run), the others stay.
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_13; mkdir -p $O
export KMC_NO_TORCH=1
for slots in $((3<<29)) $((1<<30)); do for which in none ctl f0 f1 table none; do
  KMC_DEBUG_REALLOC=$which python - $slots $which <<'PY' 2>&1 | grep -v "^\[kmc\] spec" | tee -a $O/which.txt
import os, sys
sys.path.insert(0, os.getcwd())
import kafka_specification_amd as kmc
from kafka_specification_amd.configs import HEADLINE
slots = int(sys.argv[1])
with kmc.ModelChecker(kmc.CheckerConfig(**HEADLINE, table_capacity=slots, frontier_capacity=1 << 26)) as mc:
    ks = []
    for k in range(10):
        r = mc.run()
        assert r.distinct == 279753922
        ks.append("%.2f" % (r.seconds_expand * 1e3))
    print(f"table {slots / 2**30:.2f} x 2^30, moved before every search: {sys.argv[2]:6s} k_expand {' '.join(ks)}", flush=True)
PY
done; done
