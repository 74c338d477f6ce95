#!/bin/bash
# round 6, call 6: does WHERE an 8 GiB table lies in the HBM change the random-access rate?  randbench (modes 1, 7) on the k-th of
# several 8 GiB allocations, fresh processes; then the headline with the table at the k-th allocation
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_6; mkdir -p $O
for rep in 1 2 3; do for k in 0 1 2 5 10; do
  echo "[process $rep, table = allocation $k]" | tee -a $O/randbench_placement.txt
  RANDBENCH_SKIP=$k RANDBENCH_MODES=1,7 RANDBENCH_MAX_LOG2=30 tools/membench/randbench 0 30 2>&1 | grep -v "^# table" | tee -a $O/randbench_placement.txt
done; done
export KMC_NO_TORCH=1 KMC_VERBOSE=1
python - <<'PY' 2>&1 | grep -v "^\[kmc\] spec" | tee -a $O/headline_placement.txt
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import kafka_specification_amd as kmc
from kafka_specification_amd.configs import HEADLINE
hip = ctypes.CDLL("libamdhip64.so")
keep = []
for k in range(8):
    with kmc.ModelChecker(kmc.CheckerConfig(**HEADLINE, table_capacity=1 << 30, frontier_capacity=1 << 26)) as mc:
        mc.run()
        ks = [mc.run().seconds_expand * 1e3 for _ in range(3)]
        print(f"handle {k} ({len(keep)} x 4 GiB held elsewhere): k_expand {' '.join('%.2f' % x for x in ks)} ms", flush=True)
    p = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(4 << 30)) == 0     # shift where the next handle's buffers land
    keep.append(p)
PY
