#!/bin/bash
# (KMC_TABLE_VMM / KMC_FRONTIER_VMM were the hooks of this experiment in kmc_open; what came of it is KmcEngine's seen_set_alloc and
# KMC_SEEN_SET_CHUNK_LOG2 - csrc/kmc_engine_core.cpp - with which these A/Bs are: chunks = the default, hipMalloc = KMC_SEEN_SET_CHUNK_LOG2=0)
# round 6, call 21: does the size of the seen-set's physically contiguous pieces decide the headline's level?  The table from explicit
# chunks (KMC_TABLE_VMM=log2 bytes: HIP's virtual-memory API, an experiment's hook in kmc_open) of 2 MiB / 64 MiB / 1 GiB / 2 GiB,
# against hipMalloc; four handles each inside one process, all kept alive.
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_21; mkdir -p $O
export KMC_NO_TORCH=1
for slots in $((3<<29)) $((1<<30)); do for vm in none 31 30 26 21 none 31; do
  ( [ $vm != none ] && export KMC_TABLE_VMM=$vm; timeout 600 python - $slots $vm <<'PY' 2>&1 | grep -v "^\[kmc\] spec" | tee -a $O/vmm.txt
import os, sys, time
sys.path.insert(0, os.getcwd())
import kafka_specification_amd as kmc
from kafka_specification_amd.configs import HEADLINE
slots = int(sys.argv[1])
alive, line = [], []
for k in range(4):
    t0 = time.time()
    mc = kmc.ModelChecker(kmc.CheckerConfig(**HEADLINE, table_capacity=slots, frontier_capacity=1 << 26)).__enter__()
    t_open = time.time() - t0
    r = mc.run()
    assert r.distinct == 279753922, r.distinct
    ks = [mc.run().seconds_expand * 1e3 for _ in range(2)]
    line.append("%.2f" % min(ks))
    alive.append(mc)
print(f"table {slots / 2**30:.2f} x 2^30 slots, chunks 2^{sys.argv[2]}: k_expand per handle {' '.join(line)} ms (last open {t_open:.2f} s)", flush=True)
PY
  )
done; done
