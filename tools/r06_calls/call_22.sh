#!/bin/bash
# (KMC_TABLE_VMM / KMC_FRONTIER_VMM were the hooks of this experiment in kmc_open; what came of it is KmcEngine's seen_set_alloc and
# KMC_SEEN_SET_CHUNK_LOG2 - csrc/kmc_engine_core.cpp - with which these A/Bs are: chunks = the default, hipMalloc = KMC_SEEN_SET_CHUNK_LOG2=0)
# round 6, call 22: call 21's surprise - the seen-set mapped from 2 MiB chunks ran the headline at 28.5 ms in all four handles at 8 GiB,
# where one hipMalloc gives 30.5 - 31.6.  More samples: fresh processes, chunk sizes 2^21 .. 2^25, tables of 1.0 x and 1.5 x 2^30 slots.
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_22; mkdir -p $O
export KMC_NO_TORCH=1
for rep in 1 2 3; do for slots in $((1<<30)) $((3<<29)); do for vm in none 21 22 23 25; do
  ( [ $vm != none ] && export KMC_TABLE_VMM=$vm; timeout 600 python - $slots $vm $rep <<'PY' 2>&1 | grep -v "^\[kmc\]" | tee -a $O/vmm.txt
import os, sys, time
sys.path.insert(0, os.getcwd())
import kafka_specification_amd as kmc
from kafka_specification_amd.configs import HEADLINE
slots = int(sys.argv[1])
alive, line, opens = [], [], []
for k in range(3):
    t0 = time.time()
    mc = kmc.ModelChecker(kmc.CheckerConfig(**HEADLINE, table_capacity=slots, frontier_capacity=1 << 26)).__enter__()
    opens.append("%.2f" % (time.time() - t0))
    r = mc.run()
    assert r.distinct == 279753922, r.distinct
    rs = [mc.run() for _ in range(2)]
    line.append("%.2f (+%.2f clear)" % (min(x.seconds_expand for x in rs) * 1e3, min(x.seconds_clear for x in rs) * 1e3))
    alive.append(mc)
print(f"process {sys.argv[3]}, table {slots / 2**30:.2f} x 2^30, chunks 2^{sys.argv[2]}: k_expand {' | '.join(line)} ms; open {' '.join(opens)} s", flush=True)
PY
  )
done; done; done
