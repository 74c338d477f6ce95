#!/bin/bash
# round 6, call 12: call 11 showed that the headline's mode FOLLOWS THE HANDLE (where its buffers lie), in three levels (28.4 / 30.5 /
# 31.5 ms).  Suspect: the frontier's streams - 3 planes x 8 segments read, as many written - lie at power-of-two distances (planes
# 512 MiB apart, segments 64 MiB apart) and advance together: whether they fall onto the same DRAM channels depends on the physical
# placement.  Test: the same handles with a frontier capacity that is NOT a power of two (segments 2^23 + 64 k states).
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_12; mkdir -p $O
export KMC_NO_TORCH=1
python - <<'PY' 2>&1 | grep -v "^\[kmc\] spec" | tee $O/handles.txt
import os, sys
sys.path.insert(0, os.getcwd())
import kafka_specification_amd as kmc
from kafka_specification_amd.configs import HEADLINE
for slots in (1 << 30, 3 << 29):
    for fcap in (1 << 26, (1 << 26) + 512 * 37, (1 << 26) + 512 * 1, (1 << 26) + 512 * 1021):
        print(f"== table {slots / 2**30:.2f} x 2^30 slots, frontier capacity 2^26 + {fcap - (1 << 26)} states", flush=True)
        alive, line = [], []
        for k in range(6):
            mc = kmc.ModelChecker(kmc.CheckerConfig(**HEADLINE, table_capacity=slots, frontier_capacity=fcap)).__enter__()
            r = mc.run()
            assert r.distinct == 279753922, r.distinct
            ks = [mc.run().seconds_expand * 1e3 for _ in range(2)]
            line.append("%.2f" % min(ks))
            alive.append(mc)
        print("   k_expand per handle (ms):", " ".join(line), flush=True)
        for mc in alive:
            mc.__exit__(None, None, None)
PY
