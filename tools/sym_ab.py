"""One Kafka-family configuration with and without orbit counting: step time, k_expand time, stored states; the two searches
must report the same counts.  usage: python tools/sym_ab.py MODEL N L R E [runs] [log2 table slots]      (KMC_AB_FP128=1: 128-bit seen-set entries)
(KMC_JIT_DEFINES=-DKMC_SYMM_UNROLLED_MAX=k moves the replica count from which the representative is chosen among the sorted
images — on the device only, so traces / kmc_contains of such a run are not meaningful; counts are.)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_specification_amd as kmc

m, N, L, R, E = sys.argv[1], *map(int, sys.argv[2:6])
runs = int(sys.argv[6]) if len(sys.argv) > 6 else 3
tlog = int(sys.argv[7]) if len(sys.argv) > 7 else 28
inv = ("TypeOk", "WeakIsr", "StrongIsr") if m == "Kip320" else ("TypeOk",)
seen = {}
for sym in (True, False):
    cfg = kmc.CheckerConfig(model=m, n_replicas=N, log_size=L, max_records=R, max_leader_epoch=E, invariants=inv,
                            continue_on_violation=True, symmetry=sym, wide_fingerprint=os.environ.get("KMC_AB_FP128", "0") == "1", table_capacity=1 << tlog, frontier_capacity=1 << (tlog - 3))
    with kmc.ModelChecker(cfg) as mc:
        for i in range(runs if sym else 2):
            t0 = time.time()
            r = mc.run()
            dt = time.time() - t0
            seen[sym] = (r.verdict, r.distinct, r.generated, r.depth, tuple(r.levels))
            print(json.dumps(dict(workload=f"{m},{N},{L},{R},{E}", symmetry=sym, run=i, ms_step=round(1e3 * dt, 3),
                                  ms_expand=round(1e3 * r.seconds_expand, 3), launches=r.expand_launches, distinct=r.distinct,
                                  generated=r.generated, stored=r.orbit_representatives, verdict=r.verdict)), flush=True)
print("counts equal:", seen[True] == seen[False])
