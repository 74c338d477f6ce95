"""Headline configuration (Kip320 3/6/6/2) with and without symmetry reduction (orbit counting): step time, k_expand time,
stored states; both must report the golden counts.  usage: python tools/sym_headline.py [runs] [both|sym|plain] [log2 table slots]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_specification_amd as kmc
from kafka_specification_amd.configs import HEADLINE

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
g = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                "oracle_kip320_3_6_6_2.json")))
which = sys.argv[2] if len(sys.argv) > 2 else "both"
tlog = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for sym in (True, False):
    if (sym and which == "plain") or (not sym and which == "sym"):
        continue
    cfg = kmc.CheckerConfig(**HEADLINE, symmetry=sym, table_capacity=(1 << tlog) if tlog else (1 << 28) if sym else (1 << 30),
                            frontier_capacity=(1 << 24) if sym else (1 << 26))
    with kmc.ModelChecker(cfg) as mc:
        for i in range(runs):
            t0 = time.time()
            r = mc.run()
            dt = time.time() - t0
            ok = (r.verdict, r.distinct, r.generated, r.depth, r.levels) == ("ok", g["distinct"], g["generated"], g["depth"], g["levels"])
            print(json.dumps(dict(symmetry=sym, run=i, ms_step=round(1e3 * dt, 3), ms_expand=round(1e3 * r.seconds_expand, 3),
                                  launches=r.expand_launches, distinct=r.distinct, generated=r.generated,
                                  stored=r.orbit_representatives, matches_golden=ok,
                                  states_per_s=round(r.distinct / dt))), flush=True)
