#!/usr/bin/env python3
"""Lost-successor hunt on the configuration where round 1 saw it at scale: Kip320, 7 brokers, LogSize 8 (W = 9 words,
357 action instances).  Its first honest number there (878 M states in 11 levels) replaced 197 M from the 80-VGPR build.
Runs the first BFS levels with the kernel as KMC_JIT_DEFINES builds it and compares with the oracle's prefix
(models/EXPECTED.json, Kip320_7brokers.cfg).  No oracle call: the prefix is a committed expectation."""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kafka_specification_amd import CheckerConfig, ModelChecker  # noqa: E402
want = json.load(open(os.path.join(ROOT, "models", "EXPECTED.json")))["Kip320_7brokers.cfg"]["prefix_levels"]
cfg = CheckerConfig(model="Kip320", n_replicas=7, log_size=8, max_records=8, max_leader_epoch=3, invariants=("TypeOk",),
                    max_levels=len(want), table_capacity=1 << 26, frontier_capacity=1 << 24)
with ModelChecker(cfg) as mc:
    r = mc.run()
bad = next((k for k, (a, b) in enumerate(zip(r.levels, want)) if a != b), None)
print("defines:", os.environ.get("KMC_JIT_DEFINES"), "| first differing level:", bad, "| gpu", r.levels, "| oracle", want,
      "| seconds %.2f" % r.seconds_total)
