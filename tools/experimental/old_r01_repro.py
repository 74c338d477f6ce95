#!/usr/bin/env python3
"""Round-1 lost-successor hunt (test infrastructure: calls the oracle).  Loads the library of commit 7ef7376 — the last
one whose k_expand was always compiled for 6 waves/SIMD (80 VGPRs), the build that lost six states of Kip320 with 7
replicas — from tools/experimental/old_r01/ (git archive 7ef7376 kafka_specification_amd + make; not tracked) and
compares its BFS levels with the oracle's.  usage: old_r01_repro.py Model N L R E"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools", "experimental", "old_r01"))     # the OLD package first
sys.path.insert(1, os.path.join(ROOT, "tests"))
import kmo  # noqa: E402
import kafka_specification_amd as old  # noqa: E402
assert "old_r01" in old.__file__, old.__file__
model, N, L, R, E = sys.argv[1], *map(int, sys.argv[2:6])
o = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=()))
cfg = old.CheckerConfig(model=model, n_replicas=N, log_size=L, max_records=R, max_leader_epoch=E, invariants=(),
                        table_capacity=1 << 22, frontier_capacity=1 << 20)
with old.ModelChecker(cfg) as mc:
    res = mc.run()
print("defines:", os.environ.get("KMC_JIT_DEFINES"), "| gpu distinct", res.distinct, "oracle", o.distinct,
      "| first differing level:", next((k for k, (a, b) in enumerate(zip(res.levels, o.levels)) if a != b), None),
      "| gpu levels", res.levels[:8], "oracle", o.levels[:8])
