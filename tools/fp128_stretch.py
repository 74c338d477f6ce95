#!/usr/bin/env python3
"""Kip320 3/6/6/3 (6.45 G states: the configuration whose seven narrow-table runs of round 2 lost 0, 1 or 9 states to 64-bit
fingerprint collisions, profiles/r02_ladder.jsonl) with 128-bit seen-set entries, under several hash seeds.  One JSON line
per run -> stdout.   python tools/fp128_stretch.py [seed ...]
KMC_NARROW=1: 64-bit entries.  KMC_STRETCH_TABLE_LOG2 (default 33) or KMC_STRETCH_TABLE=<slots> (any size), KMC_STRETCH_FRONTIER=<states>.  KMC_STRETCH_LEVELS=FILE: the per-level records of the LAST
run (kmc_level_stats: frontier, probes, k_expand ms -> probes/s per level) as JSON lines.  KMC_STRETCH_RUNS=n: n searches per seed
on one handle (the first touches freshly mapped memory)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("KMC_NO_TORCH", "0") != "1":
    import torch  # noqa: F401
import kafka_specification_amd as kmc

seeds = [int(x, 0) for x in sys.argv[1:]] or [0, 0x5EED2, 0xC0FFEE]
wide = os.environ.get("KMC_NARROW", "0") != "1"
tlog = int(os.environ.get("KMC_STRETCH_TABLE_LOG2", 33))
tslots = int(float(os.environ.get("KMC_STRETCH_TABLE", 1 << tlog)))       # any multiple of 64 (round 6): e.g. 12.5e9
fslots = int(float(os.environ.get("KMC_STRETCH_FRONTIER", 1 << 30)))       # the widest level holds 521,281,965 states
runs = int(os.environ.get("KMC_STRETCH_RUNS", 1))
for seed in seeds:
    cfg = kmc.CheckerConfig(model="Kip320", n_replicas=3, log_size=6, max_records=6, max_leader_epoch=3,
                            invariants=("TypeOk", "WeakIsr", "StrongIsr"), hash_seed=seed, wide_fingerprint=wide,
                            table_capacity=tslots, frontier_capacity=fslots)
    t0 = time.time()
    with kmc.ModelChecker(cfg) as mc:
        t_open = time.time() - t0
        for k in range(runs):
            t1 = time.time()
            r = mc.run()
            print(json.dumps(dict(config="Kip320 3/6/6/3", wide_fingerprint=wide, hash_seed=seed, run=k, verdict=r.verdict, distinct=r.distinct,
                                  generated=r.generated, generated_repeats=r.generated_repeats, depth=r.depth, widest_level=max(r.levels),
                                  seconds_total=round(r.seconds_total, 3), seconds_expand=round(r.seconds_expand, 3),
                                  seconds_clear=round(r.seconds_clear, 4), probes_per_s=round((r.generated - r.generated_repeats) / r.seconds_expand / 1e9, 2),
                                  table_slots=r.table_capacity, open_s=round(t_open, 2), run_wall_s=round(time.time() - t1, 2),
                                  timing=mc.timing())), flush=True)
        if os.environ.get("KMC_STRETCH_LEVELS"):
            with open(os.environ["KMC_STRETCH_LEVELS"], "w") as f:
                for st in mc.level_stats():
                    st = dict(st, generated=sum(st["generated"].values()),
                              probes_G_per_s=round(st["probes"] / max(st["expand_ms"], 1e-9) / 1e6, 2))
                    f.write(json.dumps(st) + "\n")
