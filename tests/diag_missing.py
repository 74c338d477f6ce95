#!/usr/bin/env python3
"""Diagnostic (test infrastructure: it calls the oracle): run one Kafka-family configuration on the GPU and in the C oracle, find the first BFS
level whose state sets differ, and say for each missing state which (parent, action) should have
produced it and whether the device lists it among that parent's successors (ENUM mode).
usage: python tests/diag_missing.py Kip320 7 1 1 0"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))  # kmo = the oracle's test-side binding
import kmo
from kafka_specification_amd import CheckerConfig, ModelChecker

model, N, L, R, E = sys.argv[1], *map(int, sys.argv[2:6])
o = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=()))
cfg = CheckerConfig(model=model, n_replicas=N, log_size=L, max_records=R, max_leader_epoch=E, invariants=(),
                    table_capacity=1 << 22, frontier_capacity=1 << 20)
levels = []
with ModelChecker(cfg) as mc:
    res = mc.run(progress=lambda info: levels.append({mc.unpack(r) for r in mc.frontier_states()}))
    names = mc.action_names()
    print("gpu levels", res.levels[:8], "oracle", o.levels[:8])
    idx_of = {}
    base = 0
    for k, n in enumerate(o.levels):
        want = o.level_states(k)
        if k < len(levels) and levels[k] == want:
            base += n
            continue
        missing = want - (levels[k] if k < len(levels) else set())
        extra = (levels[k] if k < len(levels) else set()) - want
        print(f"level {k}: missing {len(missing)} extra {len(extra)}")
        for i in range(base, base + n):
            st = o.state(i)
            if st in missing:
                par = o.state(kmo.lib().kmo_parent(o.h, i))
                act = kmo.lib().kmo_action(o.h, i)
                dev = [(kk, mc.unpack(w)) for (w, _fp, kk) in mc.successors(mc.pack(par))]
                listed = (act, st) in dev
                fp = mc.fingerprint(mc.pack(st))
                print(f"  missing state idx {i}: action {names[act]} listed_by_device={listed} fp={fp:016x} "
                      f"parent={par.hex()} state={st.hex()} contains={mc.contains(mc.pack(st))}")
        break
