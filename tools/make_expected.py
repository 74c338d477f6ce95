#!/usr/bin/env python3
"""Writes models/EXPECTED.json: what stock TLC should print for every models/*.cfg twin, so that anyone with a JVM can
pin this checker against the real engine (tools/verify_with_tlc.sh runs TLC and diffs).

TEST INFRASTRUCTURE: the numbers come from the C oracle (oracle/kmc_oracle.c, Oracle-B) — the exact-state CPU
restatement of the specs — not from the GPU product and not from TLC (no JVM exists in this image; parity with TLC
is unpinned until somebody runs the script).  The headline entry is taken from the committed golden fixture (the oracle
needs ~7 minutes on 8 cores for it); configurations the oracle cannot exhaust are listed with "exhaustible": false.

Per .cfg:
  module                root module TLC is started on (the .cfg sits next to <module>.tla in the verification directory)
  stop                  a default TLC run (stops at the first violation): verdict, invariant, trace_length
  exhaustive            the full reachable set (TLC -continue when an invariant is violated): distinct / generated / depth
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kmo  # noqa: E402  (the oracle's test-side binding)
from kafka_specification_amd.cfg import MODULE_TO_MODEL, parse_cfg, to_checker_config  # noqa: E402

# .cfg -> root module (TLC takes the module from the command line; several twins share one module)
MODULE_OF = {"Kip279_5brokers": "Kip279", "Kip320_7brokers": "Kip320", "LeaderInIsr": "Kip320",
             "KafkaTruncateToHighWatermark_3brokers": "KafkaTruncateToHighWatermark",
             "MCAsyncIsr_outside": "MCAsyncIsr", "MCAsyncIsr_small": "MCAsyncIsr"}
NOT_EXHAUSTIBLE = {"Kip320_7brokers": "8.8e8 states in the first 11 levels (profiles/r01_ladder.jsonl)"}
GOLDEN = {"Kip320": os.path.join(ROOT, "tests", "golden", "oracle_kip320_3_6_6_2.json"),
          "Kip279_5brokers": os.path.join(ROOT, "tests", "golden", "oracle_kip279_5_2_2_1.json"),
          "KafkaTruncateToHighWatermark_3brokers": os.path.join(ROOT, "tests", "golden", "oracle_thw_3_5_5_2.json")}


def oracle_cfg(cc, **kw):
    return kmo.make_config(cc.model, N=cc.n_replicas, L=cc.log_size, R=cc.max_records, E=cc.max_leader_epoch,
                           K=cc.n_log_records, MaxId=cc.max_id, invariants=cc.invariants,
                           check_deadlock=cc.check_deadlock, threads=os.cpu_count() or 4, **kw)


def main():
    out = {"_source": "oracle/kmc_oracle.c (Oracle-B) via tools/make_expected.py; NOT measured with TLC",
           "_how_to_verify": "tools/verify_with_tlc.sh (needs java and tla2tools.jar)"}
    for fn in sorted(os.listdir(os.path.join(ROOT, "models"))):
        if not fn.endswith(".cfg"):
            continue
        name = fn[:-4]
        module = MODULE_OF.get(name, name)
        mcfg = parse_cfg(open(os.path.join(ROOT, "models", fn)).read())
        cc = to_checker_config(module, mcfg)
        entry = {"module": module, "lowered_model": MODULE_TO_MODEL.get(module, module),
                 "invariants": list(cc.invariants), "check_deadlock": bool(cc.check_deadlock)}
        t0 = time.time()
        if name in NOT_EXHAUSTIBLE:
            entry.update(exhaustible=False, reason=NOT_EXHAUSTIBLE[name])
            # what can be pinned: the first BFS levels (TLC's "distinct states found" once its queue holds depth d+1)
            o = kmo.Run(oracle_cfg(cc, max_states=3_000_000))
            entry["prefix_levels"] = o.levels
        elif name in GOLDEN:
            g = json.load(open(GOLDEN[name]))
            entry.update(exhaustible=True,
                         stop=dict(verdict="ok" if g["verdict"] in (0, "ok") else str(g["verdict"]), invariant=None,
                                   invariants_violated_at_that_depth=[], trace_length=0),
                         exhaustive=dict(distinct=g["distinct"], generated=g["generated"], depth=g["depth"]),
                         fixture=os.path.relpath(GOLDEN[name], ROOT))
        else:
            o = kmo.Run(oracle_cfg(cc))
            entry["exhaustible"] = True
            # several invariants may fail at the depth of the first violation (in different states): TLC reports the
            # first one ITS search order meets, so any member of the list is a match
            entry["stop"] = dict(verdict=o.verdict, invariant=o.viol_inv,
                                 invariants_violated_at_that_depth=sorted(n for n, c in o.viol_count.items() if c and n in cc.invariants)
                                 if o.verdict == "invariant" else [],
                                 trace_length=o.viol_depth if o.verdict in ("invariant", "deadlock") else 0)
            if o.verdict == "ok":
                full = o
            else:   # TLC -continue: keep exploring after the first violation
                full = kmo.Run(oracle_cfg(cc, stop_on_violation=False))
            entry["exhaustive"] = dict(distinct=full.distinct, generated=full.generated, depth=full.depth)
        entry["oracle_seconds"] = round(time.time() - t0, 2)
        out[fn] = entry
        print(fn, json.dumps(entry)[:200], flush=True)
    with open(os.path.join(ROOT, "models", "EXPECTED.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
