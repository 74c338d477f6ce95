"""CLI of Oracle-R (test infrastructure): exhaustive check of one root module of the reference."""
import argparse
import json
import os
import sys

from . import REFERENCE_DIR, Checker, format_trace, parse_cfg
from .check import cfg_value
from .values import fmt


def main():
    ap = argparse.ArgumentParser(prog="python -m oracle.tlar")
    ap.add_argument("module")
    ap.add_argument("-I", dest="path", action="append", default=[], help="extra module search directory")
    ap.add_argument("-c", dest="const", action="append", default=[], help="NAME=value | NAME=a,b,c (a set of model values)")
    ap.add_argument("--cfg")
    ap.add_argument("--inv", default="")
    ap.add_argument("--constraint")
    ap.add_argument("--deadlock", action="store_true", help="report deadlock (off by default: the bounded models have terminal states)")
    ap.add_argument("--continue", dest="cont", action="store_true")
    ap.add_argument("--max-states", type=int)
    ap.add_argument("--trace", action="store_true")
    a = ap.parse_args()
    constants, invariants, constraint, init, next_, deadlock = {}, [], a.constraint, "Init", "Next", a.deadlock
    if a.cfg:
        cfg = parse_cfg(open(a.cfg).read())
        constants.update(cfg["constants"])
        invariants += cfg["invariants"]
        constraint = constraint or (cfg["constraints"][0] if cfg["constraints"] else None)
        init, next_ = cfg["init"] or init, cfg["next"] or next_
        deadlock = deadlock or cfg["check_deadlock"]
    for c in a.const:
        k, v = c.split("=", 1)
        constants[k] = cfg_value("{" + v + "}") if ("," in v or not v.lstrip("-").isdigit() and not v.startswith('"') and k.endswith("s")) else cfg_value(v)
    invariants += [x for x in a.inv.split(",") if x]
    ck = Checker(a.module, constants, a.path + [REFERENCE_DIR], init, next_)
    r = ck.run(invariants=tuple(invariants), constraint=constraint, check_deadlock=deadlock,
               stop_on_violation=not a.cont, max_states=a.max_states)
    viol = r.pop("violation")
    if viol:
        if a.trace:
            print(format_trace(viol["trace"]), file=sys.stderr)
        r["violation"] = {k: v for k, v in viol.items() if k != "trace"} | {"trace_len": len(viol["trace"])}
    r["action_generated"] = {str(k): v for k, v in r["action_generated"].items()}
    print(json.dumps(r))


if __name__ == "__main__":
    main()
