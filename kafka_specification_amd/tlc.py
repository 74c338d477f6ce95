"""`tlc2.TLC`-shaped command line [TLC-recall]:

    python -m kafka_specification_amd.tlc [-config X.cfg] [-deadlock] [-continue] [-workers N]
                                          [-fp SEED] [-gpus P] [-table SLOTS] [-frontier STATES] Spec.tla

Maps the root module's name to its lowered GPU model, reads constants / invariants from the
.cfg (default: Spec.cfg next to the module), runs the exhaustive search on the GPU and prints
TLC-style progress and summary lines.  -workers is accepted for command-line compatibility
and ignored (the GPU's waves are the workers).  No TLA+ is parsed: only the modules of
hachikuji/kafka-specification that have a Next are known.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

from .cfg import CfgError, parse_cfg, to_checker_config
from .checker import ModelChecker
from .format import format_state


def _now():
    return time.strftime("%Y-%m-%d %H:%M:%S")


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="tlc", add_help=True, prefix_chars="-")
    ap.add_argument("spec")
    ap.add_argument("-config", default=None)
    ap.add_argument("-deadlock", action="store_true", help="do NOT check for deadlock (TLC semantics of -deadlock)")
    ap.add_argument("-continue", dest="cont", action="store_true")
    ap.add_argument("-workers", default="1")
    ap.add_argument("-fp", type=int, default=0)
    ap.add_argument("-gpus", type=int, default=1, help="P > 1: P logical shards on this process's GPU (loopback); "
                                                       "real multi-GPU runs go through torch.distributed.run")
    ap.add_argument("-table", type=int, default=0)
    ap.add_argument("-frontier", type=int, default=0)
    ap.add_argument("-device", type=int, default=0)
    ap.add_argument("-notrace", action="store_true", help="do not keep predecessor links (no counterexample trace)")
    a = ap.parse_args(argv)

    module = os.path.splitext(os.path.basename(a.spec))[0]
    cfg_path = a.config or os.path.splitext(a.spec)[0] + ".cfg"
    if not os.path.exists(cfg_path):
        print(f"Error: configuration file {cfg_path} not found", file=sys.stderr)
        return 2
    try:
        mcfg = parse_cfg(open(cfg_path).read())
        over = dict(hash_seed=a.fp, device=a.device, continue_on_violation=a.cont, keep_trace=not a.notrace,
                    table_capacity=a.table, frontier_capacity=a.frontier)
        if a.deadlock:
            over["check_deadlock"] = False
        cc = to_checker_config(module, mcfg, **over)
    except CfgError as e:
        print(f"Error: {e}", file=sys.stderr)
        return 2

    print(f"kafka_specification_amd model checker (MI355X) — module {module}, config {os.path.basename(cfg_path)}")
    print(f"Running breadth-first search Model-Checking with fp seed {a.fp} on GPU {a.device}.")
    print("Computing initial states...")

    def progress(i):
        if i["depth"] == 1:
            print(f"Finished computing initial states: {i['distinct']} distinct state generated at {_now()}.")
        else:
            print(f"Progress({i['depth']}) at {_now()}: {i['generated']} states generated, "
                  f"{i['distinct']} distinct states found, {i['new_states']} states left on queue.")

    from ._native import KmcError
    try:
        if a.gpus > 1:
            from .sharded import check_loopback
            res = check_loopback(cc, a.gpus, a.device, progress)
            trace = res.trace   # walked owner by owner through the shards' predecessor tables
        else:
            with ModelChecker(cc) as mc:
                res = mc.run(progress)
                trace = []
                if res.verdict in ("invariant",) and cc.keep_trace:
                    trace = mc.trace()
                elif res.verdict == "deadlock":
                    trace = [(None, mc.unpack(mc.witness()))]
    except KmcError as e:
        print(f"Error: {e}", file=sys.stderr)
        return 3

    rc = 0
    if res.verdict == "ok":
        print("Model checking completed. No error has been found.")
    elif res.verdict == "invariant":
        where = " by the initial state" if res.violation_depth == 1 else ""
        print(f"Error: Invariant {res.violated_invariant} is violated{where}.")
        rc = 12
    elif res.verdict == "deadlock":
        print("Error: Deadlock reached.")
        rc = 11
    else:
        print(f"Error: search stopped: {res.verdict} (table {res.table_capacity} slots, "
              f"frontier {res.frontier_capacity} states)")
        rc = 1
    if trace:
        print("Error: The behavior up to this point is:")
        for k, (act, st) in enumerate(trace, 1):
            head = "<Initial predicate>" if act is None and k == 1 else f"<{act} of module {module}>"
            print(f"State {k}: {head}")
            print(format_state(cc, st))
            print()
    print(f"{res.generated} states generated, {res.distinct} distinct states found, "
          f"{res.queue_left} states left on queue.")
    print(f"The depth of the complete state graph search is {res.depth}.")
    print(f"Finished in {res.seconds_total:.3f}s ({res.distinct / max(res.seconds_total, 1e-9):,.0f} distinct states/s; "
          f"{res.seconds_expand:.3f}s in the expand kernel) at ({_now()})")
    return rc


if __name__ == "__main__":
    sys.exit(main())
