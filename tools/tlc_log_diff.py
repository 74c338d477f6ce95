#!/usr/bin/env python3
"""Compare one TLC log with models/EXPECTED.json (helper of tools/verify_with_tlc.sh).

usage: tlc_log_diff.py EXPECTED.json <cfg> <stop|exhaustive> <tlc.log>      exit 0 = equal, 1 = differs / not comparable

What is compared [TLC-recall — message formats of tlc2.output.MP]:
  stop        "Model checking completed. No error has been found."  vs  "Error: Invariant X is violated[ by the initial
              state]." / "Error: Deadlock reached."; for a violation also the length of the printed behaviour (number of
              "State k:" blocks) — breadth-first search finds a shortest counterexample; with several workers TLC may
              print one that is a state longer, which is reported as a note, not a failure.
  exhaustive  "<G> states generated, <D> distinct states found, 0 states left on queue." and "The depth of the complete
              state graph search is <depth>." (run with -continue when an invariant is violated).
"""
import json
import re
import sys


def parse(text):
    out = dict(verdict=None, invariant=None, trace_length=0, generated=None, distinct=None, left=None, depth=None)
    if "Model checking completed. No error has been found." in text:
        out["verdict"] = "ok"
    m = re.search(r"Error: Invariant (\w+) is violated", text)
    if m:
        out["verdict"], out["invariant"] = "invariant", m.group(1)
    if "Error: Deadlock reached." in text and out["verdict"] is None:
        out["verdict"] = "deadlock"
    out["trace_length"] = len(set(re.findall(r"^State (\d+):", text, flags=re.M)))
    for m in re.finditer(r"^(\d+) states generated, (\d+) distinct states found, (\d+) states left on queue\.", text, flags=re.M):
        out["generated"], out["distinct"], out["left"] = int(m.group(1)), int(m.group(2)), int(m.group(3))
    m = re.search(r"The depth of the complete state graph search is (\d+)\.", text)
    if m:
        out["depth"] = int(m.group(1))
    return out


def main():
    exp_path, cfg, mode, log = sys.argv[1:5]
    e = json.load(open(exp_path))[cfg]
    got = parse(open(log, errors="replace").read())
    tag = f"{cfg} [{mode}]"
    if not e.get("exhaustible", True):
        print(f"-- {tag}: not exhaustible ({e.get('reason')}); TLC said {got}")
        return 0
    if mode == "stop":
        want = e["stop"]
        ok = got["verdict"] == want["verdict"] and (got["invariant"] == want["invariant"] or
                                                    got["invariant"] in want.get("invariants_violated_at_that_depth", []))
        note = ""
        if ok and want["verdict"] == "invariant" and got["trace_length"] != want["trace_length"]:
            note = f" (note: TLC's behaviour has {got['trace_length']} states, the shortest has {want['trace_length']})"
            ok = got["trace_length"] in (want["trace_length"], want["trace_length"] + 1)
        print(f"{'OK' if ok else 'DIFF'} {tag}: expected {want}, TLC {dict(verdict=got['verdict'], invariant=got['invariant'], trace_length=got['trace_length'])}{note}")
        return 0 if ok else 1
    want = e["exhaustive"]
    have = dict(distinct=got["distinct"], generated=got["generated"], depth=got["depth"])
    ok = have == want and got["left"] == 0
    print(f"{'OK' if ok else 'DIFF'} {tag}: expected {want}, TLC {have} (left on queue: {got['left']})")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
