"""Oracle-R, part 4: exhaustive breadth-first search over the states the reference's text defines.

TEST INFRASTRUCTURE ONLY.  The search is level-synchronous with an exact set of full states (no fingerprints); the
result dictionary has the shape of oracle/kafka_oracle.py's `bfs` so the same tests can read both — but no code is
shared with it: this file knows nothing about Kafka, logs or replicas.

TLC semantics reproduced [TLC-recall]:
  * "generated" counts every successor the Next relation yields (one per satisfying binding), the initial states too;
  * invariants are checked on every distinct state when it is first found;
  * CONSTRAINT: a successor outside the constraint is generated and its invariants are checked, but it is neither
    added to the seen set nor explored;
  * deadlock = a state without any successor (reported only when asked for).
"""
from __future__ import annotations

import re
import time

from .interp import Interp, Loader
from .values import ModelValue, TlaEvalError, fmt


def parse_cfg(text: str):
    """The handful of TLC configuration keys the twins under models/ use."""
    text = re.sub(r"\\\*[^\n]*", "", text)
    text = re.sub(r"\(\*.*?\*\)", "", text, flags=re.S)
    toks = re.findall(r"\{[^}]*\}|\"[^\"]*\"|[A-Za-z0-9_]+|=|<-", text)
    cfg = dict(constants={}, init=None, next=None, invariants=[], constraints=[], check_deadlock=True, specification=None)
    keys = {"CONSTANT", "CONSTANTS", "INIT", "NEXT", "INVARIANT", "INVARIANTS", "CONSTRAINT", "CONSTRAINTS",
            "CHECK_DEADLOCK", "SPECIFICATION", "PROPERTY", "PROPERTIES", "SYMMETRY", "VIEW", "ACTION_CONSTRAINT"}
    i, section = 0, None
    while i < len(toks):
        t = toks[i]
        if t in keys:
            section = t
            i += 1
            continue
        if section in ("CONSTANT", "CONSTANTS"):
            if i + 2 >= len(toks) + 0 and False:
                pass
            name = t
            if toks[i + 1] not in ("=", "<-"):
                raise ValueError(f"cfg: expected '=' after {name}")
            cfg["constants"][name] = cfg_value(toks[i + 2])
            i += 3
            continue
        if section == "INIT":
            cfg["init"] = t
        elif section == "NEXT":
            cfg["next"] = t
        elif section == "SPECIFICATION":
            cfg["specification"] = t
        elif section in ("INVARIANT", "INVARIANTS"):
            cfg["invariants"].append(t)
        elif section in ("CONSTRAINT", "CONSTRAINTS"):
            cfg["constraints"].append(t)
        elif section == "CHECK_DEADLOCK":
            cfg["check_deadlock"] = t.upper() == "TRUE"
        else:
            raise ValueError(f"cfg: section {section} is outside the subset")
        i += 1
    return cfg


def cfg_value(tok: str):
    if tok.startswith("{"):
        inner = tok[1:-1].strip()
        return frozenset(cfg_value(x.strip()) for x in inner.split(",")) if inner else frozenset()
    if tok.startswith('"'):
        return tok[1:-1]
    if re.fullmatch(r"-?[0-9]+", tok):
        return int(tok)
    if tok in ("TRUE", "FALSE"):
        return tok == "TRUE"
    return ModelValue(tok)


class Checker:
    def __init__(self, root, constants, search_path, init="Init", next_="Next"):
        self.loader = Loader(search_path)
        self.interp = Interp(self.loader, root, constants)
        self.vars = self.interp.variables
        self.init, self.next = init, next_

    def next_labels(self):
        """Labels of the disjuncts of the Next relation, in source order: what `action_generated` is keyed by (the
        operator name of a disjunct that is one, else its index); [None] when Next has no disjunction."""
        d = self.interp.root.visible[self.next]
        e = d.body
        while e.kind in ("paren", "quant", "let"):
            e = e.a if e.kind == "paren" else (e.c if e.kind == "quant" else e.b)
        if e.kind != "or":
            return [None]
        out = []
        for i, it in enumerate(e.a):
            while it.kind == "paren":
                it = it.a
            out.append(it.a if it.kind in ("ident", "apply") else (it.b if it.kind == "inst" else i))
        return out

    def key(self, st: dict):
        return tuple(st[v] for v in self.vars)

    def unkey(self, k):
        return dict(zip(self.vars, k))

    def run(self, invariants=(), constraint=None, check_deadlock=False, stop_on_violation=True, max_states=None,
            keep_states=False, max_levels=None):
        ip = self.interp
        t0 = time.time()
        bad_assume = ip.check_assumes()
        if bad_assume:
            raise TlaEvalError(f"ASSUME violated: {bad_assume}")
        parent = {}
        generated = 0
        action_generated = {}
        frontier = []
        for st in ip.initial_states(self.init):
            generated += 1
            k = self.key(st)
            if k not in parent:
                parent[k] = (None, None)
                frontier.append(k)
        levels = [len(frontier)]
        level_states = [list(frontier)] if keep_states else None
        verdict, violation = "ok", None
        depth = 1
        deadlocks = 0
        outside_total = {}

        def trace_of(k):
            tr = []
            while k is not None:
                par, lab = parent[k]
                tr.append((lab, self.unkey(k)))
                k = par
            tr.reverse()
            return tr

        def check(keys):
            per, first = {}, {}
            for k in keys:
                st = self.unkey(k)
                for name in invariants:
                    if not ip.holds(st, name):
                        per[name] = per.get(name, 0) + 1
                        first.setdefault(name, k)
            return per, first

        per, first = check(frontier)
        if per:
            name = next(n for n in invariants if n in per)
            violation = dict(invariant=name, depth=1, count_at_depth=per[name], per_invariant=per,
                             trace=trace_of(first[name]))
            if stop_on_violation:
                verdict, frontier = "invariant", []
        while frontier:
            if max_levels is not None and depth >= max_levels:
                verdict = "limit" if verdict == "ok" else verdict
                break
            nxt_level = []
            outside, outside_first = {}, {}
            for k in frontier:
                st = self.unkey(k)
                succ = ip.successors(st, self.next)
                if not succ:
                    deadlocks += 1
                for lab, t in succ:
                    generated += 1
                    action_generated[lab] = action_generated.get(lab, 0) + 1
                    if constraint is not None and not ip.holds(t, constraint):
                        for name in invariants:
                            if not ip.holds(t, name):
                                outside[name] = outside.get(name, 0) + 1
                                outside_first.setdefault(name, (k, lab, t))
                        continue
                    tk = self.key(t)
                    if tk not in parent:
                        parent[tk] = (k, lab)
                        nxt_level.append(tk)
            if check_deadlock and deadlocks and verdict == "ok":
                verdict = "deadlock"
                break
            for name, cnt in outside.items():
                outside_total[name] = outside_total.get(name, 0) + cnt
            if outside and violation is None:
                name = next(n for n in invariants if n in outside)
                pk, lab, t = outside_first[name]
                violation = dict(invariant=name, depth=depth + 1, count_at_depth=outside[name],
                                 per_invariant=dict(outside), trace=trace_of(pk) + [(lab, t)], outside_constraint=True)
                if stop_on_violation:
                    verdict = "invariant"
                    break
            if not nxt_level:
                break
            depth += 1
            levels.append(len(nxt_level))
            if keep_states:
                level_states.append(nxt_level)
            per, first = check(nxt_level)
            if per and violation is None:
                name = next(n for n in invariants if n in per)
                violation = dict(invariant=name, depth=depth, count_at_depth=per[name], per_invariant=per,
                                 trace=trace_of(first[name]))
                if stop_on_violation:
                    verdict = "invariant"
                    break
            if max_states is not None and len(parent) > max_states:
                verdict = "limit"
                break
            frontier = nxt_level
        if violation is not None and verdict == "ok":
            verdict = "invariant"
        res = dict(distinct=len(parent), generated=generated, depth=depth, levels=levels,
                   action_generated=action_generated, verdict=verdict, violation=violation, deadlock_states=deadlocks,
                   outside_violations=outside_total, seconds=round(time.time() - t0, 3), variables=list(self.vars))
        if keep_states:
            res["level_states"] = [[self.unkey(k) for k in lv] for lv in level_states]
        return res


def format_trace(trace):
    out = []
    for i, (lab, st) in enumerate(trace, 1):
        out.append(f"State {i}: <{'Initial predicate' if lab is None else lab}>")
        for v, x in st.items():
            out.append(f"/\\ {v} = {fmt(x)}")
        out.append("")
    return "\n".join(out)
