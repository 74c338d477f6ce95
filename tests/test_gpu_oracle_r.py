"""GPU vs the reference's own text: the HIP engine (through the C ABI) against tests/golden/oracle_r_ladder.json, the
outputs of Oracle-R (oracle/tlar — /root/reference/*.tla parsed and evaluated; tests/golden/make_oracle_r_golden.py).

No oracle code runs here: the fixture carries distinct / generated / per-disjunct generated / depth / per-level sizes /
deadlocked states / first violation, and one sha256 per BFS level over the sorted canonical encodings of the level's
states — so the comparison is on exact state SETS, level by level, 1.3 M states in all.  The reference is absent on the
GPU box; the fixture is how "the reference, executed" gets here."""
import hashlib
import json
import os

import pytest

from kafka_specification_amd import CheckerConfig, ModelChecker

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENTRIES = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_r_ladder.json")))["entries"]


def _eid(e):
    c = "/".join(str(e[k]) for k in ("N", "L", "R", "E", "K", "MaxId") if k in e)
    return f"{e['module']}-{c}-{'+'.join(e['invariants'])}"


def _config(e):
    m = e["module"]
    inv = tuple(e["invariants"])
    cont = not e.get("stop")
    big = e["distinct"] > 200000
    caps = dict(table_capacity=1 << (24 if big else 22), frontier_capacity=1 << (21 if big else 20))
    if m == "IdSequence":
        return CheckerConfig(model=m, max_id=e["MaxId"], invariants=inv, continue_on_violation=cont, **caps)
    if m == "FiniteReplicatedLog":
        return CheckerConfig(model=m, n_replicas=e["N"], log_size=e["L"], n_log_records=e["K"], invariants=inv,
                             continue_on_violation=cont, **caps)
    if m == "MCAsyncIsr":   # N replicas, log_size = MaxOffset, max_leader_epoch = MaxVersion (the wrapper's constraint is built in)
        return CheckerConfig(model="AsyncIsr", n_replicas=e["N"], log_size=e["L"], max_leader_epoch=e["E"], invariants=inv,
                             continue_on_violation=cont, **caps)
    return CheckerConfig(model=m, n_replicas=e["N"], log_size=e["L"], max_records=e["R"], max_leader_epoch=e["E"],
                         invariants=inv, continue_on_violation=cont, **caps)


def _digest(byte_states):
    h = hashlib.sha256()
    for b in sorted(byte_states):
        h.update(b)
    return h.hexdigest()


@pytest.mark.parametrize("e", ENTRIES, ids=_eid)
def test_gpu_reproduces_the_executed_reference(e):
    digests = []
    with ModelChecker(_config(e)) as mc:
        def cb(info):
            digests.append(_digest(bytes(mc.unpack(row)) for row in mc.frontier_states()))
        res = mc.run(progress=cb)
    v = e["violation"]
    if v is None:
        assert res.violated_invariant is None and res.verdict == "ok"
    else:
        assert res.verdict == "invariant"
        assert (res.violated_invariant, res.violation_depth) == (v["invariant"], v["depth"])
        for name, cnt in v["per_invariant"].items():
            assert res.violation_count[name] == cnt
    if e.get("stop") and v is not None:
        return          # (how much of the stopping level is counted is the checker's choice, not the reference's)
    assert (res.distinct, res.generated, res.depth, res.levels, res.deadlock_states) == \
        (e["distinct"], e["generated"], e["depth"], e["levels"], e["deadlock_states"])
    got = list(res.action_generated.values())
    for k, lab in enumerate(e["actions"]):   # the engine numbers the disjuncts of Next in source order
        assert got[k] == e["action_generated"].get(lab, 0), f"disjunct {k} ({lab})"
    assert digests == e["level_digests"], "a BFS level's state set differs from the executed reference's"
