"""Shared by the CPU and GPU tests of the per-state differential against the executed reference
(tests/golden/oracle_r_successors_*.npz, written by tests/golden/make_oracle_r_successors.py).  TEST INFRASTRUCTURE ONLY."""
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INDEX = os.path.join(GOLDEN, "oracle_r_successors_index.json")


def entries():
    if not os.path.exists(INDEX):
        return []
    ix = json.load(open(INDEX))["entries"]
    return [(fn, ix[fn]) for fn in sorted(ix) if os.path.exists(os.path.join(GOLDEN, fn))]


def ids(e):
    m = e[1]
    return f"{m['module']}-{m['N']}-{m['L']}-{m['R']}-{m['E']}"


def load(fn):
    z = np.load(os.path.join(GOLDEN, fn))
    return {k: z[k] for k in z.files}


def succ_digest(records):
    """The fixture's digest of a state's successors: sha256 over the sorted multiset of (action index, canonical bytes)."""
    h = hashlib.sha256()
    for a, b in sorted(records):
        h.update(bytes([a]))
        h.update(b)
    return h.digest()[:16]


def features(b, N, L, E):
    """(deepest log, most distinct record epochs inside one log, highest hw, deepest log holding >= 2 epochs) of a state."""
    deep = eps = hw = deep2 = 0
    for r in range(N):
        o = r * (5 + L)
        end = b[o]
        e = len({(b[o + 5 + k] - 1) % (E + 1) for k in range(min(end, L)) if b[o + 5 + k]})
        deep, eps, hw = max(deep, end), max(eps, e), max(hw, b[o + 1])
        if e >= 2:
            deep2 = max(deep2, end)
    return deep, eps, hw, deep2


def compare(meta, fx, i, records, inv_bits, who):
    """One state of the fixture against what `who` computed for it."""
    state = bytes(fx["states"][i])
    per = [0] * len(meta["actions"])
    for a, _ in records:
        per[a] += 1
    assert len(records) == int(fx["nsucc"][i]) and per == [int(x) for x in fx["per_action"][i]], (
        f"{who}: state {state.hex()} has {per} successors per disjunct {meta['actions']}, the reference's text "
        f"{[int(x) for x in fx['per_action'][i]]}")
    assert succ_digest(records) == bytes(fx["digest"][i]), (
        f"{who}: the successors of {state.hex()} differ from the reference's (same counts per disjunct): "
        f"{[(meta['actions'][a], b.hex()) for a, b in sorted(records)]}")
    assert inv_bits == int(fx["inv"][i]), (
        f"{who}: invariants violated by {state.hex()}: {inv_bits:04b}, the reference's text {int(fx['inv'][i]):04b} "
        f"(bit k = {meta['invariants']}[k])")


MUTANTS_INDEX = os.path.join(GOLDEN, "oracle_r_mutants_index.json")


def mutant_entries():
    """[(file, meta)] of tests/golden/oracle_r_mutants_*.npz: the invariants on arbitrary (mutated) states, by Oracle-R."""
    if not os.path.exists(MUTANTS_INDEX):
        return []
    ix = json.load(open(MUTANTS_INDEX))["entries"]
    return [(fn, ix[fn]) for fn in sorted(ix) if os.path.exists(os.path.join(GOLDEN, fn))]


def comparable_invariants(inv_bits, undefined_bits):
    """The invariant bits of a mutated state on which the lowerings are held to the reference's text.  TypeOk (bit 0) and
    LeaderInIsr (bit 3): always.  WeakIsr / StrongIsr (bits 1, 2): on the states that satisfy TypeOk.  The text asks for
    `\\E record \\in LogRecords : HasEntry(r1, record, offset) /\\ HasEntry(r2, record, offset)` (KafkaReplication.tla:320-340);
    the lowerings compare the two slots, which is the same thing exactly when a written slot holds a member of LogRecords —
    ReplicaLog!TypeOk (FiniteReplicatedLog.tla:90-95).  On a state that breaks TypeOk the text can fail where slot equality
    holds; TLC reports TypeOk there (it is the first invariant of every .cfg twin), and no model of the reference reaches one."""
    keep = 15 & ~undefined_bits
    if inv_bits & 1:
        keep &= ~6
    return keep


def engine_model(meta):
    """(engine / C-oracle model name, CheckerConfig constants, kmo.make_config constants) of a fixture entry."""
    if meta["module"] == "MCAsyncIsr":   # AsyncIsr.tla under models/MCAsyncIsr.tla: (N, MaxOffset, MaxVersion)
        return ("AsyncIsr", dict(n_replicas=meta["N"], log_size=meta["L"], max_leader_epoch=meta["E"]),
                dict(N=meta["N"], L=meta["L"], E=meta["E"]))
    return (meta["module"], dict(n_replicas=meta["N"], log_size=meta["L"], max_records=meta["R"], max_leader_epoch=meta["E"]),
            dict(N=meta["N"], L=meta["L"], R=meta["R"], E=meta["E"]))


def is_kafka(entry):
    return entry[1]["module"] != "MCAsyncIsr"
