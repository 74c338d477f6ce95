"""AsyncIsr.tla (the tenth module of the reference) under the state constraint of
models/MCAsyncIsr.tla — CPU side: the two oracles against what the spec text pins and against
each other, the .cfg binding, and the host half of the C ABI (layout, pack/unpack).

PARITY UNPINNED: the reference holds no .cfg, no expected counts and no TLC; the pins are the
closed forms below (derived from AsyncIsr.tla by hand) and Oracle-A == Oracle-B.
"""
import os
import random

import pytest

import kmo
from oracle import kafka_oracle as A
from kafka_specification_amd import CheckerConfig, ModelChecker
from kafka_specification_amd.cfg import CfgError, parse_cfg, to_checker_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LADDER = [(1, 3, 2), (2, 1, 1), (2, 2, 2), (3, 1, 2), (3, 2, 2), (3, 2, 3), (4, 1, 2)]  # (N, MaxOffset, MaxVersion)


def canon(m, s):
    """Oracle-A state -> canonical bytes (oracle/kmc_oracle.c, include/kmc.h)."""
    cisr, cver, lisr, lver, pisr, pver, offsets, requests, updates = s
    N, V = m.N, m.MaxVersion
    mask = lambda fs: sum(1 << r for r in fs)
    rb = ((1 << N) + 7) // 8
    out = bytearray([mask(cisr), cver, mask(lisr), lver, mask(pisr), pver + 1]) + bytes(offsets)
    req = bytearray((V + 1) * rb)
    for isr, v in requests:
        assert 0 <= v <= V
        req[v * rb + (mask(isr) >> 3)] |= 1 << (mask(isr) & 7)
    upd = bytearray(V + 1)
    # the set <-> version-indexed array bijection: exactly one update per version 1..controller version
    assert sorted(v for _, v in updates) == list(range(1, cver + 1))
    for isr, v in updates:
        upd[v - 1] = mask(isr)
    return bytes(out + req + upd)


# ---- what the spec text pins --------------------------------------------------------------------
@pytest.mark.parametrize("N", [2, 3, 4, 5])
def test_first_level_has_2n_minus_1_states(N):
    # From Init (AsyncIsr.tla:137-150) exactly these are enabled: ControllerShrinkIsr for the N-1
    # non-leaders (:72-79), LeaderRequestShrinkIsr for the same (:88-100), LeaderWrite (:117-119).
    a = A.bfs(A.make_model("AsyncIsr", N=N, L=2, E=2), invariants=(), max_states=1)   # stop after level 1
    o = kmo.Run(kmo.make_config("AsyncIsr", N=N, L=2, E=2, invariants=(), max_states=2))
    assert a["levels"][:2] == [1, 2 * N - 1] == o.levels[:2]


@pytest.mark.parametrize("M", [1, 2, 5, 9])
def test_single_replica_is_a_chain(M):
    # Replicas = {Leader}: only LeaderWrite can ever fire, so the in-model states are offsets 0..M,
    # and the write from offset M (outside the constraint) is generated but neither kept nor explored.
    a = A.bfs(A.make_model("AsyncIsr", N=1, L=M, E=3), invariants=("ValidHighWatermark",))
    o = kmo.Run(kmo.make_config("AsyncIsr", N=1, L=M, E=3, invariants=("ValidHighWatermark",)))
    for distinct, generated, depth, verdict in ((a["distinct"], a["generated"], a["depth"], a["verdict"]),
                                                (o.distinct, o.generated, o.depth, o.verdict)):
        assert (distinct, generated, depth, verdict) == (M + 1, M + 2, M + 1, "ok")


@pytest.mark.parametrize("M", [1, 2, 3, 5])
def test_two_replicas_version_zero_closed_form(M):
    # N = 2, MaxVersion = 0: every controller step leaves the constraint, so in-model the controller and
    # the leader's isr never change; what varies is whether the leader has asked to shrink ({Leader}, 0)
    # and the offsets 0 <= o1 <= o0 <= M  =>  2 * (M+1)(M+2)/2 distinct states, the farthest after M
    # writes, M replications and the request.  Generated, per state: ControllerShrinkIsr (outside), LeaderRequestShrinkIsr
    # and LeaderWrite always, ControllerHandleRequest once the request exists (outside), FollowerReplicate when o1 < o0.
    distinct = (M + 1) * (M + 2)
    generated = 1 + 3 * distinct + distinct // 2 + M * (M + 1)
    a = A.bfs(A.make_model("AsyncIsr", N=2, L=M, E=0), invariants=("ValidHighWatermark",))
    o = kmo.Run(kmo.make_config("AsyncIsr", N=2, L=M, E=0, invariants=("ValidHighWatermark",)))
    assert (a["distinct"], a["generated"], a["depth"], a["verdict"]) == (distinct, generated, 2 * M + 2, "ok")
    assert (o.distinct, o.generated, o.depth, o.verdict) == (distinct, generated, 2 * M + 2, "ok")


def test_typeok_is_false_in_the_initial_state():
    # pendingVersion |-> Nil (:146) with Nil == -1 (:38) is not in Nat (:44)
    a = A.bfs(A.make_model("AsyncIsr", N=3, L=2, E=2), invariants=("TypeOk",))
    o = kmo.Run(kmo.make_config("AsyncIsr", N=3, L=2, E=2, invariants=("TypeOk",)))
    assert a["verdict"] == "invariant" and a["violation"]["depth"] == 1 and a["distinct"] == 1
    assert o.verdict == "invariant" and o.viol_inv == "TypeOk" and o.viol_depth == 1 and o.distinct == 1


# ---- Oracle-A == Oracle-B ------------------------------------------------------------------------
@pytest.mark.parametrize("N,M,V", LADDER)
def test_python_and_c_oracles_agree(N, M, V):
    inv = ("ValidHighWatermark",)
    m = A.make_model("AsyncIsr", N=N, L=M, E=V)
    a = A.bfs(m, invariants=inv, keep_states=True)
    o = kmo.Run(kmo.make_config("AsyncIsr", N=N, L=M, E=V, invariants=inv))
    assert a["verdict"] == "ok" == o.verdict          # ValidHighWatermark (:161) holds at these bounds ...
    assert a["outside_violations"] == {}              # ... also in the successors outside the constraint
    assert (a["distinct"], a["generated"], a["depth"]) == (o.distinct, o.generated, o.depth)
    assert a["levels"] == o.levels
    assert list(a["action_generated"].values()) == o.action_generated[:7]
    assert o.sb == 6 + N + (V + 1) * (((1 << N) + 7) // 8) + V + 1
    for k, states in enumerate(a["level_states"]):
        assert {canon(m, s) for s in states} == o.level_states(k)


def test_successor_multisets_agree_on_sampled_states():
    N, M, V = 3, 2, 3
    m = A.make_model("AsyncIsr", N=N, L=M, E=V)
    a = A.bfs(m, invariants=(), keep_states=True)
    cfg = kmo.make_config("AsyncIsr", N=N, L=M, E=V)
    sb = len(canon(m, m.Init()))
    for s in [s for lvl in a["level_states"] for s in lvl][::11]:
        want = sorted((ai, canon_or_outside(m, t)) for (ai, t) in m.Next(s))
        got = sorted(kmo.successors(cfg, canon(m, s), sb))
        assert got == want


def canon_or_outside(m, t):
    """canon() for a successor that may lie outside the constraint (version MaxVersion+1)."""
    cisr, cver, lisr, lver, pisr, pver, offsets, requests, updates = t
    N, V = m.N, m.MaxVersion
    mask = lambda fs: sum(1 << r for r in fs)
    rb = ((1 << N) + 7) // 8
    out = bytearray([mask(cisr), cver, mask(lisr), lver, mask(pisr), pver + 1]) + bytes(offsets)
    req = bytearray((V + 1) * rb)
    for isr, v in requests:
        req[v * rb + (mask(isr) >> 3)] |= 1 << (mask(isr) & 7)
    upd = bytearray(V + 1)
    for isr, v in updates:
        upd[v - 1] = mask(isr)
    return bytes(out + req + upd)


# ---- invariants on successors outside the constraint ------------------------------------------
def test_invariants_are_checked_on_successors_outside_the_constraint():
    # LeaderOffsetInRange (models/MCAsyncIsr.tla) is false only where offsets[Leader] = MaxOffset+1,
    # i.e. in states the constraint excludes; TLC still evaluates invariants there [TLC-recall].
    # The shortest way out is MaxOffset+1 LeaderWrites: depth MaxOffset+2.
    for N, M, V in [(3, 2, 2), (2, 3, 1), (3, 1, 2)]:
        inv = ("ValidHighWatermark", "LeaderOffsetInRange")
        m = A.make_model("AsyncIsr", N=N, L=M, E=V)
        a = A.bfs(m, invariants=inv)
        cfg = kmo.make_config("AsyncIsr", N=N, L=M, E=V, invariants=inv)
        o = kmo.Run(cfg)
        assert a["verdict"] == "invariant" == o.verdict
        assert a["violation"]["invariant"] == "LeaderOffsetInRange" == o.viol_inv
        assert a["violation"]["depth"] == M + 2 == o.viol_depth
        assert a["violation"]["outside_constraint"] and o.viol_outside
        assert a["violation"]["per_invariant"] == {k: v for k, v in o.viol_count.items() if v}
        assert a["levels"] == o.levels
        # the C oracle's witness: a real successor of its parent, outside the constraint, violating
        parent = o.state(o.viol_parent_idx)
        assert (o.viol_action, o.viol_state) in kmo.successors(cfg, parent, o.sb)
        assert o.viol_state[6] == M + 1 and not kmo.check_invariant(cfg, 2, o.viol_state)
        assert len(a["violation"]["trace"]) == M + 2
        # with -continue the search is exhaustive and the numbers are those of the plain run
        full = kmo.Run(kmo.make_config("AsyncIsr", N=N, L=M, E=V, invariants=("ValidHighWatermark",)))
        cont = kmo.Run(kmo.make_config("AsyncIsr", N=N, L=M, E=V, invariants=inv, stop_on_violation=False))
        assert (cont.distinct, cont.generated, cont.levels) == (full.distinct, full.generated, full.levels)
        assert cont.verdict == "invariant" and cont.viol_depth == M + 2


# ---- .cfg binding ------------------------------------------------------------------------------
def test_mc_async_isr_cfg_binds_to_the_constrained_model():
    for name, (n, mo, mv) in {"MCAsyncIsr.cfg": (4, 3, 4), "MCAsyncIsr_small.cfg": (3, 3, 4)}.items():
        cc = to_checker_config("MCAsyncIsr", parse_cfg(open(os.path.join(ROOT, "models", name)).read()))
        assert (cc.model, cc.n_replicas, cc.log_size, cc.max_leader_epoch) == ("AsyncIsr", n, mo, mv)
        assert cc.invariants == ("ValidHighWatermark",) and cc.check_deadlock is False
        assert cc.to_native().invariant_mask == 2
    cc = to_checker_config("MCAsyncIsr", parse_cfg(open(os.path.join(ROOT, "models", "MCAsyncIsr_outside.cfg")).read()))
    assert cc.to_native().invariant_mask == 2 | 4


def test_cfg_rejects_what_it_cannot_honour():
    text = open(os.path.join(ROOT, "models", "MCAsyncIsr_small.cfg")).read()
    with pytest.raises(CfgError, match="unbounded"):        # the bare module cannot terminate
        to_checker_config("AsyncIsr", parse_cfg(text))
    with pytest.raises(CfgError, match="CONSTRAINT StateConstraint"):
        to_checker_config("MCAsyncIsr", parse_cfg(text.replace("CONSTRAINT StateConstraint", "")))
    with pytest.raises(CfgError, match="CONSTRAINT StateConstraint"):
        to_checker_config("MCAsyncIsr", parse_cfg(text.replace("StateConstraint", "SomethingElse")))
    with pytest.raises(CfgError, match="Leader must be an element"):
        to_checker_config("MCAsyncIsr", parse_cfg(text.replace("Leader = r1", "Leader = r9")))
    with pytest.raises(CfgError, match="unknown invariant"):
        to_checker_config("MCAsyncIsr", parse_cfg(text.replace("ValidHighWatermark", "StrongIsr")))
    kip = open(os.path.join(ROOT, "models", "Kip320.cfg")).read() + "\nCONSTRAINT StateConstraint\n"
    with pytest.raises(CfgError, match="CONSTRAINT is not supported"):
        to_checker_config("Kip320", parse_cfg(kip))


# ---- host half of the C ABI ----------------------------------------------------------------------
@pytest.mark.parametrize("N,M,V", [(3, 3, 4), (4, 3, 4), (2, 5, 7), (6, 2, 1), (1, 4, 0)])
def test_pack_unpack_roundtrip_on_reachable_states(N, M, V):
    o = kmo.Run(kmo.make_config("AsyncIsr", N=N, L=M, E=V, invariants=(), max_states=3000, threads=2))
    n = min(o.distinct, 3000)
    with ModelChecker(CheckerConfig(model="AsyncIsr", n_replicas=N, log_size=M, max_leader_epoch=V, device=-1)) as mc:
        assert mc.canon_bytes == o.sb
        assert mc.action_names() == list(A.AsyncIsrModel.action_names)
        seen = {}
        for idx in random.Random(2).sample(range(n), min(n, 400)):
            b = o.state(idx)
            w = tuple(mc.pack(b))
            assert mc.unpack(w) == b and seen.setdefault(w, b) == b and mc.fingerprint(w) != 0
        # a successor outside the constraint is representable too (one spare value per bounded field)
        cfg = kmo.make_config("AsyncIsr", N=N, L=M, E=V)
        for idx in range(n):
            for _, t in kmo.successors(cfg, o.state(idx), o.sb):
                if t[6] == M + 1 or t[1] == V + 1:
                    assert mc.unpack(mc.pack(t)) == t
                    return
        assert N == 1 and M + 1 > n  # only the chain model may not reach the boundary within the sample


def test_bad_async_isr_constants_are_rejected():
    from kafka_specification_amd import _native as nat
    for kw in (dict(n_replicas=7, log_size=2, max_leader_epoch=2), dict(n_replicas=3, log_size=0, max_leader_epoch=2),
               dict(n_replicas=3, log_size=2, max_leader_epoch=8)):
        with pytest.raises(nat.KmcError):
            ModelChecker(CheckerConfig(model="AsyncIsr", device=-1, **kw))
    with pytest.raises(ValueError, match="unknown invariant"):
        CheckerConfig(model="AsyncIsr", invariants=("StrongIsr",)).to_native()
