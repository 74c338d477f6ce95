"""Oracle-R (oracle/tlar): the reference's own TLA+ text, parsed and evaluated — against the C oracle.

Two layers:
  * LIVE (needs /root/reference, i.e. the build container; skipped elsewhere): Oracle-R reads the ten .tla files and must
    equal Oracle-B on exact per-level state SETS, per-disjunct generated counts, verdicts, violation depth / counts and
    deadlock counts; plus the facts the text itself decides (LeaderInIsr false at Init, Kip279's double binding,
    AsyncIsr's TypeOk false at Init).
  * FIXTURE (runs everywhere): tests/golden/oracle_r_ladder.json holds Oracle-R's outputs on a larger ladder
    (tests/golden/make_oracle_r_golden.py, 1.3 M states); Oracle-B must reproduce every number and every level digest.
    tests/test_gpu_oracle_r.py holds the HIP engine to the same file on the GPU box, where the reference is absent.

Nothing here imports oracle/kafka_oracle.py (Oracle-A): no hand restatement of the specs sits between the reference's
text and the numbers the C oracle is held to.
"""
import json
import os

import pytest

import kmo
import oracle_r_canon as oc

REFERENCE = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "oracle_r_ladder.json")
live = pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference's .tla files are not on this box")

KAFKA = ("KafkaTruncateToHighWatermark", "Kip101", "Kip279", "Kip320", "Kip320FirstTry")
KAFKA_INV = ("TypeOk", "WeakIsr", "StrongIsr")


def oracle_b(entry, stop=None):
    m = entry["module"]
    model = "AsyncIsr" if m == "MCAsyncIsr" else m
    kw = dict(invariants=tuple(entry["invariants"]), stop_on_violation=bool(entry.get("stop")) if stop is None else stop, threads=4)
    for k in ("N", "L", "R", "E", "K", "MaxId"):
        if k in entry:
            kw[k] = entry[k]
    return kmo.Run(kmo.make_config(model, **kw))


def assert_same_as_oracle_b(r, entry, digests=None, level_bytes=None):
    """r: an Oracle-R result (live dict or fixture entry)."""
    o = oracle_b(entry)
    stop = bool(entry.get("stop"))
    v = r["violation"]
    if v is None:
        assert o.viol_inv is None
    else:
        assert (o.viol_inv, o.viol_depth) == (v["invariant"], v["depth"])
        for name, cnt in v["per_invariant"].items():
            assert o.viol_count[name] == cnt
        assert bool(v.get("outside_constraint")) == o.viol_outside
    if stop and v is not None:
        assert o.verdict == "invariant" and r["verdict"] == "invariant"
        if not v.get("outside_constraint"):   # (how much of the stopping level is counted is the checker's choice, not TLC's)
            assert r["levels"] == o.levels
        return
    cap = kmo.KMO_MAX_LEVELS   # the C oracle reports the sizes of its first 512 levels (IdSequence with MaxId 1000 has 1002)
    assert (r["distinct"], r["generated"], r["depth"], r["levels"][:cap], r["deadlock_states"]) == \
        (o.distinct, o.generated, o.depth, o.levels, o.deadlock_states)
    labels = r["actions"]
    for k, lab in enumerate(labels):  # the C oracle numbers the disjuncts of Next in source order
        assert o.action_generated[k] == r["action_generated"].get(str(lab), 0), f"disjunct {k} ({lab})"
    assert sum(r["action_generated"].values()) + 1 == r["generated"]
    if digests is not None:
        assert len(digests[:cap]) == len(o.levels)
        for k, d in enumerate(digests[:cap]):
            assert oc.level_digest(o.level_states(k)) == d, f"level {k}: state sets differ"
    if level_bytes is not None:
        for k, lv in enumerate(level_bytes):
            assert set(lv) == o.level_states(k), f"level {k}: state sets differ"


# ------------------------------------------------------------------------------------------------
# FIXTURE layer — runs everywhere
# ------------------------------------------------------------------------------------------------
def _entries():
    return json.load(open(GOLDEN))["entries"]


def _eid(e):
    c = "/".join(str(e[k]) for k in ("N", "L", "R", "E", "K", "MaxId") if k in e)
    return f"{e['module']}-{c}-{'+'.join(e['invariants'])}"


@pytest.mark.parametrize("entry", _entries(), ids=_eid)
def test_c_oracle_reproduces_oracle_r_fixture(entry):
    assert_same_as_oracle_b(entry, entry, digests=entry["level_digests"])


def test_fixture_covers_every_root_module_and_the_known_answers():
    es = _entries()
    assert {e["module"] for e in es} == set(KAFKA) | {"IdSequence", "FiniteReplicatedLog", "MCAsyncIsr"}
    by = {(_eid(e)): e for e in es}
    # closed forms derived from the text: IdSequence M+2 states; FiniteReplicatedLog (sum_{l<=L} K^l)^N
    for M in (0, 10, 1000):
        e = by[f"IdSequence-{M}-TypeOk"]
        assert (e["distinct"], e["depth"], e["generated"]) == (M + 2, M + 2, M + 2) and e["verdict"] == "ok"
    for K, want in ((1, 25), (2, 961), (3, 14641), (4, 116281)):
        e = by[f"FiniteReplicatedLog-2/4/{K}-TypeOk"]
        assert e["distinct"] == want and e["verdict"] == "ok"
    for m in KAFKA:
        for N in (2, 3):
            e = next(x for x in es if x["module"] == m and x["N"] == N and x["invariants"] == list(KAFKA_INV))
            assert e["levels"][:3] == [1, 2 * N, N * (4 * N - 2) if m == "Kip320" else N * (5 * N - 3)]
        e = by[f"{m}-2/2/2/1-LeaderInIsr"]  # KafkaReplication.tla:345 is false in the initial state (:117-119, :42)
        assert e["violation"]["depth"] == 1 and e["distinct"] == 1
    # the author's claims: Kip320 holds TypeOk / WeakIsr / StrongIsr (Kip320.tla:168-171) ...
    assert all(e["violation"] is None for e in es if e["module"] == "Kip320" and e["invariants"] == list(KAFKA_INV))
    # ... the earlier designs lose committed data for large enough bounds (prose of the four modules)
    for m in ("KafkaTruncateToHighWatermark", "Kip101", "Kip279", "Kip320FirstTry"):
        assert any(e["violation"] and e["violation"]["invariant"] in ("WeakIsr", "StrongIsr") for e in es if e["module"] == m)
        assert all((e["violation"] or {}).get("invariant") != "TypeOk" for e in es if e["module"] == m)
    # AsyncIsr's own TypeOk is false at Init: pendingVersion |-> Nil = -1 \notin Nat (AsyncIsr.tla:38,:44,:146)
    e = by["MCAsyncIsr-2/2/2-TypeOk"]
    assert e["violation"]["invariant"] == "TypeOk" and e["violation"]["depth"] == 1


def test_fixture_is_of_the_reference_revision_on_this_box():
    """Where the reference is present, the fixture must have been generated from exactly these files."""
    if not os.path.isdir(REFERENCE):
        pytest.skip("no reference here")
    import hashlib
    sha = json.load(open(GOLDEN))["spec_sha256"]
    for fn, want in sha.items():
        path = os.path.join(ROOT, fn) if fn.startswith("models/") else os.path.join(REFERENCE, fn)
        assert hashlib.sha256(open(path, "rb").read()).hexdigest() == want, fn


# ------------------------------------------------------------------------------------------------
# LIVE layer — Oracle-R reads /root/reference here and now
# ------------------------------------------------------------------------------------------------
def run_live(module, constants, invariants, constraint=None, stop=False):
    from oracle.tlar import Checker
    ck = Checker(module, constants, [os.path.join(ROOT, "models"), REFERENCE])
    r = ck.run(invariants=tuple(invariants), constraint=constraint, stop_on_violation=stop, keep_states=True)
    r["actions"] = [str(x) for x in ck.next_labels()]
    r["action_generated"] = {str(k): v for k, v in r["action_generated"].items()}
    enc = oc.encoder_for(module)
    return r, [[enc(s, constants) for s in lv] for lv in r["level_states"]]


@live
@pytest.mark.parametrize("module", KAFKA)
@pytest.mark.parametrize("N,L,R,E", [(2, 2, 2, 1), (2, 1, 2, 2)])
def test_live_kafka_modules_equal_the_c_oracle(module, N, L, R, E):
    entry = dict(module=module, N=N, L=L, R=R, E=E, invariants=list(KAFKA_INV))
    r, lv = run_live(module, oc.kafka_constants(N, L, R, E), KAFKA_INV)
    assert_same_as_oracle_b(r, entry, level_bytes=lv)


@live
def test_live_three_brokers():
    entry = dict(module="Kip320", N=3, L=1, R=1, E=1, invariants=list(KAFKA_INV))
    r, lv = run_live("Kip320", oc.kafka_constants(3, 1, 1, 1), KAFKA_INV)
    assert_same_as_oracle_b(r, entry, level_bytes=lv)
    assert r["levels"][:3] == [1, 6, 30]


@live
def test_live_standalone_modules_and_async_isr():
    from oracle.tlar import ModelValue
    r, lv = run_live("IdSequence", dict(MaxId=7), ("TypeOk",))
    assert_same_as_oracle_b(r, dict(module="IdSequence", MaxId=7, invariants=["TypeOk"]), level_bytes=lv)
    consts = dict(Replicas=frozenset(ModelValue(f"r{i}") for i in (1, 2)), LogRecords=frozenset(ModelValue(f"x{i}") for i in (1, 2)),
                  Nil=ModelValue("nil"), LogSize=3)
    r, lv = run_live("FiniteReplicatedLog", consts, ("TypeOk",))
    assert r["distinct"] == (1 + 2 + 4 + 8) ** 2
    assert_same_as_oracle_b(r, dict(module="FiniteReplicatedLog", N=2, L=3, K=2, invariants=["TypeOk"]), level_bytes=lv)
    reps = [ModelValue(f"r{i + 1}") for i in range(2)]
    consts = dict(Replicas=frozenset(reps), Leader=reps[0], MaxOffset=2, MaxVersion=2)
    for invs in (("ValidHighWatermark",), ("ValidHighWatermark", "LeaderOffsetInRange")):
        r, lv = run_live("MCAsyncIsr", consts, invs, constraint="StateConstraint")
        assert_same_as_oracle_b(r, dict(module="MCAsyncIsr", N=2, L=2, E=2, invariants=list(invs)), level_bytes=lv)
    r, _ = run_live("MCAsyncIsr", consts, ("TypeOk",), constraint="StateConstraint", stop=True)
    assert r["violation"]["invariant"] == "TypeOk" and r["violation"]["depth"] == 1


@live
def test_live_leader_in_isr_is_false_in_the_initial_state():
    for m in KAFKA:
        r, _ = run_live(m, oc.kafka_constants(2, 1, 1, 1), ("LeaderInIsr",), stop=True)
        assert r["verdict"] == "invariant" and r["violation"]["depth"] == 1 and r["distinct"] == 1


@live
def test_live_kip279_double_binding_and_kip320_double_disjunct():
    """Two places where the text yields the SAME successor twice, which TLC counts twice as generated [TLC-recall:
    Tool.getNextStates walks every disjunct that holds]: Kip279.tla:47-51 (an empty follower satisfies both disjuncts)
    and Kip320.tla:82-83 (not following the leader's epoch AND lagging)."""
    from oracle.tlar import Checker
    consts = oc.kafka_constants(2, 2, 2, 1)
    ck = Checker("Kip279", consts, [REFERENCE])
    r = ck.run(invariants=(), keep_states=True)
    doubles = 0
    for lv in r["level_states"][:6]:
        for st in lv:
            succ = [(lab, ck.key(t)) for lab, t in ck.interp.successors(st)]
            bf = [k for lab, k in succ if lab == "BecomeFollowerTruncateKip279"]
            for k in set(bf):
                n = bf.count(k)
                # the replica that became a follower is the one whose state changed; was ITS log empty before the step?
                follower_empty = any(st["replicaLog"].d[rr].d["endOffset"] == 0 and
                                     ck.unkey(k)["replicaState"].d[rr] != st["replicaState"].d[rr] for rr in consts["Replicas"])
                assert n == (2 if follower_empty else 1)
                doubles += n == 2
    assert doubles > 0
    ck = Checker("Kip320", consts, [REFERENCE])
    r = ck.run(invariants=(), keep_states=True)
    twice = 0
    for lv in r["level_states"]:
        for st in lv:
            ks = [ck.key(t) for lab, t in ck.interp.successors(st) if lab == "FencedLeaderShrinkIsr"]
            twice += sum(ks.count(k) == 2 for k in set(ks))
            assert all(ks.count(k) <= 2 for k in ks)
    assert twice > 0


@live
def test_every_reference_module_parses_and_the_parser_refuses_what_it_does_not_know():
    from oracle.tlar import TlaSyntaxError, parse_module
    names = sorted(f for f in os.listdir(REFERENCE) if f.endswith(".tla"))
    assert len(names) == 10
    for fn in names:
        m = parse_module(open(os.path.join(REFERENCE, fn)).read(), fn)
        assert m.name == fn[:-4]
    k = parse_module(open(os.path.join(REFERENCE, "KafkaReplication.tla")).read())
    assert k.extends == ["Integers", "Util"] and len(k.variables) == 6 and len(k.constants) == 4
    assert not any(d.name == "Next" for d in k.defs)            # KafkaReplication.tla defines no Next
    with pytest.raises(TlaSyntaxError):
        parse_module("---- MODULE X ----\nFoo == CASE a -> 1 [] OTHER -> 2\n====")
    with pytest.raises(TlaSyntaxError):
        parse_module("---- MODULE X ----\nFoo == \\E x : x\n====")


def test_junction_lists_and_precedence_without_the_reference():
    """The indentation rule, on text authored here (runs on any box)."""
    from oracle.tlar import Checker, parse_module
    import tempfile
    src = """---- MODULE J ----
EXTENDS Integers
VARIABLES x, y
Init == /\\ x = 0
        /\\ y = 0
A == /\\ x < 2
     /\\ \\/ x' = x + 1
        \\/ /\\ x = 1
           /\\ x' = 0
     /\\ UNCHANGED y
B == x = 2 /\\ y' = (IF y < 1 THEN y + 1 ELSE y) /\\ UNCHANGED <<x>>
Next == A \\/ B
Small == x + y * 2 <= 4 /\\ {x, y} \\subseteq 0 .. 2
====
"""
    m = parse_module(src)
    a = next(d for d in m.defs if d.name == "A").body
    assert a.kind == "and" and len(a.a) == 3 and a.a[1].kind == "or" and a.a[1].a[1].kind == "and"
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "J.tla"), "w").write(src)
        r = Checker("J", {}, [d]).run(invariants=("Small",), check_deadlock=False)
    # x: 0 -> 1 -> {2, 0}; at x = 2 only B moves (y 0 -> 1, then a self-loop)
    assert r["distinct"] == 4 and r["verdict"] == "ok" and r["action_generated"] == {"A": 3, "B": 2}


# ------------------------------------------------------------------------------------------------
# The premise of orbit counting, checked by EXECUTING the reference's text
# ------------------------------------------------------------------------------------------------
def _rename(v, m):
    """The TLA+ value v with every model value renamed through m (functions, records, tuples, sets, nested)."""
    from oracle.tlar import Fn, ModelValue
    if isinstance(v, ModelValue):
        return m.get(v, v)
    if isinstance(v, Fn):
        return Fn({_rename(k, m): _rename(x, m) for k, x in v.d.items()})
    if isinstance(v, frozenset):
        return frozenset(_rename(x, m) for x in v)
    if isinstance(v, tuple):
        return tuple(_rename(x, m) for x in v)
    assert isinstance(v, (int, str, bool)), type(v)
    return v


@live
@pytest.mark.parametrize("module", KAFKA + ("FiniteReplicatedLog",))
def test_live_every_permutation_of_replicas_is_an_automorphism(module):
    """kmc_config.symmetry stores one state per orbit of the permutations of Replicas and weighs every count by the orbit's
    size.  That rests on one claim about the SPECS: renaming the replicas maps Init to Init, the successors of a state under
    each disjunct of Next onto the successors of the renamed state under the same disjunct (with multiplicity: the doubly
    satisfied disjuncts), and keeps every invariant.  Here the claim is not argued from a reading of the text but checked
    by running the text: Oracle-R's reachable states of a 3-replica configuration, every one of the 5 non-trivial renamings."""
    import itertools
    from collections import Counter
    from oracle.tlar import Checker, ModelValue
    if module == "FiniteReplicatedLog":
        reps = [ModelValue(f"r{i}") for i in (1, 2, 3)]
        consts = dict(Replicas=frozenset(reps), LogRecords=frozenset(ModelValue(f"x{i}") for i in (1, 2)), Nil=ModelValue("nil"),
                      LogSize=2)
        invariants = ("TypeOk",)
    else:
        consts = oc.kafka_constants(3, 1, 1, 1)
        reps = sorted(consts["Replicas"], key=lambda r: r.name)
        invariants = KAFKA_INV + ("LeaderInIsr",)
    ck = Checker(module, consts, [os.path.join(ROOT, "models"), REFERENCE])
    # (nine whole levels: a renaming keeps the distance from Init, so the states of whole levels are closed under it)
    r = ck.run(invariants=(), stop_on_violation=False, keep_states=True, max_levels=9)
    ip = ck.interp
    states = [st if isinstance(st, dict) else ck.unkey(st) for lv in r["level_states"] for st in lv]
    assert len(states) > 300
    renamings = [dict(zip(reps, p)) for p in itertools.permutations(reps)][1:]
    for init in ip.initial_states():
        for m in renamings:
            assert {v: _rename(x, m) for v, x in init.items()} == init
    reachable = {ck.key(st) for st in states}
    moved = 0
    for st in states[::max(1, len(states) // 150)]:
        succ = Counter((lab, ck.key({v: x for v, x in t.items()})) for lab, t in ip.successors(st))
        held = {inv: ip.holds(st, inv) for inv in invariants}
        for m in renamings:
            image = {v: _rename(x, m) for v, x in st.items()}
            assert ck.key(image) in reachable                      # the reachable set is closed under renaming
            moved += image != st
            want = Counter((lab, ck.key({v: _rename(x, m) for v, x in ck.unkey(k).items()})) for (lab, k), n in succ.items()
                           for _ in range(n))
            got = Counter((lab, ck.key(t)) for lab, t in ip.successors(image))
            assert got == want, (module, st)
            assert {inv: ip.holds(image, inv) for inv in invariants} == held
    assert moved > 100   # and the renamings did move states
