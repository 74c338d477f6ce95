"""The oracle against everything that pins it.  The reference repository holds
no golden vectors, tests or state counts, so the pins are (1) closed-form answers derivable
from the spec text, (2) the qualitative verdicts the spec comments claim, (3) agreement of
the two independently written oracles (structural Python vs byte-structured C)."""
import json
import os

import pytest

import kmo
from oracle import kafka_oracle as A

KAFKA = ("KafkaTruncateToHighWatermark", "Kip101", "Kip279", "Kip320", "Kip320FirstTry")
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


# ---- closed forms ---------------------------------------------------------------------------
@pytest.mark.parametrize("M", [0, 1, 10, 1000])
def test_idsequence_chain(M):  # IdSequence.tla:30-43: 0 -> 1 -> ... -> MaxId+1
    r = A.bfs(A.IdSequenceModel(M))
    assert (r["distinct"], r["depth"], r["verdict"]) == (M + 2, M + 2, "ok")
    o = kmo.Run(kmo.make_config("IdSequence", MaxId=M))
    assert (o.distinct, o.depth, o.verdict, o.generated) == (M + 2, M + 2, "ok", M + 2)


@pytest.mark.parametrize("N,L,K", [(2, 4, 1), (2, 4, 2), (2, 4, 3), (2, 4, 4), (3, 2, 2), (2, 1, 5)])
def test_finite_replicated_log_closed_form(N, L, K):
    # every replica's log can become any sequence of length 0..L over K records (Append takes
    # any record, FiniteReplicatedLog.tla:116) => (sum_{l<=L} K^l)^N distinct states
    expected = sum(K ** l for l in range(L + 1)) ** N
    o = kmo.Run(kmo.make_config("FiniteReplicatedLog", N=N, L=L, K=K))
    assert o.distinct == expected and o.verdict == "ok"
    if expected <= 20000:
        r = A.bfs(A.FiniteReplicatedLogModel(N, L, K))
        assert r["distinct"] == expected and r["generated"] == o.generated and r["levels"] == o.levels


@pytest.mark.parametrize("model", KAFKA)
@pytest.mark.parametrize("N", [2, 3, 4])
def test_kafka_first_levels(model, N):
    # level 0: Init; level 1: N elections + N controller shrinks; level 2: N(5N-3), or N(4N-2)
    # for Kip320 whose fenced become-follower needs the leader to have become leader first
    o = kmo.Run(kmo.make_config(model, N=N, L=1, R=1, E=1))
    assert o.levels[0] == 1 and o.levels[1] == 2 * N
    assert o.levels[2] == (N * (4 * N - 2) if model == "Kip320" else N * (5 * N - 3))


@pytest.mark.parametrize("model", KAFKA)
def test_leader_in_isr_false_at_init(model):  # KafkaReplication.tla:42,117-119,345
    o = kmo.Run(kmo.make_config(model, N=3, L=2, R=2, E=1, invariants=("LeaderInIsr",)))
    assert (o.verdict, o.viol_inv, o.viol_depth, o.distinct) == ("invariant", "LeaderInIsr", 1, 1)
    m = A.make_model(model, N=3, L=2, R=2, E=1)
    assert not m.LeaderInIsr(m.Init())


# ---- qualitative verdicts the spec comments claim ----------------------------------------------
def test_kip320_holds_typeok_weak_strong():  # Kip320.tla:168,170,171
    for (N, L, R, E) in [(3, 2, 2, 2), (3, 3, 3, 1), (2, 3, 3, 2)]:
        o = kmo.Run(kmo.make_config("Kip320", N=N, L=L, R=R, E=E, invariants=("TypeOk", "WeakIsr", "StrongIsr")))
        assert o.verdict == "ok", (N, L, R, E)


@pytest.mark.parametrize("model", ["KafkaTruncateToHighWatermark", "Kip101", "Kip279", "Kip320FirstTry"])
def test_earlier_models_violate_strong_isr(model):
    # KafkaTruncateToHighWatermark.tla:23-27, Kip279.tla:20-23, Kip320.tla:127-132, Kip320FirstTry.tla:27-33
    o = kmo.Run(kmo.make_config(model, N=3, L=2, R=2, E=2, invariants=("TypeOk", "StrongIsr")))
    assert o.verdict == "invariant" and o.viol_inv == "StrongIsr"
    # the witness really violates it, and every state on its path is a successor of the previous
    idx = int(o.res.viol_state_idx)
    assert not kmo.check_invariant(o.cfg, 2, o.state(idx))
    while True:
        par = kmo.lib().kmo_parent(o.h, idx)
        if par < 0:
            break
        succ = {s for (_a, s) in kmo.successors(o.cfg, o.state(par), o.sb)}
        assert o.state(idx) in succ
        idx = par


@pytest.mark.parametrize("model", KAFKA)
def test_typeok_is_invariant(model):
    o = kmo.Run(kmo.make_config(model, N=3, L=2, R=2, E=2, invariants=("TypeOk",)))
    assert o.verdict == "ok"


# ---- Oracle-A == Oracle-B ----------------------------------------------------------------------
def canon(p, s):
    """Oracle-A state -> the C oracle's canonical bytes (see oracle/kmc_oracle.c header)."""
    out = bytearray()
    for r in range(p.N):
        end, recs = s.replicaLog[r]
        hw, ep, ldr, isr = s.replicaState[r]
        out += bytes([end, hw, ep + 1, 0 if ldr == A.NONE else ldr + 1, sum(1 << x for x in isr)])
        out += bytes(0 if rec == A.NIL else 1 + rec[0] * (p.E + 1) + rec[1] for rec in recs)
    q = s.quorumState
    out += bytes([s.nextRecordId, s.nextLeaderEpoch, q[0] + 1, 0 if q[1] == A.NONE else q[1] + 1,
                  sum(1 << x for x in q[2])])
    reqs = {e: (l, i) for (e, l, i) in s.leaderAndIsrRequests}
    assert sorted(reqs) == list(range(s.nextLeaderEpoch))  # the set <-> epoch-indexed array bijection
    for e in range(p.E + 1):
        if e in reqs:
            l, i = reqs[e]
            out += bytes([0 if l == A.NONE else l + 1, sum(1 << x for x in i)])
        else:
            out += bytes([0, 0])
    return bytes(out)


LADDER = [(2, 1, 1, 1), (2, 2, 2, 1), (3, 1, 1, 1), (2, 2, 1, 2), (3, 2, 2, 1), (2, 3, 2, 2)]


@pytest.mark.parametrize("model", KAFKA)
@pytest.mark.parametrize("N,L,R,E", LADDER)
def test_python_and_c_oracles_agree(model, N, L, R, E):
    inv = ("TypeOk", "WeakIsr", "StrongIsr")
    m = A.make_model(model, N=N, L=L, R=R, E=E)
    a = A.bfs(m, invariants=inv, stop_on_violation=False, keep_states=True)
    o = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=inv, stop_on_violation=False))
    assert a["distinct"] == o.distinct and a["generated"] == o.generated and a["depth"] == o.depth
    assert a["levels"] == o.levels
    assert list(a["action_generated"].values()) == o.action_generated[:len(m.action_names)]
    assert a["deadlock_states"] == o.deadlock_states
    if a["violation"]:
        assert a["violation"]["invariant"] == o.viol_inv and a["violation"]["depth"] == o.viol_depth
        assert a["violation"]["per_invariant"] == {k: v for k, v in o.viol_count.items() if v}
    else:
        assert o.viol_inv is None
    # identical reachable sets, level by level, in the canonical serialisation
    for k, states in enumerate(a["level_states"]):
        assert {canon(m.p, s) for s in states} == o.level_states(k)


@pytest.mark.parametrize("model", KAFKA)
def test_successor_multisets_agree_on_sampled_states(model):
    # differential test per state: same multiset of (action, successor) from both oracles
    N, L, R, E = 3, 2, 2, 2
    m = A.make_model(model, N=N, L=L, R=R, E=E)
    a = A.bfs(m, invariants=(), max_states=3000, keep_states=True)
    cfg = kmo.make_config(model, N=N, L=L, R=R, E=E)
    sb = N * (5 + L) + 5 + 2 * (E + 1)
    sample = [s for lvl in a["level_states"] for s in lvl][::7]
    for s in sample:
        want = sorted((ai, canon(m.p, t)) for (ai, t) in m.Next(s))
        got = sorted(kmo.successors(cfg, canon(m.p, s), sb))
        # the C oracle folds Kip279's two coinciding disjuncts (Kip279.tla:47-51) into two emits too
        assert got == want


def test_golden_fixtures_are_consistent():
    # large-configuration fixtures produced by tests/golden/make_golden.sh (C oracle, 8 threads)
    for name in ("oracle_kip320_3_5_5_2.json", "oracle_kip320_3_6_6_2.json"):
        g = json.load(open(os.path.join(GOLDEN, name)))
        assert sum(g["levels"]) == g["distinct"] and len(g["levels"]) == g["depth"]
        assert sum(g["action_generated"]) + 1 == g["generated"]
        assert g["verdict"] == 0 and g["levels"][:3] == [1, 6, 30]
    # the other models start 1, 2N, N(5N-3); Kip279 with 5 brokers and the TruncateToHW headline binding
    for name, n in (("oracle_kip279_5_2_2_1.json", 5), ("oracle_thw_3_5_5_2.json", 3), ("oracle_kip101_3_5_5_2.json", 3),
                    ("oracle_kip279_3_5_5_2.json", 3), ("oracle_kip320firsttry_3_5_5_2.json", 3)):
        g = json.load(open(os.path.join(GOLDEN, name)))
        assert sum(g["levels"]) == g["distinct"] and len(g["levels"]) == g["depth"]
        assert sum(g["action_generated"]) + 1 == g["generated"]
        assert g["verdict"] == 0 and g["levels"][:3] == [1, 2 * n, n * (5 * n - 3)]


def test_fingerprint_only_mode_of_the_c_oracle_agrees_with_its_exact_mode():
    """oracle/kmc_oracle --fp-only keeps 64-bit hashes instead of states (so that searches beyond the exact mode's RAM
    fit: the 810 M-state KafkaTruncateToHighWatermark 3/6/6/2 run that cross-checks the GPU's count).  Where both modes
    fit they must report the same numbers."""
    import subprocess
    exe = os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "kmc_oracle")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(exe)])
    runs = []
    for extra in ([], ["--fp-only", "--table-log2", "22"]):
        out = subprocess.run([exe, "--model", "Kip320", "--N", "3", "--L", "3", "--R", "3", "--E", "1", "--threads", "4",
                              "--inv", "7"] + extra, capture_output=True, text=True, timeout=300).stdout
        runs.append(json.loads(out))
    a, b = runs
    assert (a["fp_only"], b["fp_only"]) == (0, 1)
    for k in ("distinct", "generated", "depth", "levels", "action_generated", "deadlock_states", "verdict"):
        assert a[k] == b[k], k
    assert a["distinct"] == 176440


def test_fp_only_fixture_of_the_headline_truncate_to_hw_binding_equals_the_gpu_record():
    """KafkaTruncateToHighWatermark at the headline's constants (3 brokers, LogSize 6): 810,380,080 states — beyond the
    exact oracle's RAM, so the CPU side is the oracle's fingerprint-only mode (tests/golden/oracle_fp_thw_3_6_6_2.json, ten
    minutes on 8 cores).  The GPU's committed runs of the same binding (profiles/r02_ladder.jsonl, two hash seeds) report
    the same distinct / generated / depth and the same first and last levels: three unrelated 64-bit hash functions agree."""
    g = json.load(open(os.path.join(GOLDEN, "oracle_fp_thw_3_6_6_2.json")))
    assert g["fp_only"] == 1 and g["verdict"] == 0
    assert sum(g["levels"]) == g["distinct"] == 810380080 and len(g["levels"]) == g["depth"] == 52
    assert sum(g["action_generated"]) + 1 == g["generated"] == 3462005758
    assert g["levels"][:3] == [1, 6, 36]
    ladder = os.path.join(os.path.dirname(GOLDEN), "..", "profiles", "r02_ladder.jsonl")
    runs = [json.loads(l) for l in open(ladder) if "stretch_truncate_to_hw_3_6_6_2" in l]
    assert len(runs) >= 2 and len({r["config"].get("hash_seed", 0) for r in runs}) >= 2
    for r in runs:
        assert (r["verdict"], r["distinct"], r["generated"], r["depth"]) == ("ok", g["distinct"], g["generated"], g["depth"])
        assert r["levels_head"] == g["levels"][:len(r["levels_head"])] and r["levels_tail"] == g["levels"][-len(r["levels_tail"]):]
        assert r["widest_level"] == max(g["levels"])


def test_fp_only_fixtures_at_the_headline_constants_are_consistent():
    for name in ("oracle_fp_kip101_3_6_6_2.json", "oracle_fp_kip279_3_6_6_2.json", "oracle_fp_kip320firsttry_3_6_6_2.json"):
        g = json.load(open(os.path.join(GOLDEN, name)))
        assert g["fp_only"] == 1 and g["verdict"] == 0 and (g["N"], g["L"], g["R"], g["E"]) == (3, 6, 6, 2)
        assert sum(g["levels"]) == g["distinct"] and len(g["levels"]) == g["depth"]
        assert sum(g["action_generated"]) + 1 == g["generated"] and g["levels"][:3] == [1, 6, 36]
