"""GPU parity: the HIP engine (through the C ABI) against the C oracle on the same constants.
Bit-exact on every integer: distinct, generated (total and per action), depth, per-level
sizes, verdicts, violation depth/counts, and — where the level is small — the exact set of
states of every level."""
import pytest

import kmo
from kafka_specification_amd import CheckerConfig, ModelChecker

pytestmark = pytest.mark.gpu

KAFKA = ("KafkaTruncateToHighWatermark", "Kip101", "Kip279", "Kip320", "Kip320FirstTry")


def gpu_run(model, invariants=("TypeOk",), keep_levels=False, **kw):
    consts = {k: kw.pop(k) for k in list(kw) if k in ("n_replicas", "log_size", "max_records", "max_leader_epoch",
                                                      "n_log_records", "max_id")}
    cfg = CheckerConfig(model=model, invariants=invariants, table_capacity=kw.pop("table_capacity", 1 << 22),
                        frontier_capacity=kw.pop("frontier_capacity", 1 << 20), **consts, **kw)
    level_sets = []
    with ModelChecker(cfg) as mc:
        if keep_levels:
            def cb(info):
                fr = mc.frontier_states()
                level_sets.append({mc.unpack(row) for row in fr})
            res = mc.run(progress=cb)
        else:
            res = mc.run()
    return res, level_sets


def assert_same(res, o, nact):
    assert res.verdict == o.verdict
    assert res.distinct == o.distinct
    assert res.generated == o.generated
    assert res.depth == o.depth
    assert res.levels == o.levels
    assert list(res.action_generated.values()) == o.action_generated[:nact]
    assert res.deadlock_states == o.deadlock_states


@pytest.mark.parametrize("model", KAFKA)
@pytest.mark.parametrize("N,L,R,E", [(2, 2, 2, 1), (3, 2, 2, 1), (3, 1, 1, 2), (2, 3, 3, 2)])
def test_kafka_counts_and_level_sets(model, N, L, R, E):
    o = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=("TypeOk",)))
    res, level_sets = gpu_run(model, n_replicas=N, log_size=L, max_records=R, max_leader_epoch=E, keep_levels=True)
    assert_same(res, o, 10 if model == "Kip320FirstTry" else 9)
    assert len(level_sets) == len(o.levels)
    for k in range(len(o.levels)):
        assert level_sets[k] == o.level_states(k), f"level {k} state sets differ"


@pytest.mark.parametrize("model,N,L,R,E", [("Kip320", 3, 3, 3, 2), ("KafkaTruncateToHighWatermark", 3, 3, 3, 1),
                                           ("Kip279", 3, 2, 3, 2), ("Kip101", 3, 3, 2, 2),
                                           ("Kip320FirstTry", 3, 3, 3, 1)])
def test_kafka_larger_counts(model, N, L, R, E):
    o = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=("TypeOk",), threads=8))
    res, _ = gpu_run(model, n_replicas=N, log_size=L, max_records=R, max_leader_epoch=E,
                     table_capacity=1 << 25, frontier_capacity=1 << 22)
    assert_same(res, o, 10 if model == "Kip320FirstTry" else 9)


@pytest.mark.parametrize("model", KAFKA)
def test_invariant_verdicts(model):
    inv = ("TypeOk", "WeakIsr", "StrongIsr")
    o = kmo.Run(kmo.make_config(model, N=3, L=2, R=2, E=2, invariants=inv))
    res, _ = gpu_run(model, invariants=inv, n_replicas=3, log_size=2, max_records=2, max_leader_epoch=2)
    assert res.verdict == o.verdict
    assert res.violated_invariant == o.viol_inv
    if o.viol_inv:
        assert res.violation_depth == o.viol_depth
        assert res.violation_count == o.viol_count
    assert res.levels == o.levels
    assert res.generated == o.generated


@pytest.mark.parametrize("model", KAFKA)
def test_leader_in_isr_fails_at_init(model):
    res, _ = gpu_run(model, invariants=("LeaderInIsr",), n_replicas=3, log_size=2, max_records=2, max_leader_epoch=1)
    assert res.verdict == "invariant" and res.violated_invariant == "LeaderInIsr"
    assert res.violation_depth == 1 and res.distinct == 1 and res.generated == 1


def test_idsequence():
    for M in (0, 1, 10):
        res, _ = gpu_run("IdSequence", max_id=M)
        assert res.verdict == "ok" and res.distinct == M + 2 and res.depth == M + 2 and res.generated == M + 2


@pytest.mark.parametrize("K,expected", [(1, 25), (2, 961), (3, 14641), (4, 116281)])
def test_finite_replicated_log_closed_form(K, expected):
    o = kmo.Run(kmo.make_config("FiniteReplicatedLog", N=2, L=4, K=K))
    res, level_sets = gpu_run("FiniteReplicatedLog", n_replicas=2, log_size=4, n_log_records=K, keep_levels=(K <= 2))
    assert res.distinct == expected == o.distinct
    assert_same(res, o, 3)
    for k, s in enumerate(level_sets):
        assert s == o.level_states(k)


def test_seed_independence():
    a, _ = gpu_run("Kip320", n_replicas=3, log_size=2, max_records=2, max_leader_epoch=1, hash_seed=1)
    b, _ = gpu_run("Kip320", n_replicas=3, log_size=2, max_records=2, max_leader_epoch=1, hash_seed=0xDEADBEEF)
    assert (a.distinct, a.generated, a.levels) == (b.distinct, b.generated, b.levels)


@pytest.mark.parametrize("model,N,L,R,E", [("Kip320", 4, 2, 2, 1), ("Kip279", 5, 1, 1, 1), ("Kip101", 4, 2, 1, 2),
                                           ("KafkaTruncateToHighWatermark", 6, 1, 1, 1),
                                           ("Kip320FirstTry", 8, 1, 1, 0), ("Kip320", 7, 1, 1, 0),
                                           ("Kip279", 7, 1, 1, 0)])
def test_wider_replica_sets(model, N, L, R, E):
    """4 to 8 replicas: other state widths, up to 328 action instances (six words of instance bits)."""
    inv = ("TypeOk", "WeakIsr", "StrongIsr")
    o = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=inv, threads=8))
    res, _ = gpu_run(model, invariants=inv, n_replicas=N, log_size=L, max_records=R, max_leader_epoch=E,
                     table_capacity=1 << 25, frontier_capacity=1 << 23)
    assert res.verdict == o.verdict and res.violated_invariant == o.viol_inv
    assert res.levels == o.levels and res.distinct == o.distinct and res.generated == o.generated
    assert list(res.action_generated.values()) == o.action_generated[:len(res.action_generated)]
    if o.viol_inv:
        assert res.violation_depth == o.viol_depth and res.violation_count == o.viol_count


VIOLATING = ("KafkaTruncateToHighWatermark", "Kip101", "Kip279", "Kip320FirstTry")


@pytest.mark.parametrize("model,N,L,R,E", [(m, 3, 2, 2, 2) for m in VIOLATING] + [
    ("KafkaTruncateToHighWatermark", 2, 3, 3, 2), ("KafkaTruncateToHighWatermark", 3, 3, 2, 1), ("Kip101", 2, 3, 3, 2),
    ("Kip101", 3, 3, 2, 1), ("Kip279", 3, 3, 2, 1), ("Kip320FirstTry", 3, 2, 3, 2)])   # constants the oracle finds a violation for
def test_continue_on_violation_matches_oracle(model, N, L, R, E):
    """TLC -continue on the four models the reference describes as losing committed data
    (KafkaTruncateToHighWatermark.tla:23-27, Kip101.tla / Kip279.tla:20-23, Kip320FirstTry.tla:27-33): the search
    does not stop at the violation, so it exhausts the model — every count of the complete run and the depth /
    per-invariant counts of the FIRST violating level against Oracle-B with stop_on_violation = 0."""
    inv = ("TypeOk", "WeakIsr", "StrongIsr")
    o = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=inv, stop_on_violation=False, threads=8))
    plain = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=("TypeOk",), threads=8))
    res, _ = gpu_run(model, invariants=inv, n_replicas=N, log_size=L, max_records=R, max_leader_epoch=E,
                     continue_on_violation=True, table_capacity=1 << 24, frontier_capacity=1 << 21)
    assert o.verdict == "invariant", "these constants must reach a violation"
    assert (res.verdict, res.violated_invariant) == ("invariant", o.viol_inv)
    assert (res.violation_depth, res.violation_count) == (o.viol_depth, o.viol_count)
    # ... and the numbers of the complete state graph, as if no invariant had been given
    assert (res.distinct, res.generated, res.depth, res.levels) == (o.distinct, o.generated, o.depth, o.levels)
    assert (res.distinct, res.generated, res.levels) == (plain.distinct, plain.generated, plain.levels)
    assert list(res.action_generated.values()) == o.action_generated[:len(res.action_generated)]
    assert res.deadlock_states == o.deadlock_states and res.queue_left == 0


@pytest.mark.parametrize("model,N,L,R,E", [("Kip320", 2, 2, 2, 1), ("Kip279", 3, 2, 2, 1), ("Kip320FirstTry", 2, 3, 3, 2),
                                           ("KafkaTruncateToHighWatermark", 3, 1, 1, 2)])
def test_deadlock_checking_matches_oracle(model, N, L, R, E):
    """CHECK_DEADLOCK TRUE (TLC's default): the bounded models have terminal states; verdict, the depth of the first
    deadlocked level and every count up to it against Oracle-B; the witness really has no successor."""
    ocfg = kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=("TypeOk",), check_deadlock=True, threads=4)
    o = kmo.Run(ocfg)
    cfg = CheckerConfig(model=model, n_replicas=N, log_size=L, max_records=R, max_leader_epoch=E, invariants=("TypeOk",),
                        check_deadlock=True, table_capacity=1 << 22, frontier_capacity=1 << 20)
    with ModelChecker(cfg) as mc:
        res = mc.run()
        witness = mc.unpack(mc.witness()) if res.verdict == "deadlock" else None
    assert o.verdict == "deadlock" and res.verdict == "deadlock"
    assert res.levels == o.levels and res.violation_depth == len(o.levels)
    assert res.generated == o.generated and res.deadlock_states == o.deadlock_states
    assert kmo.successors(ocfg, witness, o.sb) == []


@pytest.mark.parametrize("name,model,N,L,R,E,levels", [
    ("config 4's larger twin: Kip279, 5 brokers, MaxLeaderEpoch 2", "Kip279", 5, 2, 2, 2, 10),
    ("config 5: Kip320, 7 brokers, LogSize 8", "Kip320", 7, 8, 8, 3, 7)])
def test_baseline_multi_gpu_configs_match_oracle_prefix(name, model, N, L, R, E, levels):
    """BASELINE.json config 5 at its own constants (models/Kip320_7brokers.cfg) and config 4 with MaxLeaderEpoch 2 (the
    exhaustible binding, MaxLeaderEpoch 1, has its golden-fixture test in test_gpu_sharded_and_traces.py).
    Neither is exhaustible here (6.5e9 states after 22 levels / 8.8e8 after 11), so the pin is the oracle's BFS prefix:
    level sizes, generated (total and per action) and the exact state sets of the first levels (this
    check lived in tools/run_ladder.py only)."""
    inv = ("TypeOk",)
    o = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=inv, threads=8, max_states=1_500_000))
    k = min(levels, len(o.levels) - 1)      # the oracle stops after the level that crosses max_states: compare complete levels
    sets = []
    cfg = CheckerConfig(model=model, n_replicas=N, log_size=L, max_records=R, max_leader_epoch=E, invariants=inv,
                        max_levels=k, table_capacity=1 << 26, frontier_capacity=1 << 23)
    with ModelChecker(cfg) as mc:
        res = mc.run(progress=lambda info: sets.append({mc.unpack(r) for r in mc.frontier_states()}
                                                       if info["new_states"] <= 50_000 else None))
    assert res.verdict == "level_limit" and res.levels == o.levels[:k]
    for d, s in enumerate(sets):
        if s is not None:
            assert s == o.level_states(d), f"{name}: level {d} state sets differ"
    # generated up to (not including) the expansion of the last kept level: compare through a second oracle run
    o2 = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=inv, threads=8,
                                 max_states=sum(o.levels[:k - 1]) + 1))
    if len(o2.levels) == k:   # it stopped right after producing level k
        assert res.generated == o2.generated
        assert list(res.action_generated.values()) == o2.action_generated[:len(res.action_generated)]


def test_verify_mode_regenerates_every_level_with_a_second_build(monkeypatch):
    """KMC_VERIFY=1 (CLI: -verify): a second code object of the same source (-O1, a quarter of the occupancy target)
    re-generates every level's successors; the per-action, deadlock and violation counts of the two builds must agree.
    The self-check for constants beyond any oracle, after round 1 met a build that lost successors under register
    spilling.  Here both builds are right, so the run must simply equal the oracle's."""
    monkeypatch.setenv("KMC_VERIFY", "1")
    for model, N, L, R, E in (("Kip320", 3, 2, 2, 1), ("Kip279", 5, 1, 1, 1)):
        inv = ("TypeOk", "WeakIsr", "StrongIsr")
        o = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=inv, threads=8))
        res, _ = gpu_run(model, invariants=inv, n_replicas=N, log_size=L, max_records=R, max_leader_epoch=E,
                         table_capacity=1 << 24, frontier_capacity=1 << 21)
        assert (res.verdict, res.distinct, res.generated, res.levels) == (o.verdict, o.distinct, o.generated, o.levels)


def test_verify_mode_at_a_golden_size(monkeypatch):
    """KMC_VERIFY=1 on Kip320 3/5/5/2 (75,569,791 states): the second build lowers the guards the other way (guard<K> over
    run-time bindings against the per-instance block, kmc_device.h) and must regenerate every level with the same counts
    and the same checksum of successor fingerprints; the run must equal the golden fixture."""
    import json
    import os
    monkeypatch.setenv("KMC_VERIFY", "1")
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_kip320_3_5_5_2.json")))
    res, _ = gpu_run("Kip320", invariants=("TypeOk", "WeakIsr", "StrongIsr"), n_replicas=3, log_size=5, max_records=5,
                     max_leader_epoch=2, table_capacity=1 << 28, frontier_capacity=1 << 25)
    assert (res.verdict, res.distinct, res.generated, res.depth) == ("ok", g["distinct"], g["generated"], g["depth"])
    assert res.levels == g["levels"]
