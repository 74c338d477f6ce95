"""GPU: every Kafka model at the headline's own constants against EXACT fixtures (round 4: Oracle-O, oracle/orbit_oracle.c — full
states, one per orbit of the permutations of Replicas, weighted counts: tests/golden/orbit_*_3_6_6_2.json), the plain search and
the orbit-counting one, verdicts and violation counts included; the C oracle's fingerprint-only files stay as a third witness.

The second binding SURVEY 8d names for "KafkaReplication.tla, 3 brokers, maxLogLen=6" — KafkaTruncateToHighWatermark,
whose Next (KafkaTruncateToHighWatermark.tla:33-42) is built purely from KafkaReplication.tla's actions — at the headline's
own constants.  810,380,080 distinct states: more than the exact CPU oracle can hold, so the fixture comes from the oracle's
fingerprint-only mode (tests/golden/oracle_fp_thw_3_6_6_2.json: another hash over another state encoding, another table,
another BFS).  (The file name sorts last on purpose: the largest single-GPU test of the suite.)"""
import json
import os

import pytest

from kafka_specification_amd import CheckerConfig, ModelChecker

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_truncate_to_hw_at_the_headline_constants_matches_the_fingerprint_only_oracle():
    g = json.load(open(os.path.join(GOLDEN, "oracle_fp_thw_3_6_6_2.json")))
    cfg = CheckerConfig(model="KafkaTruncateToHighWatermark", n_replicas=g["N"], log_size=g["L"], max_records=g["R"],
                        max_leader_epoch=g["E"], invariants=("TypeOk",), table_capacity=1 << 31, frontier_capacity=1 << 27)
    with ModelChecker(cfg) as mc:
        r = mc.run()
    assert r.verdict == "ok" and r.queue_left == 0
    assert (r.distinct, r.generated, r.depth, r.levels) == (g["distinct"], g["generated"], g["depth"], g["levels"])
    assert list(r.action_generated.values()) == g["action_generated"][:len(r.action_generated)]
    assert r.deadlock_states == g["deadlock_states"]
    assert sum(r.levels) == r.distinct and sum(r.action_generated.values()) + 1 == r.generated


@pytest.mark.parametrize("model,fixture", [("Kip101", "oracle_kip101_3_5_5_2.json"), ("Kip279", "oracle_kip279_3_5_5_2.json"),
                                           ("Kip320FirstTry", "oracle_kip320firsttry_3_5_5_2.json")])
def test_the_other_kafka_models_near_the_headline_size(model, fixture):
    """Kip101, Kip279 and Kip320FirstTry with 3 brokers, LogSize 5, MaxRecords 5, MaxLeaderEpoch 2 and TypeOk only (they
    violate StrongIsr by design): 161-177 M states each, against the exact oracle's fixtures (which its fingerprint-only mode
    reproduces).  With Kip320 and KafkaTruncateToHighWatermark above, every Kafka model is pinned beyond 10^8 states."""
    g = json.load(open(os.path.join(GOLDEN, fixture)))
    cfg = CheckerConfig(model=model, n_replicas=g["N"], log_size=g["L"], max_records=g["R"], max_leader_epoch=g["E"],
                        invariants=("TypeOk",), table_capacity=1 << 30, frontier_capacity=1 << 26)
    with ModelChecker(cfg) as mc:
        r = mc.run()
    assert r.verdict == "ok" and r.queue_left == 0
    assert (r.distinct, r.generated, r.depth, r.levels) == (g["distinct"], g["generated"], g["depth"], g["levels"])
    assert list(r.action_generated.values()) == g["action_generated"][:len(r.action_generated)]
    assert r.deadlock_states == g["deadlock_states"]


@pytest.mark.parametrize("model,fixture", [("Kip101", "oracle_fp_kip101_3_6_6_2.json"), ("Kip279", "oracle_fp_kip279_3_6_6_2.json"),
                                           ("Kip320FirstTry", "oracle_fp_kip320firsttry_3_6_6_2.json")])
def test_the_other_kafka_models_at_the_headline_constants(model, fixture):
    """... and at the headline's own constants (3 brokers, LogSize 6, MaxRecords 6, MaxLeaderEpoch 2; TypeOk only):
    607-655 M states each, against the C oracle's fingerprint-only mode.  Every Kafka model of the reference is then
    cross-checked at "3 brokers, maxLogLen=6" by an engine that shares nothing with the GPU's."""
    g = json.load(open(os.path.join(GOLDEN, fixture)))
    cfg = CheckerConfig(model=model, n_replicas=g["N"], log_size=g["L"], max_records=g["R"], max_leader_epoch=g["E"],
                        invariants=("TypeOk",), table_capacity=1 << 31, frontier_capacity=1 << 27)
    with ModelChecker(cfg) as mc:
        r = mc.run()
    assert r.verdict == "ok" and r.queue_left == 0
    assert (r.distinct, r.generated, r.depth, r.levels) == (g["distinct"], g["generated"], g["depth"], g["levels"])
    assert list(r.action_generated.values()) == g["action_generated"][:len(r.action_generated)]
    assert r.deadlock_states == g["deadlock_states"]


# ---- round 4: the same constants against EXACT fixtures (Oracle-O stores full states; nothing here is a hash against a hash) ----
EXACT = [("KafkaTruncateToHighWatermark", "orbit_thw_3_6_6_2.json", "oracle_fp_thw_3_6_6_2.json"),
         ("Kip101", "orbit_kip101_3_6_6_2.json", "oracle_fp_kip101_3_6_6_2.json"),
         ("Kip279", "orbit_kip279_3_6_6_2.json", "oracle_fp_kip279_3_6_6_2.json"),
         ("Kip320FirstTry", "orbit_kip320firsttry_3_6_6_2.json", "oracle_fp_kip320firsttry_3_6_6_2.json"),
         ("Kip320", "orbit_kip320_3_6_6_2.json", "oracle_kip320_3_6_6_2.json")]


@pytest.mark.parametrize("symmetry", [False, True], ids=["plain", "orbit-counting"])
@pytest.mark.parametrize("model,fixture", [(m, e) for m, e, _ in EXACT], ids=[m for m, _, _ in EXACT])
def test_every_kafka_model_at_the_headline_constants_matches_the_exact_orbit_oracle(model, fixture, symmetry):
    g = json.load(open(os.path.join(GOLDEN, fixture)))
    cfg = CheckerConfig(model=model, n_replicas=3, log_size=6, max_records=6, max_leader_epoch=2, invariants=("TypeOk",),
                        table_capacity=1 << (29 if symmetry else 31), frontier_capacity=1 << (25 if symmetry else 27),
                        symmetry=symmetry)
    with ModelChecker(cfg) as mc:
        r = mc.run()
    assert r.verdict == "ok" and r.queue_left == 0
    assert (r.distinct, r.generated, r.depth, r.levels) == (g["distinct"], g["generated"], g["depth"], g["levels"])
    assert list(r.action_generated.values()) == g["action_generated"][:len(r.action_generated)]
    assert r.deadlock_states == g["deadlock_states"]
    if symmetry:
        assert r.orbit_representatives == g["stored"]      # the two orbit searches choose representatives differently, not orbits


@pytest.mark.parametrize("symmetry", [False, True], ids=["plain", "orbit-counting"])
@pytest.mark.parametrize("model,fixture", [(m, e) for m, e, _ in EXACT], ids=[m for m, _, _ in EXACT])
def test_first_violation_at_the_headline_constants_matches_the_exact_orbit_oracle(model, fixture, symmetry):
    """TypeOk, WeakIsr and StrongIsr, stopping at the first violation: the depth and how many states of that level violate
    each invariant (Kip320: none — KafkaReplication.tla:320-340 hold, Kip320.tla:168-171)."""
    g = json.load(open(os.path.join(GOLDEN, fixture)))
    cfg = CheckerConfig(model=model, n_replicas=3, log_size=6, max_records=6, max_leader_epoch=2,
                        invariants=("TypeOk", "WeakIsr", "StrongIsr"),
                        # (Kip320 violates nothing: that leg runs the whole graph out, 279,753,922 states)
                        table_capacity=1 << (30 if model == "Kip320" and not symmetry else 28),
                        frontier_capacity=1 << (26 if model == "Kip320" and not symmetry else 25),
                        symmetry=symmetry, max_levels=0 if model == "Kip320" else 16)
    with ModelChecker(cfg) as mc:
        r = mc.run()
    depths = [d for d in g["first_violation_depth"][:3] if d]
    if not depths:
        assert r.verdict == "ok" and sum(g["violating_states"]) == 0 and r.distinct == g["distinct"]
        return
    first = min(depths)
    assert r.verdict == "invariant" and r.violation_depth == first
    want = {n: (g["violating_at_first_depth"][k] if g["first_violation_depth"][k] == first else 0)
            for k, n in enumerate(("TypeOk", "WeakIsr", "StrongIsr"))}
    got = {n: r.violation_count.get(n, 0) for n in want}
    assert got == want
    assert r.violated_invariant == next(n for n in ("TypeOk", "WeakIsr", "StrongIsr") if want[n])


@pytest.mark.parametrize("symmetry", [False, True], ids=["plain", "orbit-counting"])
def test_one_step_beyond_the_headline_kip320_with_logsize_7_matches_the_exact_orbit_oracle(symmetry):
    """Kip320, 3 brokers, LogSize 7, MaxRecords 7, MaxLeaderEpoch 2: 973,929,178 distinct states / 3,222,426,940 generated /
    depth 52 — the largest exhaustive check with an EXACT CPU witness (Oracle-O: 162,341,877 stored full states,
    tests/golden/orbit_kip320_3_7_7_2.json): level sizes, per-disjunct counts, deadlocks, no violation."""
    g = json.load(open(os.path.join(GOLDEN, "orbit_kip320_3_7_7_2.json")))
    cfg = CheckerConfig(model="Kip320", n_replicas=3, log_size=7, max_records=7, max_leader_epoch=2,
                        invariants=("TypeOk", "WeakIsr", "StrongIsr"),
                        table_capacity=1 << (30 if symmetry else 32), frontier_capacity=1 << (26 if symmetry else 28),
                        symmetry=symmetry)
    with ModelChecker(cfg) as mc:
        r = mc.run()
    assert r.verdict == "ok" and r.queue_left == 0 and sum(g["violating_states"]) == 0
    assert (r.distinct, r.generated, r.depth, r.levels) == (g["distinct"], g["generated"], g["depth"], g["levels"])
    assert list(r.action_generated.values()) == g["action_generated"][:len(r.action_generated)]
    assert r.deadlock_states == g["deadlock_states"]
    if symmetry:
        assert r.orbit_representatives == g["stored"]


def test_the_stretch_kip320_with_three_epochs_plain_search_with_wide_entries_matches_the_exact_orbit_oracle():
    """Kip320 3 / 6 / 6 / 3: 6,452,700,520 distinct states / 20,756,484,505 generated / depth 54 — the workload the frontier
    sharding is designed for and the size at which a 64-bit fingerprint loses a state (n^2 / 2^65 = 1.1: the
    default search returns ...519).  The PLAIN search with 128-bit entries (137 GB of seen-set, 1.7 s) against Oracle-O's exact
    search over orbits (tests/golden/orbit_kip320_3_6_6_3.json, round 5: 1,075,491,542 full states, no fingerprint): every
    level, every per-disjunct count, the deadlocked states.  (The orbit-counting search is held to the same file in
    tests/test_gpu_symmetry.py; until round 5 the two GPU searches only had each other.)"""
    g = json.load(open(os.path.join(GOLDEN, "orbit_kip320_3_6_6_3.json")))
    cfg = CheckerConfig(model="Kip320", n_replicas=3, log_size=6, max_records=6, max_leader_epoch=3,
                        invariants=("TypeOk", "WeakIsr", "StrongIsr"), wide_fingerprint=True,
                        table_capacity=1 << 33, frontier_capacity=1 << 30)
    with ModelChecker(cfg) as mc:
        r = mc.run()
    assert r.verdict == "ok" and r.queue_left == 0 and sum(g["violating_states"]) == 0
    assert (r.distinct, r.generated, r.depth, r.levels) == (g["distinct"], g["generated"], g["depth"], g["levels"])
    assert list(r.action_generated.values()) == g["action_generated"][:len(r.action_generated)]
    assert r.deadlock_states == g["deadlock_states"]


def test_baseline_config4_at_the_surveys_own_sizing_over_twelve_levels_matches_the_exact_oracle():
    """BASELINE config 4 (Kip279, five brokers) at SURVEY section 8(a.0)'s sizing — LogSize 4, MaxRecords 4, MaxLeaderEpoch 3, four
    words per state — where a log holds up to four epochs and FirstNonMatchingOffsetFromTail (Kip279.tla:39-45) has something to
    truncate.  Not exhaustible (three times more states per level): the first twelve levels, 318,475,476 states, against the exact C
    oracle's prefix (tests/golden/oracle_kip279_5_4_4_3_levels12.json) — level sizes, per-disjunct generated counts, deadlocks."""
    from kafka_specification_amd.configs import CONFIG4_DEEP, CONFIG4_DEEP_LEVELS
    g = json.load(open(os.path.join(GOLDEN, f"oracle_kip279_5_4_4_3_levels{CONFIG4_DEEP_LEVELS}.json")))
    cfg = CheckerConfig(**CONFIG4_DEEP, max_levels=CONFIG4_DEEP_LEVELS, table_capacity=1 << 31, frontier_capacity=1 << 28)
    with ModelChecker(cfg) as mc:
        r = mc.run()
        st = mc.level_stats()
    assert r.verdict == "level_limit" and r.queue_left == g["levels"][-1]
    assert (r.distinct, r.generated, r.depth, r.levels) == (g["distinct"], g["generated"], g["depth"], g["levels"]) and g["distinct"] == 318475476
    assert list(r.action_generated.values()) == g["action_generated"][:len(r.action_generated)]
    assert r.deadlock_states == g["deadlock_states"]
    # ... and the per-level records (kmc_level_stats) are the same search, level by level
    assert [x["new_states"] for x in st] == g["levels"][1:] and [x["frontier"] for x in st] == g["levels"][:-1]
    assert sum(sum(x["generated"].values()) for x in st) + 1 == r.generated
    assert all(x["expand_ms"] > 0 for x in st) and abs(sum(x["expand_ms"] for x in st) - 1e3 * r.seconds_expand) < 1e-3 * max(1.0, 1e3 * r.seconds_expand)
