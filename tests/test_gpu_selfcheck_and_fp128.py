"""The engine's own guards against silent wrong answers, on the GPU:

  * conservation of successors (always on): what pass 2 of k_expand dispatched must be what reached the seen-set, and what
    the seen-set claimed must be what was appended — counted on either side of the ring / flush / stager, where round 1's
    miscompiled kernel lost successors while every other counter still agreed;
  * KMC_VERIFY's checksum: the second build must reach the sink with the same multiset of successor fingerprints;
  * a fault-injection build (-DKMC_FAULT_DROP=1: one successor per launch vanishes between the ring and the sink) makes
    both fire — and shows that the per-action counts the old self-check compared do NOT notice it;
  * wide (128-bit) seen-set entries: same answers as the narrow table, traces, checkpoints and shards included; a forced
    64-bit fingerprint collision is told apart instead of losing a state.
"""
import pytest

import kmo
from kafka_specification_amd import CheckerConfig, KmcError, ModelChecker

pytestmark = pytest.mark.gpu

INV = ("TypeOk", "WeakIsr", "StrongIsr")
SMALL = dict(model="Kip320", n_replicas=3, log_size=2, max_records=2, max_leader_epoch=1)


def run(cfg, **kw):
    with ModelChecker(CheckerConfig(**cfg, invariants=INV, table_capacity=1 << 22, frontier_capacity=1 << 20, **kw)) as mc:
        return mc.run()


def test_conservation_holds_on_every_level_of_a_healthy_run():
    o = kmo.Run(kmo.make_config("Kip320", N=3, L=2, R=2, E=1, invariants=INV))
    r = run(SMALL)
    assert (r.verdict, r.distinct, r.generated) == (o.verdict, o.distinct, o.generated)
    assert 0 < r.generated_repeats < r.generated   # Kip320.tla:82-83: both reasons to shrink the ISR at once


def test_a_lost_successor_trips_the_conservation_check(monkeypatch):
    monkeypatch.setenv("KMC_JIT_DEFINES", "-DKMC_TUNING=1 -DKMC_FAULT_DROP=1")
    with pytest.raises(KmcError, match="conservation violated.*lost or invented successors"):
        run(SMALL)
    with pytest.raises(KmcError, match="conservation violated"):   # the progress-callback path checks level by level too
        with ModelChecker(CheckerConfig(**SMALL, invariants=INV, table_capacity=1 << 22, frontier_capacity=1 << 20)) as mc:
            mc.run(progress=lambda info: None)


def test_a_lost_successor_trips_the_verify_checksum_but_not_the_old_counts(monkeypatch):
    monkeypatch.setenv("KMC_JIT_DEFINES", "-DKMC_TUNING=1 -DKMC_FAULT_DROP=1")
    monkeypatch.setenv("KMC_VERIFY", "1")
    # the second build runs in DRY mode, where the fault is not injected: its successors are complete, the first build's
    # are one short — "generated / deadlock / violation counts" (all the round-2 check compared) still agree
    with pytest.raises(KmcError, match="KMC_VERIFY.*successors reaching the seen-set differ"):
        run(SMALL)


def test_a_lost_successor_is_caught_in_the_sharded_step_interface_too(monkeypatch):
    from kafka_specification_amd.sharded import check_loopback
    monkeypatch.setenv("KMC_JIT_DEFINES", "-DKMC_TUNING=1 -DKMC_FAULT_DROP=1")
    cfg = CheckerConfig(model="AsyncIsr", n_replicas=3, log_size=2, max_leader_epoch=2, invariants=("ValidHighWatermark",),
                        table_capacity=1 << 20, frontier_capacity=1 << 18)
    with pytest.raises(KmcError, match="conservation violated"):
        check_loopback(cfg, 2, 0, None, None, None)


# ---- wide fingerprints ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("model,N,L,R,E", [("Kip320", 3, 2, 2, 2), ("Kip279", 3, 2, 2, 1), ("Kip101", 2, 3, 3, 2),
                                           ("KafkaTruncateToHighWatermark", 3, 1, 1, 2), ("Kip320FirstTry", 3, 2, 2, 1)])
def test_wide_table_gives_the_oracles_answers(model, N, L, R, E):
    inv = ("TypeOk",)
    o = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=inv, threads=8))
    cfg = CheckerConfig(model=model, n_replicas=N, log_size=L, max_records=R, max_leader_epoch=E, invariants=inv,
                        table_capacity=1 << 23, frontier_capacity=1 << 21, wide_fingerprint=True)
    levels = []
    with ModelChecker(cfg) as mc:
        r = mc.run(progress=lambda info: levels.append({mc.unpack(row) for row in mc.frontier_states()}))
    assert (r.verdict, r.distinct, r.generated, r.depth, r.levels, r.deadlock_states) == \
        (o.verdict, o.distinct, o.generated, o.depth, o.levels, o.deadlock_states)
    for k in range(len(o.levels)):
        assert levels[k] == o.level_states(k)
    with ModelChecker(cfg) as mc:   # and without a callback: the chained-launch path
        r2 = mc.run()
    assert (r2.distinct, r2.generated, r2.levels) == (o.distinct, o.generated, o.levels)


def test_wide_table_traces_checkpoints_and_shards(tmp_path):
    from kafka_specification_amd.sharded import check_loopback
    inv = ("TypeOk", "StrongIsr")
    consts = dict(model="Kip279", n_replicas=3, log_size=2, max_records=3, max_leader_epoch=2)
    o = kmo.Run(kmo.make_config("Kip279", N=3, L=2, R=3, E=2, invariants=inv, threads=8))
    assert o.verdict == "invariant"
    cfg = CheckerConfig(**consts, invariants=inv, keep_trace=True, table_capacity=1 << 23, frontier_capacity=1 << 21,
                        wide_fingerprint=True)
    with ModelChecker(cfg) as mc:
        r = mc.run()
        trace = mc.trace()
    assert (r.verdict, r.violated_invariant, r.violation_depth) == ("invariant", o.viol_inv, o.viol_depth)
    assert len(trace) == o.viol_depth
    ocfg = kmo.make_config("Kip279", N=3, L=2, R=3, E=2, invariants=())
    for (_a0, s0), (a1, s1) in zip(trace, trace[1:]):   # every step is a step of the oracle's Next
        assert any(t == bytes(s1) for _k, t in kmo.successors(ocfg, bytes(s0), o.sb))
    # checkpoint at a level limit, recover into a fresh wide handle, finish: the uninterrupted run's numbers
    full = kmo.Run(kmo.make_config("Kip279", N=3, L=2, R=3, E=2, invariants=("TypeOk",), threads=8))
    c2 = CheckerConfig(**consts, invariants=("TypeOk",), table_capacity=1 << 23, frontier_capacity=1 << 21, wide_fingerprint=True)
    path = str(tmp_path / "wide.ckpt")
    from dataclasses import replace
    with ModelChecker(replace(c2, max_levels=9)) as mc:
        assert mc.run().verdict == "level_limit"
        mc.save_checkpoint(path)
    with ModelChecker(c2) as mc:
        mc.load_checkpoint(path)
        r = mc.resume()
    assert (r.verdict, r.distinct, r.generated, r.levels) == ("ok", full.distinct, full.generated, full.levels)
    with ModelChecker(replace(c2, wide_fingerprint=False)) as mc:   # a narrow handle refuses the wide file
        with pytest.raises(KmcError, match="fingerprint width|capacities"):
            mc.load_checkpoint(path)
    res = check_loopback(c2, 3, 0, None, None, None)
    assert (res.verdict, res.distinct, res.generated, res.levels) == ("ok", full.distinct, full.generated, full.levels)


def test_wide_table_tells_colliding_fingerprints_apart():
    """Distinct states with the SAME fingerprint.  A real 64-bit collision cannot be constructed on demand, so they are
    made: a test-only code object (-DKMC_TEST_FP_BITS=10) reduces every fingerprint to 10 bits — the 39,619 states of this
    model then share 1,024 fingerprints.  The narrow table merges them (at most 1,024 "distinct" states survive); the wide
    one compares the untouched 64-bit check word, walks past every collision and finds all of them."""
    import os
    o = kmo.Run(kmo.make_config("Kip320", N=3, L=2, R=2, E=1, invariants=("TypeOk",)))
    os.environ["KMC_JIT_DEFINES"] = "-DKMC_TUNING=1 -DKMC_TEST_FP_BITS=10"
    try:
        narrow = run(SMALL)
        wide = run(SMALL, wide_fingerprint=True)
    finally:
        del os.environ["KMC_JIT_DEFINES"]
    assert narrow.distinct <= 1024 < o.distinct
    assert (wide.verdict, wide.distinct, wide.generated, wide.levels) == (o.verdict, o.distinct, o.generated, o.levels)


@pytest.mark.parametrize("P", [2, 3])
def test_wide_table_across_shards_loses_no_colliding_state_to_the_sender_side_filter(P):
    """with 128-bit entries on shards the sender-side duplicate filter (on by default for 2-4 shards) remembered
    64-bit fingerprints only — a second distinct remote state with the same fingerprint was dropped at the sender and never
    met its owner's check-word comparison, unseen by the conservation law.  The filter is off under wide_fingerprint now.
    Collisions on demand (-DKMC_TEST_FP_BITS=10: 39,619 states share 1,024 fingerprints): P logical shards find all of them,
    and nothing was filtered."""
    import os
    from dataclasses import replace
    from kafka_specification_amd import sharded
    o = kmo.Run(kmo.make_config("Kip320", N=3, L=2, R=2, E=1, invariants=("TypeOk",)))
    os.environ["KMC_JIT_DEFINES"] = "-DKMC_TUNING=1 -DKMC_TEST_FP_BITS=10"
    try:
        cfg = replace(CheckerConfig(**SMALL), wide_fingerprint=True, table_capacity=1 << 18, frontier_capacity=1 << 16,
                      send_capacity=1 << 14)
        r = sharded.check_loopback(cfg, P)
        filtered = sharded.run_sharded.last_send_filtered
    finally:
        del os.environ["KMC_JIT_DEFINES"]
    assert (r.verdict, r.distinct, r.generated, r.levels) == (o.verdict, o.distinct, o.generated, o.levels)
    assert filtered == 0
