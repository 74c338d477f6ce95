"""FULL leaves (k_expand's pass 2 for wide configurations under orbit counting: csrc/kmc_kernels.h, KmcKafka::FULL_LEAVES) forced
onto configurations every level of which the oracle holds as an exact set (-DKMC_FULL_LEAVES_MIN_INSTANCES=0
-DKMC_FULL_LEAVES_PLAIN=1, code objects prebuilt by build()).

By default the path only runs under orbit counting at seven brokers (tests/test_gpu_symmetry.py and
test_gpu_zzz_oracle_r_wide.py reach it there); here the same code is held to the oracle where every state can be compared:
plain and orbit counting, the three kernels (the search's, the level-step interface's owner bucketing, the enumerator), the
meta plane of the ring (predecessor fingerprints pulled from the source lane of a pair).

Round 5 met a build of this loop whose `generated` was too large under orbit counting and differed from run to run: the
compiler had sunk a ds_bpermute into the branch of the lanes that hold a pair, where a source lane outside it reads as 0
(kmc_pull, csrc/kmc_common.h).  The counts below are the ones that were wrong."""
import pytest

import kmo
from kafka_specification_amd import CheckerConfig, ModelChecker
from kafka_specification_amd.configs import FULL_LEAVES_DEFINES, FULL_LEAVES_SMALL, FULL_LEAVES_SYMMETRY
from kafka_specification_amd.sharded import check_loopback

pytestmark = pytest.mark.gpu
INV_INDEX = {"TypeOk": 0, "WeakIsr": 1, "StrongIsr": 2, "LeaderInIsr": 3}


@pytest.fixture(autouse=True)
def forced(monkeypatch):
    monkeypatch.setenv("KMC_JIT_DEFINES", FULL_LEAVES_DEFINES)


@pytest.mark.parametrize("model,N,L,R,E", FULL_LEAVES_SMALL)
def test_every_level_is_the_oracles_set(model, N, L, R, E):
    o = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=(), threads=4))
    cfg = CheckerConfig(model=model, invariants=(), n_replicas=N, log_size=L, max_records=R, max_leader_epoch=E,
                        table_capacity=1 << 24, frontier_capacity=1 << 22)
    level_sets = []
    with ModelChecker(cfg) as mc:
        res = mc.run(progress=lambda info: level_sets.append({mc.unpack(row) for row in mc.frontier_states()}))
    assert (res.verdict, res.distinct, res.generated, res.depth, res.levels) == (o.verdict, o.distinct, o.generated, o.depth, o.levels)
    assert list(res.action_generated.values()) == o.action_generated[:len(res.action_generated)]
    assert res.deadlock_states == o.deadlock_states
    for k in range(len(o.levels)):
        assert level_sets[k] == o.level_states(k), f"level {k} state sets differ"
    inv = ("TypeOk", "WeakIsr", "StrongIsr")
    ov = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=inv, threads=4))
    with ModelChecker(CheckerConfig(model=model, invariants=inv, n_replicas=N, log_size=L, max_records=R, max_leader_epoch=E,
                                    table_capacity=1 << 24, frontier_capacity=1 << 22)) as mc:
        rv = mc.run()
    assert (rv.verdict, rv.violated_invariant) == (ov.verdict, ov.viol_inv)
    if ov.viol_inv:
        assert (rv.violation_depth, rv.violation_count) == (ov.viol_depth, ov.viol_count)


@pytest.mark.parametrize("model,N,L,R,E", FULL_LEAVES_SYMMETRY)
def test_orbit_counting_reports_the_plain_counts(model, N, L, R, E):
    """(the doubly satisfied disjuncts of Kip320.tla:82-83 / Kip279.tla:47-51 weigh in with the deficit of the pair's SOURCE
    state: the number the sunk ds_bpermute got wrong)"""
    o = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=("TypeOk",), threads=8))
    for _ in range(2):   # (the wrong build differed from run to run)
        with ModelChecker(CheckerConfig(model=model, invariants=("TypeOk",), symmetry=True, n_replicas=N, log_size=L, max_records=R,
                                        max_leader_epoch=E, table_capacity=1 << 23, frontier_capacity=1 << 21)) as mc:
            res = mc.run()
        assert (res.verdict, res.distinct, res.generated, res.depth, res.levels) == (o.verdict, o.distinct, o.generated, o.depth, o.levels)
        assert list(res.action_generated.values()) == o.action_generated[:len(res.action_generated)]
        assert res.deadlock_states == o.deadlock_states
        assert res.orbit_representatives < res.distinct


@pytest.mark.parametrize("symmetry", [False, True], ids=["plain", "orbit-counting"])
@pytest.mark.parametrize("model", ["Kip279", "Kip101"])
def test_traces_the_enumerator_and_the_owner_bucketing(model, symmetry):
    """The meta plane behind a full leaf (the predecessor fingerprint travels from the pair's source lane), the enumerator
    (kmc_successors: trace replay) and SHARDED (three logical shards, traces across them)."""
    N, L, R, E = 3, 2, 2, 2
    inv = ("TypeOk", "StrongIsr")
    o = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=inv))
    assert o.verdict == "invariant"
    base = dict(model=model, n_replicas=N, log_size=L, max_records=R, max_leader_epoch=E, symmetry=symmetry,
                table_capacity=1 << 22, frontier_capacity=1 << 20)
    with ModelChecker(CheckerConfig(**base, invariants=inv, keep_trace=True)) as mc:
        r = mc.run()
        assert (r.verdict, r.violated_invariant, r.violation_depth) == ("invariant", o.viol_inv, o.viol_depth)
        trace = mc.trace()
        names = mc.action_names()
        witness = mc.unpack(mc.witness())
        # the enumerator on a deep state: TLC's enumeration of Next, binding by binding
        got = sorted((k, mc.unpack(t)) for (t, _fp, k) in mc.successors(mc.pack(trace[-2][1])))
    assert got == sorted(kmo.successors(o.cfg, trace[-2][1], o.sb))
    assert len(trace) == r.violation_depth and trace[0] == (None, o.state(0)) and trace[-1][1] == witness
    assert not kmo.check_invariant(o.cfg, INV_INDEX[o.viol_inv], witness)
    for (_, prev), (act, cur) in zip(trace, trace[1:]):
        assert (names.index(act), cur) in kmo.successors(o.cfg, prev, o.sb)   # each step is a Next step of that action
    of = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=(), threads=4))
    rs = check_loopback(CheckerConfig(**base, invariants=(), send_capacity=1 << 18), 3)
    assert (rs.distinct, rs.generated, rs.levels) == (of.distinct, of.generated, of.levels)
    rt = check_loopback(CheckerConfig(**base, invariants=inv, keep_trace=True, send_capacity=1 << 18), 2)
    assert (rt.verdict, rt.violated_invariant, rt.violation_depth) == ("invariant", o.viol_inv, o.viol_depth)
    assert len(rt.trace) == rt.violation_depth and rt.trace[0][1] == o.state(0)
    for (_, prev), (act, cur) in zip(rt.trace, rt.trace[1:]):
        assert (names.index(act), cur) in kmo.successors(o.cfg, prev, o.sb)
