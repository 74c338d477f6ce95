"""The DEFERRED probe of the search's kernel (csrc/kmc_kernels.h: a flush issues the first probe of its batch and the batch is
completed — claims, stager — by the next flush; on by default for states of >= 8 words, i.e. seven brokers with deep logs)
forced onto configurations every level of which the oracle holds as an exact set (-DKMC_DEFER_MIN_WORDS=1, code objects
prebuilt by build()).

What the deferral must not change: which successor wins a slot does not matter, THAT exactly one does, does — two copies of a
state in consecutive batches both see an empty slot and the second one's compare-and-swap must lose; predecessor links are
written by the winner at completion; a full table and a full frontier are still reported; 128-bit entries take the undeferred
path.  By default the path runs at BASELINE config 5 (tests/test_gpu_zzz_oracle_r_wide.py, test_gpu_symmetry.py, the bench's
config5 leg: counts against the exact oracle)."""
import pytest

import kmo
from kafka_specification_amd import CheckerConfig, ModelChecker
from kafka_specification_amd.configs import DEFERRED_PROBE_DEFINES, DEFERRED_PROBE_SMALL, DEFERRED_PROBE_SYMMETRY

pytestmark = pytest.mark.gpu
INV_INDEX = {"TypeOk": 0, "WeakIsr": 1, "StrongIsr": 2, "LeaderInIsr": 3}


@pytest.fixture(autouse=True)
def forced(monkeypatch):
    monkeypatch.setenv("KMC_JIT_DEFINES", DEFERRED_PROBE_DEFINES)


@pytest.mark.parametrize("model,N,L,R,E", DEFERRED_PROBE_SMALL)
def test_every_level_is_the_oracles_set(model, N, L, R, E):
    o = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=(), threads=4))
    cfg = CheckerConfig(model=model, invariants=(), n_replicas=N, log_size=L, max_records=R, max_leader_epoch=E,
                        table_capacity=1 << 24, frontier_capacity=1 << 22)
    level_sets = []
    with ModelChecker(cfg) as mc:
        res = mc.run(progress=lambda info: level_sets.append({mc.unpack(row) for row in mc.frontier_states()}))
        again = mc.run()      # chained launches this time (no progress callback): a batch pending at the end of a launch is completed there
    assert (res.verdict, res.distinct, res.generated, res.depth, res.levels) == (o.verdict, o.distinct, o.generated, o.depth, o.levels)
    assert (again.distinct, again.generated, again.levels) == (o.distinct, o.generated, o.levels)
    assert list(res.action_generated.values()) == o.action_generated[:len(res.action_generated)]
    assert res.deadlock_states == o.deadlock_states
    for k in range(len(o.levels)):
        assert level_sets[k] == o.level_states(k), f"level {k} state sets differ"


@pytest.mark.parametrize("model,N,L,R,E", DEFERRED_PROBE_SYMMETRY)
def test_orbit_counting_reports_the_plain_counts(model, N, L, R, E):
    o = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=("TypeOk",), threads=8))
    with ModelChecker(CheckerConfig(model=model, invariants=("TypeOk",), symmetry=True, n_replicas=N, log_size=L, max_records=R,
                                    max_leader_epoch=E, table_capacity=1 << 23, frontier_capacity=1 << 21)) as mc:
        res = mc.run()
    assert (res.verdict, res.distinct, res.generated, res.depth, res.levels) == (o.verdict, o.distinct, o.generated, o.depth, o.levels)
    assert list(res.action_generated.values()) == o.action_generated[:len(res.action_generated)]
    assert res.deadlock_states == o.deadlock_states and res.orbit_representatives < res.distinct


@pytest.mark.parametrize("model", ["Kip279", "Kip101"])
def test_predecessor_links_are_written_at_completion(model):
    N, L, R, E = 3, 2, 2, 2
    inv = ("TypeOk", "StrongIsr")
    o = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=inv))
    assert o.verdict == "invariant"
    with ModelChecker(CheckerConfig(model=model, n_replicas=N, log_size=L, max_records=R, max_leader_epoch=E, invariants=inv,
                                    keep_trace=True, table_capacity=1 << 22, frontier_capacity=1 << 20)) as mc:
        r = mc.run()
        assert (r.verdict, r.violated_invariant, r.violation_depth) == ("invariant", o.viol_inv, o.viol_depth)
        trace = mc.trace()
        names = mc.action_names()
        witness = mc.unpack(mc.witness())
    assert len(trace) == r.violation_depth and trace[0] == (None, o.state(0)) and trace[-1][1] == witness
    for (_, prev), (act, cur) in zip(trace, trace[1:]):
        assert (names.index(act), cur) in kmo.successors(o.cfg, prev, o.sb)   # each step is a Next step of that action


def test_a_full_table_a_full_frontier_and_wide_entries():
    base = dict(model="Kip320", n_replicas=3, log_size=2, max_records=2, max_leader_epoch=2, invariants=("TypeOk",))
    o = kmo.Run(kmo.make_config("Kip320", N=3, L=2, R=2, E=2, invariants=("TypeOk",)))
    with ModelChecker(CheckerConfig(**base, table_capacity=1 << 12, frontier_capacity=1 << 20)) as mc:
        assert mc.run().verdict == "table_full"
    with ModelChecker(CheckerConfig(**base, table_capacity=1 << 22, frontier_capacity=1 << 10)) as mc:
        assert mc.run().verdict == "frontier_full"
    with ModelChecker(CheckerConfig(**base, wide_fingerprint=True, table_capacity=1 << 22, frontier_capacity=1 << 20)) as mc:
        r = mc.run()      # 128-bit entries: two words to wait for — the undeferred path of the same kernel
    assert (r.verdict, r.distinct, r.generated, r.levels) == (o.verdict, o.distinct, o.generated, o.levels)
