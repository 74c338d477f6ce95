"""The two walks of k_expand's pass 2 (csrc/kmc_device.h) and the layouts of the state vector behind them
(csrc/kmc_layout.h), each forced onto configurations that get another one by default (KMC_LAYOUT, read when a handle is
opened):
  * the KIND-MAJOR walk (KmcKafka::apply<K>, the binding chosen per lane at run time) on the ONE-REPLICA-PER-WORD layout —
    the automatic choice at the headline's constants — forced (rm) onto small configurations whose every level the oracle
    holds as an exact set; on the GROUPED layout — what every other Kafka configuration gets, so the rest of the -m gpu
    suite runs it — forced (rmg) onto a large golden count;
  * the INSTANCE-MAJOR walk (tight layout, inst<I>; nobody's default any more) forced (tight) onto small configurations
    with exact level sets and onto two large golden counts."""
import json
import os

import pytest

import kmo
from kafka_specification_amd import CheckerConfig, ModelChecker
from kafka_specification_amd.sharded import check_loopback
from kafka_specification_amd.configs import GROUPED_LARGE, INSTANCE_MAJOR_LARGE, INSTANCE_MAJOR_SMALL, KAFKA, KIND_MAJOR_SMALL

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture
def layout(monkeypatch):
    def set_layout(mode):
        monkeypatch.setenv("KMC_LAYOUT", mode)
    return set_layout


def run(model, N, L, R, E, invariants=("TypeOk",), keep_levels=False, **kw):
    cfg = CheckerConfig(model=model, invariants=invariants, n_replicas=N, log_size=L, max_records=R, max_leader_epoch=E,
                        table_capacity=kw.pop("table_capacity", 1 << 22), frontier_capacity=kw.pop("frontier_capacity", 1 << 20), **kw)
    level_sets = []
    with ModelChecker(cfg) as mc:
        if keep_levels:
            res = mc.run(progress=lambda info: level_sets.append({mc.unpack(row) for row in mc.frontier_states()}))
        else:
            res = mc.run()
        trace = mc.trace() if kw.get("keep_trace") and res.verdict == "invariant" else None
    return res, level_sets, trace


@pytest.mark.parametrize("model,N,L,R,E,mode", [t + ("rm",) for t in KIND_MAJOR_SMALL] + [t + ("tight",) for t in INSTANCE_MAJOR_SMALL])
def test_forced_walk_levels_are_the_oracles_sets(layout, model, N, L, R, E, mode):
    layout(mode)
    inv = ("TypeOk", "WeakIsr", "StrongIsr")
    o = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=(), threads=4))
    res, level_sets, _ = run(model, N, L, R, E, invariants=(), keep_levels=True, table_capacity=1 << 24, frontier_capacity=1 << 22)
    assert (res.verdict, res.distinct, res.generated, res.depth, res.levels) == (o.verdict, o.distinct, o.generated, o.depth, o.levels)
    assert list(res.action_generated.values()) == o.action_generated[:len(res.action_generated)]
    assert res.deadlock_states == o.deadlock_states
    for k in range(len(o.levels)):
        assert level_sets[k] == o.level_states(k), f"level {k} state sets differ"
    ov = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=inv, threads=4))
    rv, _, _ = run(model, N, L, R, E, invariants=inv, table_capacity=1 << 24, frontier_capacity=1 << 22)
    assert (rv.verdict, rv.violated_invariant) == (ov.verdict, ov.viol_inv)
    if ov.viol_inv:
        assert (rv.violation_depth, rv.violation_count) == (ov.viol_depth, ov.viol_count)


def test_kind_major_walk_is_what_the_headline_constants_get(layout):
    """No override: 3 brokers with LogSize 6 is one replica per word — forcing `rm` changes nothing there — while `tight`
    and `rmg` pack the same fields differently; all are 3 words."""
    consts = dict(model="Kip320", device=-1, n_replicas=3, log_size=6, max_records=6, max_leader_epoch=2)
    s = kmo.Run(kmo.make_config("Kip320", N=3, L=6, R=6, E=2, invariants=(), max_states=2000, threads=1)).state(1500)
    packed = {}
    for mode in ("auto", "rm", "tight", "rmg"):
        layout(mode)
        with ModelChecker(CheckerConfig(**consts)) as mc:
            assert mc.state_words == 3 and mc.unpack(mc.pack(s)) == s
            packed[mode] = tuple(mc.pack(s))
    assert packed["auto"] == packed["rm"] and len({packed["rm"], packed["tight"], packed["rmg"]}) == 3


@pytest.mark.parametrize("model", ["Kip279", "KafkaTruncateToHighWatermark"])
def test_kind_major_walk_traces_and_sharding(layout, model):
    """The meta plane of the ring (predecessor fingerprints) and the SHARDED sink behind the kind-major walk."""
    layout("rm")
    N, L, R, E = 3, 2, 2, 2
    inv = ("TypeOk", "StrongIsr")
    o = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=inv))
    assert o.verdict == "invariant"
    cfg = CheckerConfig(model=model, n_replicas=N, log_size=L, max_records=R, max_leader_epoch=E, invariants=inv,
                        keep_trace=True, table_capacity=1 << 22, frontier_capacity=1 << 20)
    with ModelChecker(cfg) as mc:
        r = mc.run()
        assert (r.verdict, r.violated_invariant, r.violation_depth) == ("invariant", o.viol_inv, o.viol_depth)
        trace = mc.trace()
        names = mc.action_names()
        witness = mc.unpack(mc.witness())
    assert len(trace) == r.violation_depth and trace[0] == (None, o.state(0)) and trace[-1][1] == witness
    for (_, prev), (act, cur) in zip(trace, trace[1:]):
        assert (names.index(act), cur) in kmo.successors(o.cfg, prev, o.sb)   # each step is a Next step of that action
    of = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=(), threads=4))
    rs = check_loopback(CheckerConfig(model=model, n_replicas=N, log_size=L, max_records=R, max_leader_epoch=E, invariants=(),
                                      table_capacity=1 << 22, frontier_capacity=1 << 20, send_capacity=1 << 18), 3)
    assert (rs.distinct, rs.generated, rs.levels) == (of.distinct, of.generated, of.levels)


@pytest.mark.parametrize("model,N,L,R,E,mode", [t + ("tight",) for t in INSTANCE_MAJOR_LARGE] + [t + ("rmg",) for t in GROUPED_LARGE])
def test_forced_walk_on_a_large_golden_count(layout, model, N, L, R, E, mode):
    layout(mode)
    name = {"Kip320": "oracle_kip320_3_5_5_2.json", "Kip279": "oracle_kip279_3_5_5_2.json"}[model]
    g = json.load(open(os.path.join(HERE, "golden", name)))
    inv = ("TypeOk", "WeakIsr", "StrongIsr") if model == "Kip320" else ("TypeOk",)
    res, _, _ = run(model, N, L, R, E, invariants=inv, table_capacity=1 << 29, frontier_capacity=1 << 26)
    assert (res.distinct, res.generated, res.depth) == (g["distinct"], g["generated"], g["depth"])
    assert res.levels == g["levels"]


def test_a_checkpoint_is_refused_under_another_layout(layout, tmp_path):
    """The same constants pack into the same number of words in more than one way: the checkpoint header records the
    arrangement, and a handle opened under another one refuses the file instead of reading garbage states."""
    from kafka_specification_amd import KmcError
    base = dict(model="Kip320", n_replicas=3, log_size=2, max_records=2, max_leader_epoch=2, invariants=("TypeOk",),
                table_capacity=1 << 22, frontier_capacity=1 << 20)
    path = str(tmp_path / "k.ckpt")
    o = kmo.Run(kmo.make_config("Kip320", N=3, L=2, R=2, E=2, invariants=("TypeOk",)))
    layout("auto")      # grouped replica-major at these constants: 2 words, like the tight packing
    with ModelChecker(CheckerConfig(**base, max_levels=6)) as mc:
        assert mc.run().verdict == "level_limit"
        mc.save_checkpoint(path)
        words = mc.state_words
    layout("tight")
    with ModelChecker(CheckerConfig(**base)) as mc:
        assert mc.state_words == words == 2
        with pytest.raises(KmcError, match="state layout"):
            mc.load_checkpoint(path)
    layout("auto")
    with ModelChecker(CheckerConfig(**base)) as mc:
        mc.load_checkpoint(path)
        r = mc.resume()
    assert (r.verdict, r.distinct, r.generated, r.levels) == (o.verdict, o.distinct, o.generated, o.levels)
