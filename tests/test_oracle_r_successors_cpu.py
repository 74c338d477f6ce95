"""The reference's text, evaluated state by state at the constants the bench binds (tests/golden/oracle_r_successors_*.npz:
Oracle-R's Next — per disjunct, with multiplicity — and the four invariants on ~20,000 deep states per binding: the five
Kafka modules at 3 brokers / LogSize 6 / MaxRecords 6 / MaxLeaderEpoch 2, Kip279 at 5/2/2/1, Kip320 at 7/8/8/3), against

  * the C oracle (oracle/kmc_oracle.c: kmo_successors / kmo_check_invariant), and
  * the DEVICE's model templates compiled for the host (tests/host_emu.cpp: every guard and effect of kmc_device.h).

This is the link the whole-BFS fixtures (logs <= 3 deep) cannot give: LookupOffsetForEpoch over logs of five and six records
with mixed epochs (Kip101.tla:27-47, Kip279.tla:27-51), the truncation cases of Kip320.tla:49-148, WeakIsr / StrongIsr at
hw >= 3 (KafkaReplication.tla:320-340) — pinned to the reference's text, not to a hand lowering.  The GPU's kmc_successors /
kmc_check_states are held to the same files by tests/test_gpu_oracle_r_successors.py."""
import os

import pytest

import host_emu
import kmo
import oracle_r_successors as ors
from kafka_specification_amd import CheckerConfig, ModelChecker

ENTRIES = ors.entries()
KAFKA_ENTRIES = [e for e in ENTRIES if ors.is_kafka(e)]


def test_the_fixtures_exist_for_every_binding_the_bench_and_baseline_name():
    have = {(m["module"], m["N"], m["L"], m["R"], m["E"]) for _, m in ENTRIES}
    for mod in ("KafkaTruncateToHighWatermark", "Kip101", "Kip279", "Kip320", "Kip320FirstTry"):
        assert (mod, 3, 6, 6, 2) in have
    assert ("Kip279", 5, 2, 2, 1) in have and ("Kip320", 7, 8, 8, 3) in have
    assert ("Kip279", 5, 4, 4, 3) in have         # config 4 at SURVEY 8(a.0)'s sizing: three epochs in a log at five brokers
    assert ("MCAsyncIsr", 4, 3, 0, 4) in have     # models/MCAsyncIsr.cfg: AsyncIsr.tla under the state constraint


@pytest.mark.parametrize("entry", KAFKA_ENTRIES, ids=ors.ids)
def test_the_sample_reaches_what_small_exhaustive_runs_cannot(entry):
    """Recomputed from the state bytes, not read from the index: deep logs with mixed epochs under a high watermark >= 3."""
    fn, m = entry
    fx = ors.load(fn)
    N, L, E = m["N"], m["L"], m["E"]
    n = len(fx["states"])
    assert n >= 20000 and len({bytes(s) for s in fx["states"]}) == n
    assert int((fx["source"] == 1).sum()) >= 5000 and int((fx["source"] == 2).sum()) >= 5000   # Oracle-R's own walks, the C oracle's
    feats = [ors.features(bytes(s), N, L, E) for s in fx["states"]]
    if L >= 5:
        assert sum(f[3] >= 5 and f[2] >= 3 for f in feats) >= 2000   # a log >= 5 deep with >= 2 epochs, some hw >= 3
        assert sum(f[0] == L for f in feats) >= 1000                  # full logs
    assert sum(f[1] >= 2 for f in feats) >= (2000 if L >= 5 else 500)   # (LogSize 2, two epochs: fewer ways to mix them)
    assert int((fx["nsucc"] == 0).sum()) >= 1                         # terminal states
    assert all(int(fx["per_action"][:, k].sum()) > 0 for k in range(len(m["actions"])))   # every disjunct of Next fires
    assert m["coverage"]["states_with_a_twice_generated_successor"] == 0 or m["module"] in ("Kip279", "Kip320", "Kip320FirstTry")


@pytest.mark.parametrize("entry", ENTRIES, ids=ors.ids)
def test_c_oracle_equals_the_executed_reference_state_by_state(entry):
    fn, m = entry
    fx = ors.load(fn)
    model, _consts, oconsts = ors.engine_model(m)
    ocfg = kmo.make_config(model, **oconsts, invariants=())
    sb = fx["states"].shape[1]
    for i in range(len(fx["states"])):
        s = bytes(fx["states"][i])
        inv = sum((0 if kmo.check_invariant(ocfg, k, s) else 1) << k for k in range(len(m["invariants"])))
        ors.compare(m, fx, i, kmo.successors(ocfg, s, sb), inv, "C oracle")


@pytest.mark.parametrize("entry", ENTRIES, ids=ors.ids)
def test_device_model_templates_equal_the_executed_reference_state_by_state(entry):
    fn, m = entry
    fx = ors.load(fn)
    model, consts, _ = ors.engine_model(m)
    cfg6 = (kmo.MODELS[model], m["N"], m["L"], m["R"], m["E"], 0)
    assert host_emu.lib().emu_words(*cfg6) > 0, "tests/host_emu.cpp does not instantiate this binding"
    step = int(os.environ.get("KMC_SUCC_FIXTURE_STRIDE", "1"))
    mask = (1 << len(m["invariants"])) - 1
    with ModelChecker(CheckerConfig(model=model, device=-1, **consts)) as mc:   # host-only handle: pack / unpack
        for i in range(0, len(fx["states"]), step):
            w = mc.pack(bytes(fx["states"][i]))
            recs = [(k, mc.unpack(t)) for (k, t) in host_emu.successors(cfg6, w)]
            ors.compare(m, fx, i, recs, host_emu.violated(cfg6, w, mask), "device templates on the host")


# ---- the invariants on ARBITRARY states (tests/golden/oracle_r_mutants_*.npz, make_oracle_r_mutants.py) -------------------------
MUTANTS = ors.mutant_entries()


def test_mutant_fixtures_make_every_invariant_fail_somewhere():
    assert len(MUTANTS) >= 7
    for fn, m in MUTANTS:
        assert m["states"] >= 3000 and all(v >= 100 for v in m["violating"]), m    # TypeOk, WeakIsr, StrongIsr, LeaderInIsr
        inv = ors.load(fn)["inv"]
        typeok = (inv & 1) == 0       # ... and WeakIsr / StrongIsr fail on states that DO satisfy TypeOk (where they are compared)
        assert int(((inv & 2) != 0)[typeok].sum()) >= 50 and int(((inv & 4) != 0)[typeok].sum()) >= 50


@pytest.mark.parametrize("entry", MUTANTS, ids=ors.ids)
def test_invariants_on_arbitrary_states_equal_the_executed_reference(entry):
    """TypeOk is true on every reachable state (and WeakIsr / StrongIsr on every reachable state of Kip320): only unreachable
    states can tell a lowered invariant from `return true`.  On 3,000 mutated deep states per binding — fields overwritten with
    values in or just outside their ranges — the C oracle's and the device templates' four predicates equal Oracle-R's, i.e.
    the reference's own KafkaReplication.tla:101,320,334,345 evaluated on that state (WeakIsr / StrongIsr where TypeOk holds:
    oracle_r_successors.comparable_invariants says why)."""
    fn, m = entry
    fx = ors.load(fn)
    ocfg = kmo.make_config(m["module"], N=m["N"], L=m["L"], R=m["R"], E=m["E"], invariants=())
    cfg6 = (kmo.MODELS[m["module"]], m["N"], m["L"], m["R"], m["E"], 0)
    consts = dict(n_replicas=m["N"], log_size=m["L"], max_records=m["R"], max_leader_epoch=m["E"])
    with ModelChecker(CheckerConfig(model=m["module"], device=-1, **consts)) as mc:
        for i in range(len(fx["states"])):
            s = bytes(fx["states"][i])
            keep = ors.comparable_invariants(int(fx["inv"][i]), int(fx["undefined"][i]))
            want = int(fx["inv"][i]) & keep
            got_c = sum((0 if kmo.check_invariant(ocfg, k, s) else 1) << k for k in range(4)) & keep
            got_d = host_emu.violated(cfg6, mc.pack(s), 15) & keep
            assert got_c == want, f"C oracle: {s.hex()} violates {got_c:04b}, the reference's text says {want:04b}"
            assert got_d == want, f"device templates: {s.hex()} violates {got_d:04b}, the reference's text says {want:04b}"


def test_config4_deep_sample_has_three_epochs_in_a_log_at_five_brokers():
    """VERDICT r5, missing 5: at Kip279 5/2/2/1 a log is at most two deep and holds two epochs — FirstNonMatchingOffsetFromTail
    (Kip279.tla:39-45) never sees three.  At 5/4/4/3 the per-state file holds >= 10,000 states with a log >= 3 deep holding >= 3
    distinct record epochs (recomputed from the state bytes)."""
    e = [x for x in ENTRIES if (x[1]["module"], x[1]["N"], x[1]["L"], x[1]["R"], x[1]["E"]) == ("Kip279", 5, 4, 4, 3)]
    assert e, "tests/golden/oracle_r_successors_kip279_5_4_4_3.npz is missing"
    fx = ors.load(e[0][0])
    N, L, E = 5, 4, 3
    n3 = 0
    for s in fx["states"]:
        b = bytes(s)
        for r in range(N):
            o = r * (5 + L)
            end = b[o]
            if end >= 3 and len({(b[o + 5 + k] - 1) % (E + 1) for k in range(min(end, L)) if b[o + 5 + k]}) >= 3:
                n3 += 1
                break
    assert n3 >= 10000, n3
    # ... and the truncating disjunct fires on thousands of them
    k = e[0][1]["actions"].index("BecomeFollowerTruncateKip279") if "BecomeFollowerTruncateKip279" in e[0][1]["actions"] else None
    assert k is not None and int(fx["per_action"][:, k].sum()) > 10000
