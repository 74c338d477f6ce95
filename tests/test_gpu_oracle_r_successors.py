"""GPU: kmc_successors and kmc_check_states — Next and the invariants as the kernels compute them, through the C ABI — on
every state of tests/golden/oracle_r_successors_*.npz, i.e. against the reference's own text executed state by state at the
headline's constants (3 brokers, LogSize 6, MaxRecords 6, MaxLeaderEpoch 2: all five Kafka modules), at BASELINE config 4's
(Kip279, 5 brokers) and config 5's (Kip320, 7 brokers, LogSize 8).  See tests/test_oracle_r_successors_cpu.py."""
import ctypes as C

import numpy as np
import pytest

import oracle_r_successors as ors
from kafka_specification_amd import CheckerConfig, ModelChecker
from kafka_specification_amd import _native as nat

pytestmark = pytest.mark.gpu
ENTRIES = ors.entries()


def test_fixtures_present():
    assert len(ENTRIES) >= 7


@pytest.mark.parametrize("symmetry", [False, True], ids=["plain", "orbit-counting"])
@pytest.mark.parametrize("entry", ENTRIES, ids=ors.ids)
def test_engine_equals_the_executed_reference_state_by_state(entry, symmetry):
    """(under orbit counting kmc_successors lists the successors themselves — the raw Next relation of the state it is
    given — so the same file applies; a quarter of the states suffices there)"""
    fn, m = entry
    if symmetry and not ors.is_kafka(entry):
        pytest.skip("AsyncIsr singles out the Leader: no orbit counting")
    fx = ors.load(fn)
    model, consts, _ = ors.engine_model(m)
    cfg = CheckerConfig(model=model, **consts, table_capacity=1 << 16, frontier_capacity=1 << 12, symmetry=symmetry)
    mask = (1 << len(m["invariants"])) - 1
    step = 4 if symmetry else 1
    with ModelChecker(cfg) as mc:
        W = mc.state_words
        lib = nat.lib()
        for i in range(0, len(fx["states"]), step):
            w = mc.pack(bytes(fx["states"][i]))
            recs = [(k, mc.unpack(t)) for (t, _fp, k) in mc.successors(w)]
            wa = (C.c_uint64 * W)(*w)
            bits = C.c_uint32()
            nat.check(lib.kmc_check_states(mc._h, wa, 1, mask, C.byref(bits)))
            ors.compare(m, fx, i, recs, int(bits.value), "HIP engine" + (" (orbit counting)" if symmetry else ""))


MUTANTS = ors.mutant_entries()


@pytest.mark.parametrize("entry", MUTANTS, ids=ors.ids)
def test_engine_invariants_on_arbitrary_states_equal_the_executed_reference(entry):
    """kmc_check_states — the predicate k_expand applies to the states it expands — on tests/golden/oracle_r_mutants_*.npz:
    deep states with fields overwritten by values in or just outside their ranges, where every one of the four invariants
    fails hundreds of times, judged by the reference's own text (Oracle-R).  WeakIsr / StrongIsr are compared where TypeOk
    holds (oracle_r_successors.comparable_invariants)."""
    fn, m = entry
    fx = ors.load(fn)
    cfg = CheckerConfig(model=m["module"], n_replicas=m["N"], log_size=m["L"], max_records=m["R"], max_leader_epoch=m["E"],
                        table_capacity=1 << 16, frontier_capacity=1 << 12)
    n = len(fx["states"])
    with ModelChecker(cfg) as mc:
        W = mc.state_words
        words = (C.c_uint64 * (n * W))()
        for i in range(n):
            words[i * W:(i + 1) * W] = mc.pack(bytes(fx["states"][i]))
        bits = (C.c_uint32 * n)()
        nat.check(nat.lib().kmc_check_states(mc._h, words, n, 15, bits))     # one call, n single-state passes
    for i in range(n):
        keep = ors.comparable_invariants(int(fx["inv"][i]), int(fx["undefined"][i]))
        assert int(bits[i]) & keep == int(fx["inv"][i]) & keep, (
            f"HIP engine: {bytes(fx['states'][i]).hex()} violates {int(bits[i]):04b}, the reference's text {int(fx['inv'][i]):04b}")
