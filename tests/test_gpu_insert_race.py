""""on GPU the only race is the CAS insert — test by inserting a known multiset from many waves and checking the
exact count".  k_insert (the receive side of the exchange, and Init) is handed a shuffled multiset of records — D distinct
synthetic states, each repeated 1 to 40 times, millions of records, every wave of a resident grid claiming at once — and the
seen-set must end up with exactly D new entries, each stored exactly once in the next frontier, whatever the interleaving:
narrow and 128-bit entries, with and without predecessor links."""
import ctypes as C

import numpy as np
import pytest

from kafka_specification_amd import CheckerConfig, ModelChecker
from kafka_specification_amd import _native as nat

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("wide,trace", [(False, False), (True, False), (False, True)], ids=["narrow", "fp128", "narrow+pred"])
def test_a_known_multiset_through_k_insert_from_every_wave_is_claimed_exactly_once(wide, trace):
    import torch
    cfg = CheckerConfig(model="Kip320", n_replicas=3, log_size=2, max_records=2, max_leader_epoch=1, invariants=(),
                        table_capacity=1 << 22, frontier_capacity=1 << 21, wide_fingerprint=wide, keep_trace=trace)
    rng = np.random.default_rng(20260930)
    D = 700_000
    with ModelChecker(cfg) as mc:
        lib, h, W = nat.lib(), mc._h, mc.state_words
        rw = W + (1 if trace else 0)                      # a record: the state's words [+ the predecessor's fingerprint]
        # synthetic "states": distinct by construction (an index in word 0, above every real state's bits) and never equal to a
        # reachable one — k_insert fingerprints and claims whatever words it is given
        states = rng.integers(0, 1 << 62, size=(D, W), dtype=np.uint64)
        states[:, 0] = (np.arange(D, dtype=np.uint64) << np.uint64(20)) | np.uint64(1 << 63)
        reps = rng.integers(1, 41, size=D)
        idx = np.repeat(np.arange(D), reps)
        rng.shuffle(idx)
        recs = np.zeros((len(idx), rw), dtype=np.uint64)
        recs[:, :W] = states[idx]
        if trace:
            recs[:, W] = 0x1234
        dev = torch.from_numpy(recs.view(np.int64)).to("cuda:0")
        nat.check(lib.kmc_step_begin(h))
        nat.check(lib.kmc_step_expand(h, None))           # level 2 of the real search: 6 states
        torch.cuda.synchronize()
        # several launches of several million records each: blocks x waves of every CU probe and claim the same slots at once
        # (the whole multiset twice: the second pass finds every state present)
        n, chunk = len(idx), 3_000_000
        for _pass in range(2):
            for at in range(0, n, chunk):
                m = min(chunk, n - at)
                nat.check(lib.kmc_step_insert(h, C.c_void_p(dev.data_ptr() + at * rw * 8), m))
        info = nat.KmcLevelInfo()
        nat.check(lib.kmc_step_finish(h, C.byref(info)))  # (the conservation law holds the books: records handed in = probed)
        assert info.error_flags == 0
        assert int(info.new_states) == 6 + D, (int(info.new_states), D, n)
        got = mc.frontier_states()
        assert got.shape == (6 + D, W)
        synthetic = got[(got[:, 0] >> np.uint64(63)) == 1]
        assert len(synthetic) == D
        order = np.argsort(synthetic[:, 0])
        assert np.array_equal(synthetic[order], states)   # every distinct record exactly once, no other
