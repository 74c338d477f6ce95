"""The plan of one BFS level's exchange (kmc_exchange_plan, include/kmc.h) is a pure function of the count matrix;
both transports under the ABI — grouped ncclSend/ncclRecv over RCCL, and device-to-device copies between logical
shards on one GPU — execute it.  No GPU is needed to check that the plans of all ranks fit together: RCCL matches
the messages of a pair in posting order, so rank s's sends to d must equal, message by message, rank d's receives
from s; receive runs must tile the receive area densely; long runs must be cut identically on both sides."""
import ctypes as C
import random

import pytest

from kafka_specification_amd import _native as nat

SUBS = nat.KMC_SEND_SUBS
MAX_WORDS = 1 << 27          # KMC_XFER_MAX_WORDS: one message stays below 1 GiB


def plan(counts, P, me, send_cap, rec_words, cap=4096):
    lib = nat.lib()
    arr = (C.c_uint64 * len(counts))(*counts)
    sends, recvs = (C.c_uint64 * (3 * cap))(), (C.c_uint64 * (3 * cap))()
    ns, nr, nrec = C.c_uint64(), C.c_uint64(), C.c_uint64()
    nat.check(lib.kmc_exchange_plan(arr, P, me, send_cap, rec_words, sends, recvs, cap, C.byref(ns), C.byref(nr), C.byref(nrec)))
    assert ns.value <= cap and nr.value <= cap
    tri = lambda a, n: [(int(a[3 * i]), int(a[3 * i + 1]), int(a[3 * i + 2])) for i in range(n)]
    return tri(sends, ns.value), tri(recvs, nr.value), int(nrec.value)


def check_world(counts, P, send_cap, rec_words):
    plans = [plan(counts, P, me, send_cap, rec_words) for me in range(P)]
    for me, (sends, recvs, nrec) in enumerate(plans):
        want = sum(counts[(s * P + me) * SUBS + sb] for s in range(P) if s != me for sb in range(SUBS))
        assert nrec == want
        # receive runs tile [0, nrec * rec_words) densely, in (source, sub-buffer) order
        at = 0
        for peer, off, words in recvs:
            assert peer != me and off == at and 0 < words <= MAX_WORDS
            at += words
        assert at == nrec * rec_words
        # every send run lies inside its (destination, sub-buffer) slot of the send area
        for peer, off, words in sends:
            assert peer != me and 0 < words <= MAX_WORDS
            slot = off // (send_cap * rec_words)
            assert slot // SUBS == peer
            assert off + words <= (slot + 1) * send_cap * rec_words
        # the sends to each destination cover exactly the announced records, sub-buffer by sub-buffer
        for d in range(P):
            if d == me:
                assert not [x for x in sends if x[0] == me]
                continue
            per_sub = [0] * SUBS
            for peer, off, words in sends:
                if peer == d:
                    per_sub[(off // (send_cap * rec_words)) % SUBS] += words
            assert per_sub == [counts[(me * P + d) * SUBS + sb] * rec_words for sb in range(SUBS)]
    # pairwise: s's sends to d, in posting order, are d's receives from s, in posting order
    for s in range(P):
        for d in range(P):
            if s == d:
                continue
            out = [w for (peer, _o, w) in plans[s][0] if peer == d]
            inn = [w for (peer, _o, w) in plans[d][1] if peer == s]
            assert out == inn, (s, d)


@pytest.mark.parametrize("P", [1, 2, 3, 4, 8])
def test_plans_of_all_ranks_fit_together(P):
    rng = random.Random(1000 + P)
    for trial in range(40):
        send_cap = rng.choice([1, 7, 64, 4096])
        rec_words = rng.choice([1, 3, 4, 10])
        counts = [0] * (P * P * SUBS)
        density = rng.choice([0.0, 0.1, 0.5, 1.0])
        for s in range(P):
            for d in range(P):
                for sb in range(SUBS):
                    if s != d and rng.random() < density:
                        counts[(s * P + d) * SUBS + sb] = rng.randint(0, send_cap)
        check_world(counts, P, send_cap, rec_words)


def test_empty_level_and_one_sided_level():
    P = 4
    check_world([0] * (P * P * SUBS), P, 128, 3)            # nothing moves: no message at all
    sends, recvs, nrec = plan([0] * (P * P * SUBS), P, 2, 128, 3)
    assert (sends, recvs, nrec) == ([], [], 0)
    counts = [0] * (P * P * SUBS)                           # BFS level 1: only the owner of Init has anything to send
    for d in range(1, P):
        counts[(0 * P + d) * SUBS + 5] = 3
    check_world(counts, P, 128, 3)
    assert plan(counts, P, 0, 128, 3)[1] == [] and len(plan(counts, P, 0, 128, 3)[0]) == 3
    assert plan(counts, P, 3, 128, 3) == ([], [(0, 0, 9)], 3)


def test_runs_longer_than_one_gib_are_cut_identically():
    P, send_cap, rec_words = 2, 1 << 26, 10                 # a full sub-buffer = 5 GiB
    counts = [0] * (P * P * SUBS)
    counts[(0 * P + 1) * SUBS + 0] = send_cap
    counts[(0 * P + 1) * SUBS + 3] = (1 << 24) + 5
    counts[(1 * P + 0) * SUBS + 7] = 1
    check_world(counts, P, send_cap, rec_words)
    sends, _, _ = plan(counts, P, 0, send_cap, rec_words)
    assert len(sends) == 5 + 2 and max(w for _p, _o, w in sends) == MAX_WORDS


def test_self_traffic_is_never_planned():
    # the diagonal of the count matrix is a shard's own successors: they take the local path inside k_expand
    P = 3
    counts = [5] * (P * P * SUBS)
    for me in range(P):
        sends, recvs, nrec = plan(counts, P, me, 64, 3)
        assert all(p != me for p, _o, _w in sends + recvs) and nrec == 2 * SUBS * 5


@pytest.mark.parametrize("P", [2, 3, 8])
def test_mock_rccl_moves_planned_runs_between_concurrent_ranks(P, tmp_path):
    """tests/mock_rccl.cpp is what stands in for librccl when the exchange under the C ABI runs with several ranks on
    one GPU (tests/test_gpu_native_exchange_threads.py).  Its matching logic is checked here on CPU (plain memory): P
    threads execute kmc_exchange_plan's runs for random count matrices — empty sub-buffers, full ones, a rank with
    nothing to send — and every receive area must hold exactly what the peers addressed to it; a mismatched pair fails."""
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    exe = str(tmp_path / "mock_rccl_selfcheck")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-w", "-o", exe,
                           os.path.join(here, "mock_rccl_selfcheck.cpp"), "-ldl"])
    lib = os.path.join(os.path.dirname(here), "kafka_specification_amd", "libkmc.so")
    p = subprocess.run([exe, lib, str(P), "12"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "selfcheck ok" in p.stdout, (p.stdout, p.stderr[-800:])
