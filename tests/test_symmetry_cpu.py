"""Symmetry reduction with orbit counting (kmc_config.symmetry), without a GPU.

The permutations of Replicas are applied here to the oracle's CANONICAL-BYTE states by a few lines of Python that share
nothing with the C++ (`permute_bytes`); against that stand the two C++ forms the product uses — KmcSymm<M>::canon, the
compile-time permutations inside the KMC_SYMM kernels (compiled for the host by tests/host_emu.cpp), and
kmc_canonical_state_generic, the run-time-layout form of the host engine (kmc_layout.h).  Then the whole orbit-counting
search is replayed on the CPU with the device's own successor function and representative function, and its WEIGHTED
counts must be the oracle's plain counts: distinct, generated, per disjunct, per level."""
import itertools
from math import factorial

import pytest

import host_emu
import kmo
from kafka_specification_amd import CheckerConfig, ModelChecker

MODEL_NAMES = {v: k for k, v in kmo.MODELS.items()}
SYMMETRIC = [c for c in host_emu.configs() if (c[0] == 1 or 2 <= c[0] <= 6) and c[1] <= 6]


def _ids(c):
    return f"{MODEL_NAMES[c[0]]}-{c[1]}-{c[2]}-{c[3]}-{c[4]}-{c[5]}" + ("", "-tight", "-rm", "-rmg")[c[6]]


def permute_bytes(model, N, L, E, b, img):
    """The state `b` (canonical bytes, include/kmc.h) with replica r renamed img[r]."""
    def mask(m):
        return sum(1 << img[i] for i in range(N) if m >> i & 1)

    def ldr(v):
        return 0 if v == 0 else img[v - 1] + 1

    out = bytearray(len(b))
    if model == 1:   # FiniteReplicatedLog: per replica [endOffset, record x LogSize]
        blk = 1 + L
        for r in range(N):
            out[img[r] * blk:(img[r] + 1) * blk] = b[r * blk:(r + 1) * blk]
        return bytes(out)
    blk = 5 + L
    for r in range(N):
        src = bytearray(b[r * blk:(r + 1) * blk])
        src[3] = ldr(src[3])
        src[4] = mask(src[4])
        out[img[r] * blk:(img[r] + 1) * blk] = src
    g = N * blk
    out[g:g + 3] = b[g:g + 3]
    out[g + 3] = ldr(b[g + 3])
    out[g + 4] = mask(b[g + 4])
    for e in range(E + 1):
        out[g + 5 + 2 * e] = ldr(b[g + 5 + 2 * e])
        out[g + 6 + 2 * e] = mask(b[g + 6 + 2 * e])
    return bytes(out)


def _consts(cfg6):
    model, N, L, R, E, K = cfg6[:6]
    return dict(n_replicas=N, log_size=L, max_records=max(R, 1), max_leader_epoch=E, n_log_records=max(K, 1))


@pytest.mark.parametrize("cfg6", SYMMETRIC, ids=_ids)
def test_representative_of_an_orbit(cfg6):
    """Per sampled reachable state: both C++ forms return the smallest packed image over all permutations, the same for
    every member of the orbit, and the order of the stabiliser — all against permutations done in Python on the bytes."""
    model, N, L, R, E, K = cfg6[:6]
    name = MODEL_NAMES[model]
    perms = list(itertools.permutations(range(N)))
    with host_emu.layout(cfg6):
        ocfg = kmo.make_config(name, N=N, L=L, R=max(R, 1), E=E, K=max(K, 1), invariants=(), max_states=20000, threads=2)
        o = kmo.Run(ocfg)
        n = min(o.distinct, 20000)
        nontrivial = 0
        with ModelChecker(CheckerConfig(model=name, device=-1, **_consts(cfg6))) as mc:
            for idx in range(0, n, max(1, n // 150)):
                s = o.state(idx)
                images = [permute_bytes(model, N, L, E, s, img) for img in perms]
                packed = [tuple(mc.pack(t)) for t in images]
                want = min(packed)                       # words compared in order, as unsigned 64-bit values
                stab = sum(1 for t in images if t == s)
                nontrivial += stab > 1
                for w in set(packed):
                    assert host_emu.canon(cfg6, w) == (stab, want), f"state {idx}: KmcSymm::canon"
                    assert host_emu.canon(cfg6, w, generic=True) == (stab, want), f"state {idx}: generic form"
                # the host library's own entry point (what kmc_contains canonicalises with)
                assert mc.canonical(packed[-1]) == (stab, want)
        assert nontrivial > 0   # Init at least is fixed by every permutation


@pytest.mark.parametrize("cfg6", [c for c in SYMMETRIC if c[6] == 0 and
                                  (c[:6] in {(2, 3, 2, 2, 1, 0), (3, 3, 2, 2, 1, 0), (5, 3, 2, 2, 1, 0), (6, 3, 2, 2, 1, 0),
                                             (4, 3, 2, 3, 1, 0), (5, 4, 1, 1, 1, 0), (4, 2, 2, 2, 2, 0), (1, 2, 4, 0, 0, 2),
                                             (1, 3, 2, 0, 0, 2)})], ids=_ids)
def test_orbit_counting_search_reproduces_the_plain_counts(cfg6):
    """Breadth-first search over orbit representatives with the device's successor and representative functions; every
    count weighted by N!/|stabiliser| of the state it belongs to.  Must equal the oracle's plain exhaustive search."""
    model, N, L, R, E, K = cfg6[:6]
    name = MODEL_NAMES[model]
    nf = factorial(N)
    o = kmo.Run(kmo.make_config(name, N=N, L=L, R=max(R, 1), E=E, K=max(K, 1), invariants=(), threads=2))
    with host_emu.layout(cfg6):
        st0, init = host_emu.canon(cfg6, host_emu.init(cfg6))
        assert st0 == nf and list(init) == host_emu.init(cfg6)    # Init is its own orbit
        seen = {init: st0}
        frontier = [init]
        levels, generated, per_kind, distinct = [], 1, [0] * 16, 0
        while frontier:
            levels.append(sum(nf // seen[s] for s in frontier))
            distinct += levels[-1]
            nxt = []
            for s in frontier:
                w = nf // seen[s]
                for kind, t in host_emu.successors(cfg6, s):
                    generated += w
                    per_kind[kind] += w
                    st, c = host_emu.canon(cfg6, t)
                    if c not in seen:
                        seen[c] = st
                        nxt.append(c)
            frontier = nxt
    assert (distinct, generated, len(levels)) == (o.distinct, o.generated, o.depth)
    assert levels == o.levels
    assert per_kind == o.action_generated[:16]
    assert len(seen) < o.distinct / (nf / 2) or N == 2 or o.distinct < 2000   # and it did reduce the search
