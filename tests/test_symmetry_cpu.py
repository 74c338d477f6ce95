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
SYMMETRIC = [c for c in host_emu.configs() if (c[0] == 1 or 2 <= c[0] <= 6) and c[1] <= 7]


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


def replica_keys_ascend(model, N, L, E, b):
    """Six replicas have 720 images, so beyond four the representative is the smallest image AMONG THOSE WHOSE REPLICA KEYS
    ASCEND with the position (KmcSymm::canon_sorted).  The key, restated on the canonical bytes: the log (highest slot most
    significant — a record's byte orders as its packed code does), then, most significant first: per LeaderAndIsr request
    from the last epoch down (its ISR holds me, it names me), how many others name me, how many others hold me, the size of
    my ISR, quorumState's ISR holds me, quorumState names me, I name nobody, my ISR holds me, I name myself, epoch, hw, end."""
    if model == 1:
        blk = 1 + L
        keys = [(tuple(reversed(b[r * blk + 1:(r + 1) * blk])), b[r * blk]) for r in range(N)]
        return all(keys[i] <= keys[i + 1] for i in range(N - 1))
    blk, g = 5 + L, N * (5 + L)
    keys = []
    for r in range(N):
        x = b[r * blk:(r + 1) * blk]
        k = [tuple(reversed(x[5:]))]
        for e in range(E, -1, -1):
            k += [b[g + 6 + 2 * e] >> r & 1, b[g + 5 + 2 * e] == r + 1]
        k += [sum(1 for o in range(N) if o != r and b[o * blk + 3] == r + 1),
              sum(1 for o in range(N) if o != r and b[o * blk + 4] >> r & 1),
              bin(x[4]).count("1"), b[g + 4] >> r & 1, b[g + 3] == r + 1, x[3] == 0, x[4] >> r & 1, x[3] == r + 1,
              x[2], x[1], x[0]]
        keys.append(tuple(k))
    return all(keys[i] <= keys[i + 1] for i in range(N - 1))


def constructed_state(rnd, model, N, L, R, E, told_apart_tie):
    """A random type-correct Kafka state (canonical bytes) with few distinct (end, hw, epoch, log) combinations — ties between
    replica keys are the point.  told_apart_tie: replicas 0 and 1 alike in every field and every count the key sees, but 0
    sits in the ISR of replica 2 and 1 in that of replica 3, and those two differ (epoch): exchanging 0 and 1 is no
    automorphism.  Then a random renaming."""
    blk, g = 5 + L, N * (5 + L)
    b = bytearray(g + 5 + 2 * (E + 1))
    proto = [rnd.randrange(0, 2) for _ in range(4)]
    for r in range(N):
        end = proto[rnd.randrange(4)] % (L + 1)
        b[r * blk:r * blk + 5] = bytes([end, rnd.randrange(0, end + 1), rnd.randrange(0, 2), rnd.randrange(0, N + 1),
                                        rnd.randrange(0, 1 << N)])
        for o in range(end):
            b[r * blk + 5 + o] = 1   # record id 0, epoch 0
    b[g:g + 5] = bytes([rnd.randrange(0, R + 1), rnd.randrange(0, E + 2), rnd.randrange(0, E + 2),
                        rnd.randrange(0, N + 1), rnd.randrange(0, 1 << N)])
    for e in range(E + 1):
        b[g + 5 + 2 * e], b[g + 6 + 2 * e] = rnd.randrange(0, N + 1), rnd.randrange(0, 1 << N)
    if told_apart_tie:
        for r in (0, 1):
            b[r * blk:(r + 1) * blk] = bytes([0, 0, 1, 0, 0] + [0] * L)
        b[2 * blk + 2], b[3 * blk + 2] = 0, 1
        for r in range(2, N):
            b[r * blk + 3] = 0 if b[r * blk + 3] in (1, 2) else b[r * blk + 3]
            b[r * blk + 4] &= ~3
        b[2 * blk + 4] |= 1
        b[3 * blk + 4] |= 2
        for at in [g + 3] + [g + 5 + 2 * e for e in range(E + 1)]:
            b[at] = 0 if b[at] in (1, 2) else b[at]
            b[at + 1] &= ~3
        b = bytearray(permute_bytes(model, N, L, E, bytes(b), rnd.sample(range(N), N)))
    return bytes(b)


UNROLLED_MAX = 3   # KMC_SYMM_UNROLLED_MAX (kmc_layout.h)


def _consts(cfg6):
    model, N, L, R, E, K = cfg6[:6]
    return dict(n_replicas=N, log_size=L, max_records=max(R, 1), max_leader_epoch=E, n_log_records=max(K, 1))


@pytest.mark.parametrize("cfg6", SYMMETRIC, ids=_ids)
def test_representative_of_an_orbit(cfg6):
    """Per sampled reachable state: both C++ forms return the smallest packed image over all permutations (five and six
    replicas: over the permutations that sort the replicas' keys), the same for every member of the orbit, and the order of
    the stabiliser — all against permutations done in Python on the bytes."""
    model, N, L, R, E, K = cfg6[:6]
    name = MODEL_NAMES[model]
    perms = list(itertools.permutations(range(N)))
    with host_emu.layout(cfg6):
        ocfg = kmo.make_config(name, N=N, L=L, R=max(R, 1), E=E, K=max(K, 1), invariants=(), max_states=20000, threads=2)
        o = kmo.Run(ocfg)
        n = min(o.distinct, 20000)
        nontrivial = 0
        with ModelChecker(CheckerConfig(model=name, device=-1, **_consts(cfg6))) as mc:
            # (seven replicas: 5040 images of every sample, and the run-time-layout form walks through all of them per call)
            for idx in range(0, n, max(1, n // (150 if N < 7 else 10))):
                s = o.state(idx)
                images = [permute_bytes(model, N, L, E, s, img) for img in perms]
                packed = [tuple(mc.pack(t)) for t in images]
                # words compared in order, as unsigned 64-bit values; beyond four replicas among the sorted images only
                want = min(p for p, t in zip(packed, images) if N <= UNROLLED_MAX or replica_keys_ascend(model, N, L, E, t))
                stab = sum(1 for t in images if t == s)
                nontrivial += stab > 1
                some = set(packed) if N < 6 else set(packed[::max(1, len(packed) // (40 if N < 7 else 6))]) | {packed[0]}
                for w in some:
                    assert host_emu.canon(cfg6, w) == (stab, want), f"state {idx}: KmcSymm::canon"
                    assert host_emu.canon(cfg6, w, generic=True) == (stab, want), f"state {idx}: generic form"
                # the host library's own entry point (what kmc_contains canonicalises with)
                assert mc.canonical(packed[-1]) == (stab, want)
        assert nontrivial > 0   # Init at least is fixed by every permutation


@pytest.mark.parametrize("cfg6", [c for c in SYMMETRIC if c[6] == 0 and
                                  (c[:6] in {(2, 3, 2, 2, 1, 0), (3, 3, 2, 2, 1, 0), (5, 3, 2, 2, 1, 0), (6, 3, 2, 2, 1, 0),
                                             (4, 3, 2, 3, 1, 0), (5, 4, 1, 1, 1, 0), (4, 2, 2, 2, 2, 0), (1, 2, 4, 0, 0, 2), (4, 5, 1, 1, 1, 0), (5, 7, 1, 1, 0, 0),
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
    print(f"{name} N={N}: {len(seen)} stored states for {o.distinct}")
    assert len(seen) < o.distinct / (nf / 2) or N == 2 or o.distinct < 2000 or N == 7   # and it did reduce the search
    assert N != 7 or len(seen) < o.distinct / 500


@pytest.mark.parametrize("cfg6", [c for c in SYMMETRIC if c[6] == 0 and c[1] > UNROLLED_MAX and c[0] != 1], ids=_ids)
def test_sorted_images_when_the_keys_do_not_tell_replicas_apart(cfg6):
    """canon_sorted's third case.  On reachable states, replicas with equal keys have always turned out to be interchangeable
    (tools/tie_stats.py), so the walk through the images of the sorted one — for ties between replicas that something the
    key does not see tells apart — never runs in the searches.  Random type-correct states do produce such ties (two
    replicas alike in everything the key counts, held by the ISRs of DIFFERENT third replicas): the device form, the
    run-time-layout form and the definition restated on the bytes must agree on them too."""
    import random
    model, N, L, R, E, K = cfg6[:6]
    name = MODEL_NAMES[model]
    rnd = random.Random(1234 + N * 100 + L)
    perms = list(itertools.permutations(range(N)))
    blk, g = 5 + L, N * (5 + L)
    told_apart = 0
    with host_emu.layout(cfg6), ModelChecker(CheckerConfig(model=name, device=-1, **_consts(cfg6))) as mc:
        samples = 120 if N < 7 else 24   # (5040 images of every sample, permuted in Python)
        for sample in range(samples):
            s = constructed_state(rnd, model, N, L, R, E, told_apart_tie=sample % 2 == 1)
            assert mc.unpack(mc.pack(s)) == s
            images = [permute_bytes(model, N, L, E, s, img) for img in perms]
            sorted_images = {t for t in images if replica_keys_ascend(model, N, L, E, t)}
            told_apart += len(sorted_images) > 1
            want = min(tuple(mc.pack(t)) for t in sorted_images)
            stab = sum(1 for t in images if t == s)
            for t in rnd.sample(images, 6) + [s]:
                w = tuple(mc.pack(t))
                assert host_emu.canon(cfg6, w) == (stab, want)
                assert host_emu.canon(cfg6, w, generic=True) == (stab, want)
    assert told_apart >= samples // 3, told_apart


def test_orbit_counting_prefix_at_baseline_config5_constants():
    """BASELINE config 5 (Kip320, 7 brokers, LogSize 8, MaxRecords 8, MaxLeaderEpoch 3 — ten words per state, 5040 images per
    orbit): the first seven levels of the orbit-counting search, replayed on the CPU with the device's successor and
    representative functions, weigh up to the exact oracle's level sizes (tests/golden/oracle_kip320_7_8_8_3_levels10.json)."""
    import json
    import os
    cfg6 = (5, 7, 8, 8, 3, 0, 0)
    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                       "oracle_kip320_7_8_8_3_levels10.json")))["levels"][:7]
    nf = factorial(7)
    with host_emu.layout(cfg6):
        st0, init = host_emu.canon(cfg6, host_emu.init(cfg6))
        assert st0 == nf
        seen = {init: st0}
        frontier, levels = [init], []
        while frontier and len(levels) < len(want):
            levels.append(sum(nf // seen[s] for s in frontier))
            if len(levels) == len(want):
                break
            nxt = []
            for s in frontier:
                for _kind, t in host_emu.successors(cfg6, s):
                    st, c = host_emu.canon(cfg6, t)
                    if c not in seen:
                        seen[c] = st
                        nxt.append(c)
            frontier = nxt
    assert levels == want
    assert len(seen) < sum(want) / 300   # 1,271,426 states from a few thousand stored ones


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _expand_disassembly(cfg):
    """The instructions of kmc_expand_* in the cached gfx950 code object of cfg (specialised here if need be: hiprtc, no GPU)."""
    import subprocess
    from kafka_specification_amd import code_object_path
    text = subprocess.run([OBJDUMP, "-d", code_object_path(cfg)], capture_output=True, text=True, check=True).stdout
    body, on = [], False
    for line in text.splitlines():
        if line.endswith(">:"):
            on = "<kmc_expand_" in line
        elif on:
            body.append(line)
    assert len(body) > 1000
    return body


@pytest.mark.skipif(not __import__("os").path.exists(OBJDUMP), reason="no llvm-objdump")
def test_orbit_deficits_leave_the_block_through_64_bit_cells():
    """Round 3's defect (oracle/orbit_oracle.c found it at 17 levels of BASELINE config 5): k_expand summed the orbit deficits
    of a launch per block in 32-bit LDS cells, and they wrapped.  No GPU here, so the width is read off the machine code:
    under orbit counting the LDS adds of kmc_expand are ds_add_u64, except the two of the plain tail (raw generated per kind,
    deadlocks / probed / won / outside / repeats: counts of a block's own successors, < 2^32 by six orders of magnitude);
    the plain search's kernel has exactly those two and nothing 64 bits wide."""
    inv = ("TypeOk", "WeakIsr", "StrongIsr")
    base = dict(model="Kip320", n_replicas=3, log_size=6, max_records=6, max_leader_epoch=2, invariants=inv)
    sym = _expand_disassembly(CheckerConfig(**base, symmetry=True))
    plain = _expand_disassembly(CheckerConfig(**base))
    count = lambda body, op: sum(1 for l in body if op in l)
    assert count(plain, "ds_add_u32") == 2 and count(plain, "ds_add_u64") == 0 and count(plain, "ds_add_rtn") == 0
    assert count(sym, "ds_add_u32") == 2, "a deficit sum goes through a 32-bit LDS cell again"
    assert count(sym, "ds_add_u64") >= 4   # per-kind deficits (one per segment of the walk), deadlocks, repeats, won
    # and the block's cells reach the control block as 64-bit global adds (corr_gen / corr_dead / corr_repeats / corr_won)
    assert count(sym, "global_atomic_add_x2") > count(plain, "global_atomic_add_x2")


def test_kernel_code_identity_ignores_the_source_hash_symbol():
    """kmc.kernel_code_sha256 = sha256 over .text, .rodata (kernel descriptors) and .note (metadata) of a code object: what
    bench.py compares before quoting a PMC summary measured on an earlier text of kmc_device.h.  On a synthetic ELF64 image:
    bytes of other sections (hiprtc's `__hip_cuid_<hash of the source>` lives in .dynstr / .strtab) do not move it, one
    byte of .text does."""
    import hashlib
    import struct
    from kafka_specification_amd.checker import elf_sections

    def image(text, dynstr):
        names = b"\0.text\0.rodata\0.note\0.dynstr\0.shstrtab\0.bss\0"
        secs = [(b".text", 1, text), (b".rodata", 1, b"R" * 64), (b".note", 7, b"N" * 40), (b".dynstr", 3, dynstr),
                (b".shstrtab", 3, names), (b".bss", 8, b"")]
        blob = bytearray(64)
        hdrs = [struct.pack("<IIQQQQIIQQ", 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)]
        for nm, typ, data in secs:
            off = len(blob)
            blob += data
            hdrs.append(struct.pack("<IIQQQQIIQQ", names.index(nm), typ, 0, 0, off, len(data) if typ != 8 else 4096, 0, 0, 1, 0))
        shoff = len(blob)
        for h in hdrs:
            blob += h
        blob[:6] = b"\x7fELF\x02\x01"
        struct.pack_into("<Q", blob, 0x28, shoff)
        struct.pack_into("<HHH", blob, 0x3A, 64, len(hdrs), 5)
        return bytes(blob)

    def ident(blob):
        sec = elf_sections(blob)
        h = hashlib.sha256()
        for n in (".text", ".rodata", ".note"):
            h.update(n.encode() + len(sec[n]).to_bytes(8, "little") + sec[n])
        return h.hexdigest()

    a = image(b"\x01\x02\x03\x04" * 16, b"\0__hip_cuid_2a988ab646cf56b6\0")
    b = image(b"\x01\x02\x03\x04" * 16, b"\0__hip_cuid_416cadc9b025235a\0")
    c = image(b"\x01\x02\x03\x05" + b"\x01\x02\x03\x04" * 15, b"\0__hip_cuid_2a988ab646cf56b6\0")
    assert a != b and ident(a) == ident(b) != ident(c)
    assert set(elf_sections(a)) == {".text", ".rodata", ".note", ".dynstr", ".shstrtab"}   # .bss holds no bytes of the file
    with pytest.raises(ValueError):
        elf_sections(b"\x7fELF\x01\x01" + bytes(64))
