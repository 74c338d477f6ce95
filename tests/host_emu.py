"""Test-side binding of tests/host_emu.cpp: the DEVICE model templates compiled for the host.
Test infrastructure only."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_emu.cpp")
LIB = os.path.join(HERE, "_host_emu.so")
CSRC = os.path.join(HERE, "..", "kafka_specification_amd", "csrc")
DEPS = [SRC] + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.startswith("kmc_") and f.endswith(".h")]
_lib = None


def build():
    if os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in DEPS):
        return
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", LIB, SRC])


def lib():
    global _lib
    if _lib is None:
        build()
        l = C.CDLL(LIB)
        six = [C.c_int] * 6
        l.emu_configs.argtypes = [C.c_int, C.POINTER(C.c_int)]
        l.emu_layout.argtypes = [C.c_int]
        l.emu_is_rm.argtypes = [C.c_int] * 5
        l.emu_words.argtypes = six
        l.emu_successors.argtypes = six + [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]
        l.emu_violated.argtypes = six + [C.POINTER(C.c_uint64), C.c_uint]
        l.emu_init.argtypes = six + [C.POINTER(C.c_uint64)]
        l.emu_kafka_reference.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_uint64), C.c_uint]
        l.emu_state_bits.argtypes = six
        l.emu_in_model.argtypes = six + [C.POINTER(C.c_uint64)]
        l.emu_canon.argtypes = six + [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        l.emu_canon_generic.argtypes = six + [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        l.emu_kind_major_check.argtypes = six + [C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _lib = l
    return _lib


LAYOUT_ENV = {0: "auto", 1: "tight", 2: "rm", 3: "rmg"}   # KMC_LAYOUT_* -> the KMC_LAYOUT value that makes the host library agree


def configs():
    """[(model, N, L, R, E, K, layout mode)] compiled into the emulation."""
    out = (C.c_int * 7)()
    n = lib().emu_configs(-1, out)
    res = []
    for i in range(n):
        lib().emu_configs(i, out)
        res.append(tuple(out))
    return res


class layout:
    """with host_emu.layout(cfg7): the emulation AND the host library (KMC_LAYOUT, read when a handle is opened) use
    the arrangement of the state vector that entry was compiled with."""

    def __init__(self, cfg7):
        self.lm = cfg7[6] if len(cfg7) > 6 else 0

    def __enter__(self):
        self.old = os.environ.get("KMC_LAYOUT")
        os.environ["KMC_LAYOUT"] = LAYOUT_ENV[self.lm]
        lib().emu_layout(self.lm)
        return self

    def __exit__(self, *exc):
        lib().emu_layout(0)
        if self.old is None:
            os.environ.pop("KMC_LAYOUT", None)
        else:
            os.environ["KMC_LAYOUT"] = self.old


def successors(cfg6, words, cap=4096):
    W = lib().emu_words(*cfg6[:6])
    w = (C.c_uint64 * W)(*words)
    out = (C.c_uint64 * (cap * (W + 1)))()
    n = lib().emu_successors(*cfg6[:6], w, out, cap)
    assert 0 <= n <= cap
    return [(int(out[i * (W + 1) + W]), tuple(int(out[i * (W + 1) + k]) for k in range(W))) for i in range(n)]


def violated(cfg6, words, mask):
    W = lib().emu_words(*cfg6[:6])
    return lib().emu_violated(*cfg6[:6], (C.c_uint64 * W)(*words), mask)


def init(cfg6):
    W = lib().emu_words(*cfg6[:6])
    w = (C.c_uint64 * W)()
    assert lib().emu_init(*cfg6[:6], w) == 0
    return [int(x) for x in w]


def in_model(cfg6, words):
    W = lib().emu_words(*cfg6[:6])
    return bool(lib().emu_in_model(*cfg6[:6], (C.c_uint64 * W)(*words)))


def kind_major_check(cfg6, words):
    """(has a kind-major form, bindings checked, bindings that differ): KmcKafka::apply<K> against inst<I> on one state."""
    W = lib().emu_words(*cfg6[:6])
    checked, bad = C.c_int(0), C.c_int(0)
    rc = lib().emu_kind_major_check(*cfg6[:6], (C.c_uint64 * W)(*words), C.byref(checked), C.byref(bad))
    assert rc >= 0
    return bool(rc), checked.value, bad.value


def kafka_reference(cfg6, words, mask):
    """Literal loop-per-slot evaluation of the Kafka invariants on packed words (tests/host_emu.cpp); -1 = undefined."""
    W = lib().emu_words(*cfg6[:6])
    return lib().emu_kafka_reference(*cfg6[:5], (C.c_uint64 * W)(*words), mask)


def state_bits(cfg6):
    return lib().emu_state_bits(*cfg6[:6])


def canon(cfg6, words, generic=False):
    """(stabiliser order, representative words) of a state's orbit under the permutations of Replicas: KmcSymm<M>::canon
    (the device's compile-time form) or, generic=True, kmc_canonical_state_generic (the host engine's run-time form)."""
    W = lib().emu_words(*cfg6[:6])
    out = (C.c_uint64 * W)()
    f = lib().emu_canon_generic if generic else lib().emu_canon
    st = f(*cfg6[:6], (C.c_uint64 * W)(*words), out)
    return st, tuple(int(x) for x in out)
