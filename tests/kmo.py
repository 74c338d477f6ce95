"""Test-side binding of the C oracle (oracle/libkmc_oracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libkmc_oracle.so")

KMO_MAX_LEVELS = 512
KMO_MAX_ACTIONS = 16
MODELS = {"IdSequence": 0, "FiniteReplicatedLog": 1, "KafkaTruncateToHighWatermark": 2, "Kip101": 3,
          "Kip279": 4, "Kip320": 5, "Kip320FirstTry": 6, "AsyncIsr": 7}
INV_BITS = {"TypeOk": 1, "WeakIsr": 2, "StrongIsr": 4, "LeaderInIsr": 8}
INV_NAMES = ("TypeOk", "WeakIsr", "StrongIsr", "LeaderInIsr")
# AsyncIsr (N replicas, L = MaxOffset, E = MaxVersion) reuses the bit positions
ASYNC_INV_BITS = {"TypeOk": 1, "ValidHighWatermark": 2, "LeaderOffsetInRange": 4}
ASYNC_INV_NAMES = ("TypeOk", "ValidHighWatermark", "LeaderOffsetInRange", "?")
KMO_MAXSB = 160
VERDICTS = ("ok", "invariant", "deadlock", "limit", "error")


class Config(C.Structure):
    _fields_ = [("model", C.c_int32), ("N", C.c_int32), ("L", C.c_int32), ("R", C.c_int32), ("E", C.c_int32),
                ("K", C.c_int32), ("MaxId", C.c_int64), ("inv_mask", C.c_uint32), ("check_deadlock", C.c_int32),
                ("stop_on_violation", C.c_int32), ("threads", C.c_int32), ("max_states", C.c_uint64)]


class Result(C.Structure):
    _fields_ = [("distinct", C.c_uint64), ("generated", C.c_uint64), ("depth", C.c_uint64),
                ("verdict", C.c_int32), ("viol_inv", C.c_int32), ("viol_depth", C.c_uint64),
                ("viol_state_idx", C.c_uint64), ("viol_count", C.c_uint64 * 4), ("deadlock_states", C.c_uint64),
                ("action_generated", C.c_uint64 * KMO_MAX_ACTIONS), ("nlevels", C.c_uint64),
                ("levels", C.c_uint64 * KMO_MAX_LEVELS), ("seconds", C.c_double),
                ("viol_outside", C.c_int32), ("viol_action", C.c_int32), ("viol_state", C.c_uint8 * KMO_MAXSB)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "libkmc_oracle.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        l = C.CDLL(LIB)
        l.kmo_run.restype = C.c_void_p
        l.kmo_run.argtypes = [C.POINTER(Config), C.POINTER(Result)]
        l.kmo_state_bytes.restype = C.c_int
        l.kmo_state_bytes.argtypes = [C.c_void_p]
        l.kmo_num_states.restype = C.c_uint64
        l.kmo_num_states.argtypes = [C.c_void_p]
        l.kmo_get_states.restype = None
        l.kmo_get_states.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        l.kmo_parent.restype = C.c_int64
        l.kmo_parent.argtypes = [C.c_void_p, C.c_uint64]
        l.kmo_action.restype = C.c_int
        l.kmo_action.argtypes = [C.c_void_p, C.c_uint64]
        l.kmo_successors.restype = C.c_int
        l.kmo_successors.argtypes = [C.POINTER(Config), C.c_void_p, C.c_void_p, C.c_int]
        l.kmo_check_invariant.restype = C.c_int
        l.kmo_check_invariant.argtypes = [C.POINTER(Config), C.c_int, C.c_void_p]
        l.kmo_free.restype = None
        l.kmo_free.argtypes = [C.c_void_p]
        _lib = l
    return _lib


def make_config(model, N=3, L=2, R=2, E=1, K=2, MaxId=10, invariants=("TypeOk",), check_deadlock=False,
                stop_on_violation=True, threads=4, max_states=0):
    mask = 0
    for n in invariants:
        mask |= (ASYNC_INV_BITS if model == "AsyncIsr" else INV_BITS)[n]
    return Config(model=MODELS[model], N=N, L=L, R=R, E=E, K=K, MaxId=MaxId, inv_mask=mask,
                  check_deadlock=int(check_deadlock), stop_on_violation=int(stop_on_violation), threads=threads,
                  max_states=max_states)


class Run:
    """One exhaustive oracle run; keeps the arena so states can be fetched per level."""

    def __init__(self, cfg: Config):
        self.cfg = cfg
        self.res = Result()
        self.h = lib().kmo_run(C.byref(cfg), C.byref(self.res))
        r = self.res
        self.distinct, self.generated, self.depth = int(r.distinct), int(r.generated), int(r.depth)
        self.verdict = VERDICTS[r.verdict]
        names = ASYNC_INV_NAMES if cfg.model == MODELS["AsyncIsr"] else INV_NAMES
        self.viol_inv = names[r.viol_inv] if r.viol_inv >= 0 else None
        self.viol_depth = int(r.viol_depth)
        self.viol_count = {names[k]: int(r.viol_count[k]) for k in range(4) if names[k] != "?"}
        # a violating successor outside the state constraint: (parent index, action, state bytes)
        self.viol_outside = bool(r.viol_outside)
        self.viol_parent_idx, self.viol_action = int(r.viol_state_idx), int(r.viol_action)
        self.deadlock_states = int(r.deadlock_states)
        self.levels = [int(r.levels[i]) for i in range(min(int(r.nlevels), KMO_MAX_LEVELS))]
        self.action_generated = [int(x) for x in r.action_generated]
        self.seconds = float(r.seconds)
        self.sb = lib().kmo_state_bytes(self.h) if self.h else 0
        self.viol_state = bytes(r.viol_state[:self.sb]) if self.viol_outside else None

    def level_states(self, k):
        """set of canonical-byte states first seen at level k (0-based)."""
        first = sum(self.levels[:k])
        n = self.levels[k]
        buf = (C.c_uint8 * (n * self.sb))()
        lib().kmo_get_states(self.h, first, n, buf)
        raw = bytes(buf)
        return {raw[i * self.sb:(i + 1) * self.sb] for i in range(n)}

    def state(self, idx):
        buf = (C.c_uint8 * self.sb)()
        lib().kmo_get_states(self.h, idx, 1, buf)
        return bytes(buf)

    def close(self):
        if self.h:
            lib().kmo_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


def successors(cfg: Config, state: bytes, sb: int, cap=4096):
    out = (C.c_uint8 * (cap * (sb + 1)))()
    st = (C.c_uint8 * sb)(*state)
    n = lib().kmo_successors(C.byref(cfg), st, out, cap)
    raw = bytes(out)
    return [(raw[i * (sb + 1)], raw[i * (sb + 1) + 1:(i + 1) * (sb + 1)]) for i in range(min(n, cap))]


def check_invariant(cfg: Config, inv_index: int, state: bytes):
    st = (C.c_uint8 * len(state))(*state)
    return bool(lib().kmo_check_invariant(C.byref(cfg), inv_index, st))
