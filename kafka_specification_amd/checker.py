"""Host-side mirror of the TLC surface for this path: a ModelChecker that takes what a TLC
`.cfg` binds (root module, CONSTANTS, INVARIANTS, CHECK_DEADLOCK) and runs the exhaustive
breadth-first search on the GPU through the C ABI (include/kmc.h).

[TLC-recall] the names follow tlc2.tool.ModelChecker / tlc2.TLC options (-deadlock,
-continue, -fp seed); TLC itself is not part of the reference repository.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Callable, Optional, Sequence

import numpy as np

from . import _native as nat


@dataclass
class CheckerConfig:
    model: str                              # root module, e.g. "Kip320"
    n_replicas: int = 3                     # |Replicas|          KafkaReplication.tla:33
    log_size: int = 2                       # LogSize             :34   (AsyncIsr: MaxOffset, AsyncIsr.tla:25)
    max_records: int = 2                    # MaxRecords          :35
    max_leader_epoch: int = 1               # MaxLeaderEpoch      :36   (AsyncIsr: MaxVersion of the state constraint)
    n_log_records: int = 2                  # |LogRecords|        FiniteReplicatedLog.tla:24 (standalone)
    max_id: int = 10                        # MaxId               IdSequence.tla:22 (standalone)
    invariants: Sequence[str] = ("TypeOk",)
    check_deadlock: bool = False            # TLC checks deadlock by default; these bounded models need -deadlock
    continue_on_violation: bool = False     # TLC -continue
    keep_trace: bool = False
    device: int = 0
    n_shards: int = 1
    shard_id: int = 0
    table_capacity: int = 0                 # 0 = auto from free HBM
    frontier_capacity: int = 0
    send_capacity: int = 0
    hash_seed: int = 0
    max_levels: int = 0
    cache_dir: Optional[str] = None
    wide_fingerprint: bool = False          # 128-bit seen-set entries (fingerprint + an independent check word): a 64-bit
                                            # fingerprint collision is recognised instead of silently merging two states
    symmetry: bool = False                  # orbit counting: store / expand one state per orbit of the permutations of Replicas,
                                            # weigh every count by the orbit's size — the plain search's numbers (TLC without a
                                            # SYMMETRY set) from ~1/|Replicas|! of the probes.  Kafka family and
                                            # FiniteReplicatedLog, at most 7 replicas (KMC_SYMMETRY_MAX_REPLICAS); one GPU or frontier-sharded (sharded.py)

    def to_native(self) -> nat.KmcConfig:
        if self.model not in nat.MODELS:
            raise ValueError(f"unknown module {self.model!r}; known: {sorted(nat.MODELS)}")
        mask = 0
        bits = nat.invariant_bits(self.model)
        for name in self.invariants:
            if name not in bits:
                raise ValueError(f"unknown invariant {name!r} for {self.model}; known: {sorted(bits)}")
            mask |= bits[name]
        return nat.KmcConfig(
            model=nat.MODELS[self.model], n_replicas=self.n_replicas, log_size=self.log_size,
            max_records=self.max_records, max_leader_epoch=self.max_leader_epoch,
            n_log_records=self.n_log_records, max_id=self.max_id, invariant_mask=mask,
            check_deadlock=int(self.check_deadlock), continue_on_violation=int(self.continue_on_violation),
            keep_trace=int(self.keep_trace), device=self.device, n_shards=self.n_shards,
            shard_id=self.shard_id, table_capacity=self.table_capacity,
            frontier_capacity=self.frontier_capacity, send_capacity=self.send_capacity,
            hash_seed=self.hash_seed, max_levels=self.max_levels,
            cache_dir=self.cache_dir.encode() if self.cache_dir else None,
            wide_fingerprint=int(self.wide_fingerprint), symmetry=int(self.symmetry))


@dataclass
class CheckResult:
    generated: int
    distinct: int
    depth: int
    queue_left: int
    verdict: str
    violated_invariant: Optional[str]
    violation_depth: int
    violation_count: dict
    violation_fp: int
    deadlock_states: int
    action_generated: dict
    levels: list
    table_capacity: int
    frontier_capacity: int
    seconds_total: float
    seconds_expand: float
    expand_launches: int
    state_words: int
    state_bits: int
    trace: list = field(default_factory=list)
    generated_repeats: int = 0   # of `generated`: successors yielded twice by two disjuncts of one binding (one probe each)
    orbit_representatives: int = 0   # CheckerConfig.symmetry: the states actually stored and expanded (else = distinct)
    seconds_inv: float = 0.0     # the invariant pass over the last, unexpanded frontier under max_levels (k_inv)
    seconds_clear: float = 0.0   # clearing the seen-set at the start of the run
    inv_launches: int = 0


def precompile(cfg: CheckerConfig, arch: str = "gfx950", mode: int = -1) -> None:
    """Specialise + cache the kernels for cfg; needs the HIP compiler but no GPU.  mode: -1 all three code objects of the
    configuration, 0 the search's own, 1 k_expand for the level-step interface, 2 k_expand as an enumerator (include/kmc.h)."""
    c = cfg.to_native()
    nat.check(nat.lib().kmc_precompile_mode(C.byref(c), arch.encode(), mode))


def compiler_identity(which: int = 0) -> int:
    """0: the hiprtc / comgr this process compiles with; 1: the pinned one (include/kmc.h, kmc_compiler_identity)."""
    return int(nat.lib().kmc_compiler_identity(which))


def code_object_path(cfg: CheckerConfig, arch: str = "gfx950") -> str:
    """The cached code object cfg's kernels are loaded from (specialised first if it is not there; no GPU needed)."""
    c = cfg.to_native()
    buf = C.create_string_buffer(4096)
    nat.check(nat.lib().kmc_code_object_path(C.byref(c), arch.encode(), buf, 4096))
    return buf.value.decode()


def elf_sections(blob: bytes) -> dict:
    """{name: bytes} of the sections of a little-endian ELF64 image that occupy file space (an AMDGPU code object)."""
    import struct
    if blob[:6] != b"\x7fELF\x02\x01":
        raise ValueError("not a little-endian ELF64 image")
    shoff, = struct.unpack_from("<Q", blob, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", blob, 0x3A)
    hdrs = [struct.unpack_from("<IIQQQQIIQQ", blob, shoff + i * shentsize) for i in range(shnum)]
    names = hdrs[shstrndx]
    strtab = blob[names[4]:names[4] + names[5]]
    out = {}
    for name_off, sh_type, _f, _a, off, size, *_ in hdrs:
        if sh_type in (0, 8):   # SHT_NULL, SHT_NOBITS
            continue
        out[strtab[name_off:strtab.index(b"\0", name_off)].decode()] = blob[off:off + size]
    return out


def kernel_code_sha256(cfg: CheckerConfig, arch: str = "gfx950") -> str:
    """Identity of the MACHINE CODE of cfg's kernels: sha256 over the code object's .text (instructions), .rodata (kernel
    descriptors: register and LDS budgets) and .note (the AMDGPU metadata: arguments, spills, occupancy inputs).  What is
    left out is what changes when OTHER builds' lines of the device header change (hiprtc names a `__hip_cuid_<hash of the
    source text>` symbol in .dynstr / .strtab): profiles stamp their numbers with this, and bench.py quotes them only if
    the kernels it just ran hash the same."""
    import hashlib
    sec = elf_sections(open(code_object_path(cfg, arch), "rb").read())
    h = hashlib.sha256()
    for name in (".text", ".rodata", ".note"):
        h.update(name.encode() + len(sec[name]).to_bytes(8, "little") + sec[name])
    return h.hexdigest()


class ModelChecker:
    def __init__(self, cfg: CheckerConfig):
        self.cfg = cfg
        self._lib = nat.lib()
        self._native_cfg = cfg.to_native()
        self._h = C.c_void_p()
        nat.check(self._lib.kmc_open(C.byref(self._native_cfg), C.byref(self._h)))
        self.model_id = nat.MODELS[cfg.model]
        self.state_words = int(self._lib.kmc_state_words(self._h))
        self.canon_bytes = int(self._lib.kmc_canon_bytes(self._h))

    # -- lifecycle ----------------------------------------------------------------------
    def close(self):
        if self._h:
            self._lib.kmc_close(self._h)
            self._h = C.c_void_p()

    def timing(self) -> dict:
        """Where the wall time outside the search went (kmc_timing: HIP initialisation, code object, allocation, first clear)."""
        t = nat.KmcTiming()
        nat.check(self._lib.kmc_timing_get(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in nat.KmcTiming._fields_}

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def action_names(self):
        n = self._lib.kmc_action_count(self.model_id)
        return [self._lib.kmc_action_name(self.model_id, k).decode() for k in range(n)]

    # -- the search ---------------------------------------------------------------------
    @staticmethod
    def _callback(progress):
        if progress is None:
            return nat.PROGRESS_CB()

        def _cb(info, _user):
            i = info.contents
            progress(dict(depth=i.depth, new_states=i.new_states, generated=i.generated_total,
                          distinct=i.distinct_total, seconds=i.seconds))
        return nat.PROGRESS_CB(_cb)

    def run(self, progress: Optional[Callable[[dict], None]] = None) -> CheckResult:
        cb = self._callback(progress)
        nat.check(self._lib.kmc_run(self._h, cb, None))
        return self.result()

    # -- checkpoint / recover (TLC -checkpoint / -recover) ----------------------------------------
    def save_checkpoint(self, path: str) -> None:
        nat.check(self._lib.kmc_checkpoint_save(self._h, path.encode()))

    def load_checkpoint(self, path: str) -> None:
        nat.check(self._lib.kmc_checkpoint_load(self._h, path.encode()))

    def resume(self, progress: Optional[Callable[[dict], None]] = None) -> CheckResult:
        cb = self._callback(progress)
        nat.check(self._lib.kmc_resume(self._h, cb, None))
        return self.result()

    def result(self) -> CheckResult:
        r = nat.KmcResult()
        nat.check(self._lib.kmc_result_get(self._h, C.byref(r)))
        nlev = int(r.n_levels)
        buf = (C.c_uint64 * max(nlev, 1))()
        self._lib.kmc_level_sizes(self._h, buf, nlev)
        names = self.action_names()
        inv_names = nat.invariant_names(self.cfg.model)
        return CheckResult(
            generated=int(r.generated), distinct=int(r.distinct), depth=int(r.depth),
            queue_left=int(r.queue_left), verdict=nat.VERDICTS[r.verdict],
            violated_invariant=(inv_names[r.violated_invariant] if r.violated_invariant >= 0 else None),
            violation_depth=int(r.violation_depth),
            violation_count={inv_names[k]: int(r.violation_count[k]) for k in range(4) if inv_names[k] != "?"},
            violation_fp=int(r.violation_fp), deadlock_states=int(r.deadlock_states),
            action_generated={names[k]: int(r.action_generated[k]) for k in range(len(names))},
            levels=[int(buf[i]) for i in range(nlev)],
            table_capacity=int(r.table_capacity), frontier_capacity=int(r.frontier_capacity),
            seconds_total=float(r.seconds_total), seconds_expand=float(r.seconds_expand),
            expand_launches=int(r.expand_launches), state_words=int(r.state_words), state_bits=int(r.state_bits),
            generated_repeats=int(r.generated_repeats), orbit_representatives=int(r.orbit_representatives),
            seconds_inv=float(r.seconds_inv), seconds_clear=float(r.seconds_clear), inv_launches=int(r.inv_launches))

    def level_stats(self) -> list:
        """One dict per expansion of the last search (kmc_level_stat): depth produced, frontier expanded, new states, generated per
        disjunct of Next, probes, deadlocks, table load, k_expand milliseconds."""
        n = int(self._lib.kmc_level_stats(self._h, None, 0))
        buf = (nat.KmcLevelStat * max(n, 1))()
        n = min(n, int(self._lib.kmc_level_stats(self._h, buf, n)))
        names = self.action_names()
        return [dict(depth=int(b.depth), frontier=int(b.frontier), new_states=int(b.new_states), stored_new=int(b.stored_new),
                     generated={names[k]: int(b.generated[k]) for k in range(len(names))}, probes=int(b.probes),
                     deadlocks=int(b.deadlocks), table_load=float(b.table_load), expand_ms=float(b.expand_ms))
                for b in buf[:n]]

    # -- states as data -------------------------------------------------------------------
    def unpack(self, words) -> bytes:
        w = (C.c_uint64 * self.state_words)(*[int(x) for x in words])
        out = (C.c_uint8 * self.canon_bytes)()
        nat.check(self._lib.kmc_unpack_state(self._h, w, out))
        return bytes(out)

    def pack(self, canon: bytes):
        c = (C.c_uint8 * self.canon_bytes)(*canon)
        w = (C.c_uint64 * self.state_words)()
        nat.check(self._lib.kmc_pack_state(self._h, c, w))
        return [int(x) for x in w]

    def fingerprint(self, words) -> int:
        w = (C.c_uint64 * self.state_words)(*[int(x) for x in words])
        return int(self._lib.kmc_fingerprint_of(self._h, w))

    def canonical(self, words):
        """(order of the stabiliser, representative words) of a packed state's orbit under the permutations of Replicas."""
        w = (C.c_uint64 * self.state_words)(*[int(x) for x in words])
        out = (C.c_uint64 * self.state_words)()
        stab = C.c_int32()
        nat.check(self._lib.kmc_canonical_state(self._h, w, out, C.byref(stab)))
        return stab.value, tuple(int(x) for x in out)

    def frontier_states(self) -> np.ndarray:
        """The last completed BFS level as an (n, state_words) uint64 array."""
        n = C.c_uint64()
        nat.check(self._lib.kmc_frontier_states(self._h, None, 0, C.byref(n)))
        out = np.zeros((n.value, self.state_words), dtype=np.uint64)
        if n.value:
            nat.check(self._lib.kmc_frontier_states(
                self._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), n.value, C.byref(n)))
        return out

    def successors(self, words):
        """[(state_words tuple, fingerprint, action kind)] of one packed state, from the device."""
        W = self.state_words
        w = (C.c_uint64 * W)(*[int(x) for x in words])
        cap = 4096
        out = np.zeros((cap, W + 2), dtype=np.uint64)
        n = C.c_uint64()
        nat.check(self._lib.kmc_successors(self._h, w, out.ctypes.data_as(C.POINTER(C.c_uint64)), cap, C.byref(n)))
        return [(tuple(int(x) for x in out[i, :W]), int(out[i, W]), int(out[i, W + 1])) for i in range(min(n.value, cap))]

    def contains(self, words) -> bool:
        """Was this packed state reached by the last run?  (FPSet.contains analogue)"""
        w = (C.c_uint64 * self.state_words)(*[int(x) for x in words])
        present = C.c_int32()
        nat.check(self._lib.kmc_contains(self._h, w, C.byref(present)))
        return bool(present.value)

    def witness(self):
        w = (C.c_uint64 * self.state_words)()
        nat.check(self._lib.kmc_witness(self._h, w))
        return [int(x) for x in w]

    def trace(self, cap: int = 4096):
        """[(action name or None, canonical state bytes)] from Init to the violation witness."""
        states = (C.c_uint8 * (cap * self.canon_bytes))()
        kinds = (C.c_int32 * cap)()
        n = C.c_uint64()
        nat.check(self._lib.kmc_trace(self._h, states, kinds, cap, C.byref(n)))
        names = self.action_names()
        out = []
        for i in range(min(n.value, cap)):
            b = bytes(states[i * self.canon_bytes:(i + 1) * self.canon_bytes])
            out.append((None if kinds[i] < 0 else names[kinds[i]], b))
        return out
