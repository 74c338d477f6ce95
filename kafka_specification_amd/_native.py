"""ctypes binding of libkmc.so (include/kmc.h).  Loading fails loudly when the in-tree
library has not been built: there is no Python or CPU fallback for the checker."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KMC_LIB_PATH") or os.path.join(_HERE, "libkmc.so")  # override: A/B two builds

KMC_MAX_KINDS = 16
KMC_MAX_SHARDS = 8
KMC_SYMMETRY_MAX_REPLICAS = 7
KMC_SEND_SUBS = 8
KMC_COMM_ID_BYTES = 128
KMC_EXCHANGE_STATS = 64

MODELS = {
    "IdSequence": 0,
    "FiniteReplicatedLog": 1,
    "KafkaTruncateToHighWatermark": 2,
    "Kip101": 3,
    "Kip279": 4,
    "Kip320": 5,
    "Kip320FirstTry": 6,
    "AsyncIsr": 7,      # under the state constraint of models/MCAsyncIsr.tla
}
INVARIANTS = {"TypeOk": 1, "WeakIsr": 2, "StrongIsr": 4, "LeaderInIsr": 8}
INVARIANT_NAMES = ("TypeOk", "WeakIsr", "StrongIsr", "LeaderInIsr")
# AsyncIsr reuses the bit positions (include/kmc.h): TypeOk, ValidHighWatermark, LeaderOffsetInRange
ASYNC_INVARIANTS = {"TypeOk": 1, "ValidHighWatermark": 2, "LeaderOffsetInRange": 4}
ASYNC_INVARIANT_NAMES = ("TypeOk", "ValidHighWatermark", "LeaderOffsetInRange", "?")


def invariant_bits(model: str) -> dict:
    return ASYNC_INVARIANTS if model == "AsyncIsr" else INVARIANTS


def invariant_names(model: str) -> tuple:
    return ASYNC_INVARIANT_NAMES if model == "AsyncIsr" else INVARIANT_NAMES
VERDICTS = ("ok", "invariant", "deadlock", "table_full", "frontier_full", "level_limit", "error")


class KmcConfig(C.Structure):
    _fields_ = [
        ("model", C.c_int32), ("n_replicas", C.c_int32), ("log_size", C.c_int32),
        ("max_records", C.c_int32), ("max_leader_epoch", C.c_int32), ("n_log_records", C.c_int32),
        ("max_id", C.c_int64), ("invariant_mask", C.c_uint32), ("check_deadlock", C.c_int32),
        ("continue_on_violation", C.c_int32), ("keep_trace", C.c_int32), ("device", C.c_int32),
        ("n_shards", C.c_int32), ("shard_id", C.c_int32),
        ("table_capacity", C.c_uint64), ("frontier_capacity", C.c_uint64), ("send_capacity", C.c_uint64),
        ("hash_seed", C.c_uint64), ("max_levels", C.c_uint64), ("cache_dir", C.c_char_p),
        ("wide_fingerprint", C.c_int32), ("symmetry", C.c_int32),
    ]


class KmcLevelInfo(C.Structure):
    _fields_ = [("depth", C.c_uint64), ("new_states", C.c_uint64), ("generated_total", C.c_uint64),
                ("distinct_total", C.c_uint64), ("seconds", C.c_double),
                ("generated_level", C.c_uint64 * KMC_MAX_KINDS), ("violation_count", C.c_uint64 * 4),
                ("violation_fp", C.c_uint64 * 4), ("outside_violation_count", C.c_uint64 * 4),
                ("outside_violation_fp", C.c_uint64 * 4), ("deadlocks_level", C.c_uint64), ("send_filtered", C.c_uint64),
                ("error_flags", C.c_uint32), ("pad_", C.c_uint32)]


class KmcTiming(C.Structure):
    _fields_ = [("hip_init_s", C.c_double), ("code_object_s", C.c_double), ("alloc_s", C.c_double), ("first_clear_s", C.c_double),
                ("open_s", C.c_double), ("device_bytes", C.c_uint64)]


class KmcResult(C.Structure):
    _fields_ = [
        ("generated", C.c_uint64), ("distinct", C.c_uint64), ("depth", C.c_uint64), ("queue_left", C.c_uint64),
        ("verdict", C.c_int32), ("violated_invariant", C.c_int32),
        ("violation_depth", C.c_uint64), ("violation_count", C.c_uint64 * 4), ("violation_fp", C.c_uint64),
        ("deadlock_states", C.c_uint64), ("action_generated", C.c_uint64 * KMC_MAX_KINDS),
        ("n_levels", C.c_uint64), ("table_capacity", C.c_uint64), ("frontier_capacity", C.c_uint64),
        ("seconds_total", C.c_double), ("seconds_expand", C.c_double), ("expand_launches", C.c_uint64),
        ("state_words", C.c_uint64), ("state_bits", C.c_uint64), ("generated_repeats", C.c_uint64),
        ("orbit_representatives", C.c_uint64),
        ("seconds_inv", C.c_double), ("seconds_clear", C.c_double), ("inv_launches", C.c_uint64),
    ]


class KmcLevelStat(C.Structure):
    _fields_ = [("depth", C.c_uint64), ("frontier", C.c_uint64), ("new_states", C.c_uint64), ("stored_new", C.c_uint64),
                ("generated", C.c_uint64 * KMC_MAX_KINDS), ("probes", C.c_uint64), ("deadlocks", C.c_uint64),
                ("table_load", C.c_double), ("expand_ms", C.c_double)]


PROGRESS_CB = C.CFUNCTYPE(None, C.POINTER(KmcLevelInfo), C.c_void_p)

# every symbol include/kmc.h declares: (name, restype, argtypes)
_H = C.c_void_p
SYMBOLS = [
    ("kmc_open", C.c_int, [C.POINTER(KmcConfig), C.POINTER(_H)]),
    ("kmc_precompile", C.c_int, [C.POINTER(KmcConfig), C.c_char_p]),
    ("kmc_precompile_mode", C.c_int, [C.POINTER(KmcConfig), C.c_char_p, C.c_int32]),
    ("kmc_code_object_path", C.c_int, [C.POINTER(KmcConfig), C.c_char_p, C.c_char_p, C.c_uint64]),
    ("kmc_run", C.c_int, [_H, PROGRESS_CB, C.c_void_p]),
    ("kmc_result_get", C.c_int, [_H, C.POINTER(KmcResult)]),
    ("kmc_timing_get", C.c_int, [_H, C.POINTER(KmcTiming)]),
    ("kmc_checkpoint_save", C.c_int, [_H, C.c_char_p]),
    ("kmc_checkpoint_load", C.c_int, [_H, C.c_char_p]),
    ("kmc_resume", C.c_int, [_H, PROGRESS_CB, C.c_void_p]),
    ("kmc_level_sizes", C.c_uint64, [_H, C.POINTER(C.c_uint64), C.c_uint64]),
    ("kmc_level_stats", C.c_uint64, [_H, C.POINTER(KmcLevelStat), C.c_uint64]),
    ("kmc_compiler_identity", C.c_int64, [C.c_int32]),
    ("kmc_close", None, [_H]),
    ("kmc_last_error", C.c_char_p, []),
    ("kmc_state_words", C.c_uint64, [_H]),
    ("kmc_canon_bytes", C.c_uint64, [_H]),
    ("kmc_unpack_state", C.c_int, [_H, C.POINTER(C.c_uint64), C.POINTER(C.c_uint8)]),
    ("kmc_pack_state", C.c_int, [_H, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)]),
    ("kmc_fingerprint_of", C.c_uint64, [_H, C.POINTER(C.c_uint64)]),
    ("kmc_canonical_state", C.c_int, [_H, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]),
    ("kmc_frontier_states", C.c_int, [_H, C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_uint64)]),
    ("kmc_successors", C.c_int, [_H, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_uint64)]),
    ("kmc_check_states", C.c_int, [_H, C.POINTER(C.c_uint64), C.c_uint64, C.c_uint32, C.POINTER(C.c_uint32)]),
    ("kmc_trace", C.c_int, [_H, C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.c_uint64, C.POINTER(C.c_uint64)]),
    ("kmc_contains", C.c_int, [_H, C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]),
    ("kmc_witness", C.c_int, [_H, C.POINTER(C.c_uint64)]),
    ("kmc_model_name", C.c_char_p, [C.c_int32]),
    ("kmc_action_name", C.c_char_p, [C.c_int32, C.c_int32]),
    ("kmc_action_count", C.c_int32, [C.c_int32]),
    ("kmc_invariant_name", C.c_char_p, [C.c_int32]),
    ("kmc_model_invariant_name", C.c_char_p, [C.c_int32, C.c_int32]),
    ("kmc_pred_of", C.c_int, [_H, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]),
    ("kmc_init_state", C.c_int, [_H, C.POINTER(C.c_uint64)]),
    ("kmc_owner_of", C.c_int32, [C.c_uint64, C.c_int32]),
    ("kmc_step_begin", C.c_int, [_H]),
    ("kmc_step_expand", C.c_int, [_H, C.POINTER(C.c_uint64)]),
    ("kmc_step_send_buffer", C.c_int, [_H, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    ("kmc_step_set_send_buffer", C.c_int, [_H, C.c_void_p, C.c_uint64]),
    ("kmc_step_insert", C.c_int, [_H, C.c_void_p, C.c_uint64]),
    ("kmc_step_finish", C.c_int, [_H, C.POINTER(KmcLevelInfo)]),
    ("kmc_step_set_verdict", C.c_int, [_H, C.c_int32]),
    ("kmc_step_resume", C.c_int, [_H]),
    ("kmc_step_check_frontier", C.c_int, [_H, C.POINTER(KmcLevelInfo)]),
    ("kmc_step_find_outside", C.c_int, [_H, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]),
    ("kmc_comm_unique_id", C.c_int, [C.POINTER(C.c_uint8)]),
    ("kmc_comm_init", C.c_int, [_H, C.POINTER(C.c_uint8)]),
    ("kmc_comm_selftest", C.c_int, [_H]),
    ("kmc_step_exchange_counts", C.c_int, [_H, C.POINTER(C.c_int64), C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]),
    ("kmc_step_expand_counts", C.c_int, [_H, C.POINTER(C.c_int64), C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_uint64)]),
    ("kmc_step_exchange_payload", C.c_int, [_H]),
    ("kmc_step_level_parts", C.c_int, [_H, C.c_int32, C.POINTER(C.c_int64), C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]),
    ("kmc_step_exchange_local", C.c_int, [C.POINTER(_H), C.c_int32, C.POINTER(C.c_int64), C.c_int32, C.POINTER(C.c_int64)]),
    ("kmc_step_deliver_local", C.c_int, [C.POINTER(_H), C.c_int32]),
    ("kmc_exchange_plan", C.c_int, [C.POINTER(C.c_uint64), C.c_int32, C.c_int32, C.c_uint64, C.c_uint64,
                                    C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_uint64),
                                    C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
]

_lib = None


class KmcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"kmc error {code}: {msg}")
        self.code = code


def lib():
    """The loaded libkmc.so (raises when it is missing — no fallback)."""
    global _lib
    if _lib is None:
        # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64 /
        # libhsa-runtime64 / libhiprtc (same SONAMEs as /opt/rocm).  If torch is loaded after
        # libkmc.so the process ends up with two HSA runtimes and torch finds no GPU; loaded
        # first, libkmc.so binds to torch's copies.  KMC_NO_TORCH=1 keeps the library on the
        # system ROCm (standalone CLI use, rocprofv3 runs).
        if os.environ.get("KMC_NO_TORCH", "0") != "1":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        if not os.path.exists(LIB_PATH):
            raise KmcError(-1, f"{LIB_PATH} is not built; run `python __graft_entry__.py` "
                               "(or make -C kafka_specification_amd/csrc)")
        l = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise KmcError(rc, lib().kmc_last_error().decode("utf-8", "replace"))
