"""MI355X-native explicit-state model checker for the Kafka replication TLA+ specs
(hachikuji/kafka-specification).  The compute path is libkmc.so (HIP, gfx950) behind the C ABI
of include/kmc.h; this package is the thin host mirror (checker, .cfg reader, tlc-shaped CLI,
multi-GPU driver)."""
from .checker import (CheckerConfig, CheckResult, ModelChecker, precompile,  # noqa: F401
                      code_object_path, kernel_code_sha256, compiler_identity)
from ._native import KmcError, MODELS, INVARIANTS  # noqa: F401
