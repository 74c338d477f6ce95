"""Named model configurations: the BASELINE.json ladder and the set of constants whose
kernels `build()` specialises ahead of time (anything else is specialised on first use)."""
from .checker import CheckerConfig

KAFKA = ("KafkaTruncateToHighWatermark", "Kip101", "Kip279", "Kip320", "Kip320FirstTry")

# BASELINE.json configs.  KafkaReplication.tla has no Next, so "KafkaReplication,
# 3 brokers, maxLogLen=6" is bound to root module Kip320; MaxRecords / MaxLeaderEpoch are not
# pinned by BASELINE.json and are stated with every number.
HEADLINE = dict(model="Kip320", n_replicas=3, log_size=6, max_records=6, max_leader_epoch=2,
                invariants=("TypeOk", "WeakIsr", "StrongIsr"))
BASELINE_CONFIGS = {
    "config0_idsequence": dict(model="IdSequence", max_id=1000, invariants=("TypeOk",)),
    "config1_finite_replicated_log": dict(model="FiniteReplicatedLog", n_replicas=2, log_size=4, n_log_records=4,
                                          invariants=("TypeOk",)),
    "config2_headline_kip320_3brokers_log6": HEADLINE,
    # exhaustible: 112,549,196 distinct states (tests/golden/oracle_kip279_5_2_2_1.json); MaxLeaderEpoch = 2 is not
    "config3_kip279_5brokers": dict(model="Kip279", n_replicas=5, log_size=2, max_records=2, max_leader_epoch=1,
                                    invariants=("TypeOk",)),
    "config4_kip320_7brokers_log8": dict(model="Kip320", n_replicas=7, log_size=8, max_records=8,
                                         max_leader_epoch=3, invariants=("TypeOk",)),
}
# BASELINE config 4 at the sizing SURVEY section 8(a.0) gives it (LogSize 4, MaxRecords 4, MaxLeaderEpoch 3: W = 4 words): not
# exhaustible — 26 M states at depth 10, growing 3x per level — so it is checked over a level budget, like config 5.  At these
# constants a log holds up to four epochs: FirstNonMatchingOffsetFromTail (Kip279.tla:39-45) has something to truncate.
CONFIG4_DEEP = dict(model="Kip279", n_replicas=5, log_size=4, max_records=4, max_leader_epoch=3, invariants=("TypeOk",))
CONFIG4_DEEP_LEVELS = 12


def precompile_list():
    out = []
    for m in KAFKA:
        for (N, L, R, E) in [(2, 2, 2, 1), (3, 2, 2, 1), (3, 1, 1, 2), (2, 3, 3, 2), (3, 2, 2, 2), (2, 2, 2, 2)]:
            out.append(dict(model=m, n_replicas=N, log_size=L, max_records=R, max_leader_epoch=E))
    out += [dict(model="Kip320", n_replicas=3, log_size=3, max_records=3, max_leader_epoch=2),
            dict(model="KafkaTruncateToHighWatermark", n_replicas=3, log_size=3, max_records=3, max_leader_epoch=1),
            dict(model="Kip279", n_replicas=3, log_size=2, max_records=3, max_leader_epoch=2),
            dict(model="Kip101", n_replicas=3, log_size=3, max_records=2, max_leader_epoch=2),
            dict(model="Kip320FirstTry", n_replicas=3, log_size=3, max_records=3, max_leader_epoch=1),
            dict(model="Kip320", n_replicas=3, log_size=4, max_records=4, max_leader_epoch=2),
            dict(model="Kip320", n_replicas=3, log_size=6, max_records=6, max_leader_epoch=1),
            dict(model="Kip320", n_replicas=3, log_size=5, max_records=5, max_leader_epoch=2),
            dict(model="KafkaTruncateToHighWatermark", n_replicas=3, log_size=5, max_records=5, max_leader_epoch=2),
            dict(model="KafkaTruncateToHighWatermark", n_replicas=3, log_size=6, max_records=6, max_leader_epoch=2)]
    out += [dict(model=m, n_replicas=3, log_size=5, max_records=5, max_leader_epoch=2) for m in ("Kip101", "Kip279", "Kip320FirstTry")]
    out += [dict(model=m, n_replicas=3, log_size=6, max_records=6, max_leader_epoch=2) for m in ("Kip101", "Kip279", "Kip320FirstTry")]
    # wider replica sets: more words per state, several 64-bit words of action-instance bits
    out += [dict(model="Kip320", n_replicas=4, log_size=2, max_records=2, max_leader_epoch=1),
            dict(model="Kip279", n_replicas=5, log_size=1, max_records=1, max_leader_epoch=1),
            dict(model="Kip101", n_replicas=4, log_size=2, max_records=1, max_leader_epoch=2),
            dict(model="KafkaTruncateToHighWatermark", n_replicas=6, log_size=1, max_records=1, max_leader_epoch=1),
            dict(model="Kip320FirstTry", n_replicas=8, log_size=1, max_records=1, max_leader_epoch=0),
            dict(model="Kip320", n_replicas=7, log_size=1, max_records=1, max_leader_epoch=0),
            dict(model="Kip279", n_replicas=7, log_size=1, max_records=1, max_leader_epoch=0)]
    out += [dict(model="KafkaTruncateToHighWatermark", n_replicas=3, log_size=2, max_records=2, max_leader_epoch=2),
            dict(model="Kip279", n_replicas=3, log_size=2, max_records=3, max_leader_epoch=1),
            dict(model="Kip320", n_replicas=3, log_size=3, max_records=3, max_leader_epoch=1),
            dict(model="Kip320", n_replicas=4, log_size=2, max_records=1, max_leader_epoch=1)]
    # -continue parity on the violating models, the models/*.cfg twins
    out += [dict(model=m, n_replicas=3, log_size=3, max_records=2, max_leader_epoch=1)
            for m in ("KafkaTruncateToHighWatermark", "Kip101", "Kip279")]
    out += [dict(model="Kip320FirstTry", n_replicas=3, log_size=2, max_records=3, max_leader_epoch=2)]
    out += [dict(model="Kip279", n_replicas=5, log_size=2, max_records=2, max_leader_epoch=2)]  # the non-exhaustible twin (prefix test)
    # BASELINE config 4 at SURVEY 8(a.0)'s sizing: logs four deep, four epochs (bench.py's config4_deep leg, the per-state fixture)
    out += [dict(model="Kip279", n_replicas=5, log_size=4, max_records=4, max_leader_epoch=3)]
    # tests/golden/oracle_r_ladder.json's 3/3/3/1 entries (round 4: the reference's text executed with logs three deep)
    out += [dict(model=m, n_replicas=3, log_size=3, max_records=3, max_leader_epoch=1) for m in KAFKA]
    out += [dict(model="Kip320", n_replicas=3, log_size=2, max_records=3, max_leader_epoch=2)]   # oracle_r_ladder.json: 1.69 M states
    # two small bindings the -m gpu suite opens (found by listing what a fresh box still had to specialise, round 4)
    out += [dict(model="Kip320", n_replicas=2, log_size=1, max_records=1, max_leader_epoch=1),
            dict(model="Kip320", n_replicas=3, log_size=2, max_records=3, max_leader_epoch=1)]
    # tests/golden/oracle_r_wide.json (the reference's text executed at 4-7 replicas): every Kafka module at 4/1/1/0 and 5/1/1/0,
    # Kip320 at 6/1/1/0 (4/2/1/1, 7/1/1/0 and the BASELINE bindings are above / below)
    out += [dict(model=m, n_replicas=N, log_size=1, max_records=1, max_leader_epoch=0) for m in KAFKA for N in (4, 5)]
    out += [dict(model="Kip320", n_replicas=6, log_size=1, max_records=1, max_leader_epoch=0)]
    for M in (0, 1, 10, 1000):
        out.append(dict(model="IdSequence", max_id=M))
    for K in (1, 2, 3, 4):
        out.append(dict(model="FiniteReplicatedLog", n_replicas=2, log_size=4, n_log_records=K))
    # AsyncIsr under the state constraint of models/MCAsyncIsr.tla: (N, MaxOffset, MaxVersion)
    for (N, M, V) in [(1, 1, 3), (1, 5, 3), (1, 9, 3), (1, 3, 2), (2, 1, 1), (2, 2, 2), (3, 1, 2), (3, 2, 2), (3, 2, 3),
                      (4, 1, 2), (3, 3, 4), (4, 2, 3), (5, 1, 2), (2, 6, 7), (2, 3, 1), (4, 2, 2), (2, 3, 7), (4, 3, 4),
                      (2, 1, 0), (2, 2, 0), (2, 3, 0), (2, 5, 0)]:
        out.append(dict(model="AsyncIsr", n_replicas=N, log_size=M, max_leader_epoch=V))
    out.append(dict(model="Kip320", n_replicas=3, log_size=6, max_records=6, max_leader_epoch=3))  # 6.45 G states: the -fp128 run
    out.append(dict(model="Kip320", n_replicas=3, log_size=7, max_records=7, max_leader_epoch=2))  # 974 M states, exact Oracle-O pin
    for name, c in BASELINE_CONFIGS.items():
        # (config 4, 7 brokers: 357 action instances, 10-word states — minutes of hiprtc time, but the -m gpu prefix test
        # needs it and a GPU box should not spend its minutes compiling)
        out.append({k: v for k, v in c.items() if k != "invariants"})
    seen, uniq = set(), []
    for c in out:
        key = tuple(sorted(c.items()))
        if key not in seen:
            seen.add(key)
            uniq.append(c)
    return uniq


def precompile_variants():
    """(config, environment) pairs `build()` also specialises: code objects that are only loaded under an environment
    switch — the second build of the differential self-check (KMC_VERIFY) and the fault-injection build the self-check
    tests run against (KMC_FAULT_DROP) — so that the GPU box spends none of its minutes in hiprtc."""
    small = dict(model="Kip320", n_replicas=3, log_size=2, max_records=2, max_leader_epoch=1)
    wide = dict(model="Kip279", n_replicas=5, log_size=1, max_records=1, max_leader_epoch=1)
    fault = {"KMC_JIT_DEFINES": "-DKMC_TUNING=1 -DKMC_FAULT_DROP=1"}
    golden = dict(model="Kip320", n_replicas=3, log_size=5, max_records=5, max_leader_epoch=2)
    return [(small, {"KMC_VERIFY": "1"}), (wide, {"KMC_VERIFY": "1"}), (golden, {"KMC_VERIFY": "1"}), (small, fault),
            (small, dict(fault, KMC_VERIFY="1")),
            (dict(model="AsyncIsr", n_replicas=3, log_size=2, max_leader_epoch=2), fault),
            (small, {"KMC_JIT_DEFINES": "-DKMC_TUNING=1 -DKMC_TEST_FP_BITS=10"})] + layout_variants() + symmetry_variants() + full_leaves_variants() + deferred_probe_variants()   # (collisions on demand for the wide-fingerprint test)


# The arrangements of the Kafka state vector (csrc/kmc_layout.h) and the two walks of k_expand's pass 2 that go with them:
# replica-major (one replica per word at the headline's constants, grouped elsewhere) + kind-major by default, tight +
# instance-major only on request.  tests/test_gpu_kind_major.py forces each form onto configurations that get another by
# default (KMC_LAYOUT=rm / rmg / tight).
KIND_MAJOR_SMALL = [(m, N, L, R, E) for m in KAFKA for (N, L, R, E) in [(3, 2, 2, 1), (2, 3, 3, 2)]] + [
    ("Kip320", 4, 2, 2, 1), ("Kip279", 5, 1, 1, 1), ("Kip101", 4, 2, 1, 1), ("Kip320", 3, 2, 2, 2),
    ("KafkaTruncateToHighWatermark", 3, 2, 2, 2), ("Kip101", 3, 2, 2, 2), ("Kip279", 3, 2, 2, 2), ("Kip320FirstTry", 3, 2, 2, 2)]
INSTANCE_MAJOR_SMALL = [(m, 3, 2, 2, 1) for m in KAFKA] + [("Kip320", 4, 2, 2, 1), ("Kip279", 5, 1, 1, 1), ("Kip320", 3, 2, 2, 2),
                                                              ("Kip279", 3, 2, 2, 2)]
INSTANCE_MAJOR_LARGE = [("Kip320", 3, 5, 5, 2), ("Kip279", 3, 5, 5, 2)]
GROUPED_LARGE = [("Kip320", 3, 5, 5, 2)]


# FULL leaves (k_expand's pass 2 under orbit counting at seven brokers: csrc/kmc_kafka.h, FULL_LEAVES) forced onto small
# configurations: tests/test_gpu_full_leaves.py
FULL_LEAVES_DEFINES = "-DKMC_FULL_LEAVES_MIN_INSTANCES=0 -DKMC_FULL_LEAVES_PLAIN=1"
FULL_LEAVES_SMALL = [(m, 3, 2, 2, 1) for m in KAFKA] + [("Kip320", 4, 2, 2, 1), ("Kip279", 5, 1, 1, 1), ("Kip320", 3, 2, 2, 2),
                                                         ("Kip320FirstTry", 2, 3, 3, 2)]
FULL_LEAVES_SYMMETRY = [(m, 3, 2, 2, 1) for m in KAFKA] + [("Kip320", 4, 2, 2, 1), ("Kip279", 5, 1, 1, 1), ("Kip320", 3, 2, 2, 2),
                                                            ("Kip279", 3, 2, 2, 2)]
FULL_LEAVES_TRACES = [("Kip279", 3, 2, 2, 2), ("Kip101", 3, 2, 2, 2)]


def full_leaves_variants():
    env = {"KMC_JIT_DEFINES": FULL_LEAVES_DEFINES}

    def c(t, **kw):
        return dict(model=t[0], n_replicas=t[1], log_size=t[2], max_records=t[3], max_leader_epoch=t[4], **kw)
    out = [(c(t), env) for t in sorted(set(FULL_LEAVES_SMALL + FULL_LEAVES_TRACES))]
    out += [(c(t, symmetry=True), env) for t in sorted(set(FULL_LEAVES_SYMMETRY + FULL_LEAVES_TRACES))]
    return out


# The deferred probe (the search's flush on states of >= 8 words: csrc/kmc_kernels.h) forced onto small configurations:
# tests/test_gpu_deferred_probe.py
DEFERRED_PROBE_DEFINES = "-DKMC_DEFER_MIN_WORDS=1"
DEFERRED_PROBE_SMALL = [(m, 3, 2, 2, 1) for m in KAFKA] + [("Kip320", 4, 2, 2, 1), ("Kip279", 5, 1, 1, 1), ("Kip320", 3, 2, 2, 2)]
DEFERRED_PROBE_SYMMETRY = [("Kip320", 3, 2, 2, 1), ("Kip279", 3, 2, 2, 2), ("Kip101", 4, 2, 1, 1), ("Kip279", 5, 1, 1, 1)]
DEFERRED_PROBE_TRACES = [("Kip279", 3, 2, 2, 2), ("Kip101", 3, 2, 2, 2)]


def deferred_probe_variants():
    env = {"KMC_JIT_DEFINES": DEFERRED_PROBE_DEFINES}

    def c(t, **kw):
        return dict(model=t[0], n_replicas=t[1], log_size=t[2], max_records=t[3], max_leader_epoch=t[4], **kw)
    out = [(c(t), env) for t in sorted(set(DEFERRED_PROBE_SMALL + DEFERRED_PROBE_TRACES))]
    out += [(c(t, symmetry=True), env) for t in sorted(set(DEFERRED_PROBE_SYMMETRY))]
    return out


# Symmetry reduction with orbit counting (CheckerConfig.symmetry; tests/test_gpu_symmetry.py, bench.py's orbit_counting leg)
SYMMETRY_KAFKA = [(m, N, L, R, E) for m in KAFKA
                  for (N, L, R, E) in [(2, 2, 2, 1), (3, 2, 2, 1), (3, 1, 1, 2), (2, 3, 3, 2), (3, 2, 2, 2), (4, 1, 1, 1)]] + [
    ("Kip320", 4, 2, 2, 1), ("Kip101", 4, 2, 1, 2), ("Kip279", 3, 2, 3, 2), ("Kip320FirstTry", 3, 3, 3, 1), ("Kip320", 3, 3, 3, 1),
    ("KafkaTruncateToHighWatermark", 3, 3, 3, 1), ("Kip320", 3, 6, 6, 2), ("Kip320", 3, 5, 5, 2), ("Kip320", 3, 6, 6, 3), ("Kip320", 3, 7, 7, 2),
    # every Kafka model at the headline's constants (the exact Oracle-O pins and the per-state differential, round 4)
    ("KafkaTruncateToHighWatermark", 3, 6, 6, 2), ("Kip101", 3, 6, 6, 2), ("Kip279", 3, 6, 6, 2), ("Kip320FirstTry", 3, 6, 6, 2),
    # five and six replicas: the adjacent-transposition walk instead of unrolled permutations (BASELINE config 4 among them)
    ("Kip279", 5, 1, 1, 1), ("Kip320", 5, 1, 1, 1), ("KafkaTruncateToHighWatermark", 6, 1, 1, 1), ("Kip101", 5, 2, 1, 1),
    ("Kip279", 5, 2, 2, 1),
    ("Kip279", 5, 4, 4, 3),   # config 4 at SURVEY's sizing: the per-state fixture under orbit counting (tests/test_gpu_oracle_r_successors.py)
    # seven replicas, BASELINE config 5 among them
    ("Kip320", 7, 1, 1, 0), ("Kip279", 7, 1, 1, 0), ("Kip320", 7, 8, 8, 3)]
SYMMETRY_LAYOUTS = [("Kip320", 3, 2, 2, 2), ("Kip279", 3, 2, 2, 1), ("Kip101", 4, 2, 1, 1)]
SYMMETRY_FRL = [(2, 4, 2), (3, 2, 2), (2, 4, 4), (4, 2, 1), (5, 1, 2)]


def symmetry_variants():
    def c(t):
        return dict(model=t[0], n_replicas=t[1], log_size=t[2], max_records=t[3], max_leader_epoch=t[4], symmetry=True)
    return ([(c(t), {}) for t in SYMMETRY_KAFKA] +
            [(c(t), {"KMC_LAYOUT": lay}) for t in SYMMETRY_LAYOUTS for lay in ("tight", "rm", "rmg")] +
            [(dict(model="FiniteReplicatedLog", n_replicas=N, log_size=L, n_log_records=K, symmetry=True), {})
             for (N, L, K) in SYMMETRY_FRL])


def layout_variants():
    def c(t):
        return dict(model=t[0], n_replicas=t[1], log_size=t[2], max_records=t[3], max_leader_epoch=t[4])
    return ([(c(t), {"KMC_LAYOUT": "rm"}) for t in KIND_MAJOR_SMALL] +
            [(c(t), {"KMC_LAYOUT": "tight"}) for t in INSTANCE_MAJOR_SMALL + INSTANCE_MAJOR_LARGE] +
            [(c(t), {"KMC_LAYOUT": "rmg"}) for t in GROUPED_LARGE])


def all_precompile_configs():
    return [CheckerConfig(**c) for c in precompile_list()]
