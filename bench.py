#!/usr/bin/env python3
"""bench.py — headline benchmark: exhaustive BFS model checking of the Kafka replication spec.

A "step" is one complete exhaustive check (Init -> fixpoint) of the headline configuration
(BASELINE.json configs[2], bound as SURVEY §8d says because KafkaReplication.tla has no Next):

    root module Kip320, Replicas = {b1,b2,b3}, LogSize = 6, MaxRecords = 6, MaxLeaderEpoch = 2,
    INVARIANTS TypeOk WeakIsr StrongIsr, CHECK_DEADLOCK FALSE, no symmetry.

MaxRecords / MaxLeaderEpoch are not pinned by BASELINE.json; the C oracle exhausts this
binding (279,753,922 distinct states, tests/golden/oracle_kip320_3_6_6_2.json), which is what
the GPU count is checked against in the same run.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N = 1: the whole search runs on one GPU inside libkmc.so (kmc_run).
N > 1 (launched by torch.distributed.run, one rank per GPU): the fingerprint space is
hash-partitioned across ranks and every BFS level exchanges successors with one RCCL
all-to-all (kafka_specification_amd/sharded.py); total work is fixed => "scaling": "strong".

Prints ONE JSON line (rank 0).  `value` = distinct states per second over the whole job.
The oracle (oracle/) is used only as the cpu_baseline leg and to check the count.
"""
import argparse
import json
import os
import sys
import time

if os.environ.get("KMC_NO_TORCH", "0") != "1":
    import torch  # first: the process must hold ONE HIP runtime (see kafka_specification_amd/_native.py)
else:
    torch = None  # profiling runs on the system ROCm, without the wheel's bundled runtime

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_BPS = 8.0e12  # MI355X HBM3E peak (MI355X_MICROARCH.md)


def device_source_sha256():
    """Identity of the device code the numbers belong to (profiles record it; a stale profile is not quoted)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("kmc_layout.h", "kmc_common.h", "kmc_models_small.h", "kmc_kafka.h", "kmc_symm.h", "kmc_sink.h", "kmc_kernels.h"):
        h.update(open(os.path.join(ROOT, "kafka_specification_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def kernel_code_sha256_of(c, symmetry=False):
    """Identity of the MACHINE CODE of the kernels configuration c runs (kmc.kernel_code_sha256: .text + kernel descriptors +
    metadata of the cached code object this very run loads).  kmc_device.h also holds the orbit-counting, verify and
    profiling builds behind #if: an edit there changes the source hash and leaves a workload's instructions as they were."""
    try:
        import kafka_specification_amd as kmc
        return kmc.kernel_code_sha256(kmc.CheckerConfig(**c, symmetry=symmetry))
    except Exception as e:   # no hiprtc, no library: the source hash alone decides
        sys.stderr.write(f"bench.py: kernel_code_sha256 unavailable ({str(e)[:120]})\n")
        return None


def kernel_compiler_of(c, symmetry=False):
    """Which compiler wrote the code object this run loads — read from the object's FILE NAME (`...-c<HIP runtime build of the
    compiler's bundle>.hsaco`: two compilers can never write the same file) — and which one is pinned (csrc/kmc_engine_internal.h,
    KMC_PINNED_COMPILER; profiles/r06_compiler_ab.txt holds the A/B behind the choice)."""
    import re
    try:
        import kafka_specification_amd as kmc
        path = kmc.code_object_path(kmc.CheckerConfig(**{k: v for k, v in c.items() if k != "max_levels"}, symmetry=symmetry))
        m = re.search(r"-c(\d+)(?:-sharded|-enum)?\.hsaco$", path)
        return {"hip_runtime_version_of_the_compiler": int(m.group(1)) if m else None, "file": os.path.basename(path),
                "pinned": kmc.compiler_identity(1), "this_process": kmc.compiler_identity(0)}
    except Exception:
        return None


def headline_kernel_code_sha256(symmetry=False):
    return kernel_code_sha256_of(headline_config(), symmetry)


def newest_profile(suffix):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    return files[-1] if files else None


def randbench_rates():
    """The chip's random-access ceilings for the seen-set's pattern, read from the newest committed
    profiles/rNN_randbench.txt (tools/membench/randbench: uniformly random 8-byte accesses over an 8 GiB table).
    -> ({mode: best G accesses/s}, file) ; nothing is hard-coded here that the file could contradict."""
    import re
    path = newest_profile("randbench.txt")
    rates = {}
    if path:
        for line in open(path):
            m = re.match(r"mode (\d+).*=\s*([0-9.]+) G/s", line)
            if m:
                k, v = int(m.group(1)), float(m.group(2)) * 1e9
                rates[k] = max(rates.get(k, 0.0), v)
    return rates, (os.path.relpath(path, ROOT) if path else None)


def calibrated_bytes_per_access(mode):
    """(DRAM bytes per random access of randbench `mode`, file) from the newest profiles/rNN_dram_bytes_per_access.txt:
    32 bytes x (TCC_EA0_RDREQ_DRAM_32B + TCC_EA0_WRREQ_WRITE_DRAM_32B + TCC_EA0_WRREQ_ATOMIC_DRAM_32B) per access, the
    gfx950 counters that say how many bytes a request really moved (FETCH_SIZE tallies a 128-byte request as 64), or None."""
    import re
    path = newest_profile("dram_bytes_per_access.txt")
    if not path:
        return None
    tot, seen = 0.0, set()
    for line in open(path):
        m = re.match(r"(TCC_EA0_RDREQ_DRAM_32B_sum|TCC_EA0_WRREQ_WRITE_DRAM_32B_sum|TCC_EA0_WRREQ_ATOMIC_DRAM_32B_sum)\s+"
                     r"mode (\d+).*?([0-9.]+) req/access", line)
        if m and int(m.group(2)) == mode:
            tot += 32.0 * float(m.group(3))
            seen.add(m.group(1))
    return (tot, os.path.relpath(path, ROOT)) if len(seen) == 3 else None


def claims_per_distinct_state():
    """(atomic requests per distinct state of the headline run, file) from the newest profiles/rNN_summary.json
    (tools/summarize_profile.py: TCC_EA0_ATOMIC_sum over the states the profiled run found), or (None, None)."""
    path = newest_profile("summary.json")
    if not path:
        return None, None
    try:
        v = json.load(open(path)).get("derived", {}).get("claims_per_distinct_state")
    except Exception:
        return None, None
    return (float(v), os.path.relpath(path, ROOT)) if v else (None, None)


def measured_traffic(code=None, level_budget=None, depth=None):
    """HBM bytes per k_expand launch from a committed PMC summary (profiles/rNN_*pmc_summary.json, newest round first) — only
    from one that was measured on the MACHINE CODE this run executes (`code` = kernel_code_sha256 of the running workload's
    kernels; the summaries carry the hash of the kernels they profiled) AND over the same search (the same level budget and
    the same depth: a per-launch average over ten geometrically growing levels says nothing about a run
    of fourteen); otherwise null, with the reason."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*pmc_summary.json")), reverse=True)
    if not files:
        return None, "no PMC summary under profiles/"
    if not code:
        return None, "kernel_code_sha256 of the running kernels unavailable: no PMC summary can be matched"
    seen, near = [], []     # near: the same machine code, measured over another search — said first
    for path in files:
        try:
            j = json.load(open(path))
        except Exception as e:
            seen.append(f"{os.path.basename(path)}: {str(e)[:40]}")
            continue
        if j.get("kernel_code_sha256") == code and j.get("hbm_bytes_per_launch"):
            run = j.get("run", {})
            if (run.get("level_budget") or None) != (level_budget or None) or (depth and run.get("depth") and run["depth"] != depth):
                near.append(f"{os.path.basename(path)}: same code, another search (level budget {run.get('level_budget')}, "
                            f"depth {run.get('depth')}; this run {level_budget}, {depth})")
                continue
            return j["hbm_bytes_per_launch"], (f"{os.path.relpath(path, ROOT)} (measured on this machine code: "
                                               f"kernel_code_sha256 {code[:16]})")
        seen.append(f"{os.path.basename(path)} {str(j.get('kernel_code_sha256'))[:12]}")
    return None, f"no PMC summary was measured on this code ({code[:12]}...) over this search: " + ", ".join(near + seen[:6])


def device_info():
    """Clocks / power state of the box, for the record: identical code has measured 37 ms and 61 ms per
    step on different boxes of the pool (every variant of a sweep equally), so a slow number needs context."""
    import subprocess
    info = {}
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showperflevel", "--json"],
                             capture_output=True, text=True, timeout=20).stdout
        card = next(iter(json.loads(out).values()))
        for k, v in card.items():
            kl = k.lower()
            if any(t in kl for t in ("sclk", "mclk", "fclk", "power", "performance level")):
                info[k] = v
    except Exception as e:  # no rocm-smi, no GPU, unexpected format: the bench does not depend on it
        info["unavailable"] = str(e)[:80]
    return info


def headline_config():
    from kafka_specification_amd.configs import HEADLINE
    return dict(HEADLINE)


def workload_name(c):
    return (f"{c['model']} N={c['n_replicas']} LogSize={c['log_size']} MaxRecords={c['max_records']} "
            f"MaxLeaderEpoch={c['max_leader_epoch']} inv={'+'.join(c['invariants'])} deadlock=off")


GOLDEN_TAG = {"Kip320": "kip320", "Kip279": "kip279", "Kip101": "kip101", "Kip320FirstTry": "kip320firsttry",
              "KafkaTruncateToHighWatermark": "thw"}


def expected_counts(c):
    """(counts, file) of the committed oracle fixture for this binding [and level budget]: the exact C oracle's
    (tests/golden/oracle_*.json) or Oracle-O's exact orbit search (orbit_*.json); None when there is none."""
    tag = GOLDEN_TAG.get(c["model"])
    if not tag:
        return None
    key = f"{tag}_{c['n_replicas']}_{c['log_size']}_{c['max_records']}_{c['max_leader_epoch']}"
    if c.get("max_levels"):
        key += f"_levels{c['max_levels']}"
    for prefix in ("oracle_", "orbit_"):
        p = os.path.join(ROOT, "tests", "golden", prefix + key + ".json")
        if os.path.exists(p):
            g = json.load(open(p))
            return dict(distinct=g["distinct"], generated=g["generated"], depth=g["depth"], levels=g.get("levels"),
                        file=os.path.relpath(p, ROOT))
    return None


def cold_start(c):
    """What a CLI user waits for (wall time-to-exhaustive): ONE fresh process of the native front end —
    kafka_specification_amd/tlc models/Kip320.tla — from exec to exit: loading libkmc.so and the cached code object, hipMalloc and
    first touch of the seen-set and the frontiers, the search, the verdict printed.  Two runs: as a user types it (traces
    kept: 8 more bytes per table slot) and with -notrace (what the timed steps above run).  Never part of `value`."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "kafka_specification_amd", "tlc")
    spec = os.path.join(ROOT, "models", c["model"] + ".tla")
    if c != headline_config() or not os.path.exists(exe):
        return None
    out = {"command": "kafka_specification_amd/tlc models/%s.tla -table %d -frontier %d -v [-notrace]" % (c["model"], 1 << 30, 1 << 26)}
    # -v: the front end's own account of its wall time (kmc_timing, include/kmc.h) — what is HIP's start-up, what the code
    # object, the allocation, the first touch of the seen-set, the search, the teardown
    pat = re.compile(r"Wall time: ([\d.]+)s in this process = ([\d.]+)s before kmc_open .* \+ ([\d.]+)s kmc_open \(HIP initialisation "
                     r"([\d.]+)s, code object ([\d.]+)s, allocation of ([\d.]+) GiB ([\d.]+)s\) \+ ([\d.]+)s kmc_run \(first clear of the "
                     r"seen-set ([\d.]+)s, search ([\d.]+)s\) \+ ([\d.]+)s verdict / trace \+ ([\d.]+)s teardown")
    for key, extra in (("wall_s", []), ("wall_s_notrace", ["-notrace"])):
        best, found, parts = None, None, None
        for _attempt in range(3):   # a fresh process each time; the smallest of three (the first pages the binaries in: the system
            #                           ROCm's libamdhip64, which a box whose Python processes only loaded PyTorch's bundled one has never read)
            t0 = time.perf_counter()
            try:
                p = subprocess.run([exe, spec, "-table", str(1 << 30), "-frontier", str(1 << 26), "-v"] + extra, capture_output=True,
                                   text=True, timeout=300)
            except Exception as e:   # the bench line does not depend on it
                out["error"] = str(e)[:120]
                break
            dt = time.perf_counter() - t0
            m = re.findall(r"(\d+) states generated, (\d+) distinct states found, (\d+) states left on queue", p.stdout)
            found = int(m[-1][1]) if m else None      # the closing line, not a progress line
            out["exit_code"] = p.returncode
            if best is None or dt < best:
                best = dt
                b = pat.search(p.stdout)
                if b:
                    v = [float(x) for x in b.groups()]
                    parts = {"in_process_s": v[0], "exec_and_load_s": max(dt - v[0], 0.0),   # fork/exec, the dynamic loader (libamdhip64), exit
                             "before_open_s": v[1], "open_s": v[2], "hip_init_s": v[3], "code_object_s": v[4],
                             "device_GiB": v[5], "alloc_s": v[6], "run_s": v[7], "first_clear_s": v[8], "search_s": v[9],
                             "verdict_s": v[10], "teardown_s": v[11]}
        out[key] = best
        out["breakdown" if key == "wall_s" else "breakdown_notrace"] = parts
        out["distinct_states" if key == "wall_s" else "distinct_states_notrace"] = found
    return out


def cpu_baseline(c, budget_states, total_states, whole="the reachable set"):
    """The C oracle (exact-state BFS, a port — TLC itself cannot run here) on all host cores,
    on a bounded prefix of the same workload: it stops after the BFS level that crosses
    `budget_states` distinct states."""
    import kmo
    # 32 threads is where the oracle peaks on the 256-thread GPU boxes (profiles/r02_oracle_scaling.txt: 17.6 M states/s
    # at 32, 14.5 at 64, 6.3 at 256 — first-touch page faults and the shared table then dominate).  Round 1 measured
    # 1.5 M/s at 32: its level-start phases ran on one thread and every new state took a shared fetch-add.
    threads = min(os.cpu_count() or 1, int(os.environ.get("KMC_CPU_THREADS", 32)))
    cfg = kmo.make_config(c["model"], N=c["n_replicas"], L=c["log_size"], R=c["max_records"],
                          E=c["max_leader_epoch"], invariants=c["invariants"], threads=threads,
                          max_states=budget_states)
    t0 = time.time()
    run = kmo.Run(cfg)
    dt = time.time() - t0
    out = dict(value=run.distinct / max(run.seconds, 1e-9), unit="distinct states/s", cores=threads, kind="port",
               sample=(f"first {run.depth} BFS levels of the same workload ({run.distinct} distinct states = "
                       f"{100.0 * run.distinct / max(total_states, 1):.1f} % of {whole}, {run.seconds:.1f} s) with "
                       f"oracle/kmc_oracle.c, {threads} threads; not TLC (no JVM on this box)"),
               seconds=round(dt, 2))
    run.close()
    return out


OPEN_S = []   # seconds every run_single handle of this process took to open, in order


def run_single(c, steps, warmup, symmetry=False, table=None, frontier=None, keep_trace=None):
    import kafka_specification_amd as kmc
    # (under symmetry the seen-set and the frontiers hold one state per orbit: a quarter of the slots keeps the same load)
    cfg = kmc.CheckerConfig(**c, device=0, symmetry=symmetry,
                            table_capacity=int(os.environ.get("KMC_BENCH_TABLE", table or ((3 << 27) if symmetry else (1 << 30)))),
                            frontier_capacity=int(os.environ.get("KMC_BENCH_FRONTIER", frontier or ((1 << 24) if symmetry else (1 << 26)))),
                            wide_fingerprint=os.environ.get("KMC_BENCH_FP128", "0") == "1",   # tuning: 128-bit entries
                            # (predecessors for counterexample traces, what both CLIs keep by default: the traces_kept leg; env: tuning)
                            keep_trace=(os.environ.get("KMC_BENCH_TRACE", "0") == "1") if keep_trace is None else keep_trace)
    results = []
    t_open = time.perf_counter()
    with kmc.ModelChecker(cfg) as mc:
        OPEN_S.append(time.perf_counter() - t_open)   # kmc_open: HIP start-up (a process's first handle), code object, allocation + spread
        for _ in range(warmup):
            mc.run()
        if torch is not None:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            results.append(mc.run())   # kmc_run returns after its stream has drained
        if torch is not None:
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return results, dt


def algorithmic_bytes_per_state(r):
    """A = 2*S + 8*g + 8 per distinct state (frontier read + write, g seen-set probes of 8 B, one 8-B claim).
    g is priced on generated - generated_repeats: TLC's "generated" counts a successor twice when two disjuncts of one binding
    hold at once (Kip320.tla:82-83, Kip279.tla:47-51); such a pair is ONE successor and one probe."""
    S = 8 * r.state_words
    probes = r.generated - getattr(r, "generated_repeats", 0)
    g = probes / max(r.distinct, 1)
    return 2 * S + 8 * g + 8, S, g, probes


# BASELINE.json configs[3] and configs[4] (SURVEY section 8d rows "config 4" and "config 5") on ONE GPU, in the driver's own run.
# (The frontier-sharded forms of the same searches are `--gpus N --workload ...`; nothing here claims a multi-GPU number.)
BASELINE_LEGS = {
    # Kip279.tla:53-62 at five brokers: exhaustible, 112,549,196 states (tests/golden/oracle_kip279_5_2_2_1.json, exact)
    "config4_kip279_5brokers": dict(
        c=dict(model="Kip279", n_replicas=5, log_size=2, max_records=2, max_leader_epoch=1, invariants=("TypeOk",)),
        table=1 << 30, frontier=1 << 26, cpu_states=0),     # cpu_baseline: the whole search (about ten seconds on 32 threads)
    # ... and at SURVEY section 8(a.0)'s own sizing of config 4 (LogSize 4, MaxRecords 4, MaxLeaderEpoch 3: four words per state),
    # where a log holds up to four epochs and Kip279's truncation (Kip279.tla:27-51) has three epochs in a log to look at: not
    # exhaustible — twelve BFS levels, 318,475,476 states (tests/golden/oracle_kip279_5_4_4_3_levels12.json, exact), "exhausted": false
    "config4_deep_kip279_5brokers_log4_levels12": dict(
        c=dict(model="Kip279", n_replicas=5, log_size=4, max_records=4, max_leader_epoch=3, invariants=("TypeOk",), max_levels=12),
        table=1 << 31, frontier=1 << 28, cpu_states=20_000_000),
    # Kip320.tla:150-159 at seven brokers, LogSize 8: nobody exhausts it — ten BFS levels, 197,561,008
    # states (tests/golden/oracle_kip320_7_8_8_3_levels10.json, exact), reported with "exhausted": false
    "config5_kip320_7brokers_levels10": dict(
        c=dict(model="Kip320", n_replicas=7, log_size=8, max_records=8, max_leader_epoch=3,
               invariants=("TypeOk", "WeakIsr", "StrongIsr"), max_levels=10),
        # (table: 1.75 x 2^30 slots — any multiple of 64 since round 6 — where k_expand + the clear of the table is smallest:
        # 2^31 as in round 5 pays 2.8 ms of clear, 2^30 pays 2 ms of longer probe chains; profiles/r06_config5.txt holds the A/B)
        table=int(os.environ.get("KMC_BENCH_TABLE5", 7 << 28)), frontier=1 << 29, cpu_states=40_000_000),
}


def step_breakdown(ms_per_step, kernel_s, inv_s, clear_s):
    """Where a step's wall time goes (HIP events on the engine's stream; the rest is host latency around the waits)."""
    return {"k_expand_ms": 1e3 * kernel_s, "k_inv_ms": 1e3 * inv_s, "clear_seen_set_ms": 1e3 * clear_s,
            "host_and_rest_ms": ms_per_step - 1e3 * (kernel_s + inv_s + clear_s)}


def baseline_leg(name, steps, warmup, with_cpu=True):
    """One BASELINE config beside the headline: the same measurement (whole kmc_run steps, inputs resident, k_expand time from HIP
    events on the engine's stream), its counts against the committed exact fixture, its own roofline block."""
    spec = BASELINE_LEGS[name]
    c = dict(spec["c"])
    try:
        results, dt = run_single(c, steps, warmup, table=spec["table"], frontier=spec["frontier"])
    except Exception as e:   # a leg never takes the headline line down with it
        return {"error": str(e)[:300]}
    r = results[-1]
    exp = expected_counts(c)
    A, S, g, probes = algorithmic_bytes_per_state(r)
    kernel_s = sum(x.seconds_expand for x in results) / len(results)
    inv_s = sum(x.seconds_inv for x in results) / len(results)
    clear_s = sum(x.seconds_clear for x in results) / len(results)
    code = kernel_code_sha256_of({k: v for k, v in c.items() if k != "max_levels"})
    cpu = None
    if with_cpu:
        try:   # the same search (or, under a level budget, a prefix of its levels) on the host cores: a port, not TLC
            cpu = cpu_baseline({k: v for k, v in c.items() if k != "max_levels"}, spec["cpu_states"], r.distinct,
                               whole="the reachable set" if not c.get("max_levels") else f"the {c['max_levels']}-level prefix")
        except Exception as e:
            cpu = {"error": str(e)[:200]}
    traffic, traffic_source = measured_traffic(code, c.get("max_levels"), r.depth)
    achieved = A * r.distinct / max(kernel_s, 1e-12)
    return {
        "workload": workload_name(c), "state_bytes": S, "level_budget": c.get("max_levels"),
        "exhausted": r.verdict != "level_limit", "verdict": r.verdict,
        "value": sum(x.distinct for x in results) / dt, "unit": "distinct states/s", "steps": steps, "warmup": warmup,
        "ms_per_step": 1e3 * dt / steps, "time_s": dt / steps,
        "distinct_states": r.distinct, "states_generated": r.generated, "seen_set_probes": probes, "depth": r.depth,
        "matches_oracle_golden": None if exp is None else (r.distinct == exp["distinct"] and r.generated == exp["generated"]
                                                           and r.depth == exp["depth"]
                                                           and (exp["levels"] is None or list(r.levels) == list(exp["levels"]))),
        "oracle_golden": exp["file"] if exp else None,
        "kernel_seconds_per_step": kernel_s, "launches_per_step": r.expand_launches,
        "step_breakdown": step_breakdown(1e3 * dt / steps, kernel_s, inv_s, clear_s),
        "table_slots": r.table_capacity, "table_load_at_end": r.distinct / max(r.table_capacity, 1),
        "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK_BPS / 1e9, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_BPS, "traffic": traffic, "traffic_source": traffic_source,
                     "algorithmic_bytes_per_distinct_state": A, "kernel": "kmc_expand_*", "kernel_code_sha256": code},
        "k_inv": inv_pass_block(c, r, inv_s, S, code),
        "cpu_baseline": cpu,
    }


def inv_pass_block(c, r, inv_s, S, code):
    """The invariant pass over the last, unexpanded frontier of a level-budgeted search (k_inv): a pure streaming read of that
    level's states — its own roofline block (algorithmic bytes = S per state of the level) and, when a committed PMC summary
    was measured on this machine code, its DRAM traffic."""
    if not c.get("max_levels") or not inv_s or not r.levels:
        return None
    n = r.levels[-1]
    alg = float(S) * n
    traffic, src = None, "no kmc_inv row in a PMC summary of this machine code"
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*summary.json")), reverse=True):
        try:
            j = json.load(open(path))
        except Exception:
            continue
        row = j.get("kmc_inv")
        if row and j.get("kernel_code_sha256") == code and row.get("hbm_bytes_per_launch"):
            traffic, src = row["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
            break
    return {"kernel": "kmc_inv_*", "states": n, "seconds": inv_s, "bound": "hbm", "achieved": alg / inv_s / 1e9,
            "peak": HBM_PEAK_BPS / 1e9, "unit": "GB/s", "frac": alg / inv_s / HBM_PEAK_BPS, "algorithmic_bytes": alg,
            "traffic": traffic, "traffic_source": src}


STRETCH = dict(model="Kip320", n_replicas=3, log_size=6, max_records=6, max_leader_epoch=3, invariants=("TypeOk", "WeakIsr", "StrongIsr"))


def stretch_1gpu_leg():
    """The one exhaustible workload that needs the HBM north_star talks about, on ONE GPU, in the driver's own run: Kip320 3/6/6/3
    (Kip320.tla:150-159 with MaxLeaderEpoch = 3, KafkaReplication.tla:36) — 6,452,700,520 distinct states, 20.76 G generated, 54
    levels — with 128-bit seen-set entries (the 64-bit table loses one state to the collision n^2 / 2^65 = 1.1 predicts) in a table
    sized to the HBM there is: 15 G slots of 16 bytes = 224 GiB, load 0.43 at the end (the largest power of two that fits, 2^33 =
    128 GiB, ends at load 0.75 and takes 1.22 s instead of 0.82: profiles/r06_stretch.txt), and two frontiers of 6.0e8 states (27 GiB;
    the widest level holds 521,281,965).  One search, counts held level by level to Oracle-O's exact
    fixture (tests/golden/orbit_kip320_3_6_6_3.json: no fingerprint anywhere).  Roofline: unit = distinct state, algorithmic bytes
    A = 2*S + 16*g + 16 (the probe reads a 16-byte entry, the claim writes one).  Never part of `value`."""
    import kafka_specification_amd as kmc
    c = dict(STRETCH)
    try:
        cfg = kmc.CheckerConfig(**c, device=0, wide_fingerprint=True,
                                table_capacity=int(os.environ.get("KMC_BENCH_STRETCH_TABLE", 15_000_000_000)),
                                frontier_capacity=int(os.environ.get("KMC_BENCH_STRETCH_FRONTIER", 600_000_000)))
        t_open = time.perf_counter()
        with kmc.ModelChecker(cfg) as mc:
            open_s = time.perf_counter() - t_open
            t0 = time.perf_counter()
            r = mc.run()
            first_wall = time.perf_counter() - t0     # includes the first touch of 128 GiB of freshly mapped HBM (timing.first_clear_s)
            timing = mc.timing()
            t0 = time.perf_counter()
            r = mc.run()                              # the timed step: memory mapped, as every other leg's steps
            dt = time.perf_counter() - t0
            levels = mc.level_stats()
    except Exception as e:   # a leg never takes the headline line down with it
        return {"workload": workload_name(c), "error": f"{type(e).__name__}: {str(e)[:300]}"}
    exp = expected_counts(c)
    S = 8 * r.state_words
    probes = r.generated - r.generated_repeats
    g = probes / max(r.distinct, 1)
    A = 2 * S + 16 * g + 16
    achieved = A * r.distinct / max(r.seconds_expand, 1e-12)
    code = kernel_code_sha256_of(c)
    traffic, traffic_source = None, "no PMC summary of this machine code with 128-bit entries"
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_stretch*pmc_summary.json")), reverse=True):
        try:
            j = json.load(open(path))
        except Exception:
            continue
        if j.get("kernel_code_sha256") == code and j.get("hbm_bytes_per_launch"):
            traffic, traffic_source = j["hbm_bytes_per_launch"], f"{os.path.relpath(path, ROOT)} (measured on this machine code)"
            break
    # the widest levels carry the run: their probe rate against the microbenchmark of the same footprint
    big = sorted(levels, key=lambda st: -st["probes"])[:8]
    big_probes, big_ms = sum(st["probes"] for st in big), sum(st["expand_ms"] for st in big)
    rates, rates_file = randbench_rates_at(33)
    return {
        "workload": workload_name(c), "entries": "128-bit (fingerprint + check word)", "table_slots": r.table_capacity,
        "table_GiB": r.table_capacity * 16 / 2 ** 30, "frontier_states": r.frontier_capacity, "table_load_at_end": r.distinct / r.table_capacity,
        "steps": 1, "warmup": 1, "time_to_exhaustive_s": dt, "ms_per_step": 1e3 * dt,
        "first_run_wall_s": first_wall, "open_s": open_s, "first_clear_s": timing.get("first_clear_s"),
        "device_GiB": timing.get("device_bytes", 0) / 2 ** 30,
        "value": r.distinct / dt, "unit": "distinct states/s",
        "distinct_states": r.distinct, "states_generated": r.generated, "seen_set_probes": probes, "depth": r.depth, "verdict": r.verdict,
        "matches_oracle_golden": None if exp is None else (r.distinct == exp["distinct"] and r.generated == exp["generated"]
                                                           and r.depth == exp["depth"]
                                                           and (exp["levels"] is None or list(r.levels) == list(exp["levels"]))),
        "oracle_golden": exp["file"] if exp else None,
        "step_breakdown": step_breakdown(1e3 * dt, r.seconds_expand, r.seconds_inv, r.seconds_clear),
        "kernel_seconds_per_step": r.seconds_expand, "launches_per_step": r.expand_launches,
        "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK_BPS / 1e9, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_BPS, "traffic": traffic, "traffic_source": traffic_source,
                     "algorithmic_bytes_per_distinct_state": A, "kernel": "kmc_expand_*", "kernel_code_sha256": code,
                     "probes_per_s": probes / max(r.seconds_expand, 1e-12),
                     "probes_per_s_widest_levels": 1e3 * big_probes / max(big_ms, 1e-9),
                     "randbench_at_this_footprint": {"mode_13_wide_mix_per_s": rates.get(13), "mode_1_loads_per_s": rates.get(1),
                                                     "mode_7_narrow_mix_per_s": rates.get(7), "source": rates_file},
                     "ratio_to_randbench_wide_mix": (1e3 * big_probes / max(big_ms, 1e-9)) / rates[13] if rates.get(13) else None},
    }


def randbench_rates_at(log2_slots):
    """{mode: G accesses/s -> accesses/s} of tools/membench/randbench at a footprint of 2^log2_slots 8-byte slots, from the newest
    committed profiles/rNN_randbench_sweep.txt that holds that footprint; ({}, None) when there is none."""
    import glob
    import re
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_randbench_sweep.txt")), reverse=True):
        rates, on = {}, False
        for line in open(path):
            m = re.match(r"# table of 2\^(\d+) slots", line)
            if m:
                on = int(m.group(1)) == log2_slots
                continue
            m = re.match(r"mode (\d+).*=\s*([0-9.]+) G/s", line)
            if m and on:
                k, v = int(m.group(1)), float(m.group(2)) * 1e9
                rates[k] = max(rates.get(k, 0.0), v)
        if rates:
            return rates, os.path.relpath(path, ROOT)
    return {}, None


def stretch_leg(a):
    """`--gpus N --stretch [MODEL,N,L,R,E]`: the workload the frontier sharding is designed for, beside the strong-scaling headline
    (default Kip320 3/6/6/3: 6,452,700,520 states, 128-bit entries; DESIGN.md section 6 holds the projection its curve is read
    against).  Every rank takes part (the exchange is collective); a failure is reported, it never takes the headline line down."""
    from kafka_specification_amd.sharded import bench_sharded
    m, n, l, rr, e = a.stretch.split(",")
    sc = dict(model=m, n_replicas=int(n), log_size=int(l), max_records=int(rr), max_leader_epoch=int(e),
              invariants=("TypeOk", "WeakIsr", "StrongIsr") if m == "Kip320" else ("TypeOk",))
    try:
        # capacities of the whole job (tools/loopback_stretch.py ran these on 8 logical shards): 2^33 table slots, 1.25 x 2^30
        # frontier states, 2^28 send records
        sres, sdt, sextra = bench_sharded(sc, 1, 0, backend=a.backend, wide_fingerprint=True,
                                          capacities=(1 << 33, (1 << 30) * 5 // 4, 1 << 28))
        x = sres[-1]
        kexp = expected_counts(sc)      # tests/golden/orbit_kip320_3_6_6_3.json: Oracle-O's exact search
        known = (kexp["distinct"], kexp["generated"], kexp["depth"]) if kexp else None
        return {"workload": workload_name(sc), "entries": "128-bit (fingerprint + check word)", "steps": 1,
                "time_to_exhaustive_s": sdt, "value": x.distinct / sdt, "unit": "distinct states/s",
                "distinct_states": x.distinct, "states_generated": x.generated, "depth": x.depth, "verdict": x.verdict,
                "matches_the_exact_oracle": None if known is None else (x.distinct, x.generated, x.depth) == known,
                **{k: v for k, v in sextra.items() if k != "per_rank"},
                "expand_kernel_seconds_max_rank": max(pr["expand_kernel_seconds_last_step"] for pr in sextra["per_rank"]),
                "exchange_bytes": sum(pr["received_bytes_last_step"] for pr in sextra["per_rank"])}
    except Exception as ex_:   # noqa: BLE001
        return {"workload": a.stretch, "error": f"{type(ex_).__name__}: {str(ex_)[:300]}"}


def self_launch(n):
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


# KMC_SEEN_SET_SPREAD (csrc/kmc_engine_core.cpp: seen_set_alloc) lays the seen-set's chunks over that many times their size of the HBM:
# a table of physically scattered chunks runs at the fast level in every process (k_expand 28.4 - 29.2 ms), a table of chunks as they
# come at the level its place in the HBM has (28.4 - 31.7: profiles/r06_chunked_seen_set.txt items 9 - 12).  It costs seconds per handle
# at open and close, so it is not the library's default and this line does not set it either: the line measures the product as a user
# gets it.  What the environment asked for is reported (config.seen_set_spread, open_s).


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-states", type=int, default=0,
                    help="cpu_baseline sample size in distinct states (default: 25 %% of the workload's reachable set; the "
                         "oracle stops after the BFS level that crosses it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--small", action="store_true", help="debug: a small configuration instead of the headline")
    ap.add_argument("--workload", default=None, metavar="MODEL,N,L,R,E",
                    help="debug / tests: another Kafka-family binding instead of the headline (never a bench line)")
    ap.add_argument("--level-budget", type=int, default=0, metavar="LEVELS",
                    help="stop after LEVELS BFS levels and report states/s over that budget with \"exhausted\": false "
                         "(SURVEY section 8d for configurations nobody can exhaust: BASELINE config 5 = --workload Kip320,7,8,8,3)")
    ap.add_argument("--symmetry", action="store_true",
                    help="the timed region runs the orbit-counting search (kmc_config.symmetry) instead of the plain one: for "
                         "profiling that kernel; the default line times the plain search and reports orbit counting beside it")
    ap.add_argument("--no-orbit-counting", action="store_true", help="skip the orbit_counting leg of the default line")
    ap.add_argument("--no-traces-leg", action="store_true", help="skip the traces_kept leg (the headline with predecessors kept, as a CLI run keeps them)")
    ap.add_argument("--no-cold-start", action="store_true", help="skip the cold_start leg (one fresh CLI process, exec to exit)")
    ap.add_argument("--no-baseline-configs", action="store_true",
                    help="skip the baseline_configs legs (BASELINE.json configs 4 and 5 on one GPU, beside the headline)")
    ap.add_argument("--no-stretch", action="store_true",
                    help="skip the stretch_1gpu leg (Kip320 3/6/6/3, 6,452,700,520 states, 128-bit entries in a 128 GiB seen-set: the "
                         "one exhaustible workload that uses the HBM)")
    ap.add_argument("--config-steps", type=int, default=3, help="timed steps of each baseline_configs leg (1 warmup before)")
    ap.add_argument("--stretch", nargs="?", const="Kip320,3,6,6,3", default=None, metavar="MODEL,N,L,R,E",
                    help="N > 1 only: one more leg beside the strong-scaling headline — the workload the frontier sharding is "
                         "designed for (default Kip320 3/6/6/3: 6,452,700,520 states, 128-bit entries, seconds on one GPU; "
                         "DESIGN.md section 6 holds the projection its curve is read against), one step, never part of `value`")
    ap.add_argument("--backend", default=os.environ.get("KMC_BENCH_BACKEND", "nccl"), choices=("nccl", "gloo"),
                    help="process-group backend of the N>1 leg: nccl (= RCCL, the product) or gloo (CPU launch-path test)")
    a = ap.parse_args()

    if a.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher — one rank per GPU under torch.distributed.run
        # on this node (the driver's own invocation shape), and pass its exit code on.  Rank 0 prints the line.
        sys.exit(self_launch(a.gpus))

    from kafka_specification_amd._native import KMC_SYMMETRY_MAX_REPLICAS
    c = headline_config()
    if a.small:
        c.update(log_size=3, max_records=3)
    if a.workload:
        m, n, l, r, e = a.workload.split(",")
        c.update(model=m, n_replicas=int(n), log_size=int(l), max_records=int(r), max_leader_epoch=int(e))
        if m != "Kip320":
            c["invariants"] = ("TypeOk",)
    if a.level_budget:
        c["max_levels"] = a.level_budget
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))

    if a.gpus > 1 or world > 1:
        from kafka_specification_amd.sharded import bench_sharded
        if world != a.gpus:
            raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}")
        results, dt, extra = bench_sharded(c, a.steps, a.warmup, backend=a.backend, symmetry=a.symmetry)
        scaling, parallelism = "strong", (f"fingerprint-sharded x{world}, all-to-all per BFS level" +
                                          (", orbit counting over the permutations of Replicas" if a.symmetry else ""))
    else:
        results, dt = run_single(c, a.steps, a.warmup, symmetry=a.symmetry)
        extra = {}
        scaling, parallelism = "strong", "1 GPU" + (", orbit counting over the permutations of Replicas" if a.symmetry else "")
    want_stretch = (a.gpus > 1 or world > 1) and a.stretch
    if rank != 0:
        if want_stretch:
            stretch_leg(a)      # (collective: every rank takes part; engines are opened under a cross-rank agreement, sharded.py)
        return

    r = results[-1]
    distinct, generated = r.distinct, r.generated
    exp = expected_counts(c)
    counts_match = None if exp is None else (distinct == exp["distinct"] and generated == exp["generated"]
                                             and r.depth == exp["depth"]
                                             and (exp["levels"] is None or list(r.levels) == list(exp["levels"])))
    total_states = sum(x.distinct for x in results)
    value = total_states / dt
    S = 8 * r.state_words
    # g = seen-set probes per distinct state.  TLC's "generated" (r.generated) counts a successor twice when two disjuncts
    # of one binding hold at once (Kip320.tla:82-83: 13.8 M of the headline's 901.9 M); such a pair is ONE successor and
    # one probe, so the algorithmic bytes are priced on generated - generated_repeats (the smaller, honest figure).
    probes = generated - getattr(r, "generated_repeats", 0)
    g = probes / max(distinct, 1)
    alg_bytes_per_state = 2 * S + 8 * g + 8          # frontier read+write, g probes, 1 claim
    kernel_s = sum(x.seconds_expand for x in results) / len(results)   # N > 1: the slowest rank's (run_sharded takes the max)
    launches = r.expand_launches
    achieved = alg_bytes_per_state * distinct / max(kernel_s, 1e-12)
    code_sha = kernel_code_sha256_of(c, symmetry=a.symmetry)
    if world == 1:   # (a level budget runs the same kernels over fewer launches; the summary's per-launch figure is then of
        #               the profiled run's own budget — the summaries name theirs in "run")
        traffic, traffic_source = measured_traffic(code_sha, a.level_budget, r.depth)
    else:
        traffic, traffic_source = None, "PMC counters are collected for single-GPU searches only"
    # Secondary view — the seen-set's probes are uniformly random 8-byte accesses, which this memory system serves
    # at a fraction of its streaming rate.  The ceilings come from tools/membench/randbench (profiles/): loads =
    # mode 1, claims = mode 3 (a load, then a CAS on the slot when it was empty: the claim sequence itself).
    # The two streams overlap in the kernel, so the bound is the larger of the two times, not their sum (round 1
    # quoted the sum with a constant the profile contradicted; it exceeded the kernel's own time).
    rates, rates_file = randbench_rates()
    cpd, cpd_file = claims_per_distinct_state()   # one CAS per new state plus the lost races, as the counters saw them
    claims = (cpd or 0.0) * distinct
    random_access = None
    if rates.get(1) and rates.get(3) and cpd:
        lb = max(probes / rates[1], claims / rates[3])
        random_access = {
            "probe_loads_per_s": probes / max(kernel_s, 1e-12), "probe_loads_per_s_ceiling": rates[1],
            "claims_per_s": claims / max(kernel_s, 1e-12), "claims_per_s_ceiling": rates[3],
            "lower_bound_s": lb, "frac_of_lower_bound": min(1.0, lb / max(kernel_s, 1e-12)),
            "source": rates_file + " (modes 1 and 3); claims per distinct state " + f"{cpd:.4f} from {cpd_file}"}
        # What actually bounds the kernel: HBM traffic of random accesses.  Every random 8-byte probe fills a whole
        # 128-byte line from DRAM (profiles/r02_request_size.txt).  The measured DRAM bytes of the run (PMC, same device
        # code) over the kernel time, against what the memory system sustains for the seen-set's own access mix —
        # randbench mode 7 (a random load, and a CAS on that slot for 35 % of the accesses; mode 3, a CAS whenever the slot
        # was empty, when no mode-7 measurement is committed): rate x DRAM bytes per access under the same counters.
        mix = 7 if rates.get(7) and calibrated_bytes_per_access(7) else 3   # mode 7: a load + a CAS for 35 % = this kernel's mix
        per_access = calibrated_bytes_per_access(mix)
        if traffic and per_access:
            dram_bps = traffic * launches / max(kernel_s, 1e-12)
            ceiling = rates[mix] * per_access[0]
            random_access.update({
                "dram_traffic_GBps": dram_bps / 1e9, "dram_traffic_frac_of_hbm_peak": dram_bps / HBM_PEAK_BPS,
                "randbench_same_mix_GBps": ceiling / 1e9,
                "ratio_to_randbench_same_mix": dram_bps / ceiling,   # ~1: a microbenchmark of the same mix, not a hard bound
                "randbench_source": f"{rates_file} mode {mix} = {rates[mix] / 1e9:.1f} G accesses/s x {per_access[0]:.1f} "
                                         f"DRAM B/access ({per_access[1]})"})
    # N > 1: every rank's own k_expand time against the algorithmic bytes of the states it owns, and what the exchange moved
    per_rank = extra.pop("per_rank", None) if isinstance(extra, dict) else None
    if per_rank:
        for pr in per_rank:
            ks = max(pr["expand_kernel_seconds_last_step"], 1e-12)
            pr["achieved_GBps"] = alg_bytes_per_state * pr["states_owned"] / ks / 1e9
            pr["frac"] = pr["achieved_GBps"] * 1e9 / HBM_PEAK_BPS
        extra["exchange_bytes_per_step"] = sum(pr["received_bytes_last_step"] for pr in per_rank)
    out = {
        "metric": "distinct states/sec + time-to-exhaustive, KafkaReplication 3-broker",
        "value": value, "unit": "distinct states/s", "n_gpus": max(a.gpus, world), "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "u64", "data": "synthetic (fully determined by model + constants; hash seed 0)",
        "config": {"workload": workload_name(c), "parallelism": parallelism, "state_bytes": S,
                   "seen_set_spread": int(os.environ.get("KMC_SEEN_SET_SPREAD", "1")),   # set-up: the table's chunks lie over this many times its size (1: as they come)
                   "open_s": OPEN_S[0] if OPEN_S else None,   # what opening the headline's handle took (HIP start-up, code object, allocation)
                   "distinct_states": distinct, "states_generated": generated, "seen_set_probes": probes, "depth": r.depth,
                   "verdict": r.verdict, "matches_oracle_golden": counts_match,
                   "exhausted": r.verdict != "level_limit", "level_budget": a.level_budget or None,
                   "stored_states": r.orbit_representatives if a.symmetry else None,
                   "time_to_exhaustive_s": dt / a.steps,
                   "step_breakdown": step_breakdown(1e3 * dt / a.steps, kernel_s, sum(x.seconds_inv for x in results) / len(results),
                                                    sum(x.seconds_clear for x in results) / len(results)) if world == 1 else None,
                   **extra},
        "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK_BPS / 1e9, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_BPS, "traffic": traffic,
                     "kernel": "kmc_expand_*", "kernel_seconds_per_step": kernel_s, "launches_per_step": launches,
                     "algorithmic_bytes_per_launch": alg_bytes_per_state * distinct / max(launches, 1),
                     "algorithmic_bytes_per_distinct_state": alg_bytes_per_state,
                     # SURVEY §8d asks for the granular figure beside the algorithmic one: what the same accesses cost
                     # at the memory's granularity — a 128-byte line fill per probe (not the 64-byte sector the survey
                     # assumed: profiles/r02_request_size.txt), a 64-byte atomic request per claim attempt
                     "line_granular_bytes_per_distinct_state": (2 * S + 128 * g + 64 * cpd) if cpd else None,
                     "line_granular_GBps": ((2 * S + 128 * g + 64 * cpd) * distinct / max(kernel_s, 1e-12) / 1e9) if cpd else None,
                     "useful_fraction_ceiling_of_a_probe": 8.0 / 128.0,
                     "traffic_source": traffic_source, "random_access": random_access, "per_rank": per_rank,
                     "device_source_sha256": device_source_sha256()[:16],
                     "kernel_code_sha256": code_sha, "kernel_compiler": kernel_compiler_of(c, a.symmetry),
                     "note": "achieved = algorithmic bytes (2*S + 8*g + 8 per distinct state) over the summed durations "
                             "of the step's per-level k_expand launches (HIP events on the engine stream).  traffic = DRAM bytes "
                             "per launch from the gfx950 request-size counters: every random 8-B probe fills one 128-B line "
                             "(6.25 % useful bytes is the ceiling for the probe part), a claim adds a 64-B atomic request.  "
                             "What bounds the kernel is the HBM traffic of these random accesses "
                             "(random_access.dram_traffic_frac_of_hbm_peak; ratio_to_randbench_same_mix ~ 1: the run moves "
                             "its lines as fast as a microbenchmark of the same load/CAS mix); neither fewer ALU instructions "
                             "nor more waves per SIMD shorten it (profiles/r02_ablation.txt, r02_occupancy_sweep.txt)"},
        "device": device_info(),
    }
    if (world == 1 and not a.symmetry and not a.no_orbit_counting and not a.level_budget and c["n_replicas"] <= KMC_SYMMETRY_MAX_REPLICAS
            and r.verdict == "ok"):
        # The same check with symmetry reduction by orbit counting (kmc_config.symmetry): one stored
        # state per orbit of the permutations of Replicas, every count weighted by the orbit's size.  It must report the
        # plain run's numbers — compared here, count by count and level by level — and is timed the same way; it is NOT
        # `value` (SURVEY rules TLC's SYMMETRY out because it changes the counts; this one does not, but it is another search).
        sres, sdt = run_single(c, a.steps, a.warmup, symmetry=True)
        sr = sres[-1]
        sk = sum(x.seconds_expand for x in sres) / len(sres)
        same = ((sr.verdict, sr.distinct, sr.generated, sr.depth, sr.levels, sr.action_generated, sr.deadlock_states,
                 sr.generated_repeats) ==
                (r.verdict, r.distinct, r.generated, r.depth, r.levels, r.action_generated, r.deadlock_states,
                 r.generated_repeats))
        s_alg = alg_bytes_per_state * sr.orbit_representatives
        s_code = kernel_code_sha256_of(c, symmetry=True)
        s_traffic, s_traffic_source = measured_traffic(s_code, None, sr.depth)
        out["orbit_counting"] = {
            "value": sum(x.distinct for x in sres) / sdt, "unit": "distinct states/s", "ms_per_step": 1e3 * sdt / a.steps,
            "time_to_exhaustive_s": sdt / a.steps, "speedup_over_plain": (dt / a.steps) / (sdt / a.steps),
            "stored_states": sr.orbit_representatives, "distinct_states": sr.distinct, "states_generated": sr.generated,
            "depth": sr.depth, "every_count_equals_the_plain_run": same,
            "matches_oracle_golden": None if exp is None else (sr.distinct == exp["distinct"] and sr.generated == exp["generated"]
                                                               and sr.depth == exp["depth"]),
            "kernel_seconds_per_step": sk, "launches_per_step": sr.expand_launches,
            "roofline": {"bound": "hbm", "achieved": s_alg / max(sk, 1e-12) / 1e9, "peak": HBM_PEAK_BPS / 1e9, "unit": "GB/s",
                         "frac": s_alg / max(sk, 1e-12) / HBM_PEAK_BPS,
                         "traffic": s_traffic, "traffic_source": s_traffic_source, "kernel_code_sha256": s_code,
                         "note": "algorithmic bytes of the STORED states (the same per-state figure) over this search's "
                                 "k_expand time; the kernel is instruction-bound here (the representative of every successor "
                                 "is the smallest of its images under the permutations: profiles/r03_symmetry.txt)"}}
    if (world == 1 and not a.symmetry and not a.level_budget and not a.no_traces_leg and not a.workload and not a.small
            and os.environ.get("KMC_BENCH_TRACE", "0") != "1" and r.verdict == "ok"):
        # The same check as a CLI user runs it: predecessors kept for counterexample traces (TLC always keeps them; `value` above
        # is the search alone, as BASELINE.json's metric is).  Since round 6 a claim's predecessor is the second word of the claim's
        # own 16-byte slot: one dirty line per claim, not two.  Never part of `value`.
        try:
            tres, tdt = run_single(c, a.config_steps, 1, keep_trace=True)
            tr = tres[-1]
            out["traces_kept"] = {
                "ms_per_step": 1e3 * tdt / len(tres), "k_expand_ms": 1e3 * sum(x.seconds_expand for x in tres) / len(tres),
                "clear_seen_set_ms": 1e3 * sum(getattr(x, "seconds_clear", 0.0) for x in tres) / len(tres),
                "value": sum(x.distinct for x in tres) / tdt, "unit": "distinct states/s", "steps": len(tres),
                "slowdown_over_the_search_alone": (tdt / len(tres)) / (dt / a.steps),
                "every_count_equals_the_plain_run": ((tr.verdict, tr.distinct, tr.generated, tr.depth, tr.levels) ==
                                                     (r.verdict, r.distinct, r.generated, r.depth, r.levels)),
                "seen_set": "16-byte slots: fingerprint + predecessor (kmc_handle::paired)"}
        except Exception as e:   # (a leg beside the headline never takes the line down)
            out["traces_kept"] = {"error": repr(e)}
    if world == 1 and not a.symmetry and not a.level_budget and not a.no_cold_start:
        # (before the legs with the large tables: the front-end process that started right after the stretch leg had handed its
        # 250 GiB back spent 0.26 - 0.35 s in its own teardown - calls 28 and 33 - against 0.02 - 0.05 s at any other time: what a CLI
        # user waits for is measured on a device that is not still digesting somebody else's release)
        cs = cold_start(c)
        if cs:
            out["cold_start"] = cs
    if (world == 1 and not a.symmetry and not a.level_budget and not a.no_baseline_configs and not a.workload and not a.small):
        # SURVEY section 8d rows "config 4" and "config 5": driver-timed here, never part of `value`
        out["baseline_configs"] = {name: baseline_leg(name, a.config_steps, 1, with_cpu=not a.no_cpu_baseline) for name in BASELINE_LEGS}
    if (world == 1 and not a.symmetry and not a.level_budget and not a.no_stretch and not a.workload and not a.small):
        out["stretch_1gpu"] = stretch_1gpu_leg()
    if want_stretch:
        # The headline line is complete here.  The stretch leg is another collective search (6.45 G states over the ranks): should
        # it hang, the headline must not be lost with it — it goes to stderr first; stdout still carries ONE line.
        sys.stderr.write("[bench.py: the headline line, before the stretch leg] " + json.dumps(out) + "\n")
        sys.stderr.flush()
        out["stretch"] = stretch_leg(a)
    if not a.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(c, a.cpu_states or max(1_000_000, distinct // 4), distinct)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
