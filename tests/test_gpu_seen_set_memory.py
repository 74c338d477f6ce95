"""Where the seen-set's memory comes from (round 6, csrc/kmc_engine_core.cpp: seen_set_alloc): a range of addresses mapped from
8 MiB physical chunks by default, one hipMalloc under KMC_SEEN_SET_CHUNK_LOG2=0 and whenever the mapping cannot be had.  The
chunk size is read once per process, so every variant is its own process (the native front end, which says under KMC_VERBOSE
which of the two it got); the answers must be the oracle's whichever memory the table lies in — with and without traces (the
predecessor table goes through the same allocator), at a capacity that is not a multiple of the chunk, and after a handle that
was closed and opened again in one process (the ranges are unmapped and released with the handle).
"""
import os
import re
import subprocess

import pytest

import kmo
from kafka_specification_amd import CheckerConfig, ModelChecker

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "kafka_specification_amd", "tlc")
INV = ("TypeOk", "WeakIsr", "StrongIsr")


def cli(env, *extra):
    e = dict(os.environ, KMC_VERBOSE="1", **env)
    # (models/Kip320.cfg binds the headline; -table / -frontier only size the buffers; 64 x 100003 slots are 48.8 MiB: not a
    #  multiple of any chunk, so the mapped range is longer than the table)
    r = subprocess.run([EXE, os.path.join(ROOT, "models", "FiniteReplicatedLog.tla"), "-table", str(64 * 100003), "-frontier", "262144",
                        *extra], capture_output=True, text=True, env=e)
    m = re.search(r"\[kmc\] seen-set: (\d+) slots x (\d+) B at (0x[0-9a-f]+) \(([^)]*)\)", r.stderr)
    assert m, r.stderr[-2000:]
    return r, m.group(4)


@pytest.mark.parametrize("env, where", [
    ({}, "mapped from chunks"),                                   # the default: 8 MiB chunks
    ({"KMC_SEEN_SET_CHUNK_LOG2": "21"}, "mapped from chunks"),    # the granularity itself
    ({"KMC_SEEN_SET_CHUNK_LOG2": "5"}, "mapped from chunks"),     # below the granularity: raised to it
    ({"KMC_SEEN_SET_CHUNK_LOG2": "0"}, "one hipMalloc"),          # switched off
    ({"KMC_SEEN_SET_CHUNK_LOG2": "44"}, "one hipMalloc"),         # a chunk no device holds (clamped to 2^40): the fallback
])
@pytest.mark.parametrize("extra", [(), ("-notrace",)])
def test_every_kind_of_seen_set_memory_gives_the_same_search(env, where, extra):
    r, got = cli(env, *extra)
    assert got == where, r.stderr[-2000:]
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    # FiniteReplicatedLog at the shipped .cfg: the closed form (tests/test_gpu_parity.py holds it level set by level set)
    assert "1190091 states generated, 116281 distinct states found, 0 states left on queue." in r.stdout


def test_handles_opened_and_closed_in_one_process_release_their_ranges():
    """Twelve handles one after the other, each with a 1 GiB seen-set + predecessor table mapped from chunks: were a closed
    handle's chunks not released, the device would run out (12 x 2 GiB + frontiers stay far below 288 GB only if they are
    — so the check is on hipMemGetInfo's free bytes, which must come back)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    free0, total = ctypes.c_size_t(), ctypes.c_size_t()
    o = kmo.Run(kmo.make_config("Kip320", N=3, L=2, R=2, E=1, invariants=INV))
    cfg = CheckerConfig(model="Kip320", n_replicas=3, log_size=2, max_records=2, max_leader_epoch=1, invariants=INV,
                        table_capacity=1 << 27, frontier_capacity=1 << 20, keep_trace=True)
    with ModelChecker(cfg) as mc:     # (the first handle also pays for the context and the code object: measured after it)
        r = mc.run()
        assert (r.distinct, r.generated) == (o.distinct, o.generated)
    assert hip.hipMemGetInfo(ctypes.byref(free0), ctypes.byref(total)) == 0
    for _ in range(12):
        with ModelChecker(cfg) as mc:
            r = mc.run()
            assert (r.distinct, r.generated) == (o.distinct, o.generated)
    free1 = ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(free1), ctypes.byref(total)) == 0
    assert free0.value - free1.value < (256 << 20), (free0.value, free1.value)   # nothing of 24 GiB was kept


def _digest(text):
    lines = text.splitlines()
    verdict = [l for l in lines if l.startswith(("Error:", "The depth")) or ("states generated," in l and not l.startswith("Progress"))]
    heads = [i for i, l in enumerate(lines) if l.startswith("State ")]
    return verdict, len(heads), lines[heads[0] + 1:heads[0] + 7], lines[heads[-1] + 1:heads[-1] + 7]


def test_predecessors_in_the_slots_give_the_trace_a_table_of_their_own_gives():
    """A run that keeps traces on 64-bit entries stores a claim's predecessor in the claim's own 16-byte slot (kmc_handle::paired,
    KMC_FLAG_PAIRED); KMC_PAIRED_SLOTS=0 is the older form, a predecessor table of its own.  Same verdict, counts, trace length,
    first and last state of the counterexample (Kip101 at the shipped .cfg violates its invariant) either way, in both front
    ends' common format; -fp128 (its slot is full: the separate table stays) agrees too."""
    args = [EXE, os.path.join(ROOT, "models", "Kip101.tla"), "-table", "4194304", "-frontier", "1048576"]
    outs = []
    for env, extra in (({}, ()), ({"KMC_PAIRED_SLOTS": "0"}, ()), ({}, ("-fp128",))):
        r = subprocess.run(args + list(extra), capture_output=True, text=True, env=dict(os.environ, KMC_VERBOSE="1", **env))
        assert r.returncode == 12, r.stdout[-1500:] + r.stderr[-1500:]
        m = re.search(r"\[kmc\] seen-set: (\d+) slots x (\d+) B", r.stderr)
        outs.append((int(m.group(2)), _digest(r.stdout)))
    assert [o[0] for o in outs] == [16, 8, 16]
    assert outs[0][1] == outs[1][1] == outs[2][1]
    assert outs[0][1][1] >= 2   # a real counterexample: more than the initial state


def test_checkpoint_of_a_trace_keeping_run_resumes_with_its_predecessors(tmp_path):
    """The checkpoint of a paired handle holds the 16-byte slots (header has_pred = 2); resumed, the search ends with the counts of
    an uninterrupted one and still reconstructs a trace back to Init across the checkpoint."""
    base = dict(model="Kip101", n_replicas=3, log_size=2, max_records=2, max_leader_epoch=2, invariants=INV,
                table_capacity=1 << 22, frontier_capacity=1 << 20, keep_trace=True)
    with ModelChecker(CheckerConfig(**base)) as mc:
        whole = mc.run()
        whole_trace = mc.trace() if whole.verdict == "invariant" else None
    assert whole.verdict == "invariant" and whole_trace
    cut = max(2, whole.depth // 2)
    path = str(tmp_path / "ck.bin")
    with ModelChecker(CheckerConfig(**base, max_levels=cut)) as mc:
        part = mc.run()
        assert part.verdict == "level_limit"
        mc.save_checkpoint(path)
    with ModelChecker(CheckerConfig(**base)) as mc:
        mc.load_checkpoint(path)
        rest = mc.resume()
        assert (rest.verdict, rest.distinct, rest.generated, rest.depth) == (whole.verdict, whole.distinct, whole.generated, whole.depth)
        tr = mc.trace()
        assert len(tr) == len(whole_trace) and tr[0][1] == whole_trace[0][1]


@pytest.mark.parametrize("spread", ["1", "4", "64", "1000"])
@pytest.mark.parametrize("extra", [(), ("-notrace",)])
def test_a_spread_seen_set_gives_the_same_search(spread, extra):
    """KMC_SEEN_SET_SPREAD (opt-in; seen_set_alloc's spacers): the table's chunks are created in clusters with one unmapped
    allocation between two clusters, released once the table is mapped.  Where the chunks lie changes no answer; the verbose line
    says how many spacers there were (none at 1; one less than the chunks' clusters otherwise; the factor is clamped to 64)."""
    r, got = cli({"KMC_SEEN_SET_SPREAD": spread}, *extra)
    assert got == "mapped from chunks", r.stderr[-2000:]
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "1190091 states generated, 116281 distinct states found, 0 states left on queue." in r.stdout
    m = re.search(r"\[kmc\] seen-set memory: (\d+) chunks of (\d+) MiB in clusters of (\d+), (\d+) spacers of ([\d.]+) GiB", r.stderr)
    assert m, r.stderr[-2000:]
    chunks, mib, per_cluster, spacers, gib = int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)), float(m.group(5))
    assert chunks * mib * (1 << 20) >= 64 * 100003 * 8 and per_cluster >= 1
    if spread == "1":
        assert spacers == 0
    else:
        clusters = (chunks + per_cluster - 1) // per_cluster
        assert spacers == clusters - 1
        assert abs(gib - (min(int(spread), 64) - 1) * per_cluster * mib / 1024) < 0.01


def test_the_spacers_of_a_spread_seen_set_go_back_at_open():
    """A handle opened with KMC_SEEN_SET_SPREAD=16 owns sixteen times its table for a moment; after kmc_open the device's free memory
    is down by the handle's own buffers only (own process: the factor is read once)."""
    code = (
        "import ctypes, os\n"
        "from kafka_specification_amd import CheckerConfig, ModelChecker\n"
        "hip = ctypes.CDLL('libamdhip64.so'); f0, f1, t = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()\n"
        "cfg = CheckerConfig(model='Kip320', n_replicas=3, log_size=2, max_records=2, max_leader_epoch=1,\n"
        "                    invariants=('TypeOk', 'WeakIsr', 'StrongIsr'), table_capacity=1 << 27, frontier_capacity=1 << 20)\n"
        "with ModelChecker(cfg) as mc: mc.run()\n"          # (context + code object paid for)
        "assert hip.hipMemGetInfo(ctypes.byref(f0), ctypes.byref(t)) == 0\n"
        "with ModelChecker(cfg) as mc:\n"
        "    assert hip.hipMemGetInfo(ctypes.byref(f1), ctypes.byref(t)) == 0\n"
        "    r = mc.run()\n"
        "print('HELD', f0.value - f1.value, r.distinct, r.generated)\n")
    r = subprocess.run(["python", "-c", code], capture_output=True, text=True, cwd=ROOT,
                       env=dict(os.environ, KMC_SEEN_SET_SPREAD="16", KMC_VERBOSE="1"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert re.search(r" [1-9]\d* spacers of 0\.\d+ GiB", r.stderr), r.stderr[-1500:]
    held, distinct, generated = map(int, re.search(r"HELD (-?\d+) (\d+) (\d+)", r.stdout).groups())
    o = kmo.Run(kmo.make_config("Kip320", N=3, L=2, R=2, E=1, invariants=INV))
    assert (distinct, generated) == (o.distinct, o.generated)
    assert held < (4 << 30), held   # 1 GiB of table + frontiers; sixteen times the table would be 16 GiB
