"""SURVEY section 5 (race detection / sanitizers): everything of this repository that can run on a CPU runs under a sanitizer here —
the C oracle (ASan + UBSan, and TSan for its threads: `make -C oracle asan tsan`) and the DEVICE model templates compiled for the
host (tests/host_emu.cpp through the stand-alone search tests/host_emu_bfs.cpp: ASan + UBSan; shifts by a field's width, reads
past a state's last word and signed overflows in the packed-field arithmetic are exactly what UBSan sees and a GPU does not
report).  Two ladder configurations each; the counts must equal the uninstrumented oracle's.  GPU sanitizers do not exist on
this pool (gpurun refuses them): the kernels' one race — the claim of a seen-set slot — is tested by counting
(tests/test_gpu_insert_race.py)."""
import json
import os
import subprocess

import pytest

import kmo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle")
REPORTS = ("ERROR: AddressSanitizer", "runtime error:", "WARNING: ThreadSanitizer", "ERROR: LeakSanitizer", "SUMMARY: ")
LADDER = [("Kip320", 3, 2, 2, 1, 7), ("Kip279", 3, 2, 2, 2, 1)]


@pytest.fixture(scope="module")
def sanitized_oracles():
    subprocess.check_call(["make", "-s", "-C", ORACLE, "kmc_oracle", "asan", "tsan"])
    return {k: os.path.join(ORACLE, "kmc_oracle" + s) for k, s in (("plain", ""), ("asan", "_asan"), ("tsan", "_tsan"))}


def run_oracle(exe, model, N, L, R, E, inv, threads):
    p = subprocess.run([exe, "--model", model, "--N", str(N), "--L", str(L), "--R", str(R), "--E", str(E), "--threads", str(threads),
                        "--inv", str(inv), "--continue"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1", TSAN_OPTIONS="halt_on_error=1"))
    assert p.returncode == 0, p.stderr[-2000:]
    assert not any(r in p.stderr for r in REPORTS), p.stderr[-2000:]
    j = json.loads(p.stdout)
    return {k: j[k] for k in ("distinct", "generated", "depth", "levels", "action_generated", "deadlock_states", "viol_count")}


@pytest.mark.parametrize("cfg", LADDER, ids=lambda c: "-".join(map(str, c[:5])))
@pytest.mark.parametrize("which", ["asan", "tsan"])
def test_the_c_oracle_is_clean_under_the_sanitizers_and_counts_the_same(sanitized_oracles, which, cfg):
    model, N, L, R, E, inv = cfg
    want = run_oracle(sanitized_oracles["plain"], model, N, L, R, E, inv, 4)
    got = run_oracle(sanitized_oracles[which], model, N, L, R, E, inv, 4)      # four threads: the level barrier, the shared table
    assert got == want


def build_sanitized_emu():
    exe = os.path.join(ROOT, "tests", "_host_emu_bfs_san")
    src = [os.path.join(ROOT, "tests", f) for f in ("host_emu_bfs.cpp", "host_emu.cpp")]
    csrc = os.path.join(ROOT, "kafka_specification_amd", "csrc")
    deps = src + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.startswith("kmc_") and f.endswith(".h")]
    if not os.path.exists(exe) or any(os.path.getmtime(exe) < os.path.getmtime(d) for d in deps):
        # (KMC_EMU_SMALL_TABLE: the instrumented build of every configuration of host_emu.cpp takes a quarter of an hour)
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-DKMC_EMU_SMALL_TABLE", "-fsanitize=address,undefined",
                               "-fno-sanitize-recover=undefined", "-o", exe, src[0]], cwd=os.path.join(ROOT, "tests"))
    return exe


@pytest.fixture(scope="module")
def sanitized_emu():
    return build_sanitized_emu()   # (tests/conftest.py builds it before pytest-xdist workers start: two must not write it at once)


# (model id, name, N, L, R, E, K, invariants, KMC_LAYOUT_* of the compiled entry: 0 automatic = tight here, 2 replica-major with the
# kind-major walk of pass 2, 3 replica-major grouped)
EMU = [(5, "Kip320", 3, 2, 2, 2, 0, 7, 0), (5, "Kip320", 3, 2, 2, 2, 0, 7, 2), (4, "Kip279", 3, 2, 2, 2, 0, 1, 2),
       (3, "Kip101", 3, 2, 2, 2, 0, 1, 3)]


@pytest.mark.parametrize("cfg", EMU, ids=lambda c: f"{c[1]}-{c[2]}-{c[3]}-{c[4]}-{c[5]}-layout{c[8]}")
def test_the_device_model_templates_are_clean_under_asan_and_ubsan(sanitized_emu, cfg):
    """A whole breadth-first search through KmcKafka::inst<I> / apply<K> / violated_pre / violated_stream as g++ compiles them,
    instrumented: no report, and the oracle's counts (the two lowerings of pass 2 are compared on every enabled binding on the way)."""
    mid, name, N, L, R, E, K, inv, lm = cfg
    p = subprocess.run([sanitized_emu] + [str(x) for x in (mid, N, L, R, E, K, inv, lm)], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"))
    assert p.returncode == 0, p.stderr[-3000:]
    assert not any(r in p.stderr for r in REPORTS), p.stderr[-3000:]
    j = json.loads(p.stdout)
    invs = tuple(n for k, n in enumerate(kmo.INV_NAMES) if inv >> k & 1)
    o = kmo.Run(kmo.make_config(name, N=N, L=L, R=R, E=E, invariants=invs, stop_on_violation=False, threads=4))
    assert (j["distinct"], j["generated"], j["depth"], j["levels"], j["deadlock_states"]) == \
        (o.distinct, o.generated, o.depth, o.levels, o.deadlock_states)
    if len(invs) == 1:   # (one invariant: the number of violating states is that invariant's count)
        assert j["violating_states"] == o.viol_count[invs[0]]
    assert j["kind_major_bindings_bad"] == 0
    o.close()
