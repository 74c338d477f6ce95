"""The measurement tooling (bench.py's profile readers, tools/summarize_profile.py) and the committed evidence under
profiles/: the bench line keeps its contract, its side numbers are derived from files that say what they say, and the
rocprofv3 summary agrees with the line it stands behind.  No GPU, no oracle."""
import csv
import importlib
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def bench():
    return importlib.import_module("bench")


def test_dram_bytes_per_access_are_read_from_the_committed_calibration(bench):
    b7, src = bench.calibrated_bytes_per_access(7)          # the kernel's own mix: a load + a CAS for 35 %
    assert src.endswith("dram_bytes_per_access.txt")
    assert 140.0 < b7 < 160.0                                # 128 B line fill + 0.35 x 64 B atomic request
    b1, _ = bench.calibrated_bytes_per_access(1)             # loads only: one 128-byte request each
    assert abs(b1 - 128.0) < 1.0
    assert bench.calibrated_bytes_per_access(42) is None     # a mode the file does not hold


def test_traffic_is_quoted_only_for_the_machine_code_it_was_measured_on(bench):
    """bench.measured_traffic(code): a committed PMC summary is quoted for a run only when it carries the kernel_code_sha256 of
    the kernels that run executes — whatever the workload (headline, orbit counting, BASELINE configs 4 and 5)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*pmc_summary.json")), reverse=True)
    stamped = [(f, json.load(open(f))) for f in files]
    stamped = [(f, j) for f, j in stamped if j.get("kernel_code_sha256")]
    assert stamped, "no PMC summary carries a code identity"
    first = {}
    for f, j in stamped:   # newest round first wins for a given code
        first.setdefault(j["kernel_code_sha256"], (f, j))
    for code, (f, j) in first.items():
        run = j.get("run", {})
        t, src = bench.measured_traffic(code, run.get("level_budget"), run.get("depth"))
        assert t == j["hbm_bytes_per_launch"] and os.path.relpath(f, ROOT) in src and code[:16] in src
        # ... and only for the SEARCH it was measured over (a per-launch average over ten geometrically growing
        # levels is not the per-launch traffic of a run with another budget)
        t_other, why_other = bench.measured_traffic(code, (run.get("level_budget") or 0) + 3, None)
        assert t_other is None and "another search" in why_other
        if "FETCH_SIZE_bytes_as_reported" in j and "atomic_bytes" in j:
            # the DRAM-unit counters, not FETCH_SIZE: reads are twice what FETCH_SIZE reports on gfx950
            assert abs(j["read_bytes"] / j["FETCH_SIZE_bytes_as_reported"] - 2.0) < 0.02
            assert abs(j["hbm_bytes"] - (j["read_bytes"] + j["write_bytes"] + j["atomic_bytes"])) < 1.0
    t2, why = bench.measured_traffic("1" * 64)          # other machine code: not quoted, and the reason says so
    assert t2 is None and "no PMC summary was measured on this code" in why and "another search" not in why
    assert bench.measured_traffic(None)[0] is None      # no compiler to ask: not quoted either


def test_a_committed_pmc_summary_belongs_to_the_headline_kernels_this_tree_builds(bench):
    """The PLAIN headline's kernels hiprtc builds from this tree (gfx950, no GPU needed) hash to what a committed summary
    records, so bench.py's default line quotes measured traffic; the same for the orbit-counting leg."""
    code = bench.headline_kernel_code_sha256()
    assert code is not None and len(code) == 64
    t, src = bench.measured_traffic(code)
    assert t, f"the headline's kernels changed since the counters were collected: re-measure ({src})"


NEWEST_ROUND = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_bench.json"))[-1][:3]


@pytest.mark.parametrize("rnd", sorted({"r02", NEWEST_ROUND}))
def test_committed_bench_line_keeps_the_contract_and_is_self_consistent(rnd):
    j = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_bench.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["higher_is_better"] is True and j["vs_baseline"] is None and j["dtype"] == "u64"
    assert "workload" in j["config"] and "model" not in j["config"] and j["config"]["matches_oracle_golden"] is True
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # achieved = algorithmic bytes per distinct state x distinct states / summed k_expand time
    distinct = j["config"]["distinct_states"]
    assert abs(r["achieved"] * 1e9 - r["algorithmic_bytes_per_distinct_state"] * distinct / r["kernel_seconds_per_step"]) \
        < 1e-6 * r["achieved"] * 1e9
    assert abs(r["algorithmic_bytes_per_launch"] * r["launches_per_step"] - r["algorithmic_bytes_per_distinct_state"] * distinct) < 1.0
    # value = distinct states of the timed steps / wall time; the kernel fits inside the step
    assert abs(j["value"] - distinct / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]
    assert r["kernel_seconds_per_step"] * 1e3 <= j["ms_per_step"]
    # measured traffic is well above the algorithmic bytes (one 128-byte line per 8-byte probe) and below the HBM peak
    assert r["traffic"] > 4 * r["algorithmic_bytes_per_launch"]
    assert r["traffic"] * r["launches_per_step"] / r["kernel_seconds_per_step"] < 8.0e12
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["unit"] == j["unit"] and c["cores"] >= 1


@pytest.mark.parametrize("rnd", sorted({"r02", NEWEST_ROUND}))
def test_rocprof_kernel_stats_agree_with_the_bench_line(rnd):
    j = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_bench.json")))
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", f"{rnd}_kernel_stats.csv"))))
    top = rows[0]
    assert top["Name"].startswith("kmc_expand_") and float(top["Percentage"]) > 90.0        # the dominant kernel
    assert int(top["Calls"]) == j["roofline"]["launches_per_step"]
    avg_prof = float(top["AverageNs"]) * 1e-9
    avg_bench = j["roofline"]["kernel_seconds_per_step"] / j["roofline"]["launches_per_step"]
    assert abs(avg_prof - avg_bench) / avg_bench < 0.05      # rocprofv3's average launch duration vs bench.py's HIP events


def test_summarize_profile_derives_dram_bytes_from_the_32_byte_unit_counters(tmp_path):
    d = tmp_path / "prof"
    (d / "trace").mkdir(parents=True)
    with open(d / "trace" / "x_kernel_trace.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp", "VGPR_Count", "LDS_Block_Size", "Scratch_Size",
                    "Workgroup_Size", "Grid_Size"])
        w.writerow(["kmc_expand_T", 1000, 2001000, 40, 0, 16, 256, 1024])
        w.writerow(["kmc_expand_T", 3000000, 5000000, 40, 0, 16, 256, 1024])
        w.writerow(["other", 0, 10, 8, 0, 0, 64, 64])

    def pmc(i, rows):
        (d / f"pmc{i}").mkdir()
        with open(d / f"pmc{i}" / "pmc_counter_collection.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value"])
            for r in rows:
                w.writerow(r)
    pmc(1, [["kmc_expand_T", "TCC_EA0_RDREQ_DRAM_32B_sum", 400], ["kmc_expand_T", "TCC_EA0_RDREQ_DRAM_32B_sum", 600],
            ["other", "TCC_EA0_RDREQ_DRAM_32B_sum", 999]])
    pmc(2, [["kmc_expand_T", "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum", 10], ["kmc_expand_T", "TCC_EA0_WRREQ_ATOMIC_DRAM_32B_sum", 20]])
    pmc(3, [["kmc_expand_T", "FETCH_SIZE", 15.625]])     # KiB: 500 requests x 64 B = half of the 32 000 bytes really read
    pmc(4, [["kmc_expand_T", "WRITE_SIZE", 0.9375]])
    out, pm = tmp_path / "summary.json", tmp_path / "pmc_summary.json"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "summarize_profile.py"), str(d), str(out), str(pm)],
                          stdout=subprocess.DEVNULL)
    s, p = json.load(open(out)), json.load(open(pm))
    assert s["dominant_kernel"] == "kmc_expand_T" and s["launches"] == 2
    assert s["derived"]["dram_read_bytes"] == 32 * 1000 and s["derived"]["dram_bytes"] == 32 * 1030
    assert p["hbm_bytes"] == 32 * 1030 and p["hbm_bytes_per_launch"] == 32 * 1030 / 2
    assert p["read_bytes"] == 2 * p["FETCH_SIZE_bytes_as_reported"]
    assert len(p["device_source_sha256"]) == 64


def test_claims_per_state_come_from_the_profile_not_from_a_literal(bench):
    """the claim stream of the roofline block was priced with a typed-in 1.115 x distinct.  It is the
    ratio of the atomic requests the memory side counted to the states the profiled run found, read from the newest
    committed summary; and no other unexplained number feeds the bench line."""
    import ast
    cpd, src = bench.claims_per_distinct_state()
    s = json.load(open(os.path.join(ROOT, src)))
    assert abs(cpd - s["counters_sum_over_launches"]["TCC_EA0_ATOMIC_sum"] / s["run"]["distinct_states"]) < 1e-12
    assert 1.0 <= cpd < 1.3                                  # one claim per new state plus the lost races
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    # structural constants only: 0 / 1 / 2 / 4, unit conversions, bytes per word, the 128-byte line and 64-byte atomic
    # request of profiles/r02_request_size.txt, the smallest cpu_baseline sample; everything measured is read from profiles/
    # (3 and 7 are randbench MODE numbers — which line of the committed file to read —, 16 the length of the sha prefix)
    allowed = {0, 1, 2, 3, 4, 7, 8, 16, 64, 128, 8.0, 128.0, 1e3, 1e9, 1e-12, 1.0, 1_000_000}
    odd = sorted({n.value for n in ast.walk(main) if isinstance(n, ast.Constant) and isinstance(n.value, (int, float))
                  and not isinstance(n.value, bool)} - allowed)
    assert odd == [], f"numeric literals in bench.main(): {odd}"


def test_ladder_tool_configurations_are_valid_bindings():
    """tools/run_ladder.py: every configuration of the ladder is a binding the ABI accepts, every oracle prefix belongs
    to one of them and starts with the known first levels 1, 2N."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    ladder = importlib.import_module("run_ladder")
    from kafka_specification_amd import CheckerConfig
    for name, c in ladder.RUNS.items():
        CheckerConfig(**c).to_native()
    for name, levels in ladder.ORACLE_PREFIX.items():
        assert name in ladder.RUNS
        assert levels[:2] == [1, 2 * ladder.RUNS[name]["n_replicas"]]


def test_one_code_object_whoever_compiles():
    """A process under the PyTorch wheel's HIP runtime (the bench, the tests) and one on the system ROCm (KMC_NO_TORCH=1: the
    native CLI, rocprofv3 runs) resolve a configuration to the SAME cached code object — the two toolchains emit different
    instructions for the same source, so a profile must load what the bench loads, not compile its own (round 4)."""
    import subprocess
    import sys
    prog = ("import sys; sys.path.insert(0, %r); import kafka_specification_amd as kmc; "
            "c = kmc.CheckerConfig(model='Kip320', n_replicas=3, log_size=2, max_records=2, max_leader_epoch=1); "
            "print(kmc.code_object_path(c, 'gfx950')); print(kmc.kernel_code_sha256(c))" % ROOT)
    outs = []
    for no_torch in ("0", "1"):
        env = dict(os.environ, KMC_NO_TORCH=no_torch)
        env.pop("KMC_JIT_DEFINES", None)
        p = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, env=env, timeout=600)
        assert p.returncode == 0, p.stderr[-800:]
        outs.append(p.stdout.strip().splitlines()[-2:])
    assert outs[0] == outs[1]


def test_every_leg_of_the_bench_line_has_its_exact_fixture_and_its_constants_compile(bench):
    """The driver's line holds BASELINE configs 4 (at two sizings) and 5 and the 6.45 G-state stretch beside the headline: each leg's
    binding has a committed exact fixture (so `matches_oracle_golden` is never null there) whose level sizes it is held to."""
    from kafka_specification_amd import CheckerConfig
    for name, spec in bench.BASELINE_LEGS.items():
        c = dict(spec["c"])
        CheckerConfig(**c).to_native()
        exp = bench.expected_counts(c)
        assert exp is not None and exp["levels"] and sum(exp["levels"]) == exp["distinct"], name
        if c.get("max_levels"):
            assert len(exp["levels"]) == c["max_levels"] == exp["depth"], name
        assert spec["table"] % 64 == 0 and spec["table"] * 0.5 > exp["distinct"] * 0.3     # (a load the leg's A/B chose, below 0.6)
    exp = bench.expected_counts(dict(bench.STRETCH))
    assert exp and exp["distinct"] == 6452700520 and exp["depth"] == 54 and exp["file"].endswith("orbit_kip320_3_6_6_3.json")
    deep = bench.BASELINE_LEGS["config4_deep_kip279_5brokers_log4_levels12"]["c"]
    assert (deep["n_replicas"], deep["log_size"], deep["max_records"], deep["max_leader_epoch"]) == (5, 4, 4, 3)   # SURVEY 8(a.0)'s sizing


def test_the_stretch_leg_is_read_against_the_microbenchmark_of_its_own_footprint(bench):
    """profiles/rNN_randbench_sweep.txt holds randbench at 2^33 eight-byte slots (64 GiB) and beyond: the ceiling the stretch leg's
    probe rate is divided by is the one measured AT that footprint, not the 8 GiB figure (missing 4)."""
    rates, src = bench.randbench_rates_at(33)
    assert src and src.endswith("randbench_sweep.txt")
    assert 40e9 < rates[1] < 60e9 and 25e9 < rates[13] < 40e9 and 25e9 < rates[7] < 40e9
    r30, _ = bench.randbench_rates_at(30)
    assert r30 and abs(r30[1] / rates[1] - 1.0) < 0.1          # random loads: the same rate at 8 GiB and at 64 GiB
    assert bench.randbench_rates_at(12) == ({}, None)


def test_step_breakdown_accounts_for_the_whole_step(bench):
    b = bench.step_breakdown(31.4, 25.8e-3, 2.4e-3, 2.78e-3)
    assert abs(sum(b.values()) - 31.4) < 1e-9 and b["k_inv_ms"] == pytest.approx(2.4) and b["host_and_rest_ms"] == pytest.approx(0.42)
