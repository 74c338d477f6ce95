"""GPU: (1) the sharded path — P logical shards on one MI355X with an in-process exchange
(bucket kernel -> exchange -> insert kernel), against the oracle; (2) counterexample traces
rebuilt from predecessor fingerprints, every step validated by the oracle; (3) the headline
configuration against the committed golden fixture and its size-independent properties."""
import json
import os

import pytest

import kmo
from kafka_specification_amd import CheckerConfig, ModelChecker
from kafka_specification_amd.sharded import check_loopback

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
INV_INDEX = {"TypeOk": 0, "WeakIsr": 1, "StrongIsr": 2, "LeaderInIsr": 3}


@pytest.mark.parametrize("exchange", ["rccl", "torch"])   # under the C ABI (device-to-device runs + one k_insert) / torch slices
@pytest.mark.parametrize("P", [2, 3, 4])
@pytest.mark.parametrize("model,N,L,R,E,inv", [("Kip320", 3, 2, 2, 1, ("TypeOk", "WeakIsr", "StrongIsr")),
                                               ("Kip279", 3, 2, 2, 2, ("TypeOk", "StrongIsr")),
                                               ("Kip320FirstTry", 2, 3, 3, 2, ("TypeOk",))])
def test_loopback_shards_match_oracle(P, model, N, L, R, E, inv, exchange, monkeypatch):
    monkeypatch.setenv("KMC_EXCHANGE", exchange)
    o = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=inv))
    cfg = CheckerConfig(model=model, n_replicas=N, log_size=L, max_records=R, max_leader_epoch=E, invariants=inv,
                        table_capacity=1 << 20, frontier_capacity=1 << 18, send_capacity=1 << 18)
    r = check_loopback(cfg, P)
    assert r.verdict == o.verdict and r.violated_invariant == o.viol_inv
    assert r.levels == o.levels and r.distinct == o.distinct and r.generated == o.generated
    assert list(r.action_generated.values()) == o.action_generated[:len(r.action_generated)]
    assert r.deadlock_states == o.deadlock_states
    if o.viol_inv:
        assert r.violation_depth == o.viol_depth and r.violation_count == o.viol_count


@pytest.mark.parametrize("model", ["KafkaTruncateToHighWatermark", "Kip101", "Kip279", "Kip320FirstTry"])
def test_counterexample_trace_is_a_real_behaviour(model):
    N, L, R, E = 3, 2, 2, 2
    inv = ("TypeOk", "StrongIsr")
    o = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=inv))
    assert o.verdict == "invariant"
    cfg = CheckerConfig(model=model, n_replicas=N, log_size=L, max_records=R, max_leader_epoch=E, invariants=inv,
                        keep_trace=True, table_capacity=1 << 22, frontier_capacity=1 << 20)
    with ModelChecker(cfg) as mc:
        r = mc.run()
        assert (r.verdict, r.violated_invariant, r.violation_depth) == ("invariant", o.viol_inv, o.viol_depth)
        trace = mc.trace()
        names = mc.action_names()
        witness = mc.unpack(mc.witness())
    assert len(trace) == r.violation_depth           # BFS => a shortest counterexample
    assert trace[0] == (None, o.state(0))            # starts at Init
    assert trace[-1][1] == witness
    assert not kmo.check_invariant(o.cfg, INV_INDEX[o.viol_inv], witness)
    for (_, prev), (act, cur) in zip(trace, trace[1:]):
        succ = kmo.successors(o.cfg, prev, o.sb)
        assert (names.index(act), cur) in succ      # each step is a Next step of that action
        assert all(kmo.check_invariant(o.cfg, INV_INDEX[i], prev) for i in inv)  # first violation is the last state


def test_deadlock_verdict():
    o = kmo.Run(kmo.make_config("Kip320", N=2, L=1, R=1, E=1, check_deadlock=True))
    cfg = CheckerConfig(model="Kip320", n_replicas=2, log_size=1, max_records=1, max_leader_epoch=1,
                        check_deadlock=True, table_capacity=1 << 16, frontier_capacity=1 << 12)
    with ModelChecker(cfg) as mc:
        r = mc.run()
        w = mc.unpack(mc.witness())
    assert r.verdict == "deadlock" == o.verdict
    assert kmo.successors(o.cfg, w, o.sb) == []


def test_frontier_and_table_overflow_are_reported():
    base = dict(model="Kip320", n_replicas=3, log_size=2, max_records=2, max_leader_epoch=1)
    with ModelChecker(CheckerConfig(**base, table_capacity=1 << 20, frontier_capacity=512)) as mc:
        assert mc.run().verdict == "frontier_full"
    with ModelChecker(CheckerConfig(**base, table_capacity=4096, frontier_capacity=1 << 16)) as mc:
        assert mc.run().verdict == "table_full"


@pytest.mark.parametrize("model,N,L,R,E", [("KafkaTruncateToHighWatermark", 3, 2, 2, 2), ("Kip101", 3, 3, 2, 2),
                                           ("Kip279", 3, 2, 3, 1), ("Kip320", 3, 3, 3, 1),
                                           ("Kip320FirstTry", 3, 2, 2, 2), ("Kip320", 4, 2, 1, 1),
                                           # wide kernels: the register budget follows the kernel (kmc_engine_codeobj.cpp,
                                           # get_code_object) after Kip320 with 7 replicas lost successors at 80 VGPRs
                                           ("Kip320", 7, 1, 1, 0), ("Kip279", 7, 1, 1, 0),
                                           ("Kip320FirstTry", 8, 1, 1, 0)])
def test_device_successors_match_oracle_on_sampled_states(model, N, L, R, E):
    """Per-state differential test: the device's Next (guards + effects of every action instance)
    against the C oracle's, on a sample of reachable states."""
    o = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=(), max_states=20000))
    with ModelChecker(CheckerConfig(model=model, n_replicas=N, log_size=L, max_records=R, max_leader_epoch=E,
                                    table_capacity=1 << 16, frontier_capacity=1 << 12)) as mc:
        for idx in range(0, min(o.distinct, 20000), 131):
            s = o.state(idx)
            got = sorted((k, mc.unpack(w)) for (w, _fp, k) in mc.successors(mc.pack(s)))
            # (round 4: kmc_successors is TLC's enumeration of Next on the state — a successor two disjuncts of one binding
            # yield is listed twice, as `generated` counts it: a multiset comparison)
            assert got == sorted(kmo.successors(o.cfg, s, o.sb))


@pytest.mark.parametrize("name", ["oracle_kip320_3_5_5_2.json", "oracle_kip320_3_6_6_2.json"])
def test_headline_size_against_golden_fixture(name):
    g = json.load(open(os.path.join(GOLDEN, name)))
    cfg = CheckerConfig(model="Kip320", n_replicas=g["N"], log_size=g["L"], max_records=g["R"],
                        max_leader_epoch=g["E"], invariants=("TypeOk", "WeakIsr", "StrongIsr"),
                        table_capacity=1 << 30, frontier_capacity=1 << 26)
    with ModelChecker(cfg) as mc:
        r = mc.run()
        r2 = mc.run()  # idempotent: a second run on the same handle gives the same answer
    assert r.verdict == "ok" and r.queue_left == 0
    assert (r.distinct, r.generated, r.depth, r.levels) == (g["distinct"], g["generated"], g["depth"], g["levels"])
    assert list(r.action_generated.values()) == g["action_generated"][:9]
    assert r.deadlock_states == g["deadlock_states"]
    assert (r2.distinct, r2.generated, r2.levels) == (r.distinct, r.generated, r.levels)
    # size-independent properties: levels partition the reachable set; generated = 1 + sum over actions
    assert sum(r.levels) == r.distinct and sum(r.action_generated.values()) + 1 == r.generated


def test_truncate_to_hw_binding_of_the_headline_against_golden_fixture():
    """SURVEY §8d's second binding of "KafkaReplication, 3 brokers": KafkaTruncateToHighWatermark (Next built purely from
    KafkaReplication.tla's actions, KafkaTruncateToHighWatermark.tla:33-42) with TypeOk, at the largest LogSize the exact
    CPU oracle still holds in RAM: 221,065,990 distinct states, 933 M generated, 736,602 deadlocked states."""
    g = json.load(open(os.path.join(GOLDEN, "oracle_thw_3_5_5_2.json")))
    cfg = CheckerConfig(model="KafkaTruncateToHighWatermark", n_replicas=g["N"], log_size=g["L"], max_records=g["R"],
                        max_leader_epoch=g["E"], invariants=("TypeOk",), table_capacity=1 << 30, frontier_capacity=1 << 26)
    with ModelChecker(cfg) as mc:
        r = mc.run()
    assert r.verdict == "ok" and r.queue_left == 0
    assert (r.distinct, r.generated, r.depth, r.levels) == (g["distinct"], g["generated"], g["depth"], g["levels"])
    assert list(r.action_generated.values()) == g["action_generated"][:len(r.action_generated)]
    assert r.deadlock_states == g["deadlock_states"]
    assert sum(r.levels) == r.distinct and sum(r.action_generated.values()) + 1 == r.generated


def test_headline_seed_independence_at_medium_size():
    cfgs = [CheckerConfig(model="Kip320", n_replicas=3, log_size=4, max_records=4, max_leader_epoch=2,
                          table_capacity=1 << 27, frontier_capacity=1 << 23, hash_seed=s) for s in (0, 12345)]
    res = []
    for c in cfgs:
        with ModelChecker(c) as mc:
            res.append(mc.run())
    assert res[0].distinct == res[1].distinct == 18731224
    assert res[0].levels == res[1].levels and res[0].generated == res[1].generated == 56197186   # 55,208,512 probed successors + 988,674 double disjuncts (Kip320.tla:82-83)


def test_cli_prints_tlc_shaped_output(capsys):
    from kafka_specification_amd import tlc
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rc = tlc.main([os.path.join(root, "models", "FiniteReplicatedLog.tla"), "-table", "1048576", "-frontier", "262144"])
    out = capsys.readouterr().out
    assert rc == 0 and "Model checking completed. No error has been found." in out
    assert "distinct states found, 0 states left on queue." in out and "116281 distinct states found" in out
    assert "The depth of the complete state graph search is" in out
    rc = tlc.main([os.path.join(root, "models", "Kip279.tla"), "-table", "4194304", "-frontier", "1048576"])
    out = capsys.readouterr().out
    o = kmo.Run(kmo.make_config("Kip279", N=3, L=2, R=2, E=2, invariants=("TypeOk", "WeakIsr", "StrongIsr")))
    assert rc == 12 and f"Error: Invariant {o.viol_inv} is violated." in out
    assert f"State {o.viol_depth}: <" in out and f"State {o.viol_depth + 1}: <" not in out
    assert "Error: The behavior up to this point is:" in out and "State 1: <Initial predicate>" in out
    assert "/\\ quorumState = [leaderEpoch |-> -1, leader |-> \"NONE\", isr |-> {b1, b2, b3}]" in out
    rc = tlc.main([os.path.join(root, "models", "LeaderInIsr.tla"), "-config", os.path.join(root, "models", "LeaderInIsr.cfg"),
                   "-table", "65536", "-frontier", "4096"])
    assert rc == 2  # module name LeaderInIsr has no lowered model: the cfg must be paired with Kip320.tla
    rc = tlc.main([os.path.join(root, "models", "Kip320.tla"), "-config", os.path.join(root, "models", "LeaderInIsr.cfg"),
                   "-table", "65536", "-frontier", "4096"])
    out = capsys.readouterr().out
    assert rc == 12 and "Error: Invariant LeaderInIsr is violated by the initial state." in out
    assert "1 states generated, 1 distinct states found" in out


@pytest.mark.parametrize("exchange", ["rccl", "torch"])
def test_rccl_single_rank_process_group_matches_oracle(exchange, monkeypatch):
    """A real RCCL communicator (world_size 1 is all one GPU allows).  "rccl": the exchange under the C ABI —
    kmc_comm_init on a unique id that travelled through the process group, kmc_comm_selftest (an all-gather and a
    grouped send/receive to itself on the engine's stream, verified: with one rank the search itself has no remote
    traffic), then the level loop through kmc_step_exchange_counts / _payload.  "torch": DistExchange."""
    monkeypatch.setenv("KMC_EXCHANGE", exchange)
    import torch
    import torch.distributed as dist
    from kafka_specification_amd.sharded import check_distributed
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:   # a free port: under pytest-xdist the two parametrisations run side by side
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            monkeypatch.setenv("MASTER_PORT", str(sk.getsockname()[1]))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        torch.cuda.set_device(0)
        o = kmo.Run(kmo.make_config("Kip320", N=3, L=3, R=3, E=2, invariants=("TypeOk",), threads=8))
        cfg = CheckerConfig(model="Kip320", n_replicas=3, log_size=3, max_records=3, max_leader_epoch=2,
                            table_capacity=1 << 24, frontier_capacity=1 << 21, send_capacity=1 << 22)
        r = check_distributed(cfg)
        assert (r.verdict, r.distinct, r.generated, r.depth) == (o.verdict, o.distinct, o.generated, o.depth)
        assert r.levels == o.levels
        if exchange == "rccl":
            from kafka_specification_amd.sharded import HipShardEngine, RcclExchange
            small = CheckerConfig(model="Kip320", n_replicas=2, log_size=2, max_records=2, max_leader_epoch=1,
                                  table_capacity=1 << 16, frontier_capacity=1 << 12)
            eng = HipShardEngine(small, 0, 1, 0)
            try:
                RcclExchange(eng, torch.device("cuda", 0)).selftest()
            finally:
                eng.close()
    finally:
        if created:
            dist.destroy_process_group()


def test_native_cli_matches_python_cli(capsys):
    """The C++ front end (kafka_specification_amd/tlc) over the same C ABI prints the same verdict,
    counts and trace states as the Python one."""
    import subprocess
    from kafka_specification_amd import tlc
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "kafka_specification_amd", "tlc")
    args = [os.path.join(root, "models", "Kip101.tla"), "-table", "4194304", "-frontier", "1048576"]
    rc_py = tlc.main(args)
    out_py = capsys.readouterr().out
    r = subprocess.run([exe] + args, capture_output=True, text=True)
    assert r.returncode == rc_py == 12

    def digest(text):
        lines = text.splitlines()
        verdict = [l for l in lines if l.startswith(("Error:", "The depth"))
                   or ("states generated," in l and not l.startswith("Progress"))]
        heads = [i for i, l in enumerate(lines) if l.startswith("State ")]
        last_state = lines[heads[-1] + 1:heads[-1] + 7]   # the witness (smallest violating fingerprint): deterministic
        return verdict, len(heads), lines[heads[0] + 1:heads[0] + 7], last_state
    # same verdict, counts, trace length, initial and final state; the path in between may differ
    # (which of several same-depth predecessors is recorded depends on who won the claim)
    assert digest(r.stdout) == digest(out_py)
    r = subprocess.run([exe, os.path.join(root, "models", "FiniteReplicatedLog.tla"), "-table", "1048576",
                        "-frontier", "262144"], capture_output=True, text=True)
    assert r.returncode == 0 and "1190091 states generated, 116281 distinct states found, 0 states left on queue." in r.stdout


def test_timing_of_a_handle_and_the_clis_account_of_their_wall_time(capsys):
    """kmc_timing (round 5): where a handle's wall time went besides the search — filled by kmc_open and the first kmc_run —
    and the closing line both front ends print under -v (bench.py's cold_start block parses the native one)."""
    import re
    import subprocess
    from kafka_specification_amd import tlc
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = CheckerConfig(model="Kip320", n_replicas=3, log_size=2, max_records=2, max_leader_epoch=1, invariants=("TypeOk",),
                        table_capacity=1 << 22, frontier_capacity=1 << 20)
    with ModelChecker(cfg) as mc:
        t0 = mc.timing()
        assert t0["open_s"] > 0 and t0["first_clear_s"] == 0.0          # nothing has run yet
        assert t0["open_s"] >= t0["hip_init_s"] + t0["code_object_s"] + t0["alloc_s"] - 1e-6
        assert t0["device_bytes"] >= (1 << 22) * 8 + 2 * (1 << 20) * 8 * mc.state_words
        mc.run()
        t1 = mc.timing()
        assert t1["first_clear_s"] > 0 and t1["open_s"] == t0["open_s"]
        mc.run()
        assert mc.timing()["first_clear_s"] == t1["first_clear_s"]    # the FIRST clear: fresh memory is touched once
    exe = os.path.join(root, "kafka_specification_amd", "tlc")
    args = [os.path.join(root, "models", "FiniteReplicatedLog.tla"), "-table", "1048576", "-frontier", "262144", "-v"]
    r = subprocess.run([exe] + args, capture_output=True, text=True)
    assert r.returncode == 0
    m = re.search(r"Wall time: ([\d.]+)s in this process = ([\d.]+)s before kmc_open .* \+ ([\d.]+)s kmc_open \(HIP initialisation "
                  r"([\d.]+)s, code object ([\d.]+)s, allocation of ([\d.]+) GiB ([\d.]+)s\) \+ ([\d.]+)s kmc_run \(first clear of the "
                  r"seen-set ([\d.]+)s, search ([\d.]+)s\) \+ ([\d.]+)s verdict / trace \+ ([\d.]+)s teardown", r.stdout)
    assert m, r.stdout[-600:]
    v = [float(x) for x in m.groups()]
    assert abs(v[0] - (v[1] + v[2] + v[7] + v[10] + v[11])) < 0.01 and v[2] >= v[3] + v[4] + v[6] - 0.002
    assert "Wall time" not in subprocess.run([exe] + args[:-1], capture_output=True, text=True).stdout   # only under -v
    assert tlc.main(args) == 0
    assert "Wall time outside the search: kmc_open" in capsys.readouterr().out


def test_both_clis_recommend_wide_entries_when_the_birthday_bound_is_large(capsys):
    """At 6.45 G states (Kip320 3/6/6/3) the 64-bit search returns one state fewer than the exact count (n^2 / 2^65 = 1.1):
    from a bound of 0.1 on, the closing estimate comes with the advice to re-run with -fp128 (or -symmetry)."""
    from kafka_specification_amd.tlc import FP128_ADVICE_ABOVE, collision_report
    big = collision_report(6452700520, 20756484505)
    assert any("Recommendation" in ln and "-fp128" in ln for ln in big) and FP128_ADVICE_ABOVE == 0.1
    assert not any("Recommendation" in ln for ln in collision_report(279753922, 901914892))        # the headline: 2e-3
    assert not any("Recommendation" in ln for ln in collision_report(6452700520, 20756484505, True))   # already wide


def test_overfull_table_stops_quickly():
    """A table far too small must end in `table_full` fast (bounded probe chains), not crawl."""
    import time
    t0 = time.time()
    with ModelChecker(CheckerConfig(model="Kip320", n_replicas=3, log_size=4, max_records=4, max_leader_epoch=2,
                                    table_capacity=1 << 20, frontier_capacity=1 << 23)) as mc:
        r = mc.run()
    assert r.verdict == "table_full" and time.time() - t0 < 20


def test_checkpoint_and_recover(tmp_path):
    """TLC -checkpoint / -recover analogue: stop at a level limit, save, load into a fresh handle,
    resume: every number equals the uninterrupted run's."""
    base = dict(model="Kip320", n_replicas=3, log_size=3, max_records=3, max_leader_epoch=1,
                invariants=("TypeOk", "WeakIsr", "StrongIsr"), table_capacity=1 << 20, frontier_capacity=1 << 17)
    with ModelChecker(CheckerConfig(**base)) as mc:
        full = mc.run()
    path = str(tmp_path / "kip320.ckpt")
    with ModelChecker(CheckerConfig(**base, max_levels=11)) as mc:
        part = mc.run()
        assert part.verdict == "level_limit" and part.depth == 11 and part.queue_left == full.levels[10]
        assert part.levels == full.levels[:11]
        mc.save_checkpoint(path)
    with ModelChecker(CheckerConfig(**base)) as mc:
        mc.load_checkpoint(path)
        rest = mc.resume()
        init = mc.pack(kmo.Run(kmo.make_config("Kip320", N=3, L=3, R=3, E=1, max_states=1)).state(0))
        assert mc.contains(init)                         # FPSet.contains analogue
        assert not mc.contains([x ^ 0x5555 for x in init])
    assert (rest.verdict, rest.distinct, rest.generated, rest.depth, rest.levels, rest.queue_left) == \
        (full.verdict, full.distinct, full.generated, full.depth, full.levels, 0)
    assert rest.action_generated == full.action_generated and rest.deadlock_states == full.deadlock_states
    # a checkpoint only fits a handle with the same constants and capacities
    from kafka_specification_amd import KmcError
    with ModelChecker(CheckerConfig(**{**base, "log_size": 2})) as mc:
        with pytest.raises(KmcError):
            mc.load_checkpoint(path)
    with ModelChecker(CheckerConfig(**{**base, "table_capacity": 1 << 21})) as mc:
        with pytest.raises(KmcError):
            mc.load_checkpoint(path)
    # the file is not trusted: a segment size beyond the capacity, a level count that would size a huge
    # vector, or a truncated body are refused before anything is copied to the device
    import struct
    raw = bytearray(open(path, "rb").read())
    from kafka_specification_amd import _native as nat
    import ctypes as C
    cfg_bytes = C.sizeof(nat.KmcConfig)
    hdr = 8 + cfg_bytes + 8 * 8                 # magic + kmc_config + 8 header words
    nlev = struct.unpack_from("<Q", raw, 8 + cfg_bytes + 5 * 8)[0]
    assert nlev == 11
    seg_off = hdr + C.sizeof(nat.KmcResult) + 8 * nlev
    bad = bytearray(raw)
    struct.pack_into("<Q", bad, seg_off, 1 << 40)          # seg_n[0]: far beyond seg_cap
    (tmp_path / "bad_seg.ckpt").write_bytes(bad)
    bad = bytearray(raw)
    struct.pack_into("<Q", bad, 8 + cfg_bytes + 5 * 8, 1 << 50)  # n_levels
    (tmp_path / "bad_levels.ckpt").write_bytes(bad)
    (tmp_path / "short.ckpt").write_bytes(raw[:len(raw) // 2])
    for name in ("bad_seg.ckpt", "bad_levels.ckpt", "short.ckpt"):
        with ModelChecker(CheckerConfig(**base)) as mc:
            with pytest.raises(KmcError):
                mc.load_checkpoint(str(tmp_path / name))
    # ... and a checkpoint is only taken at a level boundary: after a stop inside a level the table already holds the
    # rolled-back level's fingerprints, a resume from there would silently lose states
    with ModelChecker(CheckerConfig(**{**base, "table_capacity": 4096})) as mc:
        assert mc.run().verdict == "table_full"
        with pytest.raises(KmcError):
            mc.save_checkpoint(str(tmp_path / "nope.ckpt"))
    with ModelChecker(CheckerConfig(**base)) as mc:
        assert mc.run().verdict == "ok"
        with pytest.raises(KmcError):
            mc.save_checkpoint(str(tmp_path / "nope.ckpt"))


def test_level_limit_still_checks_the_last_frontier():
    """With max_levels the last frontier is not expanded; its states still get their invariant check."""
    inv = ("TypeOk", "StrongIsr")
    o = kmo.Run(kmo.make_config("Kip279", N=3, L=2, R=2, E=2, invariants=inv))
    assert o.verdict == "invariant"
    cfg = dict(model="Kip279", n_replicas=3, log_size=2, max_records=2, max_leader_epoch=2, invariants=inv,
               table_capacity=1 << 22, frontier_capacity=1 << 20)
    with ModelChecker(CheckerConfig(**cfg, max_levels=o.viol_depth)) as mc:   # stop exactly at the violating level
        r = mc.run()
    assert (r.verdict, r.violated_invariant, r.violation_depth) == ("invariant", o.viol_inv, o.viol_depth)
    assert r.violation_count == o.viol_count and r.levels == o.levels
    with ModelChecker(CheckerConfig(**cfg, max_levels=o.viol_depth - 1)) as mc:  # one level earlier: nothing found yet
        r = mc.run()
    assert r.verdict == "level_limit" and r.violated_invariant is None and r.depth == o.viol_depth - 1


def test_sender_side_filter_drops_duplicates_but_not_states():
    """The sharded path ships each remote state once per shard that generates it, not once per
    generation: same results with and without the filter, and the filter really drops copies."""
    from kafka_specification_amd import sharded
    cfg = CheckerConfig(model="Kip320", n_replicas=3, log_size=3, max_records=3, max_leader_epoch=1,
                        invariants=("TypeOk",), table_capacity=1 << 20, frontier_capacity=1 << 18,
                        send_capacity=1 << 17)
    o = kmo.Run(kmo.make_config("Kip320", N=3, L=3, R=3, E=1))
    r = check_loopback(cfg, 3)
    dropped = sharded.run_sharded.last_send_filtered
    assert (r.distinct, r.generated, r.levels) == (o.distinct, o.generated, o.levels)
    assert dropped > o.generated // 10        # g = 2.8 here: most remote copies are duplicates
    os.environ["KMC_NO_SEND_FILTER"] = "1"
    try:
        r2 = check_loopback(cfg, 3)
        assert sharded.run_sharded.last_send_filtered == 0
    finally:
        del os.environ["KMC_NO_SEND_FILTER"]
    assert (r2.distinct, r2.generated, r2.levels) == (r.distinct, r.generated, r.levels)


def test_eight_logical_shards_default_filter_policy():
    """KMC_MAX_SHARDS logical shards: the largest plan the exchange supports, and the side of the sender-filter policy
    (on for P <= 4, off beyond: profiles/r02_loopback_filter.jsonl) that the other loopback tests do not reach."""
    from kafka_specification_amd import sharded
    cfg = CheckerConfig(model="Kip320", n_replicas=3, log_size=3, max_records=3, max_leader_epoch=1,
                        invariants=("TypeOk", "WeakIsr", "StrongIsr"), table_capacity=1 << 19, frontier_capacity=1 << 17,
                        send_capacity=1 << 15)
    o = kmo.Run(kmo.make_config("Kip320", N=3, L=3, R=3, E=1, invariants=("TypeOk", "WeakIsr", "StrongIsr")))
    r = check_loopback(cfg, 8)
    assert (r.verdict, r.distinct, r.generated, r.levels) == (o.verdict, o.distinct, o.generated, o.levels)
    assert sharded.run_sharded.last_send_filtered == 0      # P = 8: no sender-side filter by default
    os.environ["KMC_SEND_FILTER"] = "1"
    try:
        r2 = check_loopback(cfg, 8)
        assert sharded.run_sharded.last_send_filtered > 0
    finally:
        del os.environ["KMC_SEND_FILTER"]
    assert (r2.distinct, r2.generated, r2.levels) == (r.distinct, r.generated, r.levels)


@pytest.mark.parametrize("P", [2, 3])
@pytest.mark.parametrize("model", ["Kip101", "Kip320FirstTry"])
def test_counterexample_trace_across_shards(P, model):
    """keep_trace with P shards: predecessor fingerprints travel with the exchanged records, the chain is
    walked owner by owner (kmc_pred_of) and replayed with the device's successor enumeration."""
    N, L, R, E = 3, 2, 2, 2
    inv = ("TypeOk", "StrongIsr")
    ocfg = kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=inv)
    o = kmo.Run(ocfg)
    assert o.verdict == "invariant"
    cfg = CheckerConfig(model=model, n_replicas=N, log_size=L, max_records=R, max_leader_epoch=E, invariants=inv,
                        keep_trace=True, table_capacity=1 << 20, frontier_capacity=1 << 18, send_capacity=1 << 18)
    r = check_loopback(cfg, P)
    assert (r.verdict, r.violated_invariant, r.violation_depth) == ("invariant", o.viol_inv, o.viol_depth)
    assert r.levels == o.levels and r.generated == o.generated
    trace = r.trace
    assert len(trace) == o.viol_depth and trace[0] == (None, o.state(0))
    with ModelChecker(CheckerConfig(model=model, n_replicas=N, log_size=L, max_records=R, max_leader_epoch=E,
                                    device=-1)) as mc:
        names = mc.action_names()
    for (_, prev), (act, cur) in zip(trace, trace[1:]):
        assert (names.index(act), cur) in kmo.successors(ocfg, prev, o.sb)
        assert all(kmo.check_invariant(ocfg, INV_INDEX[i], prev) for i in inv)
    assert not kmo.check_invariant(ocfg, INV_INDEX[o.viol_inv], trace[-1][1])


@pytest.mark.parametrize("P", [2, 3])
def test_sharded_level_limit_checks_the_last_frontier(P):
    """Sharded and single-GPU runs must agree at max_levels: the unexpanded last level gets its invariant pass
    (kmc_step_check_frontier), so a violation that first appears there is reported — with its trace — instead of
    "level_limit" (VERDICT r1 2c)."""
    model, inv = "KafkaTruncateToHighWatermark", ("TypeOk", "StrongIsr")
    ocfg = kmo.make_config(model, N=3, L=2, R=2, E=1, invariants=inv)
    o = kmo.Run(ocfg)
    assert o.verdict == "invariant"
    base = dict(model=model, n_replicas=3, log_size=2, max_records=2, max_leader_epoch=1, invariants=inv, keep_trace=True,
                table_capacity=1 << 20, frontier_capacity=1 << 18, send_capacity=1 << 18)
    with ModelChecker(CheckerConfig(**base, max_levels=o.viol_depth)) as mc:
        one = mc.run()
    r = check_loopback(CheckerConfig(**base, max_levels=o.viol_depth), P)
    assert (r.verdict, r.violated_invariant, r.violation_depth, r.violation_count) == \
        (one.verdict, one.violated_invariant, one.violation_depth, one.violation_count) == \
        ("invariant", o.viol_inv, o.viol_depth, o.viol_count)
    assert r.levels == one.levels == o.levels[:o.viol_depth]
    assert len(r.trace) == o.viol_depth and not kmo.check_invariant(ocfg, INV_INDEX[o.viol_inv], r.trace[-1][1])
    r = check_loopback(CheckerConfig(**base, max_levels=o.viol_depth - 1), P)
    assert (r.verdict, r.violated_invariant, r.levels) == ("level_limit", None, o.levels[:o.viol_depth - 1])


def test_baseline_config4_kip279_five_brokers_exhaustive_and_four_shards():
    """BASELINE.json config 4 — "Kip279.tla leader-epoch truncation, 5 brokers, 4 x MI355X frontier-sharded" — at the
    constants of models/Kip279_5brokers.cfg (LogSize 2, MaxRecords 2, MaxLeaderEpoch 1: the largest of the ladder
    that can be exhausted).  One GPU and four fingerprint-sharded logical GPUs (exchange under the C ABI; one gpurun box
    has one device) both reproduce the golden fixture of the C oracle, level by level: 112,549,196 distinct states.
    With StrongIsr in the invariants the search stops at the violation Kip279.tla:20-23 describes, depth 14."""
    g = json.load(open(os.path.join(GOLDEN, "oracle_kip279_5_2_2_1.json")))
    base = dict(model="Kip279", n_replicas=g["N"], log_size=g["L"], max_records=g["R"], max_leader_epoch=g["E"],
                invariants=("TypeOk",))
    with ModelChecker(CheckerConfig(**base, table_capacity=1 << 29, frontier_capacity=1 << 25)) as mc:
        r = mc.run()
    assert r.verdict == "ok" and r.queue_left == 0
    assert (r.distinct, r.generated, r.depth, r.levels) == (g["distinct"], g["generated"], g["depth"], g["levels"])
    assert list(r.action_generated.values()) == g["action_generated"][:9]   # Kip279's coinciding disjuncts counted twice
    assert r.deadlock_states == g["deadlock_states"]
    s = check_loopback(CheckerConfig(**base, table_capacity=1 << 27, frontier_capacity=1 << 23, send_capacity=1 << 21), 4)
    assert (s.verdict, s.distinct, s.generated, s.depth, s.levels) == ("ok", g["distinct"], g["generated"], g["depth"], g["levels"])
    assert list(s.action_generated.values()) == g["action_generated"][:9]
    inv = ("TypeOk", "WeakIsr", "StrongIsr")
    o = kmo.Run(kmo.make_config("Kip279", N=g["N"], L=g["L"], R=g["R"], E=g["E"], invariants=inv, threads=8))
    with ModelChecker(CheckerConfig(**{**base, "invariants": inv}, table_capacity=1 << 25, frontier_capacity=1 << 22)) as mc:
        v = mc.run()
    assert (v.verdict, v.violated_invariant, v.violation_depth, v.violation_count) == \
        ("invariant", "StrongIsr", 14, o.viol_count) == (o.verdict, o.viol_inv, o.viol_depth, o.viol_count)
    assert v.levels == o.levels


@pytest.mark.parametrize("P", [2, 3])
def test_sharded_checkpoint_and_recover(tmp_path, P):
    """Multi-GPU checkpoints (the gap DESIGN §8 listed after round 1): P shards stop at max_levels, every shard saves
    its own fingerprint table and frontier (kmc_checkpoint_save on a stepping handle) and the driver its global
    counters; fresh shards load them (kmc_checkpoint_load + kmc_step_resume) and finish with the numbers of the
    uninterrupted run.  A shard file does not fit another shard id."""
    base = dict(model="Kip320", n_replicas=3, log_size=3, max_records=3, max_leader_epoch=1,
                invariants=("TypeOk", "WeakIsr", "StrongIsr"), table_capacity=1 << 20, frontier_capacity=1 << 17,
                send_capacity=1 << 16)
    o = kmo.Run(kmo.make_config("Kip320", N=3, L=3, R=3, E=1, invariants=base["invariants"], threads=8))
    ckpt = str(tmp_path / "ckpt")
    part = check_loopback(CheckerConfig(**base, max_levels=11), P, checkpoint_dir=ckpt)
    assert part.verdict == "level_limit" and part.levels == o.levels[:11]
    assert sorted(os.listdir(ckpt)) == ["driver.json"] + [f"shard{i}of{P}.ckpt" for i in range(P)]
    rest = check_loopback(CheckerConfig(**base), P, resume_dir=ckpt)
    assert (rest.verdict, rest.distinct, rest.generated, rest.depth, rest.levels) == \
        (o.verdict, o.distinct, o.generated, o.depth, o.levels)
    assert list(rest.action_generated.values()) == o.action_generated[:len(rest.action_generated)]
    assert rest.deadlock_states == o.deadlock_states
    from kafka_specification_amd import KmcError
    os.replace(os.path.join(ckpt, f"shard0of{P}.ckpt"), os.path.join(ckpt, "tmp"))
    os.replace(os.path.join(ckpt, f"shard1of{P}.ckpt"), os.path.join(ckpt, f"shard0of{P}.ckpt"))
    os.replace(os.path.join(ckpt, "tmp"), os.path.join(ckpt, f"shard1of{P}.ckpt"))
    with pytest.raises(KmcError):
        check_loopback(CheckerConfig(**base), P, resume_dir=ckpt)      # shard files swapped: refused


def test_baseline_config5_seven_brokers_on_eight_logical_shards():
    """BASELINE.json config 5 (Kip320, 7 brokers, LogSize 8, MaxRecords 8, MaxLeaderEpoch 3: W = 10 words (grouped replica-major layout), 11-word exchange
    records with keep_trace) through the exchange under the ABI on P = 8 logical shards of one GPU.  The configuration is
    not exhaustible (8.8e8 states after 11 levels), so the pin is the oracle's prefix: the sizes of the first 7 levels,
    per-action generated counts, and the exact state SETS of the levels up to 50 k states, gathered over the shards."""
    from kafka_specification_amd.sharded import HipShardEngine, NativeLoopbackExchange, run_sharded
    inv, K, P = ("TypeOk",), 7, 8
    o = kmo.Run(kmo.make_config("Kip320", N=7, L=8, R=8, E=3, invariants=inv, threads=8, max_states=1_500_000))
    o2 = kmo.Run(kmo.make_config("Kip320", N=7, L=8, R=8, E=3, invariants=inv, threads=8, max_states=sum(o.levels[:K - 1]) + 1))
    assert len(o.levels) > K and len(o2.levels) == K
    cfg = CheckerConfig(model="Kip320", n_replicas=7, log_size=8, max_records=8, max_leader_epoch=3, invariants=inv,
                        keep_trace=True, max_levels=K, table_capacity=1 << 22, frontier_capacity=1 << 20, send_capacity=1 << 16)
    engines = [HipShardEngine(cfg, s, P, 0, native=True) for s in range(P)]
    sets = {}
    try:
        assert engines[0].record_words == 11 and engines[0].W == 10

        def level_done(info):      # the shards still hold the level just recorded (its expansion is in flight, not finished)
            if info["new_states"] <= 50_000:
                got = set()
                for e in engines:
                    got |= {e.mc.unpack(row) for row in e.mc.frontier_states()}
                sets[info["depth"]] = got
        res = run_sharded(engines, NativeLoopbackExchange(engines), cfg, engines[0].mc.action_names(), level_done)
    finally:
        for e in engines:
            e.close()
    assert res.verdict == "level_limit" and res.levels == o.levels[:K]
    assert res.generated == o2.generated
    assert list(res.action_generated.values()) == o2.action_generated[:len(res.action_generated)]
    assert sorted(sets) == [1, 2, 3, 4, 5]
    for d, got in sets.items():
        assert got == o.level_states(d - 1), f"level {d}: state sets differ"
