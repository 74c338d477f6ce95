"""bench.py's N>1 launch path on a CPU box: `python bench.py --gpus 2` must turn itself into a
torch.distributed.run job (it crashed at init_process_group without RANK), rendezvous on
127.0.0.1, run the sharded level loop on both ranks, and print ONE JSON line from rank 0.  There is no GPU
here, so the process group is gloo and the shard engine is the oracle-backed stand-in
(KMC_SHARD_ENGINE=shard_standin:make_engine); with the default backend the same command must fail loudly,
not fall back to a CPU path."""
import json
import os
import subprocess
import sys

import kmo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=300):
    env = dict(os.environ)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    env["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "tests"), env.get("PYTHONPATH", "")])
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          env=env, timeout=timeout, cwd=ROOT)


def test_bench_gpus2_self_launches_and_prints_one_line():
    p = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--no-cpu-baseline",
              "--workload", "Kip320,2,2,2,2"], {"KMC_SHARD_ENGINE": "shard_standin:make_engine"})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout                       # rank 0 only
    r = json.loads(lines[0])
    o = kmo.Run(kmo.make_config("Kip320", N=2, L=2, R=2, E=2, invariants=("TypeOk", "WeakIsr", "StrongIsr")))
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["steps"] == 2 and r["warmup"] == 1
    assert r["config"]["distinct_states"] == o.distinct and r["config"]["states_generated"] == o.generated
    assert r["config"]["depth"] == o.depth and r["config"]["verdict"] == "ok"
    assert r["config"]["shards"] == 2
    # what the first curve on real GPUs is read against (round 5): which exchange ran, on how many ranks RCCL's self-test
    # passed (none here: gloo, a stand-in engine), how many levels took the within-level pipeline and from which size on
    assert r["config"]["exchange"] == "DistExchange" and r["config"]["rccl_ranks"] == 0
    assert r["config"]["pipelined_levels"] == 0 and "pipeline_min_states" in r["config"] and "pipeline_parts" in r["config"]
    assert r["value"] > 0 and r["ms_per_step"] > 0
    assert "cpu_baseline" not in r                         # N>1 lines carry no CPU leg


def test_bench_gpus2_stretch_leg_rides_beside_the_headline():
    """`bench.py --gpus N --stretch [WORKLOAD]` (round 5): the workload the frontier sharding is for (default Kip320 3/6/6/3,
    128-bit entries) as one more leg of the N > 1 line — here a small stand-in workload over gloo; never part of `value`."""
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--backend", "gloo", "--no-cpu-baseline",
              "--workload", "Kip320,2,2,2,2", "--stretch", "Kip320,3,2,2,1"], {"KMC_SHARD_ENGINE": "shard_standin:make_engine"})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    o = kmo.Run(kmo.make_config("Kip320", N=3, L=2, R=2, E=1, invariants=("TypeOk", "WeakIsr", "StrongIsr")))
    st = r["stretch"]
    assert "error" not in st, st
    assert (st["distinct_states"], st["states_generated"], st["depth"], st["verdict"]) == (o.distinct, o.generated, o.depth, "ok")
    assert st["shards"] == 2 and st["matches_the_exact_oracle"] is None and st["time_to_exhaustive_s"] > 0
    o2 = kmo.Run(kmo.make_config("Kip320", N=2, L=2, R=2, E=2, invariants=("TypeOk", "WeakIsr", "StrongIsr")))
    assert r["config"]["distinct_states"] == o2.distinct      # the headline leg is untouched


def test_bench_gpus2_with_orbit_counting_prints_the_plain_searchs_numbers():
    """`bench.py --gpus 2 --symmetry`: orbit counting on every rank (round 4: the level-step interface weighs its counters) —
    the line carries the PLAIN search's counts, the stored states beside them."""
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--backend", "gloo", "--no-cpu-baseline", "--symmetry",
              "--workload", "Kip320,3,2,2,1"], {"KMC_SHARD_ENGINE": "shard_standin:make_engine"})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    o = kmo.Run(kmo.make_config("Kip320", N=3, L=2, R=2, E=1, invariants=("TypeOk", "WeakIsr", "StrongIsr")))
    assert r["config"]["distinct_states"] == o.distinct and r["config"]["states_generated"] == o.generated
    assert r["config"]["depth"] == o.depth and r["config"]["verdict"] == "ok" and r["config"]["shards"] == 2
    assert 0 < r["config"]["stored_states"] < o.distinct and "orbit counting" in r["config"]["parallelism"]


def test_bench_gpus2_without_gpus_fails_loudly():
    # the product path (backend nccl, HipShardEngine): no device => every rank raises; nothing is printed as a result
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--workload", "Kip320,2,2,2,2"])
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert "no CPU fallback" in p.stderr or "HIP" in p.stderr or "nccl" in p.stderr.lower()


def test_tlc_front_end_under_torchrun_shards_over_the_ranks():
    """`python -m torch.distributed.run --nproc-per-node P -m kafka_specification_amd.tlc Spec.tla` — how BASELINE
    configs 4 and 5 are meant to run on P GPUs: one shard per rank, every rank computes the same global result, rank 0
    alone prints it (here: gloo + the stand-in engine)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, KMC_SHARD_ENGINE="shard_standin:make_engine", KMC_BACKEND="gloo",
               PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests")]))
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "kafka_specification_amd.tlc",
                        os.path.join(ROOT, "models", "Kip320FirstTry.tla"), "-deadlock"],
                       capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    out = p.stdout
    o = kmo.Run(kmo.make_config("Kip320FirstTry", N=3, L=2, R=2, E=2, invariants=("TypeOk", "WeakIsr", "StrongIsr")))
    assert out.count("Computing initial states...") == 1                      # rank 0 only
    assert f"Error: Invariant {o.viol_inv} is violated." in out
    assert f"State {o.viol_depth}: <" in out and f"State {o.viol_depth + 1}: <" not in out
    assert f"{o.generated} states generated, {o.distinct} distinct states found" in out
    assert p.returncode != 0                                                   # TLC's exit code 12 travels through the launcher


def test_a_collective_that_never_completes_ends_the_rank_with_a_message():
    """sharded._watchdog guards the communicator's creation and its self-test: a rank that would hang inside the library
    says what it was waiting for and exits with status 3 (the launcher then ends the others); when the call returns in
    time nothing happens."""
    import subprocess
    import sys
    code = ("import os, sys, time\nsys.path.insert(0, %r)\nos.environ['KMC_COLLECTIVE_TIMEOUT'] = '1'\n"
            "from kafka_specification_amd.sharded import _watchdog\n"
            "with _watchdog('a pretend collective', 5):\n    time.sleep(%s)\nprint('returned')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code % (root, 30)], capture_output=True, text=True, timeout=60)
    assert p.returncode == 3 and "a pretend collective did not complete within 1 s" in p.stderr and "returned" not in p.stdout
    p = subprocess.run([sys.executable, "-c", code % (root, 0)], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0 and "returned" in p.stdout
