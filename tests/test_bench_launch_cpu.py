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
    assert r["value"] > 0 and r["ms_per_step"] > 0
    assert "cpu_baseline" not in r                         # N>1 lines carry no CPU leg


def test_bench_gpus2_without_gpus_fails_loudly():
    # the product path (backend nccl, HipShardEngine): no device => every rank raises; nothing is printed as a result
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--workload", "Kip320,2,2,2,2"])
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert "no CPU fallback" in p.stderr or "HIP" in p.stderr or "nccl" in p.stderr.lower()
