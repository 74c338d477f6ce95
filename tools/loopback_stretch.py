"""Kip320 3/6/6/3 (6,452,700,520 states, 1.46 s on one GPU: the configuration where sharding pays) through the sharded path
with P logical shards on ONE GPU, exchange under the C ABI (kmc_step_exchange_local / _deliver_local: the same plan RCCL
executes, with device-to-device copies).  One device does every shard's work one after the other, so the wall time says
nothing about P GPUs; what the run gives is PER SHARD: seconds inside k_expand, states owned, bytes received per level —
the inputs of DESIGN.md §6's projection for 2 / 4 / 8 real GPUs.
usage: python tools/loopback_stretch.py [P ...]      (default 8; P = 2 and 4 need the sender-side filter's memory too)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402  (first: one HIP runtime per process)
from kafka_specification_amd import CheckerConfig, sharded  # noqa: E402

WANT = dict(distinct=6452700520, generated=20756484505, depth=54)   # three hash seeds with 128-bit entries (profiles/r03_fp128.txt)
os.environ["KMC_EXCHANGE"] = "rccl"
for P in ([int(x) for x in sys.argv[1:]] or [8]):
    cfg = CheckerConfig(model="Kip320", n_replicas=3, log_size=6, max_records=6, max_leader_epoch=3,
                        invariants=("TypeOk", "WeakIsr", "StrongIsr"), table_capacity=(1 << 33) // P,
                        frontier_capacity=(1 << 30) // P * 5 // 4, send_capacity=(1 << 28) // (P * P))
    engines = [sharded.HipShardEngine(cfg, s, P, 0, native=True) for s in range(P)]
    ex = sharded.NativeLoopbackExchange(engines)
    level_new = []
    try:
        t = time.time()
        r = sharded.run_sharded(engines, ex, cfg, engines[0].action_names(), lambda info: level_new.append(info["new_states"]))
        dt = time.time() - t
        per_shard = [e.result() for e in engines]
    finally:
        for e in engines:
            e.close()
    ok = (r.distinct, r.generated, r.depth) == (WANT["distinct"], WANT["generated"], WANT["depth"])
    # a state reaches its owner once per shard that generates it (no sender-side filter beyond P = 4): remote successors
    # = generated x (P - 1) / P; 24 bytes each
    print(json.dumps(dict(config="Kip320 3/6/6/3", shards=P, verdict=r.verdict, distinct=r.distinct, generated=r.generated,
                          matches_the_single_gpu_run=ok, wall_seconds_all_shards_on_one_gpu=round(dt, 3),
                          expand_seconds_per_shard=[round(x.seconds_expand, 4) for x in per_shard],
                          expand_seconds_max_shard=round(max(x.seconds_expand for x in per_shard), 4),
                          expand_seconds_sum=round(sum(x.seconds_expand for x in per_shard), 4),
                          states_owned_per_shard=[x.distinct for x in per_shard],
                          send_filtered=sharded.run_sharded.last_send_filtered,
                          remote_successor_bytes_per_shard_estimate=int((r.generated - r.generated_repeats) * (P - 1) / P / P * 24),
                          widest_level=max(r.levels), levels=len(r.levels))), flush=True)
