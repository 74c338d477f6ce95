"""The headline configuration through the sharded path with P = 2, 4 and 8 logical shards on ONE GPU: counts and level
sizes against the committed golden fixture, once with the exchange under the C ABI (device-to-device runs + one k_insert
per shard and level) and once with the torch path (slices of a torch-owned send area routed in Python).  One device does
all shards' work one after the other, so the times say what the per-level machinery costs, not how P GPUs would scale.
usage: python tools/loopback_headline.py [P ...]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (first: one HIP runtime per process)
from kafka_specification_amd import CheckerConfig
from kafka_specification_amd.configs import HEADLINE
from kafka_specification_amd import sharded
g = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_kip320_3_6_6_2.json")))
for P in ([int(x) for x in sys.argv[1:]] or [2, 4, 8]):
    for exchange in os.environ.get("KMC_LOOPBACK_EXCHANGES", "rccl torch").split():
        os.environ["KMC_EXCHANGE"] = exchange
        cfg = CheckerConfig(**HEADLINE, table_capacity=(1 << 30) // P, frontier_capacity=(1 << 26) // P,
                            send_capacity=max(1 << 18, (1 << 25) // (P * P) * 2))
        engines = [sharded.HipShardEngine(cfg, s, P, 0, native=exchange == "rccl") for s in range(P)]
        ex = sharded.NativeLoopbackExchange(engines) if exchange == "rccl" else sharded.LoopbackExchange(P)
        try:
            sharded.run_sharded(engines, ex, cfg, engines[0].action_names())          # warm-up
            t = time.time()
            r = sharded.run_sharded(engines, ex, cfg, engines[0].action_names())
            dt = time.time() - t
        finally:
            for e in engines:
                e.close()
        ok = r.distinct == g["distinct"] and r.generated == g["generated"] and r.levels == g["levels"]
        print(json.dumps(dict(shards=P, exchange="under the C ABI" if exchange == "rccl" else "torch slices", verdict=r.verdict,
                              distinct=r.distinct, matches_golden=ok, seconds=round(dt, 4),
                              send_filtered=sharded.run_sharded.last_send_filtered,
                              sender_filter=not int(os.environ.get("KMC_NO_SEND_FILTER", "0") or 0),
                              seconds_expand_max_shard=round(r.seconds_expand, 4))), flush=True)
