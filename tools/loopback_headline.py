"""The headline configuration through the sharded path with P = 2 and 4 logical shards on ONE GPU (in-process
exchange): counts and level sizes against the committed golden fixture.  usage: python tools/loopback_headline.py"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kafka_specification_amd import CheckerConfig
from kafka_specification_amd.configs import HEADLINE
from kafka_specification_amd.sharded import check_loopback
g = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_kip320_3_6_6_2.json")))
for P in (2, 4):
    cfg = CheckerConfig(**HEADLINE, table_capacity=(1 << 30) // P, frontier_capacity=(1 << 26) // P,
                        send_capacity=(1 << 25) // (P * P) * 2)
    t = time.time()
    r = check_loopback(cfg, P)
    print(P, r.verdict, r.distinct, r.generated, r.depth, r.distinct == g["distinct"] and r.generated == g["generated"] and r.levels == g["levels"],
          round(time.time() - t, 2), "s; expand", round(r.seconds_expand, 3))
