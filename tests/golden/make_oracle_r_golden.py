#!/usr/bin/env python3
"""Writes tests/golden/oracle_r_ladder.json: the outputs of Oracle-R (oracle/tlar — the reference's own .tla text,
parsed and evaluated; no hand restatement involved) over a ladder of small configurations of every root module.

Needs /root/reference (so it runs in the build container, not on the GPU box); the fixture it writes is what carries
"the reference, executed" to the GPU box: tests/test_oracle_r_cpu.py holds the C oracle to it and
tests/test_gpu_oracle_r.py holds the HIP engine to it — counts, per-disjunct generated, verdicts, and a sha256 per BFS
level over the sorted canonical encodings of that level's states (tests/oracle_r_canon.py).

    python tests/golden/make_oracle_r_golden.py [--jobs 8] [--only small]

About 25 CPU-minutes on 8 cores for the small and medium entries; the large ones (round 4: up to 2.0 M states) take one to
three CPU-hours EACH, one process per entry.
"""
import argparse
import hashlib
import json
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

REFERENCE = "/root/reference"
KAFKA = ("KafkaTruncateToHighWatermark", "Kip101", "Kip279", "Kip320", "Kip320FirstTry")
KAFKA_INV = ("TypeOk", "WeakIsr", "StrongIsr")


def ladder(only=None):
    out = []
    for m in KAFKA:
        for c in [(2, 2, 2, 1), (2, 2, 2, 2), (2, 3, 3, 2), (3, 2, 2, 1), (3, 1, 1, 2)]:
            out.append(dict(module=m, N=c[0], L=c[1], R=c[2], E=c[3], invariants=KAFKA_INV, size="small" if c[0] == 2 else "medium"))
        out.append(dict(module=m, N=2, L=2, R=2, E=1, invariants=("LeaderInIsr",), stop=True, size="small"))
    out.append(dict(module="Kip320", N=3, L=2, R=2, E=2, invariants=KAFKA_INV, size="large"))
    out.append(dict(module="Kip320", N=3, L=2, R=3, E=2, invariants=KAFKA_INV, size="large"))   # round 4: 1,694,476 states, 1.8 CPU-hours
    for m in ("KafkaTruncateToHighWatermark", "Kip101", "Kip279", "Kip320FirstTry"):   # round 4: 1.4 - 2.0 M states, ~2 CPU-hours each
        out.append(dict(module=m, N=3, L=2, R=2, E=2, invariants=KAFKA_INV, size="large"))
    for m in KAFKA:   # round 4: three brokers with logs three deep, exhaustively (176 K - 310 K states, ~10 CPU-minutes each)
        out.append(dict(module=m, N=3, L=3, R=3, E=1, invariants=KAFKA_INV, size="large"))
    for M in (0, 10, 1000):
        out.append(dict(module="IdSequence", MaxId=M, invariants=("TypeOk",), size="small"))
    for K in (1, 2, 3):
        out.append(dict(module="FiniteReplicatedLog", N=2, L=4, K=K, invariants=("TypeOk",), size="small"))
    out.append(dict(module="FiniteReplicatedLog", N=2, L=4, K=4, invariants=("TypeOk",), size="medium"))  # BASELINE config 2
    for (N, MO, MV) in [(2, 2, 2), (3, 1, 2), (3, 2, 2), (2, 3, 1)]:
        out.append(dict(module="MCAsyncIsr", N=N, L=MO, E=MV, invariants=("ValidHighWatermark",), constraint="StateConstraint", size="small"))
    out.append(dict(module="MCAsyncIsr", N=3, L=3, E=4, invariants=("ValidHighWatermark",), constraint="StateConstraint", size="large"))  # models/MCAsyncIsr_small.cfg
    out.append(dict(module="MCAsyncIsr", N=2, L=2, E=2, invariants=("TypeOk",), constraint="StateConstraint", stop=True, size="small"))
    out.append(dict(module="MCAsyncIsr", N=3, L=2, E=2, invariants=("ValidHighWatermark", "LeaderOffsetInRange"), constraint="StateConstraint", size="small"))
    if only:
        out = [c for c in out if c["size"] in only]
    return out


def wide_ladder():
    """Four to seven replicas (tests/golden/oracle_r_wide.json): where the device's pass 1 is a straight-line block of hundreds of
    guards, states take several words and orbit counting sorts replica keys — and where BASELINE configs 4 (Kip279, 5 brokers)
    and 5 (7 brokers, LogSize 8) live.  Exhaustible bindings are run out; the two BASELINE bindings themselves, which Oracle-R
    cannot exhaust (112 M states / unbounded for practical purposes), are run over a LEVEL BUDGET (`max_levels`: exact level
    sets of the first levels only; `generated` then counts what the expanded levels produced)."""
    out = []
    for m in KAFKA:
        out.append(dict(module=m, N=4, L=1, R=1, E=0, invariants=KAFKA_INV, size="small"))
        out.append(dict(module=m, N=5, L=1, R=1, E=0, invariants=KAFKA_INV, size="medium"))
    out.append(dict(module="Kip320", N=4, L=2, R=1, E=1, invariants=KAFKA_INV, size="large"))     # 155,041 states
    out.append(dict(module="Kip320", N=6, L=1, R=1, E=0, invariants=KAFKA_INV, size="large"))     # 99,469 states
    out.append(dict(module="Kip279", N=5, L=2, R=2, E=1, invariants=KAFKA_INV, max_levels=10, size="large"))   # BASELINE config 4
    out.append(dict(module="Kip320", N=7, L=8, R=8, E=3, invariants=KAFKA_INV, max_levels=6, size="large"))    # BASELINE config 5
    out.append(dict(module="Kip320", N=7, L=1, R=1, E=0, invariants=KAFKA_INV, max_levels=11, size="large"))
    out.append(dict(module="Kip279", N=7, L=1, R=1, E=0, invariants=KAFKA_INV, max_levels=10, size="large"))
    out.append(dict(module="Kip320FirstTry", N=8, L=1, R=1, E=0, invariants=KAFKA_INV, max_levels=8, size="large"))   # the engine's widest
    out.append(dict(module="Kip320", N=7, L=1, R=1, E=0, invariants=KAFKA_INV, size="large"))   # run out: 681,871 states, an hour on one core
    return out


def constants_of(c):
    from oracle.tlar import ModelValue
    m = c["module"]
    if m == "IdSequence":
        return dict(MaxId=c["MaxId"])
    if m == "FiniteReplicatedLog":
        return dict(Replicas=frozenset(ModelValue(f"r{i + 1}") for i in range(c["N"])),
                    LogRecords=frozenset(ModelValue(f"x{i + 1}") for i in range(c["K"])), Nil=ModelValue("nil"), LogSize=c["L"])
    if m == "MCAsyncIsr":
        reps = [ModelValue(f"r{i + 1}") for i in range(c["N"])]
        return dict(Replicas=frozenset(reps), Leader=reps[0], MaxOffset=c["L"], MaxVersion=c["E"])
    return dict(Replicas=frozenset(ModelValue(f"b{i + 1}") for i in range(c["N"])), LogSize=c["L"], MaxRecords=c["R"],
                MaxLeaderEpoch=c["E"])


def run_one(c):
    from oracle.tlar import Checker
    import oracle_r_canon as oc
    consts = constants_of(c)
    ck = Checker(c["module"], consts, [os.path.join(ROOT, "models"), REFERENCE])
    t0 = time.time()
    r = ck.run(invariants=tuple(c["invariants"]), constraint=c.get("constraint"), stop_on_violation=bool(c.get("stop")),
               keep_states=True, max_levels=c.get("max_levels"))
    enc = oc.encoder_for(c["module"])
    digests = [oc.level_digest(enc(s, consts) for s in lv) for lv in r["level_states"]]
    v = r["violation"]
    e = dict(c)
    e.update(actions=[str(x) for x in ck.next_labels()], distinct=r["distinct"], generated=r["generated"], depth=r["depth"], levels=r["levels"],
             action_generated={str(k): n for k, n in r["action_generated"].items()}, deadlock_states=r["deadlock_states"],
             verdict=r["verdict"], outside_violations=r["outside_violations"], level_digests=digests,
             violation=None if v is None else dict(invariant=v["invariant"], depth=v["depth"], count_at_depth=v["count_at_depth"],
                                                   per_invariant=v["per_invariant"], trace_len=len(v["trace"]),
                                                   outside_constraint=bool(v.get("outside_constraint"))),
             seconds=round(time.time() - t0, 1))
    e["invariants"] = list(c["invariants"])
    return e


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--only", default=None, help="comma-separated sizes: small,medium,large")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "oracle_r_ladder.json"))
    ap.add_argument("--wide", action="store_true", help="the 4-7 replica ladder -> tests/golden/oracle_r_wide.json")
    a = ap.parse_args()
    if a.wide:
        cfgs = wide_ladder()
        if a.out.endswith("oracle_r_ladder.json"):
            a.out = a.out.replace("oracle_r_ladder.json", "oracle_r_wide.json")
    else:
        cfgs = ladder(a.only.split(",") if a.only else None)
    cfgs.sort(key=lambda c: {"large": 0, "medium": 1, "small": 2}[c["size"]])
    kept = []
    if a.wide and os.path.exists(a.out):   # incremental: entries of the file that the ladder still asks for are kept as they are
        ident = lambda c: (c["module"], c["N"], c["L"], c["R"], c["E"], c.get("max_levels"), tuple(c["invariants"]))
        want = {ident(c) for c in cfgs}
        kept = [e for e in json.load(open(a.out))["entries"] if ident(e) in want]
        have = {ident(e) for e in kept}
        cfgs = [c for c in cfgs if ident(c) not in have]
        print(f"{len(kept)} entries kept, {len(cfgs)} to run", flush=True)
    with ProcessPoolExecutor(max_workers=a.jobs) as ex:
        entries = kept + list(ex.map(run_one, cfgs))
    entries.sort(key=lambda e: (e["module"], e.get("N", 0), e.get("L", 0), e.get("R", 0), e.get("E", 0), e.get("K", 0),
                                e.get("MaxId", 0), e["invariants"]))
    sha = {}
    for fn in sorted(os.listdir(REFERENCE)):
        if fn.endswith(".tla"):
            sha[fn] = hashlib.sha256(open(os.path.join(REFERENCE, fn), "rb").read()).hexdigest()
    sha["models/MCAsyncIsr.tla"] = hashlib.sha256(open(os.path.join(ROOT, "models", "MCAsyncIsr.tla"), "rb").read()).hexdigest()
    doc = dict(_generated_by="tests/golden/make_oracle_r_golden.py (Oracle-R: oracle/tlar executing /root/reference/*.tla)",
               _note="level_digests[k] = sha256 over the sorted canonical byte encodings (tests/oracle_r_canon.py) of the states "
                     "first found at BFS level k; invariants are checked in continue mode unless stop is set",
               spec_sha256=sha, entries=entries)
    with open(a.out, "w") as f:
        json.dump(doc, f, indent=1)
    print(f"wrote {a.out}: {len(entries)} entries, {sum(e['distinct'] for e in entries)} states, "
          f"{sum(e['seconds'] for e in entries):.0f} CPU-seconds")


if __name__ == "__main__":
    main()
