#!/usr/bin/env python3
"""AsyncIsr.tla (under models/MCAsyncIsr.tla's state constraint) evaluated STATE BY STATE by Oracle-R at the constants of
models/MCAsyncIsr.cfg (4 replicas, MaxOffset 3, MaxVersion 4: 139,212,800 states — no interpreter exhausts it) and of
MCAsyncIsr_small.cfg (3 / 3 / 4, which Oracle-R does exhaust: oracle_r_ladder.json).  Same file format, index and tests as
tests/golden/make_oracle_r_successors.py; the three invariants are TypeOk (AsyncIsr.tla:62), ValidHighWatermark (:161) and
MCAsyncIsr's LeaderOffsetInRange.  States: random walks of Oracle-R inside the constraint + walks of the C oracle's successor
function decoded into TLA+ values; a state's successors include the ones OUTSIDE the constraint (TLC generates and
invariant-checks them; it neither stores nor explores them).

    python tests/golden/make_oracle_r_successors_async.py [--states 6000]
"""
import argparse
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
REFERENCE = "/root/reference"
INVARIANTS = ("TypeOk", "ValidHighWatermark", "LeaderOffsetInRange")


def main():
    import numpy as np
    import kmo
    import oracle_r_canon as oc
    from make_oracle_r_successors import fixture_path, succ_digest
    from oracle.tlar import Checker, ModelValue
    ap = argparse.ArgumentParser()
    ap.add_argument("--states", type=int, default=6000)
    a = ap.parse_args()
    index_path = os.path.join(ROOT, "tests", "golden", "oracle_r_successors_index.json")
    index = json.load(open(index_path))
    for (N, MO, MV) in [(4, 3, 4), (3, 3, 4)]:
        reps = [ModelValue(f"r{i + 1}") for i in range(N)]
        consts = dict(Replicas=frozenset(reps), Leader=reps[0], MaxOffset=MO, MaxVersion=MV)
        ck = Checker("MCAsyncIsr", consts, [os.path.join(ROOT, "models"), REFERENCE])
        labels = [str(x) for x in ck.next_labels()]
        lab_idx = {lab: i for i, lab in enumerate(labels)}
        ip = ck.interp
        ocfg = kmo.make_config("AsyncIsr", N=N, L=MO, E=MV, invariants=())
        rng = random.Random(4242 + N)
        inside = lambda b: b[6] <= MO and b[1] <= MV          # StateConstraint on canonical bytes
        init_st = next(iter(ip.initial_states(ck.init)))
        init_b = oc.async_state_bytes(init_st, consts)
        sb = len(init_b)
        merged = {}

        def evaluate(st, b, src):
            inv = sum((0 if ip.holds(st, n) else 1) << k for k, n in enumerate(INVARIANTS))
            succ = ip.successors(st, ck.next)
            recs = [(lab_idx[str(lab)], oc.async_state_bytes(t, consts)) for lab, t in succ]
            merged.setdefault(b, (src, inv, recs))
            return succ, recs

        # (a) Oracle-R's own walks, inside the constraint
        while len(merged) < a.states // 2:
            st, b = init_st, init_b
            for _ in range(rng.randint(8, 45)):
                succ, recs = evaluate(st, b, 1)
                nxt = [(s, r) for s, r in zip(succ, recs) if inside(r[1])]
                if not nxt:
                    break
                (_, st), (_, b) = rng.choice(nxt)
        # (b) walks of the C oracle's successor function, decoded
        pool = set()
        while len(pool) < a.states:
            b = init_b
            for step in range(rng.randint(10, 60)):
                nxt = [t for _a, t in kmo.successors(ocfg, b, sb) if inside(t)]
                if not nxt:
                    break
                b = rng.choice(nxt)
                if step >= 6:
                    pool.add(b)
        for b in sorted(pool):
            if len(merged) >= a.states:
                break
            if b not in merged:
                st = oc.async_state_from_bytes(b, consts)
                assert oc.async_state_bytes(st, consts) == b
                evaluate(st, b, 2)
        keys = sorted(merged)
        n, na = len(keys), len(labels)
        states = np.frombuffer(b"".join(keys), dtype=np.uint8).reshape(n, sb)
        inv = np.array([merged[k][1] for k in keys], dtype=np.uint8)
        src = np.array([merged[k][0] for k in keys], dtype=np.uint8)
        nsucc = np.array([len(merged[k][2]) for k in keys], dtype=np.uint16)
        per = np.zeros((n, na), dtype=np.uint16)
        dig = np.zeros((n, 16), dtype=np.uint8)
        outside = 0
        for i, k in enumerate(keys):
            for ai, t in merged[k][2]:
                per[i, ai] += 1
                outside += 0 if inside(t) else 1
            dig[i] = np.frombuffer(succ_digest(merged[k][2]), dtype=np.uint8)
        path = fixture_path("MCAsyncIsr", N, MO, 0, MV)
        np.savez_compressed(path, states=states, inv=inv, nsucc=nsucc, per_action=per, digest=dig, source=src)
        cov = dict(states=n, successors=int(nsucc.sum()), successors_outside_the_constraint=outside,
                   from_oracle_r_walks=int((src == 1).sum()), from_c_oracle_walks=int((src == 2).sum()),
                   violating=[int((inv >> k & 1).sum()) for k in range(3)], deadlocked=int((nsucc == 0).sum()),
                   per_action_successors={lab: int(per[:, i].sum()) for lab, i in lab_idx.items()})
        index["entries"][os.path.basename(path)] = dict(module="MCAsyncIsr", N=N, L=MO, R=0, E=MV, actions=labels,
                                                        invariants=list(INVARIANTS), coverage=cov)
        print(os.path.basename(path), cov, flush=True)
    json.dump(index, open(index_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
