#!/usr/bin/env python3
"""Writes tests/golden/oracle_r_mutants_<module>_<N>_<L>_<R>_<E>.npz: the four invariants of KafkaReplication.tla (:101, :320,
:334, :345) evaluated by Oracle-R — the reference's text, executed — on ARBITRARY states: deep reachable states of the
per-state fixtures (tests/golden/oracle_r_successors_*.npz) with one to three fields overwritten by values in or just outside
their ranges.

Why: TypeOk never fails on a reachable state, and at Kip320 neither do WeakIsr / StrongIsr — reachable states alone cannot tell a
lowered invariant from `return true`.  Here every invariant is false on hundreds of states per binding, and it is the reference's
own predicate that says so (the C oracle, the device templates on the host and the GPU's kmc_check_states are held to the file:
tests/test_oracle_r_successors_cpu.py, tests/test_gpu_oracle_r_successors.py).

A mutant is kept only if (a) it survives the packed layout unchanged (a value the bit fields cannot hold would reach the engines
as another state) and (b) it can be written as TLA+ values at all (a leader index beyond the replicas names nobody).  An invariant
whose evaluation raises (a function applied outside its domain: hw beyond LogSize, ...) is marked UNDEFINED for that state and not
compared — TLC would stop with an error there, not answer.

    python tests/golden/make_oracle_r_mutants.py [--per-binding 3000]

File format: states u8[n, sb], inv u8[n] (bit k = invariant k VIOLATED), undefined u8[n] (bit k = invariant k raised).
"""
import argparse
import json
import os
import random
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

REFERENCE = "/root/reference"
INVARIANTS = ("TypeOk", "WeakIsr", "StrongIsr", "LeaderInIsr")


def mutate(b, N, L, R, E, rng):
    """One to three fields of a canonical-byte state overwritten with a value from 0 .. (its maximum + 1)."""
    b = bytearray(b)
    rs = 5 + L
    g = N * rs
    top_rec = R * (E + 1)
    for _ in range(rng.choice((1, 1, 2, 3))):
        kind = rng.randrange(9)
        r = rng.randrange(N)
        if kind == 0:
            b[r * rs + 0] = rng.randrange(L + 2)          # endOffset
        elif kind == 1:
            b[r * rs + 1] = rng.randrange(L + 2)          # hw
        elif kind == 2:
            b[r * rs + 2] = rng.randrange(E + 3)          # leaderEpoch + 1
        elif kind == 3:
            b[r * rs + 3] = rng.randrange(N + 1)          # leader + 1 (a name must exist to write the state down)
        elif kind == 4:
            b[r * rs + 4] = rng.randrange(1 << N)         # isr
        elif kind == 5:
            b[r * rs + 5 + rng.randrange(L)] = rng.randrange(top_rec + 2)   # a record slot (0 = Nil), also beyond endOffset
        elif kind == 6:
            b[g + rng.randrange(3)] = rng.randrange(max(R, E + 1) + 3)      # nextRecordId / nextLeaderEpoch / quorum epoch + 1
        elif kind == 7:
            b[g + 3] = rng.randrange(N + 1)               # quorumState.leader + 1
        else:
            b[g + 4] = rng.randrange(1 << N)              # quorumState.isr
    return bytes(b)


def main():
    import numpy as np
    import oracle_r_canon as oc
    import oracle_r_successors as ors
    from oracle.tlar import Checker
    from oracle.tlar.values import TlaEvalError
    from kafka_specification_amd import CheckerConfig, ModelChecker
    ap = argparse.ArgumentParser()
    ap.add_argument("--per-binding", type=int, default=3000)
    a = ap.parse_args()
    index = {}
    for fn, m in ors.entries():
        N, L, R, E = m["N"], m["L"], m["R"], m["E"]
        fx = ors.load(fn)
        consts = oc.kafka_constants(N, L, R, E)
        ck = Checker(m["module"], consts, [os.path.join(ROOT, "models"), REFERENCE])
        rng = random.Random(zlib.crc32(fn.encode()))
        out, seen = [], set()
        with ModelChecker(CheckerConfig(model=m["module"], n_replicas=N, log_size=L, max_records=R, max_leader_epoch=E, device=-1)) as mc:
            tries = 0
            while len(out) < a.per_binding and tries < 40 * a.per_binding:
                tries += 1
                s = mutate(bytes(fx["states"][rng.randrange(len(fx["states"]))]), N, L, R, E, rng)
                if s in seen or mc.unpack(mc.pack(s)) != s:
                    continue
                seen.add(s)
                # the requests array must still be a history: entries at or beyond nextLeaderEpoch are zero in the encoding
                g = N * (5 + L)
                if s[g + 1] > E + 1 or any(s[g + 5 + 2 * e] or s[g + 6 + 2 * e] for e in range(min(s[g + 1], E + 1), E + 1)):
                    continue
                try:
                    st = oc.kafka_state_from_bytes(s, consts)
                except (KeyError, AssertionError):
                    continue
                inv = undef = 0
                for k, name in enumerate(INVARIANTS):
                    try:
                        if not ck.interp.holds(st, name):
                            inv |= 1 << k
                    except TlaEvalError:
                        undef |= 1 << k
                out.append((s, inv, undef))
        out.sort()
        n = len(out)
        states = np.frombuffer(b"".join(s for s, _, _ in out), dtype=np.uint8).reshape(n, -1)
        inv = np.array([i for _, i, _ in out], dtype=np.uint8)
        undef = np.array([u for _, _, u in out], dtype=np.uint8)
        path = os.path.join(ROOT, "tests", "golden", fn.replace("oracle_r_successors_", "oracle_r_mutants_"))
        np.savez_compressed(path, states=states, inv=inv, undefined=undef)
        index[os.path.basename(path)] = dict(module=m["module"], N=N, L=L, R=R, E=E, states=n,
                                             violating=[int((inv >> k & 1).sum()) for k in range(4)],
                                             undefined=[int((undef >> k & 1).sum()) for k in range(4)])
        print(os.path.basename(path), index[os.path.basename(path)], flush=True)
    json.dump(dict(_generated_by="tests/golden/make_oracle_r_mutants.py (Oracle-R on mutated states)", entries=index),
              open(os.path.join(ROOT, "tests", "golden", "oracle_r_mutants_index.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
