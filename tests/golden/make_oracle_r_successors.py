#!/usr/bin/env python3
"""Writes tests/golden/oracle_r_successors_<module>_<N>_<L>_<R>_<E>.npz (+ one .json index): the reference's own text
(Oracle-R, oracle/tlar executing /root/reference/*.tla) evaluated STATE BY STATE at the constants the bench and BASELINE.json
bind — where no exhaustive run of an interpreter can go (280 M - 810 M states at 3 brokers / LogSize 6; unbounded for practical
purposes at 7 brokers / LogSize 8).

For every sampled state: the reference's Next relation, disjunct by disjunct and with multiplicity (a binding that satisfies
two disjuncts is generated twice, as TLC generates it), and the four invariants (KafkaReplication.tla:101,320,334,345).
The C oracle (tests/test_oracle_r_successors_cpu.py), the device's model templates compiled for the host (same file) and
the HIP engine's kmc_successors / kmc_check_states on the GPU (tests/test_gpu_oracle_r_successors.py) are each held to it.

Where the states come from (no BFS is needed for a per-state differential — Next and the invariants are defined on any state):
  (a) random walks of Oracle-R itself from Init, 10-60 steps, every state on the way kept (its successors are what the
      walk needs anyway);
  (b) walks of the C oracle's successor function (fast: a million steps a minute), biased towards long logs, several
      leader epochs inside one log and high watermarks >= 3, the states then DECODED into TLA+ values
      (tests/oracle_r_canon.kafka_state_from_bytes) and handed to Oracle-R;
  (c) uniform samples of the C oracle's first BFS levels (every reachable state of small depth has the same chance).
A state of (b)/(c) is whatever the C oracle believes reachable; were it wrong about that, the comparison on that state
would still be a valid one.

    python tests/golden/make_oracle_r_successors.py [--jobs 4] [--states 20000] [--only Kip320:3,6,6,2]

Needs /root/reference; about 40 CPU-minutes for the seven bindings (Oracle-R evaluates ~100 states/s at 3 brokers, ~20/s at 7).

File format (numpy .npz, compressed): states u8[n, sb] (canonical bytes, sorted), inv u8[n] (bit k = invariant k VIOLATED:
TypeOk, WeakIsr, StrongIsr, LeaderInIsr), nsucc u16[n], per_action u16[n, n_actions] (successors per Next disjunct, in the
module's order), digest u8[n, 16] = sha256 over the sorted multiset of (action index byte + successor's canonical bytes),
first 16 bytes; source u8[n] (1 = Oracle-R walk, 2 = C-oracle walk, 3 = C-oracle level sample).
"""
import argparse
import hashlib
import json
import os
import random
import sys
import time
import zlib
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

REFERENCE = "/root/reference"
INVARIANTS = ("TypeOk", "WeakIsr", "StrongIsr", "LeaderInIsr")
BINDINGS = [("KafkaTruncateToHighWatermark", 3, 6, 6, 2), ("Kip101", 3, 6, 6, 2), ("Kip279", 3, 6, 6, 2), ("Kip320", 3, 6, 6, 2),
            ("Kip320FirstTry", 3, 6, 6, 2), ("Kip279", 5, 2, 2, 1), ("Kip320", 7, 8, 8, 3),
            # round 6: BASELINE config 4 at SURVEY 8(a.0)'s own sizing — five brokers whose logs hold up to four epochs, so that
            # FirstNonMatchingOffsetFromTail (Kip279.tla:39-45) sees three epochs in a log at five brokers (missing 5)
            ("Kip279", 5, 4, 4, 3)]
# bindings whose biased sample is chosen for logs >= 3 deep holding >= 3 epochs (the older ones keep their own criterion, so
# that re-running the script reproduces their committed files)
THREE_EPOCHS = {("Kip279", 5, 4, 4, 3)}


def deep3(b, N, L, E):
    """deepest log of the state that holds >= 3 distinct record epochs (0: none)"""
    best = 0
    for r in range(N):
        o = r * (5 + L)
        end = b[o]
        if len({(b[o + 5 + k] - 1) % (E + 1) for k in range(min(end, L)) if b[o + 5 + k]}) >= 3:
            best = max(best, end)
    return best


def fixture_path(module, N, L, R, E):
    return os.path.join(ROOT, "tests", "golden", f"oracle_r_successors_{module.lower()}_{N}_{L}_{R}_{E}.npz")


def succ_digest(records):
    """records: iterable of (action index, successor canonical bytes) — a multiset."""
    h = hashlib.sha256()
    for a, b in sorted(records):
        h.update(bytes([a]))
        h.update(b)
    return h.digest()[:16]


def features(b, N, L, E):
    """What a state exercises, read off its canonical bytes: (deepest log, most distinct record epochs inside one log,
    highest hw, deepest log that holds >= 2 epochs)."""
    deep = eps = hw = deep2 = 0
    for r in range(N):
        o = r * (5 + L)
        end = b[o]
        e = len({(b[o + 5 + k] - 1) % (E + 1) for k in range(min(end, L)) if b[o + 5 + k]})
        deep, eps, hw = max(deep, end), max(eps, e), max(hw, b[o + 1])
        if e >= 2:
            deep2 = max(deep2, end)
    return deep, eps, hw, deep2


def score(b, N, L, E):
    deep, eps, hw, deep2 = features(b, N, L, E)
    return sum(b[r * (5 + L)] for r in range(N)) + 2 * hw + 3 * eps + deep2


def pick(succ, rng, greedy, N, L, E):
    """One step of a walk: uniformly at random, or (with probability `greedy`) one of the best-scoring successors."""
    if rng.random() >= greedy:
        return rng.choice(succ)
    sc = [score(s[1], N, L, E) for s in succ]
    m = max(sc)
    return rng.choice([s for s, v in zip(succ, sc) if v == m])


# ---- phase 1: states from the C oracle (cheap) ------------------------------------------------------------------------

def c_oracle_states(module, N, L, R, E, n_walk, n_level, seed):
    import kmo
    cfg = kmo.make_config(module, N=N, L=L, R=R, E=E, invariants=())
    rng = random.Random(seed)
    sb = N * (5 + L) + 5 + 2 * (E + 1)
    # (c) level samples: the first levels that fit a small budget
    run = kmo.Run(kmo.make_config(module, N=N, L=L, R=R, E=E, invariants=(), max_states=300_000))
    levels = [sorted(run.level_states(k)) for k in range(len(run.levels))]
    init = levels[0][0]
    pool = [s for lv in levels[3:] for s in lv]
    level_sample = set(rng.sample(pool, min(n_level, len(pool))))
    run.close()
    # (b) biased walks; keep the states from step 10 on, favouring the interesting ones when thinning
    seen, walk_states = set(), []
    walks = 0
    while len(walk_states) < 6 * n_walk and walks < 200_000:
        walks += 1
        greedy = rng.choice((0.0, 0.2, 0.5, 0.8))
        st, depth, maxd = init, 0, rng.randint(12, 70)
        while depth < maxd:
            succ = kmo.successors(cfg, st, sb)
            if not succ:
                break
            st = pick(succ, rng, greedy, N, L, E)[1]
            depth += 1
            if depth >= 10 and st not in seen and st not in level_sample:
                seen.add(st)
                walk_states.append(st)
    want = min(L, 5)
    if (module, N, L, R, E) in THREE_EPOCHS:
        # Three epochs inside one log need three leaders in a row, each fetching its predecessor's records before it writes its own
        # (~16 specific steps; MaxRecords = 4 records exist in all): uniform walks almost never get there (145 of 78,000 walk states).
        # So: greedy walks towards "more distinct epochs in a log" until there are seeds, then short random walks FROM the seeds —
        # their descendants keep the three-epoch log unless a truncation removes it (which is the path to exercise).
        def score3(b):
            f = features(b, N, L, E)
            return 100 * f[1] + 10 * deep3(b, N, L, E) + sum(b[r * (5 + L)] for r in range(N)) + f[2]
        seeds, tries = set(), 0
        while len(seeds) < 400 and tries < 40_000:
            tries += 1
            st = init
            for _ in range(rng.randint(18, 48)):
                succ = kmo.successors(cfg, st, sb)
                if not succ:
                    break
                if rng.random() < 0.85:
                    sc = [score3(x[1]) for x in succ]
                    m = max(sc)
                    st = rng.choice([x for x, v in zip(succ, sc) if v == m])[1]
                else:
                    st = rng.choice(succ)[1]
                if deep3(st, N, L, E) >= 3:
                    seeds.add(st)
        seeds = sorted(seeds)
        rich_set = set(seeds)
        target = int(0.85 * n_walk)
        guard = 0
        while len(rich_set) < 2 * target and seeds and guard < 400_000:
            guard += 1
            st = rng.choice(seeds)
            for _ in range(rng.randint(1, 25)):
                succ = kmo.successors(cfg, st, sb)
                if not succ:
                    break
                st = rng.choice(succ)[1]
                if deep3(st, N, L, E) >= 3 and st not in level_sample:
                    rich_set.add(st)
        rich = sorted(rich_set)
        rng.shuffle(rich)
        rich = rich[:target]
    else:
        rich = [s for s in walk_states if (lambda f: f[3] >= want and f[2] >= min(3, L))(features(s, N, L, E))]
        rng.shuffle(rich)
        rich = rich[:n_walk // 2]
    rest = list(set(walk_states) - set(rich))
    rest.sort()
    rng.shuffle(rest)
    chosen = rich + rest[:n_walk - len(rich)]
    return init, [(s, 2) for s in chosen] + [(s, 3) for s in sorted(level_sample)]


# ---- phase 2: Oracle-R (expensive, in worker processes) ----------------------------------------------------------------

_CK = {}


def checker(module, N, L, R, E):
    key = (module, N, L, R, E)
    if key not in _CK:
        from oracle.tlar import Checker
        import oracle_r_canon as oc
        consts = oc.kafka_constants(N, L, R, E)
        ck = Checker(module, consts, [os.path.join(ROOT, "models"), REFERENCE])
        labels = [str(x) for x in ck.next_labels()]
        _CK[key] = (ck, consts, {lab: i for i, lab in enumerate(labels)}, labels)
    return _CK[key]


def evaluate(ck, consts, lab_idx, st):
    """One state (TLA+ values) through the reference's text -> (inv bits, [(action index, successor bytes)], successors)."""
    import oracle_r_canon as oc
    ip = ck.interp
    inv = 0
    for k, name in enumerate(INVARIANTS):
        if not ip.holds(st, name):
            inv |= 1 << k
    succ = ip.successors(st, ck.next)
    recs = [(lab_idx[str(lab)], oc.kafka_state_bytes(t, consts)) for lab, t in succ]
    return inv, recs, succ


def task_eval(args):
    (module, N, L, R, E), chunk = args
    import oracle_r_canon as oc
    ck, consts, lab_idx, _ = checker(module, N, L, R, E)
    out = []
    for b, src in chunk:
        st = oc.kafka_state_from_bytes(b, consts)
        assert oc.kafka_state_bytes(st, consts) == b
        inv, recs, _ = evaluate(ck, consts, lab_idx, st)
        out.append((b, src, inv, recs))
    return out


def task_walk(args):
    (module, N, L, R, E), seed, n_states = args
    import oracle_r_canon as oc
    ck, consts, lab_idx, _ = checker(module, N, L, R, E)
    rng = random.Random(seed)
    init = next(iter(ck.interp.initial_states(ck.init)))
    out, seen = [], set()
    while len(out) < n_states:
        greedy = rng.choice((0.0, 0.0, 0.3, 0.6))
        st, depth, maxd = init, 0, rng.randint(10, 60)
        while depth <= maxd and len(out) < n_states:
            b = oc.kafka_state_bytes(st, consts)
            inv, recs, succ = evaluate(ck, consts, lab_idx, st)
            if depth >= 4 and b not in seen:      # (the first levels are in every walk; (c) samples them)
                seen.add(b)
                out.append((b, 1, inv, recs))
            if not succ:
                break
            by_bytes = list(zip(succ, recs))
            choice = pick([(sr, r[1]) for sr, r in by_bytes], rng, greedy, N, L, E)
            st = choice[0][1]
            depth += 1
    return out


def main():
    import numpy as np
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=max(1, (os.cpu_count() or 2) // 2))
    ap.add_argument("--states", type=int, default=20000, help="states per binding")
    ap.add_argument("--only", default=None, help="Module:N,L,R,E[;...]")
    a = ap.parse_args()
    todo = BINDINGS
    if a.only:
        want = {(m, *map(int, c.split(","))) for m, c in (x.split(":") for x in a.only.split(";"))}
        todo = [b for b in BINDINGS if b in want]
    index_path = os.path.join(ROOT, "tests", "golden", "oracle_r_successors_index.json")
    index = json.load(open(index_path)) if os.path.exists(index_path) else {"entries": {}}
    sha = {fn: hashlib.sha256(open(os.path.join(REFERENCE, fn), "rb").read()).hexdigest()
           for fn in sorted(os.listdir(REFERENCE)) if fn.endswith(".tla")}
    with ProcessPoolExecutor(max_workers=a.jobs) as ex:
        for bind in todo:
            module, N, L, R, E = bind
            t0 = time.time()
            n_r = a.states * 2 // 5
            n_c = a.states - n_r
            init, cstates = c_oracle_states(module, N, L, R, E, n_c - n_c // 10, n_c // 10, seed=zlib.crc32(repr(bind).encode()) & 0xFFFF)
            t1 = time.time()
            chunk = 250
            tasks = [ex.submit(task_eval, (bind, cstates[i:i + chunk])) for i in range(0, len(cstates), chunk)]
            per_walk = 200
            tasks += [ex.submit(task_walk, (bind, 1000 + w, per_walk)) for w in range((n_r + per_walk - 1) // per_walk)]
            merged = {}
            for t in tasks:
                for b, src, inv, recs in t.result():
                    if b in merged:
                        assert merged[b][1:] == (inv, recs) or sorted(merged[b][2]) == sorted(recs)
                    else:
                        merged[b] = (src, inv, recs)
            _, consts, lab_idx, labels = checker(module, N, L, R, E)
            keys = sorted(merged)
            n, sb, na = len(keys), len(keys[0]), len(labels)
            states = np.frombuffer(b"".join(keys), dtype=np.uint8).reshape(n, sb)
            inv = np.array([merged[k][1] for k in keys], dtype=np.uint8)
            src = np.array([merged[k][0] for k in keys], dtype=np.uint8)
            nsucc = np.array([len(merged[k][2]) for k in keys], dtype=np.uint16)
            per = np.zeros((n, na), dtype=np.uint16)
            dig = np.zeros((n, 16), dtype=np.uint8)
            for i, k in enumerate(keys):
                for ai, _ in merged[k][2]:
                    per[i, ai] += 1
                dig[i] = np.frombuffer(succ_digest(merged[k][2]), dtype=np.uint8)
            path = fixture_path(module, N, L, R, E)
            np.savez_compressed(path, states=states, inv=inv, nsucc=nsucc, per_action=per, digest=dig, source=src)
            feats = [features(k, N, L, E) for k in keys]
            want = min(L, 5)
            cov = dict(states=n, successors=int(nsucc.sum()), from_oracle_r_walks=int((src == 1).sum()),
                       from_c_oracle_walks=int((src == 2).sum()), from_c_oracle_levels=int((src == 3).sum()),
                       deadlocked=int((nsucc == 0).sum()),
                       violating=[int((inv >> k & 1).sum()) for k in range(4)],
                       log_depth_ge_5=sum(f[0] >= want for f in feats),
                       two_epochs_in_a_log=sum(f[1] >= 2 for f in feats), three_epochs_in_a_log=sum(f[1] >= 3 for f in feats),
                       hw_ge_3=sum(f[2] >= min(3, L) for f in feats),
                       deep_mixed_epoch_log_and_hw_ge_3=sum(f[3] >= want and f[2] >= min(3, L) for f in feats),
                       log_ge_3_deep_with_three_epochs=sum(deep3(k, N, L, E) >= 3 for k in keys),
                       per_action_successors={lab: int(per[:, i].sum()) for lab, i in lab_idx.items()},
                       states_with_a_twice_generated_successor=sum(
                           len(set(merged[k][2])) < len(merged[k][2]) or
                           len({r[1] for r in merged[k][2]}) < len(merged[k][2]) for k in keys))
            index["entries"][os.path.basename(path)] = dict(module=module, N=N, L=L, R=R, E=E, actions=labels,
                                                            invariants=list(INVARIANTS), coverage=cov,
                                                            seconds=round(time.time() - t0, 1))
            print(f"{module} {N}/{L}/{R}/{E}: {n} states, {cov['successors']} successors, "
                  f"{cov['deep_mixed_epoch_log_and_hw_ge_3']} deep+mixed+hw>=3, sampling {t1 - t0:.0f}s, total {time.time() - t0:.0f}s",
                  flush=True)
            index["spec_sha256"] = sha
            index["_generated_by"] = ("tests/golden/make_oracle_r_successors.py (Oracle-R: oracle/tlar executing "
                                      "/root/reference/*.tla, state by state)")
            json.dump(index, open(index_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
