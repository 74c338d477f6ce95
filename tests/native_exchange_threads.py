"""P ranks of the exchange under the C ABI as P host THREADS of one process on one GPU, with tests/_mock_rccl.so in
librccl's place (KMC_RCCL_LIB).  RCCL refuses two ranks on one device and a gpurun box has one GPU, so this is the only way
kmc_comm_init / kmc_comm_selftest / kmc_step_exchange_counts / kmc_step_exchange_payload ever run with more than one
concurrent rank: each thread drives its own HipShardEngine through sharded.run_sharded exactly as a torch.distributed.run
rank does; only the transport under the nccl* calls is the stand-in (tests/mock_rccl.cpp).  Checked against the C oracle.
Run by tests/test_gpu_native_exchange_threads.py in a fresh process (the engine binds "RCCL" once per process).

usage: native_exchange_threads.py MODEL N L R E P inv1,inv2 [trace] [levels=K] [pipeline=PARTS] [symmetry] [golden=FILE] [table=LOG2] [frontier=LOG2] [send=LOG2]   -> one JSON line, exit code 0 when everything agrees
(levels=K: stop after K BFS levels and compare with the oracle's prefix — for constants nothing can exhaust;
 pipeline=PARTS: every level runs as a pipeline of PARTS parts, kmc_step_level_parts, instead of in one shot)"""
import ctypes as C
import json
import os
import sys
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
os.environ["KMC_RCCL_LIB"] = os.path.join(HERE, "_mock_rccl.so")     # before the engine's first look for librccl
os.environ["KMC_EXCHANGE"] = "rccl"
_parts = next((int(a[9:]) for a in sys.argv[8:] if a.startswith("pipeline=")), 0)
if _parts:   # every level as a pipeline of that many parts (kmc_step_level_parts); read when sharded is imported
    os.environ["KMC_PIPELINE_PARTS"] = str(_parts)
    os.environ["KMC_PIPELINE_MIN_STATES"] = "0"
else:
    os.environ["KMC_PIPELINE_PARTS"] = "1"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402,F401  (first: one HIP runtime per process)
import kmo  # noqa: E402  (the oracle: checker only)
from kafka_specification_amd import CheckerConfig, _native as nat, sharded  # noqa: E402


class Shared:
    """The small reductions a rank normally does through torch.distributed, between threads."""

    def __init__(self, n):
        self.n = n
        self.barrier = threading.Barrier(n, timeout=60)
        self.slots = [None] * n

    def all_reduce(self, rank, value, op):
        self.slots[rank] = value
        self.barrier.wait()
        out = op(list(self.slots))
        self.barrier.wait()
        return out


class ThreadRcclExchange(sharded.RcclExchange):
    """RcclExchange without torch.distributed: the unique id is handed over in memory, the rare small reductions
    (trace reconstruction) go through `Shared`.  exchange() / _deliver() / selftest() are inherited unchanged."""

    def __init__(self, engine, uid_bytes, shared, rank):
        self.engine, self.lib, self.shared, self.rank, self.world = engine, nat.lib(), shared, rank, shared.n
        uid = (C.c_uint8 * nat.KMC_COMM_ID_BYTES)(*uid_bytes)
        nat.check(self.lib.kmc_comm_init(engine.mc.handle, uid))     # returns when every rank has joined

    def all_reduce_sum(self, stats):
        mine = np.sum(np.stack(stats), axis=0)
        return self.shared.all_reduce(self.rank, mine, lambda xs: np.sum(np.stack(xs), axis=0))

    def all_reduce_max(self, x):
        return self.shared.all_reduce(self.rank, x, max)

    def barrier(self):
        self.shared.barrier.wait()


def main():
    model, N, L, R, E, P = sys.argv[1], *map(int, sys.argv[2:7])
    inv = tuple(x for x in sys.argv[7].split(",") if x)
    trace = "trace" in sys.argv[8:]
    K = next((int(a[7:]) for a in sys.argv[8:] if a.startswith("levels=")), 0)
    opt = lambda name, dflt: next((int(a[len(name) + 1:]) for a in sys.argv[8:] if a.startswith(name + "=")), dflt)
    symmetry = "symmetry" in sys.argv[8:]      # orbit counting on every rank (kmc_step_finish weighs, run_sharded sums)
    golden = next((a[7:] for a in sys.argv[8:] if a.startswith("golden=")), None)   # an Oracle-O fixture instead of a live oracle run
    if golden:
        g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", golden)))
        assert (g["model"], g["N"], g["L"], g["R"], g["E"]) == (model, N, L, R, E) and K and len(g["levels"]) >= K

        class _G:   # the two views the comparison below reads: o.levels (prefix), o2.generated / action_generated (after K levels)
            levels = g["levels"] + [0]
            generated = g["generated"]
            action_generated = g["action_generated"]
            viol_inv = None
        assert len(g["levels"]) == K, "the fixture's generated counts belong to exactly its own number of levels"
        o = o2 = _G
    elif K:   # a prefix: the oracle stops after the level that crosses max_states; o2 stops right after producing level K
        o = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=inv, threads=8, max_states=1_500_000))
        assert len(o.levels) > K
        o2 = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=inv, threads=8, max_states=sum(o.levels[:K - 1]) + 1))
        assert len(o2.levels) == K
    else:
        o = kmo.Run(kmo.make_config(model, N=N, L=L, R=R, E=E, invariants=inv))
    consts = dict(n_replicas=N, log_size=L, max_leader_epoch=E)      # AsyncIsr: (N, MaxOffset, MaxVersion); R is unused
    if model != "AsyncIsr":
        consts["max_records"] = R
    cfg = CheckerConfig(model=model, **consts, invariants=inv, keep_trace=trace, table_capacity=1 << opt("table", 22 if K else 20),
                        frontier_capacity=1 << opt("frontier", 20 if K else 18), send_capacity=1 << opt("send", 16), max_levels=K,
                        symmetry=symmetry)
    lib = nat.lib()
    uid = (C.c_uint8 * nat.KMC_COMM_ID_BYTES)()
    nat.check(lib.kmc_comm_unique_id(uid))          # main thread: also the one-time binding of "librccl"
    engines = [sharded.HipShardEngine(cfg, r, P, 0, native=True) for r in range(P)]
    assert all(e.native for e in engines)
    shared, results, errors, exchanges = Shared(P), [None] * P, [None] * P, [None] * P

    def rank_main(r):
        try:
            ex = exchanges[r] = ThreadRcclExchange(engines[r], bytes(uid), shared, r)
            ex.selftest()                           # all-gather + grouped send/receive ring across ALL ranks, verified
            results[r] = sharded.run_sharded([engines[r]], ex, cfg, engines[r].action_names())
        except BaseException as e:  # noqa: BLE001 — reported below; the other ranks run into the mock's time-out
            errors[r] = f"{type(e).__name__}: {e}"
            shared.barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300 if K else 90)
    hung = [r for r, t in enumerate(threads) if t.is_alive()]
    out = dict(model=model, N=N, L=L, R=R, E=E, P=P, trace=trace, errors=errors, hung=hung)
    ok = not hung and not any(errors)
    if ok:
        r0 = results[0]
        same_everywhere = all((r.verdict, r.distinct, r.generated, r.levels, r.violated_invariant, r.violation_depth) ==
                              (r0.verdict, r0.distinct, r0.generated, r0.levels, r0.violated_invariant, r0.violation_depth)
                              for r in results)
        if K:
            matches = (r0.verdict == "level_limit" and r0.levels == o.levels[:K] and r0.generated == o2.generated and
                       list(r0.action_generated.values()) == o2.action_generated[:len(r0.action_generated)] and
                       r0.violated_invariant is None)
        else:
            matches = (r0.verdict == o.verdict and r0.distinct == o.distinct and r0.generated == o.generated and
                       r0.levels == o.levels and r0.deadlock_states == o.deadlock_states and r0.violated_invariant == o.viol_inv)
        if o.viol_inv and not K:
            matches = matches and r0.violation_depth == o.viol_depth and r0.violation_count == o.viol_count
        out["pipelined_levels"] = _parts and min(getattr(ex, "pipelined_levels", 0) for ex in exchanges)
        if symmetry:
            out["stored"] = r0.orbit_representatives
            if golden:
                matches = matches and r0.orbit_representatives == g["stored"] and r0.deadlock_states == g["deadlock_states"]
            else:
                matches = matches and 0 < r0.orbit_representatives < max(r0.distinct, 2)
        out.update(verdict=r0.verdict, distinct=r0.distinct, generated=r0.generated, depth=r0.depth,
                   same_on_every_rank=same_everywhere, matches_oracle=matches,
                   exchange=type(engines[0]).__name__ + " + ThreadRcclExchange over " + os.path.basename(os.environ["KMC_RCCL_LIB"]))
        ok = same_everywhere and matches
        if trace and o.viol_inv:
            tr = r0.trace
            out["trace_len"] = len(tr)
            # every rank reconstructed the same behaviour, it has the oracle's length and ends in a violating state
            ok = ok and len(tr) == o.viol_depth and all([(a, bytes(s)) for a, s in r.trace] == [(a, bytes(s)) for a, s in tr]
                                                        for r in results)
    if not hung:
        for e in engines:
            e.close()
    print("RESULT " + json.dumps(out), flush=True)
    os._exit(0 if ok else 1)      # no interpreter tear-down with possibly blocked threads


if __name__ == "__main__":
    main()
