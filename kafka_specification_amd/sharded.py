"""Multi-GPU BFS: the fingerprint space is hash-partitioned across shards (one shard per GPU,
one process per GPU) and every BFS level exchanges successor states with one all-to-all.

    owner(state) = (bits 40..63 of the fingerprint) * P >> 24      (csrc/kmc_device.h: kmc_owner)

Per level, on every shard:  expand the local frontier, bucketing each successor into the send
area of its owner (k_expand, mode SHARDED)  ->  all-to-all-v of packed states (W words, plus the
predecessor fingerprint when traces are kept; the owner recomputes the fingerprint because it must
later expand the state)  ->  the owner probes/inserts what it received, checks invariants on
the winners and appends them to its next frontier (k_insert).  A small statistics vector per
shard (new states, generated, violations, deadlocks, error flags) rides on the send-count
all-gather of the NEXT level, whose sum decides termination and the verdict identically on every
rank: one collective and one device sync fewer per BFS level than a separate all-reduce.

The level logic is written against two small interfaces so that it can be exercised without
GPUs:
  * an *engine* (begin / expand / insert / finish) — `HipShardEngine` drives libkmc.so's
    kmc_step_* entry points; the CPU tests plug in an oracle-backed stand-in;
  * an *exchange* — `DistExchange` is torch.distributed (RCCL on GPUs, gloo in the CPU
    tests), `LoopbackExchange` routes between several engines living in one process (P
    logical shards on one GPU; RCCL refuses two ranks on one device).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import time
from dataclasses import replace
from typing import List, Optional, Sequence

import numpy as np

from . import _native as nat
from .checker import CheckerConfig, CheckResult, ModelChecker

N_STATS = 32  # [new, generated[16], viol[4], deadlocks, err_table, err_frontier, send_overflow, ...]


class HipShardEngine:
    """One shard on one GPU through the kmc_step_* C ABI."""

    def __init__(self, cfg: CheckerConfig, shard_id: int, n_shards: int, device: int, native: bool = True):
        """native=True (the product): the send and receive areas belong to the engine and the per-level
        exchange runs under the C ABI (RcclExchange / NativeLoopbackExchange) — no torch object is involved.
        native=False: the send area is a torch tensor whose filled slices are handed to a torch.distributed
        collective (DistExchange) or routed in process (LoopbackExchange)."""
        native = native and os.environ.get("KMC_EXCHANGE", "rccl") != "torch"
        self.native = native
        subs = nat.KMC_SEND_SUBS
        scap = cfg.send_capacity or max(1 << 13, (cfg.frontier_capacity or (1 << 22)) * 4 // (n_shards * subs))
        self.send_cap = scap
        self.cfg = replace(cfg, n_shards=n_shards, shard_id=shard_id, device=device, send_capacity=scap)
        self.shard_id, self.n_shards = shard_id, n_shards
        self.mc = ModelChecker(self.cfg)
        self.lib = nat.lib()
        self.W = self.mc.state_words
        self.record_words = self.W + (1 if cfg.keep_trace else 0)  # the predecessor fingerprint travels only for traces
        self.torch = None
        if not native:
            import torch
            self.torch = torch
            self.device = torch.device("cuda", device)
            # the send area belongs to torch so that slices of it can be handed to the collective:
            # [destination][sub-buffer][record]; block b of k_expand fills sub-buffer b % KMC_SEND_SUBS
            self.send = torch.zeros((n_shards, subs, scap, self.record_words), dtype=torch.int64, device=self.device)
            nat.check(self.lib.kmc_step_set_send_buffer(self.mc.handle, C.c_void_p(self.send.data_ptr()), scap))
        self._last = None
        self._viol_fp = [0, 0, 0, 0]
        self._oviol_fp = [0, 0, 0, 0]

    def close(self):
        if self.torch is not None:
            self.torch.cuda.synchronize()
        self.mc.close()

    def begin(self):
        nat.check(self.lib.kmc_step_begin(self.mc.handle))
        r = self.mc.result()
        st = np.zeros(N_STATS, dtype=np.int64)
        st[0] = r.levels[-1] if r.levels else 0
        st[16] = r.generated  # the initial state counts as generated on its owner
        return st

    def expand(self):
        """-> per destination, the list of filled sub-buffer slices (each a contiguous [n, W+1] view)."""
        subs = nat.KMC_SEND_SUBS
        counts = (C.c_uint64 * (nat.KMC_MAX_SHARDS * subs))()
        nat.check(self.lib.kmc_step_expand(self.mc.handle, counts))
        if self.native:
            return None          # the engine keeps the counts; the exchange under the ABI moves the runs
        return [[self.send[d, sb, :int(counts[d * subs + sb])] for sb in range(subs) if counts[d * subs + sb]]
                for d in range(self.n_shards)]

    def insert(self, records):
        n = int(records.shape[0])
        if n:
            assert records.is_contiguous()
            nat.check(self.lib.kmc_step_insert(self.mc.handle, C.c_void_p(records.data_ptr()), n))

    def finish(self):
        """Statistics of the expansion just completed: st[0] new states of the produced level,
        st[1..15] generated per action, st[17..20] invariant violations among the states of the
        EXPANDED level, st[21] deadlocked states of the expanded level, st[22..23] error flags, st[24]
        filtered sends, st[25..28] violating successors outside the state constraint (AsyncIsr)."""
        info = nat.KmcLevelInfo()
        nat.check(self.lib.kmc_step_finish(self.mc.handle, C.byref(info)))
        st = np.zeros(N_STATS, dtype=np.int64)
        st[0] = info.new_states
        for k in range(15):
            st[1 + k] = info.generated_level[k]
        for k in range(4):
            st[17 + k] = info.violation_count[k]
            st[25 + k] = info.outside_violation_count[k]
        st[21] = info.deadlocks_level
        st[24] = info.send_filtered
        st[22] = 1 if info.error_flags & 2 else 0
        st[23] = 1 if info.error_flags & (1 | 4) else 0
        self._viol_fp = [int(info.violation_fp[k]) for k in range(4)]
        self._oviol_fp = [int(info.outside_violation_fp[k]) for k in range(4)]
        return st

    def save_checkpoint(self, path: str):
        """This shard's table / frontier at the level boundary it stands on (between finish() and expand())."""
        nat.check(self.lib.kmc_step_set_verdict(self.mc.handle, nat.VERDICTS.index("level_limit")))
        nat.check(self.lib.kmc_checkpoint_save(self.mc.handle, path.encode()))

    def load_checkpoint(self, path: str) -> int:
        """-> the size of the level this shard holds after the load (its share of the last recorded level)."""
        nat.check(self.lib.kmc_checkpoint_load(self.mc.handle, path.encode()))
        nat.check(self.lib.kmc_step_resume(self.mc.handle))
        lv = self.mc.result().levels
        return lv[-1] if lv else 0

    def check_frontier(self):
        """Invariant-only pass over the current, unexpanded frontier (the last level under max_levels):
        st[17..20] = violations among its states."""
        info = nat.KmcLevelInfo()
        nat.check(self.lib.kmc_step_check_frontier(self.mc.handle, C.byref(info)))
        st = np.zeros(N_STATS, dtype=np.int64)
        for k in range(4):
            st[17 + k] = info.violation_count[k]
        self._viol_fp = [int(info.violation_fp[k]) for k in range(4)]
        return st

    def outside_violation_fp(self, inv_index: int) -> int:
        """Smallest fingerprint of the violating successors OUTSIDE the state constraint this shard generated in
        the last finished expansion (0 = none)."""
        return self._oviol_fp[inv_index]

    def find_outside(self, fp: int):
        """(packed words, parent fingerprint) of the outside-the-constraint successor fp if this shard's retired
        level generates it, else None.  Valid between finish() and the next expand()."""
        words, parent, found = (C.c_uint64 * self.W)(), C.c_uint64(), C.c_int32()
        nat.check(self.lib.kmc_step_find_outside(self.mc.handle, C.c_uint64(fp), words, C.byref(parent), C.byref(found)))
        return ([int(x) for x in words], int(parent.value)) if found.value else None

    def result(self) -> CheckResult:
        return self.mc.result()

    def action_names(self):
        return self.mc.action_names()

    # -- trace reconstruction across shards (keep_trace) ------------------------------------------
    def violation_fp(self, inv_index: int) -> int:
        """Smallest fingerprint of this shard's states of the last expanded level that violate invariant k."""
        return self._viol_fp[inv_index]

    def owner(self, fp: int) -> int:
        return int(self.lib.kmc_owner_of(C.c_uint64(fp), self.n_shards))   # the partition k_expand buckets by (kmc_owner)

    def pred_of(self, fp: int):
        """Predecessor fingerprint recorded for fp in this shard's table, None when fp is not here."""
        pred, found = C.c_uint64(), C.c_int32()
        nat.check(self.lib.kmc_pred_of(self.mc.handle, C.c_uint64(fp), C.byref(pred), C.byref(found)))
        return int(pred.value) if found.value else None

    def init_words(self):
        w = (C.c_uint64 * self.W)()
        nat.check(self.lib.kmc_init_state(self.mc.handle, w))
        return [int(x) for x in w]

    def successors(self, words):
        return self.mc.successors(words)            # [(words, fingerprint, action kind)], ENUM mode: no table access

    def fingerprint(self, words) -> int:
        return self.mc.fingerprint(words)

    def canonical(self, words) -> bytes:
        return self.mc.unpack(words)


class LoopbackExchange:
    """P engines in one process; the all-to-all is a routing of tensor references."""

    def __init__(self, n):
        self.n = n

    def exchange(self, sends, stats):
        """-> (global statistics of the PREVIOUS expansion, deliver) ; deliver() routes the payload."""
        return np.sum(np.stack(stats), axis=0), lambda: self.all_to_all(sends)

    def all_to_all(self, sends):  # sends[i][d] = chunks from shard i for shard d  ->  recvs[d] = all chunks for d
        return [[c for i in range(self.n) for c in _chunks(sends[i][d])] for d in range(self.n)]

    def all_reduce_sum(self, stats):
        return np.sum(np.stack(stats), axis=0)

    def all_reduce_max(self, x: float) -> float:
        return x

    def barrier(self):
        pass


class NativeLoopbackExchange:
    """P native HipShardEngines in one process on one GPU: kmc_step_exchange_local / kmc_step_deliver_local move
    every run with a device-to-device copy, following the same plan (kmc_exchange_plan) the RCCL transport
    executes, and each shard inserts what it received with one k_insert."""

    def __init__(self, engines):
        self.engines = list(engines)
        self.n = len(self.engines)
        self.lib = nat.lib()
        self.handles = (C.c_void_p * self.n)(*[e.mc.handle for e in self.engines])

    def exchange(self, sends, stats):
        flat = np.ascontiguousarray(np.stack(stats), dtype=np.int64)
        out = np.zeros(N_STATS, dtype=np.int64)
        nat.check(self.lib.kmc_step_exchange_local(self.handles, self.n, flat.ctypes.data_as(C.POINTER(C.c_int64)),
                                                   N_STATS, out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out, self._deliver

    def _deliver(self):
        nat.check(self.lib.kmc_step_deliver_local(self.handles, self.n))
        return [[] for _ in self.engines]     # already inserted under the ABI

    def all_reduce_sum(self, stats):
        return np.sum(np.stack(stats), axis=0)

    def all_reduce_max(self, x: float) -> float:
        return x

    def barrier(self):
        pass


class _watchdog:
    """A collective that never completes (a rank missing, a transport that does not come up) blocks inside the library,
    where no exception can reach it, and a multi-GPU job that hangs tells its user nothing.  After KMC_COLLECTIVE_TIMEOUT
    seconds (default 180) the rank says what it was waiting for and exits with status 3; the launcher then ends the other
    ranks.  Used around the communicator's creation and its self-test — the two places where "never" shows first."""

    def __init__(self, what, rank):
        self.what, self.rank = what, rank
        self.limit = float(os.environ.get("KMC_COLLECTIVE_TIMEOUT", "180"))

    def _give_up(self):
        import sys
        print(f"[kmc] rank {self.rank}: {self.what} did not complete within {self.limit:.0f} s — giving up instead of hanging",
              file=sys.stderr, flush=True)
        os._exit(3)

    def __enter__(self):
        import threading
        self.timer = threading.Timer(self.limit, self._give_up)
        self.timer.daemon = True
        self.timer.start()
        return self

    def __exit__(self, *exc):
        self.timer.cancel()
        return False


class RcclExchange:
    """One native HipShardEngine per process, one process per GPU: the per-level exchange under the C ABI
    (kmc_step_exchange_counts / kmc_step_exchange_payload: an RCCL all-gather of counts + statistics, grouped
    ncclSend/ncclRecv from the send area into the receive area, one k_insert, all on the engine's stream).
    torch.distributed is used for the bootstrap only (the unique id travels through the default process
    group) and for the rare small reductions of trace reconstruction and bench timing."""

    def __init__(self, engine, device):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.engine, self.device = engine, device
        self.lib = nat.lib()
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        uid = (C.c_uint8 * nat.KMC_COMM_ID_BYTES)()
        box = [None]
        if self.rank == 0:
            # a failure here (librccl cannot be loaded) must still reach the other ranks' broadcast, or they wait forever
            try:
                nat.check(self.lib.kmc_comm_unique_id(uid))
                box = [bytes(uid)]
            except nat.KmcError as e:
                box = [str(e)]
        dist.broadcast_object_list(box, src=0)
        if not isinstance(box[0], bytes):
            raise nat.KmcError(2, f"rank 0 could not create an RCCL unique id: {box[0]}")
        uid = (C.c_uint8 * nat.KMC_COMM_ID_BYTES)(*box[0])
        with _watchdog("ncclCommInitRank (every rank of the job has to join the communicator)", self.rank):
            nat.check(self.lib.kmc_comm_init(engine.mc.handle, uid))
        self.level_bytes = []   # bytes this rank received per BFS level (observability: bench.py --gpus N reports them)

    def selftest(self):
        with _watchdog("the exchange self-test (an all-gather and a send/receive ring over RCCL on the engine's stream)",
                       getattr(self, "rank", "?")):
            nat.check(self.lib.kmc_comm_selftest(self.engine.mc.handle))

    def exchange(self, sends, stats):
        (st,) = stats
        st = np.ascontiguousarray(st, dtype=np.int64)
        out = np.zeros(N_STATS, dtype=np.int64)
        nrecv = C.c_uint64()
        nat.check(self.lib.kmc_step_exchange_counts(self.engine.mc.handle, st.ctypes.data_as(C.POINTER(C.c_int64)), N_STATS,
                                                    out.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(nrecv)))
        return out, self._deliver

    # A level whose GLOBAL frontier (the same number on every rank) has at least PIPELINE_MIN_STATES states runs as a pipeline
    # of PIPELINE_PARTS parts under the ABI (kmc_step_level_parts: the wire of a part hides behind the expansion of the next
    # one), smaller levels in one shot.  Every part costs a host wait, so the threshold is far above the headline's widest
    # level (29 M states): the pipeline is for the configurations that take seconds.  KMC_PIPELINE_PARTS=1
    # switches it off, KMC_PIPELINE_MIN_STATES moves the threshold (the tests run it on everything).
    PIPELINE_PARTS = int(os.environ.get("KMC_PIPELINE_PARTS", "4"))
    PIPELINE_MIN_STATES = int(os.environ.get("KMC_PIPELINE_MIN_STATES", str(1 << 26)))

    def expand_and_exchange(self, engines, stats, level_states=0):
        """The level's expansion and its count exchange in one call under the ABI (kmc_step_expand_counts): the send counts
        go from k_expand's control block into the all-gather on the device, and the host waits once.  `level_states`: the
        global size of the newest level whose statistics have been REDUCED (identical on every rank) — the level before the
        one being expanded, whose own size is still in `stats`, unreduced: the choice between one shot and pipeline therefore
        lags one BFS level (levels grow by less than 6 x here, so the first level past the threshold runs one-shot), and the
        first level after a resume (0) is never pipelined."""
        (st,) = stats
        st = np.ascontiguousarray(st, dtype=np.int64)
        out = np.zeros(N_STATS, dtype=np.int64)
        nrecv = C.c_uint64()
        if not hasattr(self, "level_bytes"):
            self.level_bytes = []
        parts = type(self).PIPELINE_PARTS
        if parts > 1 and level_states >= type(self).PIPELINE_MIN_STATES and self.world > 1:
            nat.check(self.lib.kmc_step_level_parts(self.engine.mc.handle, parts, st.ctypes.data_as(C.POINTER(C.c_int64)), N_STATS,
                                                    out.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(nrecv)))
            self.level_bytes.append(int(nrecv.value) * self.engine.record_words * 8)
            self.pipelined_levels = getattr(self, "pipelined_levels", 0) + 1
            return out, lambda: [[]]          # the payload has travelled and is being inserted: nothing left to deliver
        nat.check(self.lib.kmc_step_expand_counts(self.engine.mc.handle, st.ctypes.data_as(C.POINTER(C.c_int64)), N_STATS,
                                                  out.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(nrecv), None))
        self.level_bytes.append(int(nrecv.value) * self.engine.record_words * 8)
        return out, self._deliver

    def _deliver(self):
        nat.check(self.lib.kmc_step_exchange_payload(self.engine.mc.handle))
        return [[]]

    def all_reduce_sum(self, stats):
        torch, dist = self.torch, self.dist
        t = torch.from_numpy(np.sum(np.stack(stats), axis=0)).to(self.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def all_reduce_max(self, x: float) -> float:
        torch, dist = self.torch, self.dist
        t = torch.tensor([x], dtype=torch.float64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def barrier(self):
        self.dist.barrier()


class DistExchange:
    """torch.distributed: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
    One engine per process."""

    def __init__(self, device=None, record_words: Optional[int] = None):
        """`record_words` is the engine's exchange record size (state words, +1 when predecessor
        fingerprints travel).  It must be given by the caller — every rank has to size its receive
        buffers and its rounds identically, also a rank that has nothing to send in this level (then
        there is no outgoing chunk to read a width from: that was a bug, non-owner ranks of BFS level 1
        allocated 1-word receive records and the collective's counts diverged)."""
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.device = device if device is not None else torch.device("cpu")
        self.record_words = record_words

    # all_to_all_single corrupts messages above 2 GiB on this RCCL / torch build (probe:
    # tools/a2a_probe.py — half of a 2.24 GiB message arrives wrong), so a level's exchange is
    # cut into rounds of at most ROUND_BYTES per rank.
    ROUND_BYTES = 1 << 30

    def exchange(self, sends, stats):
        """One all-gather carries this level's send counts AND the statistics of the previous
        expansion (a collective and a device sync fewer per BFS level).  Returns the summed
        statistics and a `deliver` closure that performs the payload all-to-all."""
        torch, dist = self.torch, self.dist
        (mine,) = sends  # one engine per process: mine[d] = chunks for destination d
        (st,) = stats
        P = self.world
        chunks = [_chunks(x) for x in mine]
        counts = [sum(int(c.shape[0]) for c in cs) for cs in chunks]
        row = torch.tensor(counts + [int(x) for x in st], dtype=torch.int64, device=self.device)
        rows = [torch.empty_like(row) for _ in range(P)]
        dist.all_gather(rows, row)                       # (gloo has no all_gather_into_tensor)
        m = torch.stack(rows).cpu()
        allc = m[:, :P]                                  # allc[s][d]: records s sends to d
        global_stats = m[:, P:].sum(dim=0).numpy()
        return global_stats, lambda: self._deliver(chunks, allc)

    def all_to_all(self, sends):  # counts exchange + payload, without piggy-backed statistics
        _, deliver = self.exchange(sends, [np.zeros(N_STATS, dtype=np.int64)])
        return deliver()

    def _deliver(self, chunks, allc):
        torch, dist = self.torch, self.dist
        P, me = self.world, self.rank
        words = self.record_words
        if words is None:
            raise RuntimeError("DistExchange needs the engine's record_words (a rank with nothing to send cannot infer it)")
        for cs in chunks:
            for c in cs:
                if int(c.shape[1]) != words:
                    raise RuntimeError(f"exchange record width {int(c.shape[1])} != engine record_words {words}")
        dtype = torch.int64
        if int(allc.max().item()) == 0:
            return [[]]                                  # nothing moves this level (every successor was local)
        per_dst = [torch.cat(cs, dim=0) if len(cs) > 1 else (cs[0] if cs else torch.empty((0, words), dtype=dtype,
                   device=self.device)) for cs in chunks]
        budget = max(1, self.ROUND_BYTES // (words * 8 * P))  # records per (source, destination) per round
        rounds = max(1, -(-int(allc.max().item()) // budget))
        out = []
        for r in range(rounds):
            in_split = [int(min(max(int(allc[me][d]) - r * budget, 0), budget)) for d in range(P)]
            out_split = [int(min(max(int(allc[s][me]) - r * budget, 0), budget)) for s in range(P)]
            send = torch.cat([per_dst[d][r * budget:r * budget + in_split[d]] for d in range(P)], dim=0).contiguous()
            recv = torch.empty((sum(out_split), words), dtype=dtype, device=self.device)
            dist.all_to_all_single(recv, send, output_split_sizes=out_split, input_split_sizes=in_split)
            out.append(recv)
        if self.device.type == "cuda":
            # libkmc launches k_insert on its own stream: the collective (torch / RCCL streams) must
            # have landed before the engine reads the receive buffers
            torch.cuda.synchronize(self.device)
        return [out]

    def all_reduce_sum(self, stats):
        torch, dist = self.torch, self.dist
        t = torch.from_numpy(np.sum(np.stack(stats), axis=0)).to(self.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def all_reduce_max(self, x: float) -> float:
        torch, dist = self.torch, self.dist
        t = torch.tensor([x], dtype=torch.float64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def barrier(self):
        self.dist.barrier()


def _chunks(x):
    """A destination's payload is one tensor or a list of tensors."""
    return list(x) if isinstance(x, (list, tuple)) else [x]


def run_sharded(engines: Sequence, exchange, cfg: CheckerConfig, action_names: List[str],
                progress=None, checkpoint_dir: Optional[str] = None, resume_dir: Optional[str] = None) -> CheckResult:
    """Level-synchronous BFS over all shards.  Returns the global result (identical on every
    rank).  `engines` are this process's shards.

    checkpoint_dir: when the run stops at max_levels (verdict level_limit) every shard saves its table / frontier
    there (shard<i>of<P>.ckpt, kmc_checkpoint_save) and the global counters go to driver.json — TLC's -checkpoint for a
    sharded search.  resume_dir: load such a checkpoint into shards opened with the same constants, capacities and
    shard ids and continue as if the search had never stopped (-recover); cfg.max_levels of this run applies."""
    import json
    t0 = time.perf_counter()
    inv_names = tuple(n for n in nat.invariant_names(cfg.model) if n != "?")
    levels, generated, deadlocks = [], 0, 0
    filtered = 0  # remote successors dropped by the sender-side filters (all shards)
    action_generated = [0] * nat.KMC_MAX_KINDS
    verdict, viol_inv, viol_depth, viol_count = "ok", None, 0, {n: 0 for n in inv_names}

    def absorb(st, parent_depth):
        """st describes the expansion of the level at `parent_depth`.  Returns (stop, new)."""
        nonlocal generated, deadlocks, verdict, viol_inv, viol_depth, viol_count, filtered
        filtered += int(st[24])
        if viol_inv is None:
            counts = {n: int(st[17 + k]) for k, n in enumerate(inv_names)}
            hit = [n for n in inv_names if n in cfg.invariants and counts[n]]
            if hit:
                viol_inv, viol_depth, viol_count = hit[0], parent_depth, counts
                verdict = "invariant"
                if not cfg.continue_on_violation:
                    return True, 0          # the level this expansion produced is rolled back
        if viol_inv is None:
            # successors outside the state constraint that violate an invariant (AsyncIsr): they sit
            # one level below the expanded states, which is why those were looked at first
            counts = {n: int(st[25 + k]) for k, n in enumerate(inv_names)}
            hit = [n for n in inv_names if n in cfg.invariants and counts[n]]
            if hit:
                viol_inv, viol_depth, viol_count = hit[0], parent_depth + 1, counts
                verdict = "invariant"
                if not cfg.continue_on_violation:
                    return True, 0
        for k in range(15):
            action_generated[k] += int(st[1 + k])
            generated += int(st[1 + k])
        generated += int(st[16])
        deadlocks += int(st[21])
        if st[22]:
            verdict = "table_full"
            return True, 0
        if st[23]:
            verdict = "frontier_full"
            return True, 0
        if cfg.check_deadlock and st[21] and verdict == "ok":
            verdict, viol_depth = "deadlock", parent_depth
            return True, 0
        return False, int(st[0])

    # The statistics of expansion k travel with the send counts of expansion k+1 (one all-gather),
    # so expansion k+1 has already run locally when its predecessor's verdict becomes known: after
    # a stop, or once the global frontier is empty, that speculative expansion is simply not
    # counted (on an empty frontier it did nothing at all).
    # Exception: a model with a state constraint whose traces are wanted.  A violating successor OUTSIDE the
    # constraint is in no table; its words and its parent can only be recovered from the level that generated
    # it, which the speculative expansion would already have overwritten.  Those runs reduce the statistics
    # first and expand afterwards (one small collective more per level).
    pipelined = not (cfg.keep_trace and cfg.model == "AsyncIsr")
    zeros = [np.zeros(N_STATS, dtype=np.int64) for _ in engines]
    P = engines[0].n_shards
    if resume_dir:
        drv = json.load(open(os.path.join(resume_dir, "driver.json")))
        if drv["n_shards"] != P or drv["model"] != cfg.model:
            raise ValueError(f"checkpoint in {resume_dir} is for {drv['model']} on {drv['n_shards']} shards")
        # the shards come back holding the last recorded level, complete and unexpanded: it is recorded again from
        # the sizes they report, everything before it comes from the driver file
        pending = []
        for e in engines:
            st = np.zeros(N_STATS, dtype=np.int64)
            st[0] = e.load_checkpoint(os.path.join(resume_dir, f"shard{e.shard_id}of{P}.ckpt"))
            pending.append(st)
        levels = list(drv["levels"][:-1])
        generated, deadlocks, filtered = drv["generated"], drv["deadlocks"], drv["filtered"]
        action_generated = list(drv["action_generated"])
        depth = len(levels)
        pending_depth = depth
    else:
        pending = [e.begin() for e in engines]   # per-engine statistics not yet reduced ...
        pending_depth = 0                        # ... of the expansion of this level (0: Init's insertion)
        depth = 0                                # levels recorded so far
    max_levels = cfg.max_levels or (1 << 62)
    new = 0
    while True:
        # the engines hold level depth+1 (complete); like kmc_run, level l is expanded iff l < max_levels
        can_expand = depth + 1 < max_levels
        if can_expand and pipelined and hasattr(exchange, "expand_and_exchange"):
            # fused under the ABI: one host wait (or, for a very wide level, a pipeline of parts: RcclExchange).  `new` is the
            # global size of the last level RECORDED (the one before the level being expanded, whose statistics are still in
            # `pending`): the same on every rank, one level behind.
            # (under orbit counting `new` is the WEIGHTED count, |Replicas|! times what the shards store, expand and ship: the
            # choice between one shot and pipeline is about records on the wire, so it sees the stored representatives)
            stored = new // math.factorial(cfg.n_replicas) if cfg.symmetry else new
            st, deliver = exchange.expand_and_exchange(engines, pending, level_states=stored)
        elif can_expand and pipelined:
            sends = [e.expand() for e in engines]
            st, deliver = exchange.exchange(sends, pending)
        else:
            st, deliver = exchange.all_reduce_sum(pending), None
        stopped, produced = absorb(st, pending_depth)
        if stopped:
            break
        new = produced
        if new == 0:
            break
        depth += 1
        levels.append(new)
        if progress:
            progress(dict(depth=depth, new_states=new, generated=generated, distinct=sum(levels)))
        if not can_expand:
            break
        if not pipelined:
            sends = [e.expand() for e in engines]
            _, deliver = exchange.exchange(sends, zeros)
        recvs = deliver()
        for e, rs in zip(engines, recvs):
            for r in rs:
                e.insert(r)
        pending = [e.finish() for e in engines]
        pending_depth = depth
    if depth >= max_levels and new > 0 and verdict in ("ok", "invariant") and viol_inv is None:
        # like kmc_run: the last frontier is not expanded, so its states get their invariant check now
        st = exchange.all_reduce_sum([e.check_frontier() for e in engines])
        counts = {n: int(st[17 + k]) for k, n in enumerate(inv_names)}
        hit = [n for n in inv_names if n in cfg.invariants and counts[n]]
        if hit:
            viol_inv, viol_depth, viol_count, verdict = hit[0], depth, counts, "invariant"
    if depth >= max_levels and new > 0 and verdict == "ok":
        verdict = "level_limit"
        if checkpoint_dir:
            os.makedirs(checkpoint_dir, exist_ok=True)
            for e in engines:
                e.save_checkpoint(os.path.join(checkpoint_dir, f"shard{e.shard_id}of{P}.ckpt"))
            if any(e.shard_id == 0 for e in engines):
                with open(os.path.join(checkpoint_dir, "driver.json"), "w") as f:
                    json.dump(dict(model=cfg.model, n_shards=P, levels=levels, generated=generated, deadlocks=deadlocks,
                                   filtered=filtered, action_generated=action_generated), f)
            exchange.barrier()
    trace = []
    if verdict == "invariant" and cfg.keep_trace:
        outside = viol_depth > len(levels)   # a witness outside the state constraint: in no shard's table
        trace = _sharded_trace(engines, exchange, inv_names.index(viol_inv), action_names, outside)
    local = [e.result() for e in engines]
    tail = exchange.all_reduce_sum([np.array([sum(r.generated_repeats for r in local),
                                              sum(r.orbit_representatives for r in local)], dtype=np.int64)])
    run_sharded.last_send_filtered = filtered  # observability for tests / bench
    return CheckResult(
        generated=generated, distinct=sum(levels), depth=len(levels),
        queue_left=(levels[-1] if verdict != "ok" else 0), verdict=verdict, violated_invariant=viol_inv,
        violation_depth=viol_depth, violation_count=viol_count, violation_fp=0, deadlock_states=deadlocks,
        action_generated={n: action_generated[k] for k, n in enumerate(action_names)}, levels=levels,
        table_capacity=sum(r.table_capacity for r in local), frontier_capacity=sum(r.frontier_capacity for r in local),
        seconds_total=time.perf_counter() - t0, seconds_expand=max(r.seconds_expand for r in local),
        expand_launches=max(r.expand_launches for r in local), state_words=local[0].state_words,
        state_bits=local[0].state_bits, trace=trace,
        generated_repeats=int(tail[0]),
        # CheckerConfig.symmetry: every shard weighs its own counts (kmc_step_finish: N! x stored - deficits), so the sums
        # above are the plain search's numbers; what the shards actually stored and expanded, summed:
        orbit_representatives=int(tail[1]))


def _split64(x):
    return [x & 0xFFFFFFFF, x >> 32]


def _min_fp_over_shards(engines, exchange, values):
    """values: {shard_id: fingerprint or 0} of this process's engines -> the smallest non-zero one over all shards."""
    P = engines[0].n_shards
    v = np.zeros(2 * P, dtype=np.int64)
    for sid, fp in values.items():
        v[2 * sid:2 * sid + 2] = _split64(fp)
    v = exchange.all_reduce_sum([v])
    cands = [int(v[2 * i]) | (int(v[2 * i + 1]) << 32) for i in range(P)]
    cands = [c for c in cands if c]
    return min(cands) if cands else 0


def _sharded_trace(engines, exchange, inv_index, action_names, outside=False):
    """Counterexample of a sharded run (keep_trace): [(action name or None, canonical bytes)] from Init to
    the witness.  Every rank executes the same steps: (1) the witness is the smallest violating
    fingerprint over all shards (one small reduction); (2) the predecessor chain is walked owner by
    owner — the owner of a fingerprint looks it up in its table (kmc_pred_of), one small reduction per
    step carries the answer to everybody; (3) the chain is replayed forward from Init with the device's
    own successor enumeration, locally, picking at each step the successor whose fingerprint is next —
    what TLC does with its trace file [TLC-recall]."""
    tail = []     # an outside-the-constraint witness: [its fingerprint]; the chain proper starts at its parent
    if outside:
        # the witness is the smallest violating outside-fingerprint over all shards; every shard that generates
        # it from its retired level offers a parent, the smallest parent fingerprint wins
        wfp = _min_fp_over_shards(engines, exchange, {e.shard_id: e.outside_violation_fp(inv_index) for e in engines})
        if not wfp:
            return []
        offers = {}
        for e in engines:
            hit = e.find_outside(wfp)
            offers[e.shard_id] = hit[1] if hit else 0
        start = _min_fp_over_shards(engines, exchange, offers)
        if not start:
            raise RuntimeError(f"trace: no shard generates the outside-constraint witness {wfp:016x}")
        tail = [wfp]
    else:
        start = _min_fp_over_shards(engines, exchange, {e.shard_id: e.violation_fp(inv_index) for e in engines})
        if not start:
            return []
    chain = [start]
    for _guard in range(1 << 16):
        fp = chain[-1]
        a = np.zeros(3, dtype=np.int64)
        for e in engines:
            if e.owner(fp) == e.shard_id:
                pred = e.pred_of(fp)
                if pred is not None:
                    a[0], a[1], a[2] = 1, *_split64(pred)
        a = exchange.all_reduce_sum([a])
        if int(a[0]) != 1:
            raise RuntimeError(f"trace: fingerprint {fp:016x} is in no shard's table")
        pred = int(a[1]) | (int(a[2]) << 32)
        if pred == 0:
            break
        chain.append(pred)
    chain.reverse()
    chain += tail
    e0 = engines[0]
    cur = e0.init_words()
    if e0.fingerprint(cur) != chain[0]:
        raise RuntimeError("trace does not start at Init")
    out = [(None, e0.canonical(cur))]
    for want in chain[1:]:
        for words, fp, kind in e0.successors(cur):
            if fp == want:
                cur = list(words)
                out.append((action_names[kind], e0.canonical(cur)))
                break
        else:
            raise RuntimeError("trace replay lost the path")
    return out


def check_loopback(cfg: CheckerConfig, n_shards: int, device: int = 0, progress=None, checkpoint_dir=None,
                   resume_dir=None) -> CheckResult:
    """P logical shards on ONE GPU with an in-process exchange (tests the bucket / insert kernels
    and the level logic without RCCL)."""
    native = os.environ.get("KMC_EXCHANGE", "rccl") != "torch"
    engines = [HipShardEngine(cfg, s, n_shards, device, native=native) for s in range(n_shards)]
    try:
        ex = NativeLoopbackExchange(engines) if native else LoopbackExchange(n_shards)
        return run_sharded(engines, ex, cfg, engines[0].mc.action_names(), progress, checkpoint_dir, resume_dir)
    finally:
        for e in engines:
            e.close()


def make_engine_and_exchange(cfg: CheckerConfig, rank: int, world: int, local: int, device):
    """The shard engine and per-level exchange of a one-shard-per-rank job.

    On GPUs the product is a native HipShardEngine with the exchange under the C ABI (RcclExchange).  Before it is
    used, every rank runs kmc_comm_selftest on the fresh communicator — an all-gather and a grouped send/receive ring
    across ALL ranks on the engine's stream, verified — and the ranks agree (one small all-reduce) on whether it
    worked everywhere.  If any rank could not bring it up (librccl not loadable, a failed RCCL call, a wrong
    pattern), all of them fall back TOGETHER to the torch.distributed exchange (DistExchange, still RCCL, still the
    GPU — there is no CPU path) and say so on stderr; a rank-local decision would desynchronise the collectives.
    KMC_EXCHANGE=torch selects that exchange from the start; a stand-in engine on gloo always gets it."""
    import sys
    factory = _engine_factory()
    want_native = device.type == "cuda" and factory is HipShardEngine and os.environ.get("KMC_EXCHANGE", "rccl") != "torch"
    import torch
    import torch.distributed as dist
    if not want_native:
        # (the same agreement as below: a rank that cannot open its engine — out of memory at the stretch leg's capacities, a cold
        # cache that does not compile — must not leave the others waiting in the first collective of the search: ADVICE r5)
        eng, err = None, ""
        try:
            eng = factory(cfg, rank, world, local)
        except Exception as e:   # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        if dist.is_initialized() and world > 1:
            flag = torch.tensor([0 if err else 1], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item()) == 1
        else:
            ok = not err
        if not ok:
            if eng is not None:
                eng.close()
            raise RuntimeError(f"rank {rank}: a shard engine could not be created on some rank"
                               f"{' (here: ' + err + ')' if err else ''}; no rank continues")
        return eng, DistExchange(device, eng.record_words)
    eng, ex, err = None, None, ""
    try:
        eng = HipShardEngine(cfg, rank, world, local, native=True)
    except Exception as e:   # noqa: BLE001 — out of memory, a kernel that does not compile: the other ranks must hear of it
        err = f"{type(e).__name__}: {e}"
    # agree on the engines BEFORE the unique-id broadcast inside RcclExchange: a rank that skipped that broadcast would leave
    # the others waiting in it for ever (the watchdog only covers ncclCommInitRank and the self-test)
    flag = torch.tensor([0 if err else 1], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        if eng is not None:
            eng.close()
        raise RuntimeError(f"rank {rank}: a shard engine could not be created on some rank"
                           f"{' (here: ' + err + ')' if err else ''}; no rank continues")
    try:
        ex = RcclExchange(eng, device)
        ex.selftest()
    except Exception as e:   # noqa: BLE001 — any failure here must reach the agreement below, not kill one rank
        err = f"{type(e).__name__}: {e}"
    flag = torch.tensor([0 if err else 1], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        return eng, ex
    if rank == 0 or err:
        print(f"[kmc] rank {rank}: the exchange under the C ABI is not usable on every rank"
              f"{' (' + err + ')' if err else ''}; all ranks fall back to the torch.distributed exchange", file=sys.stderr)
    if eng is not None:
        eng.close()
    eng = HipShardEngine(cfg, rank, world, local, native=False)
    return eng, DistExchange(device, eng.record_words)


def check_distributed(cfg: CheckerConfig, progress=None, checkpoint_dir=None, resume_dir=None) -> CheckResult:
    """One shard per rank of the default process group (launch with torch.distributed.run)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    local = int(os.environ.get("LOCAL_RANK", rank))
    device = torch.device("cuda", local) if dist.get_backend() == "nccl" else torch.device("cpu")
    eng, ex = make_engine_and_exchange(cfg, rank, world, local, device)
    try:
        return run_sharded([eng], ex, cfg, eng.action_names(), progress, checkpoint_dir, resume_dir)
    finally:
        eng.close()


def _engine_factory():
    """The shard engine of the N>1 legs.  Always HipShardEngine (libkmc.so on a GPU) unless
    KMC_SHARD_ENGINE="module:callable" names a stand-in with the same (cfg, shard, n_shards, device)
    signature — used by the CPU test of bench.py's launch path (tests/test_bench_launch_cpu.py), which has
    no GPU to give the real engine."""
    spec = os.environ.get("KMC_SHARD_ENGINE")
    if not spec:
        return HipShardEngine
    import importlib
    mod, _, name = spec.partition(":")
    return getattr(importlib.import_module(mod), name)


def bench_sharded(c: dict, steps: int, warmup: int, backend: str = "nccl", symmetry: bool = False, wide_fingerprint: bool = False,
                  capacities=None):
    """bench.py's N>1 leg: strong scaling of the headline check over the ranks of this job.
    backend "nccl" (= RCCL over xGMI) is the product; "gloo" exists so that the launch / rendezvous /
    timing / rank-0-prints path can be exercised on a CPU box with a stand-in engine."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    on_gpu = backend == "nccl"
    if on_gpu and not torch.cuda.is_available():
        # said here, before the process group is created, so that the message names the cause
        raise RuntimeError(f"bench.py --gpus {os.environ.get('WORLD_SIZE', '?')}: no HIP device is visible to rank "
                           f"{os.environ.get('RANK', '?')}; the multi-GPU leg runs one shard per GPU over RCCL "
                           "(there is no CPU fallback)")
    if not dist.is_initialized():
        dist.init_process_group(backend=backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    local = int(os.environ.get("LOCAL_RANK", rank))
    if on_gpu:
        if not torch.cuda.is_available() or torch.cuda.device_count() <= local:
            raise RuntimeError(f"bench.py --gpus {world}: rank {rank} needs HIP device {local}, "
                               f"{torch.cuda.device_count() if torch.cuda.is_available() else 0} visible "
                               "(there is no CPU fallback)")
        torch.cuda.set_device(local)
    device = torch.device("cuda", local) if on_gpu else torch.device("cpu")
    per = max(1, world)
    # per rank: table for its 1/P of the states at load <= 0.5, frontier for its share of the widest
    # level (2.6e7 states) with 2x slack, send sub-buffers for its share of that level's successors
    # (symmetry: orbit counting on every rank — successors travel as representatives, each rank weighs its own counters)
    # (`capacities` = (table, frontier, send) for the whole job, divided here: bench.py's stretch leg brings its own)
    tab, fro, snd = capacities or (1 << 30, 1 << 26, 1 << 25)
    cfg = CheckerConfig(**c, symmetry=symmetry, wide_fingerprint=wide_fingerprint,
                        table_capacity=int(os.environ.get("KMC_BENCH_TABLE", max(1 << 27, tab // per))),
                        frontier_capacity=int(os.environ.get("KMC_BENCH_FRONTIER", max(1 << 23, fro // per))),
                        send_capacity=int(os.environ.get("KMC_BENCH_SEND", max(1 << 18, snd // (per * per)))))
    eng, ex = make_engine_and_exchange(cfg, rank, world, local, device)
    names = eng.action_names()
    results = []

    def sync():
        dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    try:
        for _ in range(warmup):
            run_sharded([eng], ex, cfg, names)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            results.append(run_sharded([eng], ex, cfg, names))
        sync()
        dt = ex.all_reduce_max(time.perf_counter() - t0)
        # what each rank did in the timed steps, gathered for rank 0's line: seconds inside k_expand (HIP events on the
        # rank's engine stream), states it owns, bytes it received over the exchange, BFS levels that moved data
        own = eng.result()
        lb = getattr(ex, "level_bytes", [])
        mine = np.zeros((world, 4), dtype=np.float64)
        mine[rank] = [own.seconds_expand, float(own.distinct), float(sum(lb[-len(results[-1].levels):]) if lb else 0),
                      float(sum(1 for b in lb[-len(results[-1].levels):] if b))]
        if world > 1:
            t = torch.from_numpy(mine).to(device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            per_rank = t.cpu().numpy()
        else:
            per_rank = mine
    finally:
        eng.close()
    # what the first curve on real GPUs will be read against: did the exchange under the C ABI come up on every rank (the
    # self-test in make_engine_and_exchange ran before anything was timed: an all-gather and a send / receive ring over RCCL,
    # verified; a failure anywhere makes ALL ranks fall back to DistExchange), and how many levels took the within-level pipeline
    extra = {"shards": world, "exchange": type(ex).__name__,
             "rccl_ranks": world if isinstance(ex, RcclExchange) else 0,
             "pipelined_levels": int(getattr(ex, "pipelined_levels", 0)),
             "pipeline_min_states": int(getattr(type(ex), "PIPELINE_MIN_STATES", 0)),
             "pipeline_parts": int(getattr(type(ex), "PIPELINE_PARTS", 0)),
             "per_rank": [dict(rank=i, expand_kernel_seconds_last_step=float(per_rank[i][0]), states_owned=int(per_rank[i][1]),
                               received_bytes_last_step=int(per_rank[i][2]), levels_with_traffic=int(per_rank[i][3]))
                          for i in range(world)]}
    return results, dt, extra
