"""Oracle-backed stand-in for HipShardEngine (test infrastructure: it calls the oracle).  The CPU tests
plug it into kafka_specification_amd.sharded.run_sharded and — through KMC_SHARD_ENGINE=shard_standin:make_engine —
into bench.py's N>1 leg, so that the level logic, the exchange and the launch path run under a real gloo process
group on a box without GPUs.  The product engine is HipShardEngine (libkmc.so), covered by the -m gpu tests."""
import hashlib

import numpy as np
import torch

import kmo
from kafka_specification_amd import _native as nat
from kafka_specification_amd.checker import CheckerConfig
from kafka_specification_amd.sharded import N_STATS

INV_INDEX = {"TypeOk": 0, "WeakIsr": 1, "StrongIsr": 2, "LeaderInIsr": 3,
             "ValidHighWatermark": 1, "LeaderOffsetInRange": 2}   # AsyncIsr reuses the positions


class OracleShardEngine:
    """Oracle-backed stand-in for HipShardEngine: same begin/expand/insert/finish interface, plus the
    trace hooks (violation_fp / owner / pred_of / init_words / successors / fingerprint / canonical).
    A "packed state" here is the canonical byte string padded to 8-byte words."""

    def __init__(self, cfg: CheckerConfig, rank, world, hip_shaped=True):
        # hip_shaped: expand() returns what HipShardEngine.expand() returns — per destination a LIST of
        # filled sub-buffer slices, and an EMPTY list for a destination that gets nothing (no zero-row tensor
        # to read a record width from: ADVICE r1, DistExchange must not infer the width from the data)
        self.hip_shaped = hip_shaped
        self.cfg, self.rank, self.world = cfg, rank, world
        self.shard_id, self.n_shards = rank, world
        self.kcfg = kmo.make_config(cfg.model, N=cfg.n_replicas, L=cfg.log_size, R=cfg.max_records,
                                    E=cfg.max_leader_epoch, invariants=())
        probe = kmo.Run(kmo.make_config(cfg.model, N=cfg.n_replicas, L=cfg.log_size, R=cfg.max_records,
                                        E=cfg.max_leader_epoch, invariants=(), max_states=1))
        self.sb = probe.sb
        self.init = probe.state(0)
        probe.close()
        self.words = (self.sb + 7) // 8
        self.rec_words = self.words + (1 if cfg.keep_trace else 0)   # the predecessor fingerprint travels for traces
        self.seen, self.frontier, self.next = {}, [], []             # seen: state -> predecessor fingerprint
        self.level = 0
        self._viol_fp = [0, 0, 0, 0]
        self._oviol_fp = [0, 0, 0, 0]
        self.retired = []                                             # the level the last finish() retired
        # CheckerConfig.symmetry: this stand-in weighs as kmc_step_finish does — only orbit representatives are kept, shipped
        # and expanded; what a state's expansion counts is multiplied by its orbit's size N!/|Stab|, a new state counts its
        # orbit's size where it is admitted (its owner).  Representative and stabiliser come from the product's HOST-ONLY
        # handle (kmc_canonical_state needs no device); the successor function stays the oracle's.
        self.sym = bool(getattr(cfg, "symmetry", False))
        self.next_weight = 0
        self.weight = {}                                              # stored state -> its orbit's size
        if self.sym:
            import math
            from dataclasses import replace
            from kafka_specification_amd import ModelChecker
            self.nfact = math.factorial(cfg.n_replicas)
            self._mc = ModelChecker(replace(cfg, device=-1, symmetry=False, n_shards=1, shard_id=0))
        self.reset_level()

    def _canon(self, s: bytes):
        """-> (representative of s's orbit as canonical bytes, the orbit's size)"""
        if not self.sym:
            return s, 1
        stab, rep = self._mc.canonical(self._mc.pack(s))
        return self._mc.unpack(rep), self.nfact // stab

    def reset_level(self):
        self.st = np.zeros(N_STATS, dtype=np.int64)

    @property
    def record_words(self):
        return self.rec_words

    def action_names(self):
        n = 10 if self.cfg.model == "Kip320FirstTry" else 7 if self.cfg.model == "AsyncIsr" else 9
        return [f"a{k}" for k in range(n)]

    def close(self):
        pass

    @staticmethod
    def fp(state: bytes) -> int:
        return int.from_bytes(hashlib.blake2b(state, digest_size=8).digest(), "little") | 1

    def owner(self, fp: int) -> int:
        return (fp >> 40) % self.world

    def _enc(self, items):  # items: [(state, predecessor fingerprint)]
        buf = np.zeros((len(items), self.rec_words * 8), dtype=np.uint8)
        for i, (s, pred) in enumerate(items):
            buf[i, :self.sb] = np.frombuffer(s, dtype=np.uint8)
            if self.cfg.keep_trace:
                buf[i, self.words * 8:] = np.frombuffer(pred.to_bytes(8, "little"), dtype=np.uint8)
        return torch.from_numpy(buf.view(np.int64).reshape(len(items), self.rec_words))

    def _admit(self, s: bytes, pred: int):
        if s in self.seen:
            return
        self.seen[s] = pred
        self.next.append(s)
        w = self._canon(s)[1]
        self.weight[s] = w
        self.next_weight += w

    def begin(self):
        self.seen, self.frontier, self.next = {}, [], []   # like kmc_step_begin: a fresh search
        self.reset_level()
        self.weight, self.next_weight = {}, 0
        init = self._canon(self.init)[0]
        if self.owner(self.fp(init)) == self.rank:
            self._admit(init, 0)
            self.st[16] = 1
        return self._close_level()

    def expand(self):
        self.reset_level()
        viol = [0, 0, 0, 0]
        oviol = [0, 0, 0, 0]
        buckets = [[] for _ in range(self.world)]
        for s in self.frontier:
            fps = self.fp(s)
            w = self.weight.get(s, 1)     # everything this expansion counts stands for the whole orbit of s
            for name in self.cfg.invariants:  # like the GPU engine: a state is checked when it is expanded
                if not kmo.check_invariant(self.kcfg, INV_INDEX[name], s):
                    k = INV_INDEX[name]
                    self.st[17 + k] += w
                    viol[k] = fps if viol[k] == 0 else min(viol[k], fps)
            succ = kmo.successors(self.kcfg, s, self.sb)
            if not succ:
                self.st[21] += w
            for a, t in succ:
                self.st[1 + a] += w
                t = self._canon(t)[0]     # (symmetry) successors travel as representatives
                if self.cfg.model == "AsyncIsr" and (t[6] > self.cfg.log_size or t[1] > self.cfg.max_leader_epoch):
                    # outside the state constraint: invariant-checked, neither kept nor shipped
                    for name in self.cfg.invariants:
                        if not kmo.check_invariant(self.kcfg, INV_INDEX[name], t):
                            k = INV_INDEX[name]
                            self.st[25 + k] += 1
                            oviol[k] = self.fp(t) if oviol[k] == 0 else min(oviol[k], self.fp(t))
                    continue
                buckets[self.owner(self.fp(t))].append((t, fps))
        self._viol_next = viol
        self._oviol_next = oviol
        self._expanding = list(self.frontier)
        if not self.hip_shaped:
            return [self._enc(b) for b in buckets]
        subs = nat.KMC_SEND_SUBS
        return [[self._enc(b[sb::subs]) for sb in range(subs) if b[sb::subs]] for b in buckets]

    def insert(self, records):
        raw = records.contiguous().numpy().view(np.uint8).reshape(records.shape[0], self.rec_words * 8)
        for i in range(raw.shape[0]):
            pred = int.from_bytes(raw[i, self.words * 8:].tobytes(), "little") if self.cfg.keep_trace else 0
            self._admit(raw[i, :self.sb].tobytes(), pred)

    def finish(self):
        self._viol_fp = self._viol_next
        self._oviol_fp = self._oviol_next
        self.retired = self._expanding
        return self._close_level()

    def save_checkpoint(self, path):
        import pickle
        with open(path, "wb") as f:
            pickle.dump(dict(seen=self.seen, frontier=self.frontier), f)

    def load_checkpoint(self, path):
        import pickle
        d = pickle.load(open(path, "rb"))
        self.seen, self.frontier, self.next = d["seen"], d["frontier"], []
        self.reset_level()
        return len(self.frontier)

    def check_frontier(self):
        """Invariant-only pass over the current, unexpanded frontier (the last level under max_levels)."""
        st = np.zeros(N_STATS, dtype=np.int64)
        viol = [0, 0, 0, 0]
        for s in self.frontier:
            for name in self.cfg.invariants:
                if not kmo.check_invariant(self.kcfg, INV_INDEX[name], s):
                    k = INV_INDEX[name]
                    st[17 + k] += self.weight.get(s, 1)
                    viol[k] = self.fp(s) if viol[k] == 0 else min(viol[k], self.fp(s))
        self._viol_fp = viol
        return st

    def outside_violation_fp(self, k):
        return self._oviol_fp[k]

    def find_outside(self, fp):
        best = None
        for s in self.retired:
            for _a, t in kmo.successors(self.kcfg, s, self.sb):
                if self.fp(t) == fp and (best is None or self.fp(s) < best[1]):
                    best = (self._pack(t), self.fp(s))
        return best

    def _close_level(self):
        self.frontier, self.next = self.next, []
        self.st[0] = self.next_weight if self.sym else len(self.frontier)
        self.next_weight = 0
        return self.st

    def result(self):
        from kafka_specification_amd.checker import CheckResult
        return CheckResult(0, len(self.seen), 0, 0, "ok", None, 0, {}, 0, 0, {}, [], 0, 0, 0.0, 0.0, 0, self.words, 0,
                           orbit_representatives=len(self.seen))

    # -- trace hooks -----------------------------------------------------------------------------
    def violation_fp(self, k):
        return self._viol_fp[k]

    def pred_of(self, fp):
        for s, pred in self.seen.items():
            if self.fp(s) == fp:
                return pred
        return None

    def _pack(self, s: bytes):
        return [int(x) for x in np.frombuffer(s.ljust(self.words * 8, b"\0"), dtype=np.uint64)]

    def canonical(self, words) -> bytes:
        return np.array(words, dtype=np.uint64).tobytes()[:self.sb]

    def init_words(self):
        return self._pack(self.init)

    def fingerprint(self, words):
        return self.fp(self.canonical(words))

    def successors(self, words):
        return [(tuple(self._pack(t)), self.fp(t), a) for a, t in kmo.successors(self.kcfg, self.canonical(words), self.sb)]


def make_engine(cfg: CheckerConfig, rank, world, _device):
    """KMC_SHARD_ENGINE factory: (cfg, shard, n_shards, device) like HipShardEngine."""
    return OracleShardEngine(cfg, rank, world)
