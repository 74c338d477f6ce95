"""Test-side glue for Oracle-R (oracle/tlar): TLA+ state values -> the canonical byte string the C oracle
(oracle/kmc_oracle.c, header comment) and the engine's kmc_unpack_state use, plus per-level digests.

TEST INFRASTRUCTURE ONLY.  Written against the byte layout documented in oracle/kmc_oracle.c; it imports nothing from
oracle/kafka_oracle.py (Oracle-A) — the point of Oracle-R is that no hand restatement of the specs sits between the
reference's text and the numbers.
"""
import hashlib

from oracle.tlar import Fn, ModelValue


def replica_order(constants):
    """Model values b1 < b2 < ... (by name) -> indices 0..N-1; AsyncIsr: Leader first, the others by name."""
    reps = sorted(constants["Replicas"], key=lambda m: m.name)
    if "Leader" in constants:
        reps = [constants["Leader"]] + [r for r in reps if r != constants["Leader"]]
    return {r: i for i, r in enumerate(reps)}


def _mask(s, idx):
    m = 0
    for r in s:
        m |= 1 << idx[r]
    return m


def kafka_state_bytes(st, constants):
    """vars of KafkaReplication.tla:75 -> N*(5+L) + 5 + 2*(E+1) bytes."""
    idx = replica_order(constants)
    N, L, E = len(idx), constants["LogSize"], constants["MaxLeaderEpoch"]
    out = bytearray(N * (5 + L) + 5 + 2 * (E + 1))
    for r, i in idx.items():
        log, rs = st["replicaLog"].d[r].d, st["replicaState"].d[r].d
        b = i * (5 + L)
        out[b + 0] = log["endOffset"]
        out[b + 1] = rs["hw"]
        out[b + 2] = rs["leaderEpoch"] + 1
        out[b + 3] = 0 if rs["leader"] == "NONE" else idx[rs["leader"]] + 1
        out[b + 4] = _mask(rs["isr"], idx)
        for o in range(L):
            rec = log["records"].d[o]
            out[b + 5 + o] = 0 if rec == -1 else 1 + rec.d["id"] * (E + 1) + rec.d["epoch"]
    g = N * (5 + L)
    q = st["quorumState"].d
    out[g + 0] = st["nextRecordId"]
    out[g + 1] = st["nextLeaderEpoch"]
    out[g + 2] = q["leaderEpoch"] + 1
    out[g + 3] = 0 if q["leader"] == "NONE" else idx[q["leader"]] + 1
    out[g + 4] = _mask(q["isr"], idx)
    seen = set()
    for req in st["leaderAndIsrRequests"]:
        e = req.d["leaderEpoch"]
        assert e not in seen and 0 <= e < st["nextLeaderEpoch"], "requests are not an epoch-indexed history"
        seen.add(e)
        out[g + 5 + 2 * e] = 0 if req.d["leader"] == "NONE" else idx[req.d["leader"]] + 1
        out[g + 6 + 2 * e] = _mask(req.d["isr"], idx)
    assert len(seen) == st["nextLeaderEpoch"]
    return bytes(out)


def frl_state_bytes(st, constants):
    """FiniteReplicatedLog standalone: per replica [end, rec[0..L-1]], record codes 1..K by model-value name."""
    reps = sorted(constants["Replicas"], key=lambda m: m.name)
    recs = {r: k + 1 for k, r in enumerate(sorted(constants["LogRecords"], key=lambda m: m.name))}
    L = constants["LogSize"]
    out = bytearray(len(reps) * (1 + L))
    for i, r in enumerate(reps):
        log = st["logs"].d[r].d
        out[i * (1 + L)] = log["endOffset"]
        for o in range(L):
            x = log["records"].d[o]
            out[i * (1 + L) + 1 + o] = 0 if x == constants["Nil"] else recs[x]
    return bytes(out)


def idseq_state_bytes(st, constants):
    return int(st["nextId"]).to_bytes(8, "little")


def async_state_bytes(st, constants):
    """AsyncIsr under models/MCAsyncIsr.tla; E = MaxVersion bounds the version-indexed parts of the layout."""
    idx = replica_order(constants)
    N, E = len(idx), constants["MaxVersion"]
    rb = ((1 << N) + 7) // 8
    a_req = 6 + N
    a_upd = a_req + (E + 1) * rb
    out = bytearray(a_upd + E + 1)
    c, l = st["controllerState"].d, st["leaderState"].d
    out[0], out[1] = _mask(c["isr"], idx), c["version"]
    out[2], out[3], out[4], out[5] = _mask(l["isr"], idx), l["version"], _mask(l["pendingIsr"], idx), l["pendingVersion"] + 1
    for r, i in idx.items():
        out[6 + i] = l["offsets"].d[r]
    for m in st["requests"]:
        v, isr = m.d["version"], _mask(m.d["isr"], idx)
        out[a_req + v * rb + (isr >> 3)] |= 1 << (isr & 7)
    seen = set()
    for u in st["updates"]:
        v = u.d["version"]
        assert v not in seen and 1 <= v <= c["version"], "updates are not a version-indexed history"
        seen.add(v)
        out[a_upd + v - 1] = _mask(u.d["isr"], idx)
    assert len(seen) == c["version"]
    return bytes(out)


def encoder_for(module):
    if module == "IdSequence":
        return idseq_state_bytes
    if module == "FiniteReplicatedLog":
        return frl_state_bytes
    if module in ("AsyncIsr", "MCAsyncIsr"):
        return async_state_bytes
    return kafka_state_bytes


def level_digest(byte_states):
    """sha256 over the sorted canonical byte strings of one BFS level."""
    h = hashlib.sha256()
    for b in sorted(byte_states):
        h.update(b)
    return h.hexdigest()


def kafka_constants(N, L, R, E):
    return dict(Replicas=frozenset(ModelValue(f"b{i + 1}") for i in range(N)), LogSize=L, MaxRecords=R, MaxLeaderEpoch=E)


def kafka_state_from_bytes(b, constants):
    """Inverse of kafka_state_bytes: canonical bytes -> the TLA+ values of KafkaReplication.tla:75's variables (what Oracle-R's
    evaluator takes as a state).  Lets the reference's text be evaluated on ANY state — a deep sample of a C-oracle walk, an
    arbitrary bit pattern — not only on those its own search reaches."""
    idx = replica_order(constants)
    name = {i: r for r, i in idx.items()}
    N, L, E = len(idx), constants["LogSize"], constants["MaxLeaderEpoch"]
    assert len(b) == N * (5 + L) + 5 + 2 * (E + 1)

    def unmask(m):
        return frozenset(name[i] for i in range(N) if m >> i & 1)

    def leader(x):
        return "NONE" if x == 0 else name[x - 1]

    logs, states = {}, {}
    for i in range(N):
        o = i * (5 + L)
        recs = {}
        for k in range(L):
            c = b[o + 5 + k]
            recs[k] = -1 if c == 0 else Fn({"id": (c - 1) // (E + 1), "epoch": (c - 1) % (E + 1)})
        logs[name[i]] = Fn({"endOffset": b[o], "records": Fn(recs)})
        states[name[i]] = Fn({"hw": b[o + 1], "leaderEpoch": b[o + 2] - 1, "leader": leader(b[o + 3]), "isr": unmask(b[o + 4])})
    g = N * (5 + L)
    reqs = frozenset(Fn({"leaderEpoch": e, "leader": leader(b[g + 5 + 2 * e]), "isr": unmask(b[g + 6 + 2 * e])})
                     for e in range(b[g + 1]))
    return dict(replicaLog=Fn(logs), replicaState=Fn(states), nextRecordId=b[g], nextLeaderEpoch=b[g + 1],
                quorumState=Fn({"leaderEpoch": b[g + 2] - 1, "leader": leader(b[g + 3]), "isr": unmask(b[g + 4])}),
                leaderAndIsrRequests=reqs)


def async_state_from_bytes(b, constants):
    """Inverse of async_state_bytes: canonical bytes -> the TLA+ values of AsyncIsr.tla:30-34's variables."""
    idx = replica_order(constants)
    name = {i: r for r, i in idx.items()}
    N, E = len(idx), constants["MaxVersion"]
    rb = ((1 << N) + 7) // 8
    a_req = 6 + N
    a_upd = a_req + (E + 1) * rb
    assert len(b) == a_upd + E + 1

    def unmask(m):
        return frozenset(name[i] for i in range(N) if m >> i & 1)

    controller = Fn({"isr": unmask(b[0]), "version": b[1]})
    leader = Fn({"isr": unmask(b[2]), "version": b[3], "pendingIsr": unmask(b[4]), "pendingVersion": b[5] - 1,
                 "offsets": Fn({name[i]: b[6 + i] for i in range(N)})})
    requests = frozenset(Fn({"isr": unmask(m), "version": v}) for v in range(E + 1) for m in range(1 << N)
                         if b[a_req + v * rb + (m >> 3)] >> (m & 7) & 1)
    updates = frozenset(Fn({"isr": unmask(b[a_upd + v - 1]), "version": v}) for v in range(1, b[1] + 1))
    return dict(controllerState=controller, leaderState=leader, requests=requests, updates=updates)
