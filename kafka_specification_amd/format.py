"""Pretty-printing of canonical-byte states as TLA+ values, the way TLC prints a trace state
(`/\\ var = value`) [TLC-recall].  The byte layout is documented in DESIGN.md §3."""
from __future__ import annotations


def _set(mask, names):
    return "{" + ", ".join(names[r] for r in range(len(names)) if mask >> r & 1) + "}"


def format_state(cfg, b: bytes) -> str:
    m = cfg.model
    if m == "IdSequence":
        return f"nextId = {int.from_bytes(b[:8], 'little')}"
    if m == "FiniteReplicatedLog":
        N, L = cfg.n_replicas, cfg.log_size
        reps = [f"r{r + 1}" for r in range(N)]
        parts = []
        for r in range(N):
            blk = b[r * (1 + L):(r + 1) * (1 + L)]
            recs = ", ".join("Nil" if c == 0 else f"x{c}" for c in blk[1:])
            parts.append(f"{reps[r]} :> [endOffset |-> {blk[0]}, records |-> <<{recs}>>]")
        return "logs = (" + " @@ ".join(parts) + ")"
    if m == "AsyncIsr":
        return _format_async_isr(cfg, b)
    N, L, E = cfg.n_replicas, cfg.log_size, cfg.max_leader_epoch
    reps = [f"b{r + 1}" for r in range(N)]
    rs = 5 + L

    def ldr(x):
        return '"NONE"' if x == 0 else reps[x - 1]

    logs, states = [], []
    for r in range(N):
        blk = b[r * rs:(r + 1) * rs]
        recs = ", ".join("-1" if c == 0 else f"[id |-> {(c - 1) // (E + 1)}, epoch |-> {(c - 1) % (E + 1)}]"
                         for c in blk[5:])
        logs.append(f"{reps[r]} :> [endOffset |-> {blk[0]}, records |-> <<{recs}>>]")
        states.append(f"{reps[r]} :> [hw |-> {blk[1]}, leaderEpoch |-> {blk[2] - 1}, leader |-> {ldr(blk[3])}, "
                      f"isr |-> {_set(blk[4], reps)}]")
    g = b[N * rs:]
    reqs = [f"[leaderEpoch |-> {e}, leader |-> {ldr(g[5 + 2 * e])}, isr |-> {_set(g[6 + 2 * e], reps)}]"
            for e in range(g[1])]
    return "\n".join([
        "/\\ replicaLog = (" + " @@ ".join(logs) + ")",
        "/\\ replicaState = (" + " @@ ".join(states) + ")",
        f"/\\ nextRecordId = {g[0]}",
        f"/\\ nextLeaderEpoch = {g[1]}",
        "/\\ leaderAndIsrRequests = {" + ", ".join(reqs) + "}",
        f"/\\ quorumState = [leaderEpoch |-> {g[2] - 1}, leader |-> {ldr(g[3])}, isr |-> {_set(g[4], reps)}]",
    ])


def _format_async_isr(cfg, b: bytes) -> str:
    """AsyncIsr.tla:31-35; canonical bytes as in include/kmc.h.  Replica 0 is `Leader`."""
    N, V = cfg.n_replicas, cfg.max_leader_epoch
    reps = [f"r{r + 1}" for r in range(N)]
    rb = ((1 << N) + 7) // 8
    q = b[6 + N:6 + N + (V + 1) * rb]
    u = b[6 + N + (V + 1) * rb:]
    reqs = [f"[isr |-> {_set(mask, reps)}, version |-> {v}]"
            for v in range(V + 1) for mask in range(1 << N) if q[v * rb + (mask >> 3)] >> (mask & 7) & 1]
    upds = [f"[isr |-> {_set(u[v - 1], reps)}, version |-> {v}]" for v in range(1, min(b[1], V + 1) + 1)]
    offs = " @@ ".join(f"{reps[r]} :> {b[6 + r]}" for r in range(N))
    return "\n".join([
        f"/\\ controllerState = [isr |-> {_set(b[0], reps)}, version |-> {b[1]}]",
        f"/\\ leaderState = [isr |-> {_set(b[2], reps)}, version |-> {b[3]}, pendingIsr |-> {_set(b[4], reps)}, "
        f"pendingVersion |-> {b[5] - 1}, offsets |-> ({offs})]",
        "/\\ requests = {" + ", ".join(reqs) + "}",
        "/\\ updates = {" + ", ".join(upds) + "}",
    ])
