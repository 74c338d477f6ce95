"""Oracle-A: structural, line-by-line Python restatement of the Kafka replication TLA+ specs.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` may be imported by the product
package; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may use it, and there only as the checker.

PARITY UNPINNED: the engine this path replaces is TLC (``tla2tools.jar``), which is not in
/root/reference, has no version pinned by the reference, and cannot run here (no JVM).
The reference holds no state counts, golden vectors or tests.  What pins this oracle is
(1) the spec text it restates operator by operator (cited ``File.tla:line`` below),
(2) closed-form known answers derivable from the text (tests/test_oracle_known_answers.py),
(3) agreement with the independently written C oracle (oracle/kmc_oracle.c).

States are plain Python values (tuples / frozensets), *not* bit-packed:

  replicaLog   : tuple over replicas of (endOffset, records)   records: tuple of len LogSize,
                 each NIL (-1) or (id, epoch)                     FiniteReplicatedLog.tla:41-44
  replicaState : tuple over replicas of (hw, leaderEpoch, leader, isr)  KafkaReplication.tla:96-99
  nextLeaderEpoch, nextRecordId : ints                           KafkaReplication.tla:77-78
  leaderAndIsrRequests : frozenset of (leaderEpoch, leader, isr) KafkaReplication.tla:65
  quorumState  : (leaderEpoch, leader, isr)                      KafkaReplication.tla:87-89

Replicas are the integers 0..N-1 (standing for model values b1..bN); ``NONE`` is the string
"NONE" exactly as KafkaReplication.tla:38; ``NIL`` is -1 exactly as KafkaReplication.tla:39.

"generated" counts one successor per satisfying binding of the existentials / disjuncts that
contain primed variables, which is how TLC's interpreter enumerates them [TLC-recall]; the
initial state counts as generated too.
"""
from __future__ import annotations

from collections import namedtuple
from itertools import combinations

NONE = "NONE"   # KafkaReplication.tla:38
NIL = -1        # KafkaReplication.tla:39

State = namedtuple(
    "State",
    "replicaLog replicaState nextLeaderEpoch nextRecordId leaderAndIsrRequests quorumState",
)  # vars, KafkaReplication.tla:75

MODELS = ("KafkaTruncateToHighWatermark", "Kip101", "Kip279", "Kip320", "Kip320FirstTry")
INVARIANTS = ("TypeOk", "WeakIsr", "StrongIsr", "LeaderInIsr")


class Params:
    """CONSTANTS Replicas, LogSize, MaxRecords, MaxLeaderEpoch (KafkaReplication.tla:32-36)."""

    def __init__(self, N, L, R, E):
        assert N >= 1 and L >= 1 and R >= 1 and E >= 0  # ASSUME MaxLeaderEpoch \in Nat (:43)
        self.N, self.L, self.R, self.E = N, L, R, E
        self.Replicas = tuple(range(N))
        self.Offsets = tuple(range(L))                      # FiniteReplicatedLog.tla:37
        self.RecordIdSet = tuple(range(0, R))               # RecordSeq!IdSet = 0..MaxRecords-1 (:78, IdSequence.tla:28)
        self.EpochIdSet = tuple(range(0, E + 1))            # LeaderEpochSeq!IdSet = 0..MaxLeaderEpoch (:77)

    def __repr__(self):
        return f"Params(N={self.N}, L={self.L}, R={self.R}, E={self.E})"


# ----------------------------------------------------------------------------------------
# Util.tla
# ----------------------------------------------------------------------------------------
def Max(s):  # Util.tla:22  (CHOOSE on an empty set is an error in TLC; callers guard)
    assert len(s) > 0
    return max(s)


def Min(s):  # Util.tla:23
    assert len(s) > 0
    return min(s)


# ----------------------------------------------------------------------------------------
# FiniteReplicatedLog.tla, instantiated as ReplicaLog with logs <- replicaLog
# (KafkaReplication.tla:84).  Each helper takes the tuple of logs.
# ----------------------------------------------------------------------------------------
def IsEmpty(logs, r):  # FiniteReplicatedLog.tla:46
    return logs[r][0] == 0


def IsFull(p, logs, r):  # :48
    return logs[r][0] == p.L


def HasEntry(logs, r, record, offset):  # :50-52 — conjunct order protects records[offset]
    end, recs = logs[r]
    return offset < end and recs[offset] == record


def IsLatestEntry(logs, r, record, offset):  # :54-57
    end, recs = logs[r]
    return (not IsEmpty(logs, r)) and offset == end - 1 and record == recs[offset]


def GetLatestRecord(logs, r):  # :59-62
    end, recs = logs[r]
    return NIL if IsEmpty(logs, r) else recs[end - 1]


def GetEndOffset(logs, r):  # :66
    return logs[r][0]


def GetRecordAtOffset(logs, r, offset):  # :70
    return logs[r][1][offset]


def GetAllEntries(logs, r):  # :72-76  set of (offset, record)
    end, recs = logs[r]
    if end == 0:
        return frozenset()
    return frozenset((o, recs[o]) for o in range(0, end))


def HasOffset(logs, r, offset):  # :78
    return offset < logs[r][0]


def LogTypeOk(p, logs, log_records):  # :80-95 (GetWrittenOffsets / GetUnwrittenOffsets / ReplicaLogTypeOk / TypeOk)
    for r in range(len(logs)):
        end, recs = logs[r]
        if not (0 <= end <= p.L):                       # log \in LogType: endOffset
            return False
        if len(recs) != p.L:
            return False
        for o in range(p.L):                            # records \in [Offsets -> LogRecords \union {Nil}]
            if not (recs[o] == NIL or recs[o] in log_records):
                return False
        for o in range(0, end):                         # written offsets hold LogRecords
            if recs[o] not in log_records:
                return False
        for o in range(end, p.L):                       # unwritten offsets hold Nil
            if recs[o] != NIL:
                return False
    return True


def Append(p, logs, r, record, offset):  # :99-103 -> new logs or None when disabled
    end, recs = logs[r]
    if IsFull(p, logs, r) or offset != end:
        return None
    nrecs = recs[:offset] + (record,) + recs[offset + 1:]
    return logs[:r] + ((end + 1, nrecs),) + logs[r + 1:]


def TruncateTo(p, logs, r, newEnd):  # :105-109 — disabled (not clamped) when newEnd > end
    end, recs = logs[r]
    if not (newEnd <= end):
        return None
    nrecs = tuple(recs[o] if o < newEnd else NIL for o in range(p.L))
    return logs[:r] + ((newEnd, nrecs),) + logs[r + 1:]


def ReplicateTo(p, logs, log_records, frm, to):  # :111-113 -> list of new logs (one per binding)
    out = []
    for offset in p.Offsets:
        for record in log_records:
            if HasEntry(logs, frm, record, offset):
                nl = Append(p, logs, to, record, offset)
                if nl is not None:
                    out.append(nl)
    return out


# ----------------------------------------------------------------------------------------
# KafkaReplication.tla
# ----------------------------------------------------------------------------------------
class Kafka:
    """All operators of the KafkaReplication family over one Params."""

    def __init__(self, p: Params, model: str):
        assert model in MODELS
        self.p, self.model = p, model
        # LogRecords == [id : RecordSeq!IdSet, epoch : LeaderEpochSeq!IdSet]   :82
        self.LogRecords = frozenset((i, e) for i in p.RecordIdSet for e in p.EpochIdSet)
        self.LogRecordsList = sorted(self.LogRecords)
        self.ReplicaOpt = frozenset(p.Replicas) | {NONE}             # :85
        self.LeaderEpochOpt = frozenset(p.EpochIdSet) | {NIL}        # :86
        self.SubsetReplicas = frozenset(
            frozenset(c) for k in range(p.N + 1) for c in combinations(p.Replicas, k)
        )
        self.next_actions = {
            # KafkaTruncateToHighWatermark.tla:33-42
            "KafkaTruncateToHighWatermark": (
                self.ControllerElectLeader, self.ControllerShrinkIsr, self.BecomeLeader,
                self.LeaderExpandIsr, self.LeaderShrinkIsr, self.LeaderWrite,
                self.LeaderIncHighWatermark, self.BecomeFollowerTruncateToHighWatermark,
                self.FollowerReplicate),
            # Kip101.tla:49-58
            "Kip101": (
                self.ControllerElectLeader, self.ControllerShrinkIsr, self.BecomeLeader,
                self.LeaderExpandIsr, self.LeaderShrinkIsr, self.LeaderWrite,
                self.LeaderIncHighWatermark, self.BecomeFollowerTruncateKip101,
                self.FollowerReplicate),
            # Kip279.tla:53-62
            "Kip279": (
                self.ControllerElectLeader, self.ControllerShrinkIsr, self.BecomeLeader,
                self.LeaderExpandIsr, self.LeaderShrinkIsr, self.LeaderWrite,
                self.LeaderIncHighWatermark, self.BecomeFollowerTruncateKip279,
                self.FollowerReplicate),
            # Kip320.tla:150-159
            "Kip320": (
                self.ControllerElectLeader, self.ControllerShrinkIsr, self.BecomeLeader,
                self.FencedLeaderExpandIsr, self.FencedLeaderShrinkIsr, self.LeaderWrite,
                self.FencedLeaderIncHighWatermark, self.FencedBecomeFollowerAndTruncate,
                self.FencedFollowerFetch),
            # Kip320FirstTry.tla:159-169
            "Kip320FirstTry": (
                self.ControllerElectLeader, self.ControllerShrinkIsr, self.BecomeLeader,
                self.LeaderExpandIsrBetterFencing, self.LeaderShrinkIsrBetterFencing,
                self.LeaderWrite, self.ImprovedLeaderIncHighWatermark, self.BecomeFollower,
                self.FollowerFetch, self.FollowerTruncate),
        }[model]
        self.action_names = tuple(a.__name__ for a in self.next_actions)

    # -- type sets ------------------------------------------------------------------
    def in_QuorumState(self, q):  # :87-89
        ep, ldr, isr = q
        return ep in self.LeaderEpochOpt and ldr in self.ReplicaOpt and isr in self.SubsetReplicas

    def in_ReplicaState(self, rs):  # :96-99
        hw, ep, ldr, isr = rs
        return (0 <= hw <= self.p.L and ep in self.LeaderEpochOpt
                and ldr in self.ReplicaOpt and isr in self.SubsetReplicas)

    def TypeOk(self, s: State):  # :101-107
        p = self.p
        return (0 <= s.nextLeaderEpoch <= p.E + 1                      # LeaderEpochSeq!TypeOk, IdSequence.tla:43
                and 0 <= s.nextRecordId <= (p.R - 1) + 1               # RecordSeq!TypeOk
                and LogTypeOk(p, s.replicaLog, self.LogRecords)        # ReplicaLog!TypeOk
                and len(s.replicaState) == p.N
                and all(self.in_ReplicaState(rs) for rs in s.replicaState)
                and self.in_QuorumState(s.quorumState)
                and all(self.in_QuorumState(q) for q in s.leaderAndIsrRequests))

    def Init(self):  # :109-120
        p = self.p
        empty_log = (0, tuple(NIL for _ in p.Offsets))                 # FiniteReplicatedLog.tla:43-44,97
        return State(
            replicaLog=tuple(empty_log for _ in p.Replicas),
            replicaState=tuple((0, NIL, NONE, frozenset()) for _ in p.Replicas),
            nextLeaderEpoch=0,                                         # IdSequence.tla:37
            nextRecordId=0,
            leaderAndIsrRequests=frozenset(),
            quorumState=(NIL, NONE, frozenset(p.Replicas)),
        )

    # -- predicates -------------------------------------------------------------------
    @staticmethod
    def ReplicaPresumesLeadership(s, r):  # :126
        return s.replicaState[r][2] == r

    @staticmethod
    def ReplicaIsFollowing(s, follower, leader):  # :127
        return s.replicaState[follower][2] == leader

    def IsTrueLeader(self, s, leader):  # :128-131
        return (s.quorumState[1] == leader
                and self.ReplicaPresumesLeadership(s, leader)
                and s.replicaState[leader][1] == s.quorumState[0])

    # -- controller -------------------------------------------------------------------
    def ControllerUpdateIsr(self, s, newLeader, newIsr):  # :138-145 -> list of partial updates
        out = []
        for newLeaderEpoch in self.p.EpochIdSet:
            # LeaderEpochSeq!NextId(newLeaderEpoch): IdSequence.tla:30-33
            if newLeaderEpoch <= self.p.E and newLeaderEpoch == s.nextLeaderEpoch:
                newControllerState = (newLeaderEpoch, newLeader, newIsr)
                out.append(dict(
                    nextLeaderEpoch=s.nextLeaderEpoch + 1,
                    quorumState=newControllerState,
                    leaderAndIsrRequests=s.leaderAndIsrRequests | {newControllerState},
                ))
        return out

    def ControllerShrinkIsr(self, s):  # :158-168
        out = []
        q_ep, q_ldr, q_isr = s.quorumState
        for replica in self.p.Replicas:
            if q_ldr == replica and q_isr == frozenset({replica}):
                out += self.ControllerUpdateIsr(s, NONE, q_isr)
            if q_ldr == replica and q_isr != frozenset({replica}):
                out += self.ControllerUpdateIsr(s, NONE, q_isr - {replica})
            if q_ldr != replica and replica in q_isr:
                out += self.ControllerUpdateIsr(s, q_ldr, q_isr - {replica})
        return [s._replace(**u) for u in out]  # UNCHANGED <<nextRecordId, replicaLog, replicaState>>

    def ControllerElectLeader(self, s):  # :176-179
        out = []
        q_ep, q_ldr, q_isr = s.quorumState
        for newLeader in sorted(q_isr):
            if q_ldr != newLeader:
                out += self.ControllerUpdateIsr(s, newLeader, q_isr)
        return [s._replace(**u) for u in out]

    # -- replica actions --------------------------------------------------------------
    def BecomeLeader(self, s):  # :186-195
        out = []
        for req in sorted(s.leaderAndIsrRequests, key=_req_key):
            r_ep, leader, r_isr = req
            if leader != NONE and r_ep > s.replicaState[leader][1]:
                old = s.replicaState[leader]
                new = (old[0], r_ep, leader, r_isr)
                out.append(s._replace(replicaState=_set(s.replicaState, leader, new)))
        return out

    def LeaderWrite(self, s):  # :202-207
        p, out = self.p, []
        for replica in p.Replicas:
            for id_ in p.RecordIdSet:
                for offset in p.Offsets:
                    if not self.ReplicaPresumesLeadership(s, replica):
                        continue
                    # RecordSeq!NextId(id): IdSequence.tla:30-33 with MaxId <- MaxRecords-1
                    if not (id_ <= p.R - 1 and id_ == s.nextRecordId):
                        continue
                    record = (id_, s.replicaState[replica][1])
                    nl = Append(p, s.replicaLog, replica, record, offset)
                    if nl is None:
                        continue
                    out.append(s._replace(replicaLog=nl, nextRecordId=s.nextRecordId + 1))
        return out

    def QuorumUpdateLeaderAndIsr(self, s, leader, newIsr):  # :213-217 -> state or None
        if not (self.IsTrueLeader(s, leader) and s.quorumState[1] == leader):
            return None
        q = s.quorumState
        rs = s.replicaState[leader]
        return s._replace(
            quorumState=(q[0], q[1], newIsr),
            replicaState=_set(s.replicaState, leader, (rs[0], rs[1], rs[2], newIsr)),
        )

    def IsFollowerCaughtUp(self, s, leader, follower, endOffset):  # :219-225
        if not self.ReplicaIsFollowing(s, follower, leader):
            return False
        if endOffset == 0:
            return True
        if endOffset > 0:
            offset = endOffset - 1
            return any(HasEntry(s.replicaLog, leader, record, offset)
                       and HasOffset(s.replicaLog, follower, offset)
                       for record in self.LogRecordsList)
        return False

    def LeaderShrinkIsr(self, s):  # :233-239
        out = []
        for leader in self.p.Replicas:
            isr = s.replicaState[leader][3]
            endOffset = GetEndOffset(s.replicaLog, leader)
            for replica in sorted(isr - {leader}):
                if not self.IsFollowerCaughtUp(s, leader, replica, endOffset):
                    t = self.QuorumUpdateLeaderAndIsr(s, leader, isr - {replica})
                    if t is not None:
                        out.append(t)
        return out

    def LeaderExpandIsr(self, s):  # :248-254
        out = []
        for leader in self.p.Replicas:
            isr = s.replicaState[leader][3]
            leaderHw = s.replicaState[leader][0]
            for replica in sorted(frozenset(self.p.Replicas) - isr):
                if self.IsFollowerCaughtUp(s, leader, replica, leaderHw):
                    t = self.QuorumUpdateLeaderAndIsr(s, leader, isr | {replica})
                    if t is not None:
                        out.append(t)
        return out

    def LeaderIncHighWatermark(self, s):  # :264-271
        out = []
        for offset in self.p.Offsets:
            for leader in self.p.Replicas:
                if not self.ReplicaPresumesLeadership(s, leader):
                    continue
                if offset != s.replicaState[leader][0]:
                    continue
                if all(self.ReplicaIsFollowing(s, f, leader) and HasOffset(s.replicaLog, f, offset)
                       for f in s.replicaState[leader][3]):
                    rs = s.replicaState[leader]
                    out.append(s._replace(
                        replicaState=_set(s.replicaState, leader, (rs[0] + 1, rs[1], rs[2], rs[3]))))
        return out

    def BecomeFollowerAndTruncateTo(self, s, leader, replica, truncationOffset):  # :281-294
        out = []
        for req in sorted(s.leaderAndIsrRequests, key=_req_key):
            r_ep, r_ldr, r_isr = req
            if not (leader != replica):
                continue
            if not (r_ldr == leader):
                continue
            if not (r_ep > s.replicaState[replica][1]):
                continue
            logs_options = []
            if leader == NONE:
                logs_options.append(s.replicaLog)             # UNCHANGED replicaLog
            if leader != NONE:
                nl = TruncateTo(self.p, s.replicaLog, replica, truncationOffset)
                if nl is not None:
                    logs_options.append(nl)
            for nl in logs_options:
                old = s.replicaState[replica]
                new = (Min({truncationOffset, old[0]}), r_ep, leader, r_isr)
                out.append(s._replace(replicaLog=nl,
                                      replicaState=_set(s.replicaState, replica, new)))
        return out

    def FollowerReplicate(self, s):  # :302-310
        out = []
        for follower in self.p.Replicas:
            for leader in self.p.Replicas:
                if not self.ReplicaPresumesLeadership(s, leader):
                    continue
                if not self.ReplicaIsFollowing(s, follower, leader):
                    continue
                for nl in ReplicateTo(self.p, s.replicaLog, self.LogRecordsList, leader, follower):
                    newEndOffset = GetEndOffset(s.replicaLog, follower) + 1
                    leaderHw = s.replicaState[leader][0]
                    followerHw = Min({leaderHw, newEndOffset})
                    rs = s.replicaState[follower]
                    out.append(s._replace(
                        replicaLog=nl,
                        replicaState=_set(s.replicaState, follower, (followerHw, rs[1], rs[2], rs[3]))))
        return out

    # -- invariants -------------------------------------------------------------------
    def _isr_prefix_ok(self, s, r1, members):
        hw = s.replicaState[r1][0]
        if hw == 0:
            return True
        for r2 in members:
            for offset in range(0, hw):
                if not any(HasEntry(s.replicaLog, r1, rec, offset) and HasEntry(s.replicaLog, r2, rec, offset)
                           for rec in self.LogRecordsList):
                    return False
        return True

    def WeakIsr(self, s):  # :320-326
        return all((not self.ReplicaPresumesLeadership(s, r1))
                   or self._isr_prefix_ok(s, r1, s.replicaState[r1][3])
                   for r1 in self.p.Replicas)

    def StrongIsr(self, s):  # :334-340
        return all((not self.ReplicaPresumesLeadership(s, r1))
                   or self._isr_prefix_ok(s, r1, s.quorumState[2])
                   for r1 in self.p.Replicas)

    @staticmethod
    def LeaderInIsr(s):  # :345
        return s.quorumState[1] in s.quorumState[2]

    # -- KafkaTruncateToHighWatermark.tla ---------------------------------------------
    def BecomeFollowerTruncateToHighWatermark(self, s):  # KafkaTruncateToHighWatermark.tla:29-31
        out = []
        for leader in self.p.Replicas:
            for replica in self.p.Replicas:
                replicaHw = s.replicaState[replica][0]
                out += self.BecomeFollowerAndTruncateTo(s, leader, replica, replicaHw)
        return out

    # -- Kip101.tla -------------------------------------------------------------------
    def OffsetsWithLargerEpochs(self, s, replica, epoch):  # Kip101.tla:27-29
        return frozenset(o for (o, rec) in GetAllEntries(s.replicaLog, replica) if rec[1] > epoch)

    def LookupOffsetForEpoch(self, s, leader, follower, epoch):  # Kip101.tla:31-39
        if IsEmpty(s.replicaLog, leader):
            return s.replicaState[follower][0]
        if GetLatestRecord(s.replicaLog, leader)[1] == epoch:
            return GetEndOffset(s.replicaLog, leader)
        larger = self.OffsetsWithLargerEpochs(s, leader, epoch)
        if larger == frozenset():
            return s.replicaState[follower][0]
        return Min(larger)

    def BecomeFollowerTruncateKip101(self, s):  # Kip101.tla:41-47
        out = []
        for leader in self.p.Replicas:
            for replica in self.p.Replicas:
                if IsEmpty(s.replicaLog, replica):
                    out += self.BecomeFollowerAndTruncateTo(s, leader, replica, 0)
                for record in self.LogRecordsList:
                    # IsLatestRecord: FiniteReplicatedLog.tla:64
                    if any(IsLatestEntry(s.replicaLog, replica, record, o) for o in self.p.Offsets):
                        offset = self.LookupOffsetForEpoch(s, leader, replica, record[1])
                        out += self.BecomeFollowerAndTruncateTo(s, leader, replica, offset)
        return out

    # -- Kip279.tla -------------------------------------------------------------------
    def MatchingOffsets(self, s, replica1, replica2):  # Kip279.tla:27-30
        return frozenset(o for (o, rec) in GetAllEntries(s.replicaLog, replica1)
                         if HasEntry(s.replicaLog, replica2, rec, o))

    def FirstNonMatchingOffsetFromTail(self, s, leader, follower):  # Kip279.tla:39-45
        if IsEmpty(s.replicaLog, leader):
            return 0
        matching = self.MatchingOffsets(s, follower, leader)
        if matching == frozenset():
            return 0
        return Max(matching) + 1

    def BecomeFollowerTruncateKip279(self, s):  # Kip279.tla:47-51
        out = []
        for leader in self.p.Replicas:
            for replica in self.p.Replicas:
                if IsEmpty(s.replicaLog, replica):
                    out += self.BecomeFollowerAndTruncateTo(s, leader, replica, 0)
                offset = self.FirstNonMatchingOffsetFromTail(s, leader, replica)
                out += self.BecomeFollowerAndTruncateTo(s, leader, replica, offset)
        return out

    # -- Kip320.tla -------------------------------------------------------------------
    def IsFollowingLeaderEpoch(self, s, leader, follower):  # Kip320.tla:39-42
        return (self.ReplicaPresumesLeadership(s, leader)
                and s.replicaState[follower][2] == leader
                and s.replicaState[follower][1] == s.replicaState[leader][1])

    def _replicate_and_update_hw(self, s, leader, follower):
        """Shared tail of FollowerReplicate (:305-309), FencedFollowerFetch (Kip320.tla:51-55),
        FollowerFetch (Kip320FirstTry.tla:106-110)."""
        out = []
        for nl in ReplicateTo(self.p, s.replicaLog, self.LogRecordsList, leader, follower):
            newEndOffset = GetEndOffset(s.replicaLog, follower) + 1
            leaderHw = s.replicaState[leader][0]
            followerHw = Min({leaderHw, newEndOffset})
            rs = s.replicaState[follower]
            out.append(s._replace(
                replicaLog=nl,
                replicaState=_set(s.replicaState, follower, (followerHw, rs[1], rs[2], rs[3]))))
        return out

    def FencedFollowerFetch(self, s):  # Kip320.tla:49-56
        out = []
        for follower in self.p.Replicas:
            for leader in self.p.Replicas:
                if self.IsFollowingLeaderEpoch(s, leader, follower):
                    out += self._replicate_and_update_hw(s, leader, follower)
        return out

    def FencedLeaderIncHighWatermark(self, s):  # Kip320.tla:63-70
        out = []
        for leader in self.p.Replicas:
            leaderHw = s.replicaState[leader][0]
            if not HasOffset(s.replicaLog, leader, leaderHw):
                continue
            if all(self.IsFollowingLeaderEpoch(s, leader, f) and HasOffset(s.replicaLog, f, leaderHw)
                   for f in s.replicaState[leader][3]):
                rs = s.replicaState[leader]
                out.append(s._replace(
                    replicaState=_set(s.replicaState, leader, (rs[0] + 1, rs[1], rs[2], rs[3]))))
        return out

    def FencedLeaderShrinkIsr(self, s):  # Kip320.tla:78-85
        out = []
        for leader in self.p.Replicas:
            isr = s.replicaState[leader][3]
            leaderEndOffset = GetEndOffset(s.replicaLog, leader)
            for follower in sorted(isr - {leader}):
                # :82-83 is a DISJUNCTION in front of the primed conjunct :84.  TLC's next-state enumeration walks
                # every disjunct that holds and continues with the rest of the conjunction from each of them
                # (Tool.getNextStates, OPCODE_lor [TLC-recall]), so when both hold the same successor is generated
                # twice.  (Found by Oracle-R, oracle/tlar, which executes the module's text that way.)
                for holds in ((not self.IsFollowingLeaderEpoch(s, leader, follower)),
                              GetEndOffset(s.replicaLog, follower) < leaderEndOffset):
                    if holds:
                        t = self.QuorumUpdateLeaderAndIsr(s, leader, isr - {follower})
                        if t is not None:
                            out.append(t)
        return out

    def HasHighWatermarkReachedCurrentEpoch(self, s, leader):  # Kip320.tla:87-92 / Kip320FirstTry.tla:122-127
        hw = s.replicaState[leader][0]
        if hw == GetEndOffset(s.replicaLog, leader):
            return True
        return any(HasEntry(s.replicaLog, leader, record, hw)
                   and record[1] == s.replicaState[leader][1]
                   for record in self.LogRecordsList)

    def HasFollowerReachedHighWatermark(self, s, leader, follower):  # Kip320.tla:94-98
        hw = s.replicaState[leader][0]
        return hw == 0 or (hw > 0 and HasOffset(s.replicaLog, follower, hw - 1))

    def FencedLeaderExpandIsr(self, s):  # Kip320.tla:110-117
        out = []
        for leader in self.p.Replicas:
            isr = s.replicaState[leader][3]
            for follower in sorted(frozenset(self.p.Replicas) - isr):
                if (self.IsFollowingLeaderEpoch(s, leader, follower)
                        and self.HasFollowerReachedHighWatermark(s, leader, follower)
                        and self.HasHighWatermarkReachedCurrentEpoch(s, leader)):
                    t = self.QuorumUpdateLeaderAndIsr(s, leader, isr | {follower})
                    if t is not None:
                        out.append(t)
        return out

    def FencedBecomeFollowerAndTruncate(self, s):  # Kip320.tla:134-148
        out = []
        for leader in self.p.Replicas:          # leader \in Replicas, so `leader = None` (:138) is dead
            for replica in self.p.Replicas:
                for req in sorted(s.leaderAndIsrRequests, key=_req_key):
                    r_ep, r_ldr, r_isr = req
                    if not (leader != replica and r_ldr == leader and r_ep > s.replicaState[replica][1]):
                        continue
                    if leader == NONE:                                       # :138-140 (unreachable)
                        new = (s.replicaState[replica][0], r_ep, r_ldr, r_isr)
                        out.append(s._replace(replicaState=_set(s.replicaState, replica, new)))
                    if leader != NONE:                                       # :141-147
                        if not self.ReplicaPresumesLeadership(s, leader):
                            continue
                        if not (s.replicaState[leader][1] == r_ep):
                            continue
                        truncationOffset = self.FirstNonMatchingOffsetFromTail(s, leader, replica)
                        newHighWatermark = Min({truncationOffset, s.replicaState[replica][0]})
                        nl = TruncateTo(self.p, s.replicaLog, replica, truncationOffset)
                        if nl is None:
                            continue
                        new = (newHighWatermark, r_ep, r_ldr, r_isr)          # BecomeFollower :119-124
                        out.append(s._replace(replicaLog=nl,
                                              replicaState=_set(s.replicaState, replica, new)))
        return out

    # -- Kip320FirstTry.tla -----------------------------------------------------------
    def IsFollowerCaughtUpToLeaderEpoch(self, s, leader, follower, endOffset):  # Kip320FirstTry.tla:49-57
        if not self.ReplicaPresumesLeadership(s, leader):
            return False
        if not self.ReplicaIsFollowing(s, follower, leader):
            return False
        if endOffset == 0:
            return True
        if endOffset > 0:
            offset = endOffset - 1
            for record in self.LogRecordsList:
                if (HasEntry(s.replicaLog, leader, record, offset)
                        and HasOffset(s.replicaLog, follower, offset)
                        and GetRecordAtOffset(s.replicaLog, follower, offset)[1] == record[1]):
                    return True
        return False

    def FollowerNeedsTruncation(self, s, follower, leader):  # Kip320FirstTry.tla:64-69
        if GetEndOffset(s.replicaLog, follower) > GetEndOffset(s.replicaLog, leader):
            return True
        for record in self.LogRecordsList:
            for offset in self.p.Offsets:
                if (IsLatestEntry(s.replicaLog, follower, record, offset)
                        and HasOffset(s.replicaLog, leader, offset)
                        and GetRecordAtOffset(s.replicaLog, leader, offset)[1] != record[1]):
                    return True
        return False

    def FollowerTruncate(self, s):  # Kip320FirstTry.tla:75-82
        out = []
        for leader in self.p.Replicas:
            for follower in self.p.Replicas:
                if not (self.ReplicaPresumesLeadership(s, leader)
                        and self.ReplicaIsFollowing(s, follower, leader)
                        and self.FollowerNeedsTruncation(s, follower, leader)):
                    continue
                truncationOffset = self.FirstNonMatchingOffsetFromTail(s, leader, follower)
                nl = TruncateTo(self.p, s.replicaLog, follower, truncationOffset)
                if nl is None:
                    continue
                rs = s.replicaState[follower]
                out.append(s._replace(
                    replicaLog=nl,
                    replicaState=_set(s.replicaState, follower,
                                      (Min({truncationOffset, rs[0]}), rs[1], rs[2], rs[3]))))
        return out

    def ImprovedLeaderIncHighWatermark(self, s):  # Kip320FirstTry.tla:90-97
        out = []
        for leader in self.p.Replicas:
            if not self.ReplicaPresumesLeadership(s, leader):
                continue
            leaderHw = s.replicaState[leader][0]
            for record in self.LogRecordsList:
                if not HasEntry(s.replicaLog, leader, record, leaderHw):
                    continue
                if all(self.IsFollowerCaughtUpToLeaderEpoch(s, leader, f, leaderHw + 1)
                       for f in s.replicaState[leader][3]):
                    rs = s.replicaState[leader]
                    out.append(s._replace(
                        replicaState=_set(s.replicaState, leader, (rs[0] + 1, rs[1], rs[2], rs[3]))))
        return out

    def FollowerFetch(self, s):  # Kip320FirstTry.tla:103-111
        out = []
        for follower in self.p.Replicas:
            for leader in self.p.Replicas:
                followerEndOffset = GetEndOffset(s.replicaLog, follower)
                if self.IsFollowerCaughtUpToLeaderEpoch(s, leader, follower, followerEndOffset):
                    out += self._replicate_and_update_hw(s, leader, follower)
        return out

    def LeaderShrinkIsrBetterFencing(self, s):  # Kip320FirstTry.tla:114-120
        out = []
        for leader in self.p.Replicas:
            isr = s.replicaState[leader][3]
            endOffset = GetEndOffset(s.replicaLog, leader)
            for replica in sorted(isr - {leader}):
                if not self.IsFollowerCaughtUpToLeaderEpoch(s, leader, replica, endOffset):
                    t = self.QuorumUpdateLeaderAndIsr(s, leader, isr - {replica})
                    if t is not None:
                        out.append(t)
        return out

    def LeaderExpandIsrBetterFencing(self, s):  # Kip320FirstTry.tla:134-141
        out = []
        for leader in self.p.Replicas:
            isr = s.replicaState[leader][3]
            leaderHw = s.replicaState[leader][0]
            for replica in sorted(frozenset(self.p.Replicas) - isr):
                if (self.IsFollowerCaughtUpToLeaderEpoch(s, leader, replica, leaderHw)
                        and self.HasHighWatermarkReachedCurrentEpoch(s, leader)):
                    t = self.QuorumUpdateLeaderAndIsr(s, leader, isr | {replica})
                    if t is not None:
                        out.append(t)
        return out

    def BecomeFollower(self, s):  # Kip320FirstTry.tla:148-157
        out = []
        for leader in self.p.Replicas:
            for replica in self.p.Replicas:
                for req in sorted(s.leaderAndIsrRequests, key=_req_key):
                    r_ep, r_ldr, r_isr = req
                    if leader != replica and r_ldr == leader and r_ep > s.replicaState[replica][1]:
                        old = s.replicaState[replica]
                        out.append(s._replace(
                            replicaState=_set(s.replicaState, replica, (old[0], r_ep, leader, r_isr))))
        return out

    # -- Next -------------------------------------------------------------------------
    def Next(self, s):
        """All successors, one per satisfying binding, as (action_index, state)."""
        out = []
        for ai, act in enumerate(self.next_actions):
            for t in act(s):
                out.append((ai, t))
        return out

    def invariant(self, name):
        return getattr(self, name)


def _set(tup, i, v):
    return tup[:i] + (v,) + tup[i + 1:]


def _req_key(req):
    ep, ldr, isr = req
    return (ep, -1 if ldr == NONE else ldr, tuple(sorted(isr)))


# ----------------------------------------------------------------------------------------
# IdSequence.tla standalone and FiniteReplicatedLog.tla standalone
# ----------------------------------------------------------------------------------------
class IdSequenceModel:
    """IdSequence.tla:22-45.  State = nextId (int)."""
    action_names = ("Next",)

    def __init__(self, MaxId):
        assert MaxId >= 0  # ASSUME MaxId \in Nat (:24)
        self.MaxId = MaxId

    def Init(self):  # :37
        return 0

    def Next(self, s):  # :39 with NextId :30-33.  Only id = nextId can satisfy the conjunction.
        out = []
        id_ = s
        if 0 <= id_ <= self.MaxId:          # id \in IdSet, id <= MaxId, id = nextId
            out.append((0, s + 1))
        return out

    def TypeOk(self, s):  # :43
        return 0 <= s <= self.MaxId + 1

    def invariant(self, name):
        return getattr(self, name)


class FiniteReplicatedLogModel:
    """FiniteReplicatedLog.tla:22-122 standalone.  LogRecords = 0..K-1 stand for K model
    values, Nil = NIL.  State = tuple over replicas of (endOffset, records)."""
    action_names = ("Append", "TruncateTo", "ReplicateTo")

    def __init__(self, N, L, K):
        self.p = Params(N, L, 1, 0)
        self.K = K
        self.LogRecords = tuple(range(K))

    def Init(self):  # :97
        return tuple((0, tuple(NIL for _ in self.p.Offsets)) for _ in self.p.Replicas)

    def Next(self, s):  # :115-118
        p, out = self.p, []
        for replica in p.Replicas:
            for record in self.LogRecords:
                for offset in p.Offsets:
                    nl = Append(p, s, replica, record, offset)
                    if nl is not None:
                        out.append((0, nl))
            for offset in p.Offsets:
                nl = TruncateTo(p, s, replica, offset)
                if nl is not None:
                    out.append((1, nl))
            for other in p.Replicas:
                if other != replica:
                    for nl in ReplicateTo(p, s, self.LogRecords, replica, other):
                        out.append((2, nl))
        return out

    def TypeOk(self, s):  # :95
        return LogTypeOk(self.p, s, frozenset(self.LogRecords))

    def invariant(self, name):
        return getattr(self, name)


# ----------------------------------------------------------------------------------------
# AsyncIsr.tla (standalone, EXTENDS Integers, Util — AsyncIsr.tla:20)
# ----------------------------------------------------------------------------------------
class AsyncIsrModel:
    r"""AsyncIsr.tla:22-162.  The spec is unbounded (version: Nat, offsets: [Replicas -> Nat],
    :40-56; LeaderWrite :117-119 increments forever; MaxOffset :25 only feeds the unused
    `Offsets` :37), so it is checked under an explicit state CONSTRAINT that the reference
    does not contain (models/MCAsyncIsr.tla):
        leaderState.offsets[Leader] <= MaxOffset /\ controllerState.version <= MaxVersion
    State = (cisr, cver, lisr, lver, pisr, pver, offsets, requests, updates); sets are
    frozensets, messages are (isr, version) pairs, Leader is replica 0."""
    action_names = ("ControllerShrinkIsr", "ControllerHandleRequest", "LeaderRequestShrinkIsr",
                    "LeaderRequestExpandIsr", "LeaderWrite", "LeaderHandleUpdate", "FollowerReplicate")

    def __init__(self, N, MaxOffset, MaxVersion):
        assert MaxOffset > 0 and N >= 1  # ASSUME :27-29 (Leader \in Replicas by construction)
        self.N, self.MaxOffset, self.MaxVersion = N, MaxOffset, MaxVersion
        self.Replicas = tuple(range(N))
        self.Leader = 0

    def Init(self):  # :137-150
        R = frozenset(self.Replicas)
        return (R, 0, R, 0, frozenset(), -1, tuple(0 for _ in self.Replicas), frozenset(), frozenset())

    def constraint(self, s):  # models/MCAsyncIsr.tla — not part of the reference
        return s[6][self.Leader] <= self.MaxOffset and s[1] <= self.MaxVersion

    def HighWatermark(self, s):  # :58-60
        cisr, cver, lisr, lver, pisr, pver, offsets, requests, updates = s
        potentialIsr = lisr | pisr
        return Min({offsets[replica] for replica in potentialIsr})

    def TypeOk(self, s):  # :62-66 with LeaderState :40-46, ControllerState :48-51, Message :53-56
        cisr, cver, lisr, lver, pisr, pver, offsets, requests, updates = s
        R = set(self.Replicas)
        nat = lambda x: isinstance(x, int) and x >= 0
        msg = lambda m: m[0] <= R and nat(m[1])
        return (cisr <= R and nat(cver)
                and lisr <= R and nat(lver) and pisr <= R and nat(pver)   # pendingVersion = Nil = -1 fails here (:38,:44,:146)
                and all(nat(o) for o in offsets)
                and all(msg(m) for m in requests) and all(msg(m) for m in updates))

    def ValidHighWatermark(self, s):  # :161-162
        cisr, offsets = s[0], s[6]
        hw = self.HighWatermark(s)
        return all(offsets[replica] >= hw for replica in cisr)

    def LeaderOffsetInRange(self, s):  # models/MCAsyncIsr.tla (not in the reference): offsets[Leader] \in Offsets (:37)
        return 0 <= s[6][self.Leader] <= self.MaxOffset

    def Next(self, s):  # :152-159
        cisr, cver, lisr, lver, pisr, pver, offsets, requests, updates = s
        Leader, out = self.Leader, []
        # ControllerShrinkIsr :72-79 (ControllerWriteIsr :68-70)
        for replica in self.Replicas:
            if replica != Leader and replica in cisr:
                version = cver + 1
                isr = cisr - {replica}
                out.append((0, (isr, version, lisr, lver, pisr, pver, offsets, requests, updates | {(isr, version)})))
        # ControllerHandleRequest :81-86
        for message in sorted(requests, key=lambda m: (m[1], sorted(m[0]))):
            if message[1] == cver:
                version = cver + 1
                out.append((1, (message[0], version, lisr, lver, pisr, pver, offsets, requests,
                                updates | {(message[0], version)})))
        # LeaderRequestShrinkIsr :88-100
        for replica in sorted(lisr):
            if replica != Leader:
                isr = lisr - {replica}
                version = lver
                out.append((2, (cisr, cver, lisr, lver, pisr | isr, version, offsets,
                                requests | {(isr, version)}, updates)))
        # LeaderRequestExpandIsr :102-115
        for replica in self.Replicas:
            if replica not in lisr and offsets[replica] >= self.HighWatermark(s):
                isr = lisr | {replica}
                version = lver
                out.append((3, (cisr, cver, lisr, lver, pisr | isr, version, offsets,
                                requests | {(isr, version)}, updates)))
        # LeaderWrite :117-119 — always enabled
        out.append((4, (cisr, cver, lisr, lver, pisr, pver, _set(offsets, Leader, offsets[Leader] + 1),
                        requests, updates)))
        # LeaderHandleUpdate :121-129
        for update in sorted(updates, key=lambda m: (m[1], sorted(m[0]))):
            if update[1] > lver:
                out.append((5, (cisr, cver, update[0], update[1], frozenset(), -1, offsets, requests, updates)))
        # FollowerReplicate :131-135
        for replica in self.Replicas:
            if replica != Leader and offsets[replica] < offsets[Leader]:
                out.append((6, (cisr, cver, lisr, lver, pisr, pver, _set(offsets, replica, offsets[replica] + 1),
                                requests, updates)))
        return out

    def invariant(self, name):
        return getattr(self, name)


# ----------------------------------------------------------------------------------------
# Level-synchronous exhaustive BFS (what TLC's Worker loop does, [TLC-recall]).
# ----------------------------------------------------------------------------------------
def bfs(model, invariants=("TypeOk",), check_deadlock=False, stop_on_violation=True,
        max_states=None, keep_states=False):
    """Returns a dict:
      distinct, generated (incl. the initial state), depth (number of BFS levels, TLC's
      "depth of the complete state graph search"), levels (new states per level),
      action_generated {name: n}, verdict in {"ok","invariant","deadlock","limit"},
      violation {invariant, depth, count_at_depth, trace} or None,
      violations_per_level_at_stop {inv: count} for the stopping level.
    The search is level-synchronous: on a violation the whole level is still completed,
    so every number is deterministic.
    """
    init = model.Init()
    parent = {init: (None, None)}
    levels = [1]
    generated = 1
    action_generated = {n: 0 for n in model.action_names}
    frontier = [init]
    verdict, violation = "ok", None
    level_states = [[init]] if keep_states else None

    def check(states, depth):
        per_inv = {}
        first = {}
        for st in states:
            for name in invariants:
                if not model.invariant(name)(st):
                    per_inv[name] = per_inv.get(name, 0) + 1
                    first.setdefault(name, st)
        return per_inv, first

    per_inv, first = check(frontier, 1)
    depth = 1
    deadlocks = 0
    if per_inv and stop_on_violation:
        name = next(n for n in invariants if n in per_inv)
        violation = dict(invariant=name, depth=1, count_at_depth=per_inv[name],
                         per_invariant=per_inv, trace=[(None, init)])
        verdict = "invariant"
        frontier = []
    constraint = getattr(model, "constraint", None)
    outside_total = {}
    while frontier:
        nxt = []
        outside, outside_first = {}, {}
        for s in frontier:
            succ = model.Next(s)
            if not succ:
                deadlocks += 1
            for ai, t in succ:
                generated += 1
                action_generated[model.action_names[ai]] += 1
                if constraint is not None and not constraint(t):
                    # TLC CONSTRAINT [TLC-recall: ModelChecker.doNext]: a successor outside the
                    # model is neither fingerprinted nor queued, but its invariants ARE checked
                    # (every time it is generated, since it is never "seen")
                    for name in invariants:
                        if not model.invariant(name)(t):
                            outside[name] = outside.get(name, 0) + 1
                            outside_first.setdefault(name, (s, ai, t))
                    continue
                if t not in parent:
                    parent[t] = (s, ai)
                    nxt.append(t)
        if check_deadlock and deadlocks and verdict == "ok":
            verdict = "deadlock"
            break
        for name, cnt in outside.items():
            outside_total[name] = outside_total.get(name, 0) + cnt
        if outside and violation is None:
            # violating successors outside the constraint sit at depth+1 and are seen while the
            # current level is expanded, i.e. before the in-model states of depth+1 are checked
            name = next(n for n in invariants if n in outside)
            par, ai, st = outside_first[name]
            trace = [(model.action_names[ai], st)]
            cur = par
            while cur is not None:
                pp, pa = parent[cur]
                trace.append((None if pa is None else model.action_names[pa], cur))
                cur = pp
            trace.reverse()
            violation = dict(invariant=name, depth=depth + 1, count_at_depth=outside[name],
                             per_invariant=dict(outside), trace=trace, outside_constraint=True)
            if stop_on_violation:
                verdict = "invariant"
                break
        if not nxt:
            break
        depth += 1
        levels.append(len(nxt))
        if keep_states:
            level_states.append(nxt)
        per_inv, first = check(nxt, depth)
        if per_inv and violation is None:
            name = next(n for n in invariants if n in per_inv)
            st = first[name]
            trace = []
            cur = st
            while cur is not None:
                par, ai = parent[cur]
                trace.append((None if ai is None else model.action_names[ai], cur))
                cur = par
            trace.reverse()
            violation = dict(invariant=name, depth=depth, count_at_depth=per_inv[name],
                             per_invariant=per_inv, trace=trace)
            if stop_on_violation:
                verdict = "invariant"
                break
        if max_states is not None and len(parent) > max_states:
            verdict = "limit"
            break
        frontier = nxt
    if violation is not None and verdict == "ok":
        verdict = "invariant"
    res = dict(distinct=len(parent), generated=generated, depth=depth, levels=levels,
               action_generated=action_generated, verdict=verdict, violation=violation,
               deadlock_states=deadlocks, outside_violations=outside_total)
    if keep_states:
        res["level_states"] = level_states
    return res


def make_model(model, **c):
    if model == "IdSequence":
        return IdSequenceModel(c["MaxId"])
    if model == "FiniteReplicatedLog":
        return FiniteReplicatedLogModel(c["N"], c["L"], c["K"])
    if model == "AsyncIsr":  # L = MaxOffset, E = MaxVersion (the constraint's two bounds)
        return AsyncIsrModel(c["N"], c["L"], c["E"])
    return Kafka(Params(c["N"], c["L"], c["R"], c["E"]), model)


if __name__ == "__main__":
    import argparse, json, time
    ap = argparse.ArgumentParser()
    ap.add_argument("model")
    ap.add_argument("--N", type=int, default=3)
    ap.add_argument("--L", type=int, default=2)
    ap.add_argument("--R", type=int, default=2)
    ap.add_argument("--E", type=int, default=1)
    ap.add_argument("--K", type=int, default=2)
    ap.add_argument("--MaxId", type=int, default=10)
    ap.add_argument("--inv", default="TypeOk")
    ap.add_argument("--continue", dest="cont", action="store_true")
    ap.add_argument("--max-states", type=int, default=None)
    a = ap.parse_args()
    m = make_model(a.model, N=a.N, L=a.L, R=a.R, E=a.E, K=a.K, MaxId=a.MaxId)
    t0 = time.time()
    r = bfs(m, invariants=tuple(x for x in a.inv.split(",") if x), stop_on_violation=not a.cont,
            max_states=a.max_states)
    r["seconds"] = round(time.time() - t0, 3)
    if r["violation"]:
        r["violation"] = {k: v for k, v in r["violation"].items() if k != "trace"} | {
            "trace_len": len(r["violation"]["trace"])}
    print(json.dumps(r))
