/* Oracle-B: plain-C CPU restatement of the Kafka replication TLA+ specs + an exact
 * (full-state) level-synchronous BFS.  TEST INFRASTRUCTURE ONLY: it is the checker for the
 * HIP engine and the timed "cpu_baseline" of bench.py; the product never links or calls it.
 *
 * PARITY UNPINNED: the reference (/root/reference) is ten .tla files; the engine on this
 * path is TLC (tla2tools.jar), absent, unpinned and unrunnable here (no JVM).  This oracle
 * is pinned by (1) the spec text it restates (File.tla:line cited per function), (2) the
 * closed-form known answers in tests/, (3) agreement with oracle/kafka_oracle.py, which was
 * written separately and enumerates the TLA+ existentials literally.
 *
 * State representation (deliberately NOT the GPU's bit packing): one byte per field.
 *   per replica r, block of 5+L bytes at r*(5+L):
 *     [0] endOffset  [1] hw  [2] leaderEpoch+1 (Nil=-1 -> 0)  [3] leader+1 (None -> 0)
 *     [4] isr bitmask  [5+o] record at offset o: 0 = Nil, else 1 + id*(E+1) + epoch
 *   globals at N*(5+L):
 *     [0] nextRecordId [1] nextLeaderEpoch [2] quorum.leaderEpoch+1 [3] quorum.leader+1
 *     [4] quorum.isr   [5+2e],[6+2e] request with leaderEpoch e: leader+1, isr  (e<nextLeaderEpoch;
 *     zero otherwise).  leaderAndIsrRequests is a set, but ControllerUpdateIsr
 *     (KafkaReplication.tla:138-145) is its only writer and always adds the record whose
 *     leaderEpoch is the old nextLeaderEpoch, so the set is in bijection with this array.
 * This byte string is the "canonical serialization" the tests compare state sets on.
 *
 * Build: see oracle/Makefile (gcc -O2 -pthread; -DKMO_MAIN adds the CLI).
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <sys/mman.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define KMO_MAX_LEVELS 512
#define KMO_MAX_ACTIONS 16
#define KMO_MAXN 8
#define KMO_MAXSB 160

enum { M_IDSEQ = 0, M_FRL = 1, M_TRUNC_HW = 2, M_KIP101 = 3, M_KIP279 = 4, M_KIP320 = 5, M_KIP320_FIRST = 6, M_ASYNC_ISR = 7 };
enum { INV_TYPEOK = 0, INV_WEAKISR = 1, INV_STRONGISR = 2, INV_LEADERINISR = 3 };
enum { INV_VALIDHW = 1 }; /* AsyncIsr: invariant index 1 is ValidHighWatermark (AsyncIsr.tla:161) */
enum { V_OK = 0, V_INVARIANT = 1, V_DEADLOCK = 2, V_LIMIT = 3, V_ERROR = 4 };

typedef struct {
    int32_t model, N, L, R, E, K;
    int64_t MaxId;
    uint32_t inv_mask;
    int32_t check_deadlock, stop_on_violation, threads;
    uint64_t max_states;
} kmo_config;

typedef struct {
    uint64_t distinct, generated, depth;
    int32_t verdict, viol_inv;
    uint64_t viol_depth, viol_state_idx;
    uint64_t viol_count[4];
    uint64_t deadlock_states;
    uint64_t action_generated[KMO_MAX_ACTIONS];
    uint64_t nlevels;
    uint64_t levels[KMO_MAX_LEVELS];
    double seconds;
    /* a violating successor OUTSIDE the state constraint (AsyncIsr only): it is in no level, so
     * it is handed over here together with its parent (viol_state_idx) and the action */
    int32_t viol_outside, viol_action;
    uint8_t viol_state[KMO_MAXSB];
} kmo_result;

typedef struct {
    int model, N, L, R, E, K;
    int64_t MaxId;
    int rstride, goff, sb, nact;
    int EP1; /* E+1 */
    int a_req, a_rb, a_upd; /* AsyncIsr: byte offset of the requests bitset, bytes per version, offset of the updates array */
} P;

typedef void (*emit_fn)(void *ctx, int action, const uint8_t *succ);

/* ---- field access ------------------------------------------------------------------ */
#define REP(s, r) ((s) + (r) * p->rstride)
#define END(s, r) (REP(s, r)[0])
#define HW(s, r) (REP(s, r)[1])
#define EP1(s, r) (REP(s, r)[2])
#define LDR1(s, r) (REP(s, r)[3])
#define ISR(s, r) (REP(s, r)[4])
#define REC(s, r, o) (REP(s, r)[5 + (o)])
#define G(s) ((s) + p->goff)
#define NEXTREC(s) (G(s)[0])
#define NEXTEP(s) (G(s)[1])
#define Q_EP1(s) (G(s)[2])
#define Q_LDR1(s) (G(s)[3])
#define Q_ISR(s) (G(s)[4])
#define REQ_LDR1(s, e) (G(s)[5 + 2 * (e)])
#define REQ_ISR(s, e) (G(s)[6 + 2 * (e)])
#define BIT(r) ((uint8_t)(1u << (r)))
#define REC_EPOCH(code) (((code)-1) % p->EP1) /* record.epoch of a non-Nil record code */
static inline int imin(int a, int b) { return a < b ? a : b; }

/* ==================================================================================== */
/* IdSequence.tla standalone (:22-45): state = 8-byte little-endian nextId                */
/* ==================================================================================== */
static void idseq_expand(const P *p, const uint8_t *s, emit_fn emit, void *ctx) {
    uint64_t id;
    memcpy(&id, s, 8);
    /* Next == \E id \in IdSet : NextId(id)  (:39); NextId (:30-33): id<=MaxId /\ id=nextId */
    if ((int64_t)id <= p->MaxId) {
        uint8_t t[8];
        uint64_t n = id + 1;
        memcpy(t, &n, 8);
        emit(ctx, 0, t);
    }
}
static int idseq_typeok(const P *p, const uint8_t *s) { /* :43 */
    uint64_t id;
    memcpy(&id, s, 8);
    return (int64_t)id <= p->MaxId + 1;
}

/* ==================================================================================== */
/* FiniteReplicatedLog.tla standalone (:97-118): per replica [end, rec[0..L-1]],          */
/* record codes 0 = Nil, 1..K = the K elements of LogRecords                              */
/* ==================================================================================== */
#define FEND(s, r) ((s)[(r) * (1 + p->L)])
#define FREC(s, r, o) ((s)[(r) * (1 + p->L) + 1 + (o)])
static void frl_expand(const P *p, const uint8_t *s, emit_fn emit, void *ctx) {
    uint8_t t[KMO_MAXSB];
    for (int r = 0; r < p->N; r++) {
        int end = FEND(s, r);
        /* \E record, offset : Append(replica, record, offset)  (:116, :99-103) */
        if (end < p->L)
            for (int k = 1; k <= p->K; k++) {
                memcpy(t, s, p->sb);
                FREC(t, r, end) = (uint8_t)k;
                FEND(t, r) = (uint8_t)(end + 1);
                emit(ctx, 0, t);
            }
        /* \E offset \in Offsets : TruncateTo(replica, offset)  (:117, :105-109) */
        for (int o = 0; o < p->L; o++)
            if (o <= end) {
                memcpy(t, s, p->sb);
                for (int x = o; x < p->L; x++) FREC(t, r, x) = 0;
                FEND(t, r) = (uint8_t)o;
                emit(ctx, 1, t);
            }
        /* \E other # replica : ReplicateTo(replica, other)  (:118, :111-113) */
        for (int to = 0; to < p->N; to++)
            if (to != r) {
                int eto = FEND(s, to);
                if (eto < end && eto < p->L) {
                    memcpy(t, s, p->sb);
                    FREC(t, to, eto) = FREC(s, r, eto);
                    FEND(t, to) = (uint8_t)(eto + 1);
                    emit(ctx, 2, t);
                }
            }
    }
}
static int frl_typeok(const P *p, const uint8_t *s) { /* :90-95 */
    for (int r = 0; r < p->N; r++) {
        int end = FEND(s, r);
        if (end > p->L) return 0;
        for (int o = 0; o < p->L; o++) {
            int c = FREC(s, r, o);
            if (c > p->K) return 0;
            if (o < end ? c == 0 : c != 0) return 0;
        }
    }
    return 1;
}

/* ==================================================================================== */
/* KafkaReplication.tla                                                                  */
/* ==================================================================================== */
static void kafka_init(const P *p, uint8_t *s) { /* KafkaReplication.tla:109-120 */
    memset(s, 0, p->sb);
    Q_ISR(s) = (uint8_t)((1u << p->N) - 1); /* quorumState.isr = Replicas (:119) */
}

static inline int presumes(const P *p, const uint8_t *s, int r) { return LDR1(s, r) == r + 1; } /* :126 */
static inline int is_true_leader(const P *p, const uint8_t *s, int l) {                         /* :128-131 */
    return Q_LDR1(s) == l + 1 && presumes(p, s, l) && EP1(s, l) == Q_EP1(s);
}

/* ControllerUpdateIsr (:138-145); newLeader1 = leader+1 or 0 for None */
static void controller_update_isr(const P *p, const uint8_t *s, int newLeader1, int newIsr, int action,
                                  emit_fn emit, void *ctx) {
    int e = NEXTEP(s);
    if (e > p->E) return; /* LeaderEpochSeq!NextId: id <= MaxId (IdSequence.tla:31) */
    uint8_t t[KMO_MAXSB];
    memcpy(t, s, p->sb);
    Q_EP1(t) = (uint8_t)(e + 1);
    Q_LDR1(t) = (uint8_t)newLeader1;
    Q_ISR(t) = (uint8_t)newIsr;
    REQ_LDR1(t, e) = (uint8_t)newLeader1;
    REQ_ISR(t, e) = (uint8_t)newIsr;
    NEXTEP(t) = (uint8_t)(e + 1);
    emit(ctx, action, t);
}

static void ControllerElectLeader(const P *p, const uint8_t *s, int a, emit_fn emit, void *ctx) { /* :176-179 */
    for (int r = 0; r < p->N; r++)
        if ((Q_ISR(s) & BIT(r)) && Q_LDR1(s) != r + 1) controller_update_isr(p, s, r + 1, Q_ISR(s), a, emit, ctx);
}

static void ControllerShrinkIsr(const P *p, const uint8_t *s, int a, emit_fn emit, void *ctx) { /* :158-168 */
    for (int r = 0; r < p->N; r++) {
        if (Q_LDR1(s) == r + 1) {
            if (Q_ISR(s) == BIT(r))
                controller_update_isr(p, s, 0, Q_ISR(s), a, emit, ctx);
            else
                controller_update_isr(p, s, 0, Q_ISR(s) & ~BIT(r), a, emit, ctx);
        } else if (Q_ISR(s) & BIT(r)) {
            controller_update_isr(p, s, Q_LDR1(s), Q_ISR(s) & ~BIT(r), a, emit, ctx);
        }
    }
}

static void BecomeLeader(const P *p, const uint8_t *s, int a, emit_fn emit, void *ctx) { /* :186-195 */
    uint8_t t[KMO_MAXSB];
    for (int e = 0; e < NEXTEP(s); e++) {
        int l1 = REQ_LDR1(s, e);
        if (l1 == 0) continue; /* leader # None, tested first (:188) */
        int l = l1 - 1;
        if (e + 1 > EP1(s, l)) {
            memcpy(t, s, p->sb);
            EP1(t, l) = (uint8_t)(e + 1);
            LDR1(t, l) = (uint8_t)l1;
            ISR(t, l) = REQ_ISR(s, e);
            emit(ctx, a, t);
        }
    }
}

static void LeaderWrite(const P *p, const uint8_t *s, int a, emit_fn emit, void *ctx) { /* :202-207 */
    uint8_t t[KMO_MAXSB];
    int id = NEXTREC(s);
    if (id > p->R - 1) return; /* RecordSeq!NextId, MaxId <- MaxRecords-1 (:78) */
    for (int r = 0; r < p->N; r++) {
        if (!presumes(p, s, r)) continue;
        int end = END(s, r);
        if (end >= p->L) continue; /* Append: ~IsFull, offset = endOffset (FiniteReplicatedLog.tla:99-103) */
        if (EP1(s, r) == 0) abort(); /* a presumed leader always holds an epoch >= 0 */
        memcpy(t, s, p->sb);
        REC(t, r, end) = (uint8_t)(1 + id * p->EP1 + (EP1(s, r) - 1));
        END(t, r) = (uint8_t)(end + 1);
        NEXTREC(t) = (uint8_t)(id + 1);
        emit(ctx, a, t);
    }
}

/* QuorumUpdateLeaderAndIsr (:213-217) */
static void quorum_update(const P *p, const uint8_t *s, int l, int newIsr, int a, emit_fn emit, void *ctx) {
    if (!is_true_leader(p, s, l)) return;
    uint8_t t[KMO_MAXSB];
    memcpy(t, s, p->sb);
    Q_ISR(t) = (uint8_t)newIsr;
    ISR(t, l) = (uint8_t)newIsr;
    emit(ctx, a, t);
}

/* IsFollowerCaughtUp (:219-225).  The \E record is satisfied by the record the leader holds
 * at `offset` whenever offset < end_leader (TypeOk keeps written slots in LogRecords). */
static int is_follower_caught_up(const P *p, const uint8_t *s, int l, int f, int endOffset) {
    if (LDR1(s, f) != l + 1) return 0;
    if (endOffset == 0) return 1;
    int o = endOffset - 1;
    return o < END(s, l) && o < END(s, f);
}

static void LeaderShrinkIsr(const P *p, const uint8_t *s, int a, emit_fn emit, void *ctx) { /* :233-239 */
    for (int l = 0; l < p->N; l++) {
        int isr = ISR(s, l), endOffset = END(s, l);
        for (int r = 0; r < p->N; r++)
            if (r != l && (isr & BIT(r)) && !is_follower_caught_up(p, s, l, r, endOffset))
                quorum_update(p, s, l, isr & ~BIT(r), a, emit, ctx);
    }
}

static void LeaderExpandIsr(const P *p, const uint8_t *s, int a, emit_fn emit, void *ctx) { /* :248-254 */
    for (int l = 0; l < p->N; l++) {
        int isr = ISR(s, l), leaderHw = HW(s, l);
        for (int r = 0; r < p->N; r++)
            if (!(isr & BIT(r)) && is_follower_caught_up(p, s, l, r, leaderHw))
                quorum_update(p, s, l, isr | BIT(r), a, emit, ctx);
    }
}

static void LeaderIncHighWatermark(const P *p, const uint8_t *s, int a, emit_fn emit, void *ctx) { /* :264-271 */
    uint8_t t[KMO_MAXSB];
    for (int l = 0; l < p->N; l++) {
        if (!presumes(p, s, l)) continue;
        int hw = HW(s, l);
        if (hw > p->L - 1) continue; /* offset \in Offsets */
        int ok = 1;
        for (int f = 0; f < p->N && ok; f++)
            if (ISR(s, l) & BIT(f)) ok = (LDR1(s, f) == l + 1) && (hw < END(s, f));
        if (!ok) continue;
        memcpy(t, s, p->sb);
        HW(t, l) = (uint8_t)(hw + 1);
        emit(ctx, a, t);
    }
}

/* FiniteReplicatedLog!TruncateTo applied in place; returns 0 when disabled (:105-109) */
static int truncate_to(const P *p, uint8_t *t, int r, int newEnd) {
    if (newEnd > END(t, r)) return 0;
    for (int o = newEnd; o < p->L; o++) REC(t, r, o) = 0;
    END(t, r) = (uint8_t)newEnd;
    return 1;
}

/* BecomeFollowerAndTruncateTo (:281-294); leader \in Replicas in every caller so the
 * `leader = None` branch (:285-286) cannot be taken. */
static void become_follower_and_truncate_to(const P *p, const uint8_t *s, int l, int r, int off, int a,
                                            emit_fn emit, void *ctx) {
    if (l == r) return;
    uint8_t t[KMO_MAXSB];
    for (int e = 0; e < NEXTEP(s); e++) {
        if (REQ_LDR1(s, e) != l + 1) continue;
        if (!(e + 1 > EP1(s, r))) continue;
        memcpy(t, s, p->sb);
        if (!truncate_to(p, t, r, off)) continue;
        EP1(t, r) = (uint8_t)(e + 1);
        LDR1(t, r) = (uint8_t)(l + 1);
        ISR(t, r) = REQ_ISR(s, e);
        HW(t, r) = (uint8_t)imin(off, HW(s, r));
        emit(ctx, a, t);
    }
}

/* shared tail of FollowerReplicate (:305-309) / FencedFollowerFetch / FollowerFetch:
 * ReplicateTo(leader, follower) (FiniteReplicatedLog.tla:111-113) + follower hw update */
static void replicate_and_update_hw(const P *p, const uint8_t *s, int l, int f, int a, emit_fn emit, void *ctx) {
    int ef = END(s, f);
    if (!(ef < END(s, l) && ef < p->L)) return;
    uint8_t t[KMO_MAXSB];
    memcpy(t, s, p->sb);
    REC(t, f, ef) = REC(s, l, ef);
    END(t, f) = (uint8_t)(ef + 1);
    HW(t, f) = (uint8_t)imin(HW(s, l), ef + 1);
    emit(ctx, a, t);
}

static void FollowerReplicate(const P *p, const uint8_t *s, int a, emit_fn emit, void *ctx) { /* :302-310 */
    for (int f = 0; f < p->N; f++)
        for (int l = 0; l < p->N; l++)
            if (presumes(p, s, l) && LDR1(s, f) == l + 1) replicate_and_update_hw(p, s, l, f, a, emit, ctx);
}

/* invariants ------------------------------------------------------------------------- */
static int kafka_typeok(const P *p, const uint8_t *s) { /* :101-107 */
    int full = (1 << p->N) - 1, maxcode = p->R * p->EP1;
    if (NEXTEP(s) > p->E + 1) return 0;  /* LeaderEpochSeq!TypeOk (IdSequence.tla:43) */
    if (NEXTREC(s) > p->R) return 0;     /* RecordSeq!TypeOk */
    for (int r = 0; r < p->N; r++) {
        int end = END(s, r);
        if (end > p->L) return 0; /* ReplicaLog!TypeOk (FiniteReplicatedLog.tla:90-95) */
        for (int o = 0; o < p->L; o++) {
            int c = REC(s, r, o);
            if (c > maxcode) return 0;
            if (o < end ? c == 0 : c != 0) return 0;
        }
        if (HW(s, r) > p->L || EP1(s, r) > p->E + 1 || LDR1(s, r) > p->N || (ISR(s, r) & ~full)) return 0;
    }
    if (Q_EP1(s) > p->E + 1 || Q_LDR1(s) > p->N || (Q_ISR(s) & ~full)) return 0;
    for (int e = 0; e < NEXTEP(s); e++)
        if (REQ_LDR1(s, e) > p->N || (REQ_ISR(s, e) & ~full)) return 0;
    return 1;
}

static int isr_prefix_ok(const P *p, const uint8_t *s, int r1, int members) {
    int hw = HW(s, r1);
    if (hw == 0) return 1;
    for (int r2 = 0; r2 < p->N; r2++) {
        if (!(members & BIT(r2))) continue;
        for (int o = 0; o < hw; o++) {
            /* \E record : HasEntry(r1,record,o) /\ HasEntry(r2,record,o) */
            if (!(o < END(s, r1) && o < END(s, r2) && REC(s, r1, o) == REC(s, r2, o))) return 0;
        }
    }
    return 1;
}
static int kafka_weakisr(const P *p, const uint8_t *s) { /* :320-326 */
    for (int r1 = 0; r1 < p->N; r1++)
        if (presumes(p, s, r1) && !isr_prefix_ok(p, s, r1, ISR(s, r1))) return 0;
    return 1;
}
static int kafka_strongisr(const P *p, const uint8_t *s) { /* :334-340 */
    for (int r1 = 0; r1 < p->N; r1++)
        if (presumes(p, s, r1) && !isr_prefix_ok(p, s, r1, Q_ISR(s))) return 0;
    return 1;
}
static int kafka_leaderinisr(const P *p, const uint8_t *s) { /* :345 */
    int l1 = Q_LDR1(s);
    return l1 != 0 && (Q_ISR(s) & BIT(l1 - 1)) != 0;
}

/* KafkaTruncateToHighWatermark.tla:29-31 */
static void BecomeFollowerTruncateToHighWatermark(const P *p, const uint8_t *s, int a, emit_fn emit, void *ctx) {
    for (int l = 0; l < p->N; l++)
        for (int r = 0; r < p->N; r++) become_follower_and_truncate_to(p, s, l, r, HW(s, r), a, emit, ctx);
}

/* Kip101.tla:27-39 */
static int lookup_offset_for_epoch(const P *p, const uint8_t *s, int l, int f, int epoch) {
    int el = END(s, l);
    if (el == 0) return HW(s, f);
    if (REC_EPOCH(REC(s, l, el - 1)) == epoch) return el;
    for (int o = 0; o < el; o++) /* Min(OffsetsWithLargerEpochs) */
        if (REC_EPOCH(REC(s, l, o)) > epoch) return o;
    return HW(s, f);
}
static void BecomeFollowerTruncateKip101(const P *p, const uint8_t *s, int a, emit_fn emit, void *ctx) { /* Kip101.tla:41-47 */
    for (int l = 0; l < p->N; l++)
        for (int r = 0; r < p->N; r++) {
            int er = END(s, r);
            if (er == 0)
                become_follower_and_truncate_to(p, s, l, r, 0, a, emit, ctx);
            else
                become_follower_and_truncate_to(
                    p, s, l, r, lookup_offset_for_epoch(p, s, l, r, REC_EPOCH(REC(s, r, er - 1))), a, emit, ctx);
        }
}

/* Kip279.tla:27-45 */
static int first_non_matching_offset_from_tail(const P *p, const uint8_t *s, int l, int f) {
    if (END(s, l) == 0) return 0;
    int best = 0; /* Max(matching)+1, or 0 when empty */
    for (int o = 0; o < END(s, f); o++)
        if (o < END(s, l) && REC(s, l, o) == REC(s, f, o)) best = o + 1;
    return best;
}
static void BecomeFollowerTruncateKip279(const P *p, const uint8_t *s, int a, emit_fn emit, void *ctx) { /* Kip279.tla:47-51 */
    for (int l = 0; l < p->N; l++)
        for (int r = 0; r < p->N; r++) {
            if (END(s, r) == 0) become_follower_and_truncate_to(p, s, l, r, 0, a, emit, ctx);
            become_follower_and_truncate_to(p, s, l, r, first_non_matching_offset_from_tail(p, s, l, r), a, emit, ctx);
        }
}

/* Kip320.tla ------------------------------------------------------------------------- */
static int is_following_leader_epoch(const P *p, const uint8_t *s, int l, int f) { /* Kip320.tla:39-42 */
    return presumes(p, s, l) && LDR1(s, f) == l + 1 && EP1(s, f) == EP1(s, l);
}
static void FencedFollowerFetch(const P *p, const uint8_t *s, int a, emit_fn emit, void *ctx) { /* Kip320.tla:49-56 */
    for (int f = 0; f < p->N; f++)
        for (int l = 0; l < p->N; l++)
            if (is_following_leader_epoch(p, s, l, f)) replicate_and_update_hw(p, s, l, f, a, emit, ctx);
}
static void FencedLeaderIncHighWatermark(const P *p, const uint8_t *s, int a, emit_fn emit, void *ctx) { /* Kip320.tla:63-70 */
    uint8_t t[KMO_MAXSB];
    for (int l = 0; l < p->N; l++) {
        int hw = HW(s, l);
        if (!(hw < END(s, l))) continue;
        int ok = 1;
        for (int f = 0; f < p->N && ok; f++)
            if (ISR(s, l) & BIT(f)) ok = is_following_leader_epoch(p, s, l, f) && hw < END(s, f);
        if (!ok) continue;
        memcpy(t, s, p->sb);
        HW(t, l) = (uint8_t)(hw + 1);
        emit(ctx, a, t);
    }
}
static void FencedLeaderShrinkIsr(const P *p, const uint8_t *s, int a, emit_fn emit, void *ctx) { /* Kip320.tla:78-85 */
    for (int l = 0; l < p->N; l++) {
        int isr = ISR(s, l);
        for (int f = 0; f < p->N; f++)
            if (f != l && (isr & BIT(f))) {
                /* :82-83 is a disjunction of two state predicates in front of the primed conjunct :84: TLC's
                 * next-state enumeration continues from EVERY disjunct that holds (Tool.getNextStates, OPCODE_lor
                 * [TLC-recall]), so the successor is generated once per disjunct — twice when both hold.
                 * (Found by Oracle-R, oracle/tlar, which executes the module's text that way.) */
                if (!is_following_leader_epoch(p, s, l, f)) quorum_update(p, s, l, isr & ~BIT(f), a, emit, ctx);
                if (END(s, f) < END(s, l)) quorum_update(p, s, l, isr & ~BIT(f), a, emit, ctx);
            }
    }
}
static int hw_reached_current_epoch(const P *p, const uint8_t *s, int l) { /* Kip320.tla:87-92, Kip320FirstTry.tla:122-127 */
    int hw = HW(s, l);
    if (hw == END(s, l)) return 1;
    return hw < END(s, l) && REC_EPOCH(REC(s, l, hw)) == EP1(s, l) - 1;
}
static int follower_reached_hw(const P *p, const uint8_t *s, int l, int f) { /* Kip320.tla:94-98 */
    int hw = HW(s, l);
    return hw == 0 || hw - 1 < END(s, f);
}
static void FencedLeaderExpandIsr(const P *p, const uint8_t *s, int a, emit_fn emit, void *ctx) { /* Kip320.tla:110-117 */
    for (int l = 0; l < p->N; l++) {
        int isr = ISR(s, l);
        for (int f = 0; f < p->N; f++)
            if (!(isr & BIT(f)) && is_following_leader_epoch(p, s, l, f) && follower_reached_hw(p, s, l, f) &&
                hw_reached_current_epoch(p, s, l))
                quorum_update(p, s, l, isr | BIT(f), a, emit, ctx);
    }
}
static void FencedBecomeFollowerAndTruncate(const P *p, const uint8_t *s, int a, emit_fn emit, void *ctx) { /* Kip320.tla:134-148 */
    uint8_t t[KMO_MAXSB];
    for (int l = 0; l < p->N; l++)
        for (int r = 0; r < p->N; r++) {
            if (l == r) continue;
            for (int e = 0; e < NEXTEP(s); e++) {
                if (REQ_LDR1(s, e) != l + 1 || !(e + 1 > EP1(s, r))) continue;
                if (!presumes(p, s, l) || EP1(s, l) != e + 1) continue; /* :142-143 */
                int off = first_non_matching_offset_from_tail(p, s, l, r);
                memcpy(t, s, p->sb);
                if (!truncate_to(p, t, r, off)) continue;
                EP1(t, r) = (uint8_t)(e + 1); /* BecomeFollower (:119-124) */
                LDR1(t, r) = (uint8_t)(l + 1);
                ISR(t, r) = REQ_ISR(s, e);
                HW(t, r) = (uint8_t)imin(off, HW(s, r));
                emit(ctx, a, t);
            }
        }
}

/* Kip320FirstTry.tla ----------------------------------------------------------------- */
static int caught_up_to_leader_epoch(const P *p, const uint8_t *s, int l, int f, int endOffset) { /* :49-57 */
    if (!presumes(p, s, l) || LDR1(s, f) != l + 1) return 0;
    if (endOffset == 0) return 1;
    int o = endOffset - 1;
    return o < END(s, l) && o < END(s, f) && REC_EPOCH(REC(s, f, o)) == REC_EPOCH(REC(s, l, o));
}
static int follower_needs_truncation(const P *p, const uint8_t *s, int f, int l) { /* :64-69 */
    if (END(s, f) > END(s, l)) return 1;
    if (END(s, f) == 0) return 0;
    int o = END(s, f) - 1; /* IsLatestEntry(follower, record, offset) */
    return o < END(s, l) && REC_EPOCH(REC(s, l, o)) != REC_EPOCH(REC(s, f, o));
}
static void FollowerTruncate(const P *p, const uint8_t *s, int a, emit_fn emit, void *ctx) { /* :75-82 */
    uint8_t t[KMO_MAXSB];
    for (int l = 0; l < p->N; l++)
        for (int f = 0; f < p->N; f++) {
            if (!(presumes(p, s, l) && LDR1(s, f) == l + 1 && follower_needs_truncation(p, s, f, l))) continue;
            int off = first_non_matching_offset_from_tail(p, s, l, f);
            memcpy(t, s, p->sb);
            if (!truncate_to(p, t, f, off)) continue;
            HW(t, f) = (uint8_t)imin(off, HW(s, f));
            emit(ctx, a, t);
        }
}
static void ImprovedLeaderIncHighWatermark(const P *p, const uint8_t *s, int a, emit_fn emit, void *ctx) { /* :90-97 */
    uint8_t t[KMO_MAXSB];
    for (int l = 0; l < p->N; l++) {
        if (!presumes(p, s, l)) continue;
        int hw = HW(s, l);
        if (!(hw < END(s, l))) continue; /* \E record : HasEntry(leader, record, leaderHw) */
        int ok = 1;
        for (int f = 0; f < p->N && ok; f++)
            if (ISR(s, l) & BIT(f)) ok = caught_up_to_leader_epoch(p, s, l, f, hw + 1);
        if (!ok) continue;
        memcpy(t, s, p->sb);
        HW(t, l) = (uint8_t)(hw + 1);
        emit(ctx, a, t);
    }
}
static void FollowerFetch(const P *p, const uint8_t *s, int a, emit_fn emit, void *ctx) { /* :103-111 */
    for (int f = 0; f < p->N; f++)
        for (int l = 0; l < p->N; l++)
            if (caught_up_to_leader_epoch(p, s, l, f, END(s, f))) replicate_and_update_hw(p, s, l, f, a, emit, ctx);
}
static void LeaderShrinkIsrBetterFencing(const P *p, const uint8_t *s, int a, emit_fn emit, void *ctx) { /* :114-120 */
    for (int l = 0; l < p->N; l++) {
        int isr = ISR(s, l), endOffset = END(s, l);
        for (int r = 0; r < p->N; r++)
            if (r != l && (isr & BIT(r)) && !caught_up_to_leader_epoch(p, s, l, r, endOffset))
                quorum_update(p, s, l, isr & ~BIT(r), a, emit, ctx);
    }
}
static void LeaderExpandIsrBetterFencing(const P *p, const uint8_t *s, int a, emit_fn emit, void *ctx) { /* :134-141 */
    for (int l = 0; l < p->N; l++) {
        int isr = ISR(s, l), leaderHw = HW(s, l);
        for (int r = 0; r < p->N; r++)
            if (!(isr & BIT(r)) && caught_up_to_leader_epoch(p, s, l, r, leaderHw) && hw_reached_current_epoch(p, s, l))
                quorum_update(p, s, l, isr | BIT(r), a, emit, ctx);
    }
}
static void BecomeFollower(const P *p, const uint8_t *s, int a, emit_fn emit, void *ctx) { /* :148-157 */
    uint8_t t[KMO_MAXSB];
    for (int l = 0; l < p->N; l++)
        for (int r = 0; r < p->N; r++) {
            if (l == r) continue;
            for (int e = 0; e < NEXTEP(s); e++) {
                if (REQ_LDR1(s, e) != l + 1 || !(e + 1 > EP1(s, r))) continue;
                memcpy(t, s, p->sb);
                EP1(t, r) = (uint8_t)(e + 1);
                LDR1(t, r) = (uint8_t)(l + 1);
                ISR(t, r) = REQ_ISR(s, e);
                emit(ctx, a, t);
            }
        }
}


/* ==================================================================================== */
/* AsyncIsr.tla (standalone).  Unbounded as written (version: Nat, offsets: [Replicas -> Nat],
 * :40-56; LeaderWrite :117-119), so it runs under the explicit state constraint of
 * models/MCAsyncIsr.tla (NOT part of the reference):
 *     leaderState.offsets[Leader] <= MaxOffset /\ controllerState.version <= MaxVersion
 * with MaxOffset = L and MaxVersion = E here; Leader is replica 0.  One byte per field:
 *   [0] controllerState.isr  [1] controllerState.version
 *   [2] leaderState.isr [3] .version [4] .pendingIsr [5] .pendingVersion+1 (Nil = -1 -> 0)
 *   [6+r] leaderState.offsets[r]
 *   requests (a set of [isr, version], versions 0..E): per version a bitset over the 2^N isr
 *     masks, a_rb bytes each;
 *   updates: only the controller adds to it, always [isr, version = old version + 1] (:68-79,
 *     :81-86), so the set is in bijection with the array, indexed by version 1..E+1, of the isr
 *     written at that version (zero while version > controllerState.version).
 * Successors outside the constraint carry offsets[Leader] = L+1 or version = E+1: both fit.  */
/* ==================================================================================== */
#define A_CISR(s) ((s)[0])
#define A_CVER(s) ((s)[1])
#define A_LISR(s) ((s)[2])
#define A_LVER(s) ((s)[3])
#define A_PISR(s) ((s)[4])
#define A_PVER1(s) ((s)[5])
#define A_OFF(s, r) ((s)[6 + (r)])
#define A_UPD(s, v) ((s)[p->a_upd + (v)-1]) /* v in 1..E+1 */
static inline int a_has_req(const P *p, const uint8_t *s, int v, int isr) {
    return s[p->a_req + v * p->a_rb + (isr >> 3)] >> (isr & 7) & 1;
}
static inline void a_add_req(const P *p, uint8_t *s, int v, int isr) {
    s[p->a_req + v * p->a_rb + (isr >> 3)] |= (uint8_t)(1u << (isr & 7));
}
static void async_init(const P *p, uint8_t *s) { /* :137-150 */
    memset(s, 0, p->sb);
    A_CISR(s) = (uint8_t)((1u << p->N) - 1);
    A_LISR(s) = (uint8_t)((1u << p->N) - 1);
    A_PVER1(s) = 0; /* pendingVersion |-> Nil */
}
static int async_high_watermark(const P *p, const uint8_t *s) { /* :58-60; Leader is always in leaderState.isr, so the set is non-empty */
    int potential = A_LISR(s) | A_PISR(s), hw = 1 << 30;
    for (int r = 0; r < p->N; r++)
        if (potential >> r & 1) hw = imin(hw, A_OFF(s, r));
    return hw;
}
static int async_in_model(const P *p, const uint8_t *s) { return A_OFF(s, 0) <= p->L && A_CVER(s) <= p->E; }
static int async_typeok(const P *p, const uint8_t *s) { /* :62-66: pendingVersion \in Nat (:44) is false while it is Nil (:38) */
    (void)p;
    return A_PVER1(s) != 0;
}
static int async_valid_hw(const P *p, const uint8_t *s) { /* :161-162 */
    const int hw = async_high_watermark(p, s);
    for (int r = 0; r < p->N; r++)
        if ((A_CISR(s) >> r & 1) && A_OFF(s, r) < hw) return 0;
    return 1;
}
static void async_expand(const P *p, const uint8_t *s, emit_fn emit, void *ctx) { /* Next :152-159 */
    uint8_t t[KMO_MAXSB];
    const int cver = A_CVER(s), lver = A_LVER(s);
    /* ControllerShrinkIsr :72-79 (ControllerWriteIsr :68-70) */
    for (int r = 1; r < p->N; r++)
        if (A_CISR(s) >> r & 1) {
            memcpy(t, s, p->sb);
            A_CISR(t) = (uint8_t)(A_CISR(s) & ~(1u << r));
            A_CVER(t) = (uint8_t)(cver + 1);
            A_UPD(t, cver + 1) = A_CISR(t);
            emit(ctx, 0, t);
        }
    /* ControllerHandleRequest :81-86 */
    if (cver <= p->E)
        for (int isr = 0; isr < (1 << p->N); isr++)
            if (a_has_req(p, s, cver, isr)) {
                memcpy(t, s, p->sb);
                A_CISR(t) = (uint8_t)isr;
                A_CVER(t) = (uint8_t)(cver + 1);
                A_UPD(t, cver + 1) = (uint8_t)isr;
                emit(ctx, 1, t);
            }
    /* LeaderRequestShrinkIsr :88-100 */
    for (int r = 1; r < p->N; r++)
        if (A_LISR(s) >> r & 1) {
            const int isr = A_LISR(s) & ~(1 << r);
            memcpy(t, s, p->sb);
            a_add_req(p, t, lver, isr);
            A_PISR(t) = (uint8_t)(A_PISR(s) | isr);
            A_PVER1(t) = (uint8_t)(lver + 1);
            emit(ctx, 2, t);
        }
    /* LeaderRequestExpandIsr :102-115 */
    {
        const int hw = async_high_watermark(p, s);
        for (int r = 0; r < p->N; r++)
            if (!(A_LISR(s) >> r & 1) && A_OFF(s, r) >= hw) {
                const int isr = A_LISR(s) | (1 << r);
                memcpy(t, s, p->sb);
                a_add_req(p, t, lver, isr);
                A_PISR(t) = (uint8_t)(A_PISR(s) | isr);
                A_PVER1(t) = (uint8_t)(lver + 1);
                emit(ctx, 3, t);
            }
    }
    /* LeaderWrite :117-119 */
    memcpy(t, s, p->sb);
    A_OFF(t, 0)++;
    emit(ctx, 4, t);
    /* LeaderHandleUpdate :121-129 */
    for (int v = lver + 1; v <= cver; v++) {
        memcpy(t, s, p->sb);
        A_LISR(t) = A_UPD(s, v);
        A_LVER(t) = (uint8_t)v;
        A_PISR(t) = 0;
        A_PVER1(t) = 0;
        emit(ctx, 5, t);
    }
    /* FollowerReplicate :131-135 */
    for (int r = 1; r < p->N; r++)
        if (A_OFF(s, r) < A_OFF(s, 0)) {
            memcpy(t, s, p->sb);
            A_OFF(t, r)++;
            emit(ctx, 6, t);
        }
}

typedef void (*action_fn)(const P *, const uint8_t *, int, emit_fn, void *);
static const action_fn NEXT_TRUNC_HW[] = {/* KafkaTruncateToHighWatermark.tla:33-42 */
                                          ControllerElectLeader, ControllerShrinkIsr, BecomeLeader, LeaderExpandIsr,
                                          LeaderShrinkIsr, LeaderWrite, LeaderIncHighWatermark,
                                          BecomeFollowerTruncateToHighWatermark, FollowerReplicate};
static const action_fn NEXT_KIP101[] = {/* Kip101.tla:49-58 */
                                        ControllerElectLeader, ControllerShrinkIsr, BecomeLeader, LeaderExpandIsr,
                                        LeaderShrinkIsr, LeaderWrite, LeaderIncHighWatermark,
                                        BecomeFollowerTruncateKip101, FollowerReplicate};
static const action_fn NEXT_KIP279[] = {/* Kip279.tla:53-62 */
                                        ControllerElectLeader, ControllerShrinkIsr, BecomeLeader, LeaderExpandIsr,
                                        LeaderShrinkIsr, LeaderWrite, LeaderIncHighWatermark,
                                        BecomeFollowerTruncateKip279, FollowerReplicate};
static const action_fn NEXT_KIP320[] = {/* Kip320.tla:150-159 */
                                        ControllerElectLeader, ControllerShrinkIsr, BecomeLeader, FencedLeaderExpandIsr,
                                        FencedLeaderShrinkIsr, LeaderWrite, FencedLeaderIncHighWatermark,
                                        FencedBecomeFollowerAndTruncate, FencedFollowerFetch};
static const action_fn NEXT_KIP320_FIRST[] = {/* Kip320FirstTry.tla:159-169 */
                                              ControllerElectLeader, ControllerShrinkIsr, BecomeLeader,
                                              LeaderExpandIsrBetterFencing, LeaderShrinkIsrBetterFencing, LeaderWrite,
                                              ImprovedLeaderIncHighWatermark, BecomeFollower, FollowerFetch,
                                              FollowerTruncate};

static void model_expand(const P *p, const uint8_t *s, emit_fn emit, void *ctx) {
    const action_fn *acts;
    switch (p->model) {
    case M_IDSEQ: idseq_expand(p, s, emit, ctx); return;
    case M_FRL: frl_expand(p, s, emit, ctx); return;
    case M_ASYNC_ISR: async_expand(p, s, emit, ctx); return;
    case M_TRUNC_HW: acts = NEXT_TRUNC_HW; break;
    case M_KIP101: acts = NEXT_KIP101; break;
    case M_KIP279: acts = NEXT_KIP279; break;
    case M_KIP320: acts = NEXT_KIP320; break;
    default: acts = NEXT_KIP320_FIRST; break;
    }
    for (int a = 0; a < p->nact; a++) acts[a](p, s, a, emit, ctx);
}
/* TLC CONSTRAINT: only AsyncIsr has one */
static int model_in_model(const P *p, const uint8_t *s) { return p->model == M_ASYNC_ISR ? async_in_model(p, s) : 1; }
static int model_invariant(const P *p, int inv, const uint8_t *s) {
    if (p->model == M_IDSEQ) return inv == INV_TYPEOK ? idseq_typeok(p, s) : 1;
    if (p->model == M_FRL) return inv == INV_TYPEOK ? frl_typeok(p, s) : 1;
    if (p->model == M_ASYNC_ISR) /* 2: LeaderOffsetInRange of models/MCAsyncIsr.tla (not in the reference) */
        return inv == INV_TYPEOK ? async_typeok(p, s) : inv == INV_VALIDHW ? async_valid_hw(p, s) : inv == 2 ? A_OFF(s, 0) <= p->L : 1;
    switch (inv) {
    case INV_TYPEOK: return kafka_typeok(p, s);
    case INV_WEAKISR: return kafka_weakisr(p, s);
    case INV_STRONGISR: return kafka_strongisr(p, s);
    default: return kafka_leaderinisr(p, s);
    }
}

/* ==================================================================================== */
/* Exact-state BFS engine                                                                */
/* ==================================================================================== */
#define CHUNK_BITS 20
#define CHUNK (1u << CHUNK_BITS)
#define MAX_CHUNKS (1u << 16)

typedef struct {
    P p;
    kmo_config cfg;
    int rs; /* record size = sb + 4 (parent) + 1 (action) */
    uint8_t *chunks[MAX_CHUNKS];
    pthread_mutex_t chunk_mu;
    _Atomic uint64_t nstates;
    _Atomic uint64_t *table;
    uint64_t cap; /* power of two */
    /* per-level shared cursors */
    _Atomic uint64_t cursor;
    uint64_t lo, hi;
    int table_overflow;
} Engine;

#define IDX_BLOCK 64 /* arena indices a worker reserves at a time */
typedef struct {
    Engine *e;
    uint64_t parent;
    uint64_t blk_next, blk_end; /* reserved, not yet used arena indices [blk_next, blk_end) */
    uint64_t generated, succ_of_state, deadlocks;
    uint64_t action_generated[KMO_MAX_ACTIONS];
    /* violating successors outside the constraint seen by this worker */
    uint64_t out_cnt[4], out_parent[4];
    int out_action[4];
    uint8_t out_state[4][KMO_MAXSB];
} Worker;

static inline uint8_t *rec_ptr(Engine *e, uint64_t idx) {
    return e->chunks[idx >> CHUNK_BITS] + (idx & (CHUNK - 1)) * (uint64_t)e->rs;
}
static void ensure_chunk(Engine *e, uint64_t idx) {
    uint64_t c = idx >> CHUNK_BITS;
    if (c >= MAX_CHUNKS) {
        fprintf(stderr, "kmc_oracle: arena exhausted\n");
        abort();
    }
    if (__atomic_load_n(&e->chunks[c], __ATOMIC_ACQUIRE)) return;
    pthread_mutex_lock(&e->chunk_mu);
    if (!e->chunks[c]) {
        uint8_t *m = malloc((uint64_t)CHUNK * e->rs);
        if (!m) {
            fprintf(stderr, "kmc_oracle: out of memory\n");
            abort();
        }
        __atomic_store_n(&e->chunks[c], m, __ATOMIC_RELEASE);
    }
    pthread_mutex_unlock(&e->chunk_mu);
}

static inline uint64_t hash_bytes(const uint8_t *b, int n) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
    int i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, b + i, 8);
        h = (h ^ w) * 0xff51afd7ed558ccdull;
        h ^= h >> 32;
    }
    uint64_t w = 0;
    if (i < n) memcpy(&w, b + i, n - i);
    h = (h ^ w) * 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 29;
    h *= 0xbf58476d1ce4e5b9ull;
    h ^= h >> 32;
    return h;
}

/* Fingerprint-only mode (CLI --fp-only, or KMO_FP_ONLY=1): the table holds this file's own 64-bit hash of a state's
 * canonical bytes instead of a reference to the stored state, so states of finished levels can be thrown away and a
 * search far beyond the exact mode's RAM fits (8 B per state in the table + two levels of states).  It is NOT exact any
 * more — two states with one hash lose a state, n^2 / 2^65 expected — but it is an engine-independent second opinion on
 * the GPU's counts where the exact mode cannot go: another hash function over another state encoding, another table,
 * another BFS.  Parent links (traces) are meaningless in this mode.  --table-log2 N fixes the table size (no growth). */
static int g_fp_only = 0;
static int g_table_log2 = 0;
/* The exact mode's hash_bytes only spreads states over the table (a state is confirmed by memcmp), and one
 * multiply-xorshift round per 8 bytes is enough for that.  It is NOT enough to BE the state: on the 810 M-state
 * KafkaTruncateToHighWatermark 3/6/6/2 run it lost 23,131 states to systematic collisions (the same lesson the GPU's
 * fingerprint taught at 7.6e7 states).  Fingerprint-only mode therefore absorbs every 8 bytes through
 * a full-avalanche finaliser — murmur3's fmix64, not the splitmix64 constants of the GPU's kmc_fingerprint, over this
 * file's byte-per-field encoding, not the GPU's bit packing: an independent function of an independent representation. */
static inline uint64_t fmix64(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return k;
}
static inline uint64_t hash_bytes_strong(const uint8_t *b, int n) {
    uint64_t h = fmix64(0x2545F4914F6CDD1Dull + (uint64_t)n);
    int i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, b + i, 8);
        h = fmix64(h ^ w) + 0x9E3779B97F4A7C15ull;
    }
    if (i < n) {
        uint64_t w = 0;
        memcpy(&w, b + i, n - i);
        h = fmix64(h ^ w) + 0x9E3779B97F4A7C15ull;
    }
    return fmix64(h);
}
#define SLOT_BUSY 1ull
/* returns 1 when the state was new (and stores it with parent/action) */
/* Arena indices are handed out in blocks of IDX_BLOCK per worker: one shared fetch-add per new state serialised the
 * whole expansion once more than ~16 threads hammered that cache line (256 threads: 0.14 M states/s against 5 M/s on
 * 16).  The unused tail of each worker's last block leaves holes at the end of a level; compact_level() fills them
 * before the level is used, so a level stays a dense index range. */
static uint64_t take_index(Engine *e, uint64_t *blk_next, uint64_t *blk_end) {
    if (!blk_next) return atomic_fetch_add(&e->nstates, 1);
    if (*blk_next == *blk_end) {
        *blk_next = atomic_fetch_add(&e->nstates, IDX_BLOCK);
        *blk_end = *blk_next + IDX_BLOCK;
    }
    return (*blk_next)++;
}
static int engine_insert(Engine *e, const uint8_t *st, uint64_t parent, int action, uint64_t *blk_next, uint64_t *blk_end) {
    int sb = e->p.sb;
    uint64_t h = g_fp_only ? hash_bytes_strong(st, sb) : hash_bytes(st, sb);
    uint64_t tag = (h >> 40) << 40; /* high 24 bits */
    uint64_t i = h & (e->cap - 1);
    for (uint64_t probes = 0;;) {
        if (probes > e->cap) {
            e->table_overflow = 1;
            return 0;
        }
        uint64_t v = atomic_load_explicit(&e->table[i], memory_order_acquire);
        if (g_fp_only) {
            const uint64_t hv = h < 2 ? h + 2 : h; /* never 0 (empty) */
            if (v == 0) {
                uint64_t exp = 0;
                if (atomic_compare_exchange_strong(&e->table[i], &exp, hv)) {
                    uint64_t idx = take_index(e, blk_next, blk_end);
                    ensure_chunk(e, idx);
                    uint8_t *r = rec_ptr(e, idx);
                    memcpy(r, st, sb);
                    memset(r + sb, 0xFF, 4);
                    r[sb + 4] = (uint8_t)action;
                    return 1;
                }
                continue; /* somebody else took the slot: re-read it */
            }
            if (v == hv) return 0;
            probes++;
            i = (i + 1) & (e->cap - 1);
            continue;
        }
        if (v == 0) {
            uint64_t exp = 0;
            if (atomic_compare_exchange_strong(&e->table[i], &exp, SLOT_BUSY)) {
                uint64_t idx = take_index(e, blk_next, blk_end);
                ensure_chunk(e, idx);
                uint8_t *r = rec_ptr(e, idx);
                memcpy(r, st, sb);
                uint32_t par = (uint32_t)parent;
                memcpy(r + sb, &par, 4);
                r[sb + 4] = (uint8_t)action;
                atomic_store_explicit(&e->table[i], tag | (idx + 2), memory_order_release);
                return 1;
            }
            continue; /* re-read the slot */
        }
        if (v == SLOT_BUSY) continue; /* writer in flight: spin */
        if ((v & 0xFFFFFF0000000000ull) == tag) {
            uint64_t idx = (v & 0xFFFFFFFFFFull) - 2;
            if (memcmp(rec_ptr(e, idx), st, sb) == 0) return 0;
        }
        probes++; /* spins on a BUSY slot are not probes */
        i = (i + 1) & (e->cap - 1);
    }
}

/* Static slices of an index range on T threads (the level-start phases: invariants, re-hashing).  The BFS used to run
 * both on the calling thread alone, which capped the whole search near the single-thread rate whatever `threads` was
 * (8 threads: 1.4x; the expansion itself scales). */
typedef struct {
    Engine *e;
    void (*fn)(void *job, uint64_t lo, uint64_t hi);
    void *arg;
    uint64_t lo, hi;
} Slice;
static void *slice_main(void *a) {
    Slice *sl = a;
    sl->fn(sl->arg, sl->lo, sl->hi);
    return NULL;
}
/* args: T contiguous per-thread argument records of `stride` bytes */
static void run_slices(int T, uint64_t lo, uint64_t hi, void (*fn)(void *, uint64_t, uint64_t), void *args, size_t stride) {
    if (T <= 1 || hi - lo < 4096) {
        fn(args, lo, hi);
        return;
    }
    pthread_t th[256];
    Slice sl[256];
    if (T > 256) T = 256;
    for (int t = 0; t < T; t++) {
        sl[t].fn = fn;
        sl[t].arg = (char *)args + (size_t)t * stride;
        sl[t].lo = lo + (hi - lo) * (uint64_t)t / (uint64_t)T;
        sl[t].hi = lo + (hi - lo) * (uint64_t)(t + 1) / (uint64_t)T;
        pthread_create(&th[t], NULL, slice_main, &sl[t]);
    }
    for (int t = 0; t < T; t++) pthread_join(th[t], NULL);
}

typedef struct {
    Engine *e;
    uint64_t cap;
} RehashJob;
static void rehash_slice(void *a, uint64_t lo, uint64_t hi) {
    RehashJob *j = a;
    Engine *e = j->e;
    const uint64_t cap = j->cap;
    for (uint64_t idx = lo; idx < hi; idx++) {
        const uint8_t *st = rec_ptr(e, idx);
        uint64_t h = hash_bytes(st, e->p.sb);
        uint64_t tag = (h >> 40) << 40, i = h & (cap - 1);
        for (;;) {
            uint64_t exp = 0;
            if (atomic_load_explicit(&e->table[i], memory_order_relaxed) == 0 &&
                atomic_compare_exchange_strong(&e->table[i], &exp, tag | (idx + 2)))
                break;
            i = (i + 1) & (cap - 1);
        }
    }
}
static void engine_grow(Engine *e, uint64_t want, int T) {
    if (g_fp_only && e->cap) {
        /* the table cannot be rebuilt from states that are no longer kept: it has its final size from the start */
        if (atomic_load(&e->nstates) > e->cap - e->cap / 8) {
            fprintf(stderr, "kmc_oracle: --fp-only table of 2^%d slots is full; rerun with a larger --table-log2\n", g_table_log2);
            abort();
        }
        return;
    }
    if (g_fp_only) want = 1ull << (g_table_log2 ? g_table_log2 : 28);
    uint64_t cap = e->cap ? e->cap : 1024;
    while (cap < want) cap <<= 1;
    if (cap == e->cap) return;
    if (e->cap && cap < 4 * e->cap && cap < (1ull << 30)) cap = 4 * e->cap; /* below 8 GiB: grow by 4, re-hash half as often */
    /* a fresh anonymous mapping is zero-filled; ask for huge pages (512x fewer first-touch faults — they, not the
     * re-hashing, were most of the "table growth" time) and let the re-hashing threads do the first touch */
    if (e->table) munmap((void *)e->table, e->cap * sizeof(uint64_t));
    void *m = mmap(NULL, cap * sizeof(uint64_t), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) {
        fprintf(stderr, "kmc_oracle: out of memory (table)\n");
        abort();
    }
#ifdef MADV_HUGEPAGE
    madvise(m, cap * sizeof(uint64_t), MADV_HUGEPAGE);
#endif
    e->table = m;
    e->cap = cap;
    uint64_t n = atomic_load(&e->nstates);
    RehashJob jobs[256];
    for (int t = 0; t < 256; t++) { jobs[t].e = e; jobs[t].cap = cap; }
    run_slices(T, 0, n, rehash_slice, jobs, sizeof(RehashJob));
}

typedef struct {
    Engine *e;
    uint32_t inv_mask;
    uint64_t cnt[4], first[4];
} InvJob;
static void invariant_slice(void *a, uint64_t lo, uint64_t hi) {
    InvJob *j = a;
    for (uint64_t idx = lo; idx < hi; idx++)
        for (int inv = 0; inv < 4; inv++)
            if ((j->inv_mask >> inv & 1) && !model_invariant(&j->e->p, inv, rec_ptr(j->e, idx)))
                if (j->cnt[inv]++ == 0) j->first[inv] = idx;
}

static void worker_emit(void *ctx, int action, const uint8_t *succ) {
    Worker *w = ctx;
    w->generated++;
    w->succ_of_state++;
    w->action_generated[action]++;
    if (!model_in_model(&w->e->p, succ)) {
        /* TLC CONSTRAINT [TLC-recall, ModelChecker.doNext]: not fingerprinted, not queued, but the
         * invariants are evaluated — every time the state is generated, since it is never "seen" */
        for (int inv = 0; inv < 4; inv++)
            if ((w->e->cfg.inv_mask >> inv & 1) && !model_invariant(&w->e->p, inv, succ)) {
                if (w->out_cnt[inv]++ == 0 ||
                    memcmp(succ, w->out_state[inv], w->e->p.sb) < 0) { /* keep the smallest: deterministic */
                    w->out_parent[inv] = w->parent;
                    w->out_action[inv] = action;
                    memcpy(w->out_state[inv], succ, w->e->p.sb);
                }
            }
        return;
    }
    engine_insert(w->e, succ, w->parent, action, &w->blk_next, &w->blk_end);
}

/* After a level's expansion: the reserved-but-unused indices of every worker (the tail of its last block) are holes in
 * [lo, top).  States from the top of the range move into the holes below the new top, and their table entries follow. */
static void compact_level(Engine *e, Worker *ws, int T, uint64_t lo) {
    uint64_t top = atomic_load(&e->nstates), nholes = 0;
    for (int t = 0; t < T; t++) nholes += ws[t].blk_end - ws[t].blk_next;
    if (nholes == 0) return;
    const uint64_t new_top = top - nholes;
    /* mark the holes (an index can be a hole at most once; the ranges are disjoint) */
    uint8_t *is_hole = calloc(top - lo + 1, 1);
    for (int t = 0; t < T; t++)
        for (uint64_t i = ws[t].blk_next; i < ws[t].blk_end; i++) is_hole[i - lo] = 1;
    uint64_t src = top; /* walks down over the valid states at or above new_top */
    for (uint64_t dst = lo; dst < new_top; dst++) {
        if (!is_hole[dst - lo]) continue;
        do { src--; } while (is_hole[src - lo]);
        /* src >= new_top holds a state: move it to dst and repoint its table entry */
        uint8_t *from = rec_ptr(e, src), *to = rec_ptr(e, dst);
        memcpy(to, from, e->rs);
        if (g_fp_only) continue; /* the table holds hashes, not references */
        uint64_t h = hash_bytes(to, e->p.sb), tag = (h >> 40) << 40, i = h & (e->cap - 1);
        for (;;) {
            uint64_t v = atomic_load_explicit(&e->table[i], memory_order_relaxed);
            if ((v & 0xFFFFFFFFFFull) == src + 2 && (v & 0xFFFFFF0000000000ull) == tag) {
                atomic_store_explicit(&e->table[i], tag | (dst + 2), memory_order_relaxed);
                break;
            }
            i = (i + 1) & (e->cap - 1);
        }
    }
    free(is_hole);
    atomic_store(&e->nstates, new_top);
}

static void *worker_main(void *arg) {
    Worker *w = arg;
    Engine *e = w->e;
    const uint64_t GRAB = 256;
    for (;;) {
        uint64_t b = atomic_fetch_add(&e->cursor, GRAB);
        if (b >= e->hi) break;
        uint64_t end = b + GRAB < e->hi ? b + GRAB : e->hi;
        for (uint64_t idx = b; idx < end; idx++) {
            uint8_t st[KMO_MAXSB];
            memcpy(st, rec_ptr(e, idx), e->p.sb);
            w->parent = idx;
            w->succ_of_state = 0;
            model_expand(&e->p, st, worker_emit, w);
            if (w->succ_of_state == 0) w->deadlocks++;
        }
    }
    return NULL;
}

static int setup_params(P *p, const kmo_config *c) {
    memset(p, 0, sizeof *p);
    p->model = c->model;
    p->N = c->N; p->L = c->L; p->R = c->R; p->E = c->E; p->K = c->K; p->MaxId = c->MaxId;
    p->EP1 = c->E + 1;
    if (c->model == M_IDSEQ) {
        p->sb = 8; p->nact = 1;
        return c->MaxId >= 0;
    }
    if (c->N < 1 || c->N > KMO_MAXN || c->L < 1) return 0;
    if (c->model == M_ASYNC_ISR) { /* L = MaxOffset, E = MaxVersion */
        if (c->N > 6 || c->E < 0 || c->E > 14 || c->L > 250) return 0;
        p->a_rb = ((1 << c->N) + 7) / 8;
        p->a_req = 6 + c->N;
        p->a_upd = p->a_req + (c->E + 1) * p->a_rb;
        p->sb = p->a_upd + c->E + 1;
        p->nact = 7;
        return p->sb <= KMO_MAXSB;
    }
    if (c->model == M_FRL) {
        p->sb = c->N * (1 + c->L); p->nact = 3;
        return p->sb <= KMO_MAXSB && c->K >= 1 && c->K < 255;
    }
    if (c->R < 1 || c->E < 0 || c->R * (c->E + 1) > 254) return 0;
    p->rstride = 5 + c->L;
    p->goff = c->N * p->rstride;
    p->sb = p->goff + 5 + 2 * (c->E + 1);
    p->nact = c->model == M_KIP320_FIRST ? 10 : 9;
    return p->sb <= KMO_MAXSB && c->model >= M_TRUNC_HW && c->model <= M_KIP320_FIRST;
}

void *kmo_run(const kmo_config *cfg, kmo_result *res) {
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    memset(res, 0, sizeof *res);
    if (getenv("KMO_FP_ONLY") && atoi(getenv("KMO_FP_ONLY"))) g_fp_only = 1;
    if (getenv("KMO_TABLE_LOG2")) g_table_log2 = atoi(getenv("KMO_TABLE_LOG2"));
    Engine *e = calloc(1, sizeof *e);
    e->cfg = *cfg;
    if (!setup_params(&e->p, cfg)) {
        res->verdict = V_ERROR;
        free(e);
        return NULL;
    }
    e->rs = e->p.sb + 5;
    pthread_mutex_init(&e->chunk_mu, NULL);
    int T = cfg->threads > 0 ? cfg->threads : 1;
    engine_grow(e, 1 << 16, 1);

    uint8_t init[KMO_MAXSB];
    if (e->p.model == M_IDSEQ || e->p.model == M_FRL)
        memset(init, 0, e->p.sb); /* IdSequence.tla:37 / FiniteReplicatedLog.tla:97 */
    else if (e->p.model == M_ASYNC_ISR)
        async_init(&e->p, init);
    else
        kafka_init(&e->p, init);
    engine_insert(e, init, 0xFFFFFFFFu, 255, NULL, NULL);
    res->generated = 1;
    res->viol_inv = -1;

    uint64_t lo = 0, hi = 1;
    double t_inv = 0, t_grow = 0, t_exp = 0; /* KMO_TIMING=1: where a run's time goes */
    struct timespec ta, tb;
#define KMO_TICK(acc)                                                             \
    do {                                                                          \
        clock_gettime(CLOCK_MONOTONIC, &tb);                                      \
        acc += (tb.tv_sec - ta.tv_sec) + 1e-9 * (tb.tv_nsec - ta.tv_nsec);        \
        ta = tb;                                                                  \
    } while (0)
    clock_gettime(CLOCK_MONOTONIC, &ta);
    Worker *ws = calloc(T, sizeof(Worker));
    pthread_t *th = calloc(T, sizeof(pthread_t));
    for (;;) {
        /* invariant check on the new level [lo,hi) */
        res->levels[res->nlevels < KMO_MAX_LEVELS ? res->nlevels : KMO_MAX_LEVELS - 1] = hi - lo;
        res->nlevels++;
        res->depth = res->nlevels;
        if (cfg->inv_mask && res->viol_inv < 0) {
            /* slices are ordered, so the first violator of the lowest slice is the smallest index: the result is
             * the serial loop's, whatever T is */
            uint64_t cnt[4] = {0, 0, 0, 0}, first[4] = {0, 0, 0, 0};
            InvJob *ij = calloc(256, sizeof(InvJob));
            for (int t = 0; t < 256; t++) { ij[t].e = e; ij[t].inv_mask = cfg->inv_mask; }
            run_slices(T, lo, hi, invariant_slice, ij, sizeof(InvJob));
            for (int t = 0; t < 256; t++)
                for (int inv = 0; inv < 4; inv++)
                    if (ij[t].cnt[inv]) {
                        if (cnt[inv] == 0) first[inv] = ij[t].first[inv];
                        cnt[inv] += ij[t].cnt[inv];
                    }
            free(ij);
            KMO_TICK(t_inv);
            for (int inv = 0; inv < 4; inv++)
                if (cnt[inv]) {
                    res->viol_inv = inv;
                    res->viol_depth = res->nlevels;
                    res->viol_state_idx = first[inv];
                    memcpy(res->viol_count, cnt, sizeof cnt);
                    break;
                }
            if (res->viol_inv >= 0 && cfg->stop_on_violation) {
                res->verdict = V_INVARIANT;
                break;
            }
        }
        if (cfg->max_states && hi > cfg->max_states) {
            res->verdict = V_LIMIT;
            break;
        }
        /* keep the table sparse enough to absorb this level's growth */
        clock_gettime(CLOCK_MONOTONIC, &ta);
        engine_grow(e, 4 * (hi + 6 * (hi - lo)), T);
        KMO_TICK(t_grow);
        e->lo = lo; e->hi = hi;
        atomic_store(&e->cursor, lo);
        for (int t = 0; t < T; t++) {
            memset(&ws[t], 0, sizeof(Worker));
            ws[t].e = e;
            if (T == 1)
                worker_main(&ws[t]);
            else
                pthread_create(&th[t], NULL, worker_main, &ws[t]);
        }
        uint64_t dl = 0;
        for (int t = 0; t < T; t++)
            if (T > 1) pthread_join(th[t], NULL);
        compact_level(e, ws, T, hi);
        KMO_TICK(t_exp);
        if (cfg->inv_mask && res->viol_inv < 0) { /* violating successors outside the constraint: depth nlevels+1 */
            uint64_t cnt[4] = {0, 0, 0, 0};
            int best[4] = {-1, -1, -1, -1};
            for (int t = 0; t < T; t++)
                for (int inv = 0; inv < 4; inv++)
                    if (ws[t].out_cnt[inv]) {
                        cnt[inv] += ws[t].out_cnt[inv];
                        if (best[inv] < 0 || memcmp(ws[t].out_state[inv], ws[best[inv]].out_state[inv], e->p.sb) < 0)
                            best[inv] = t;
                    }
            for (int inv = 0; inv < 4; inv++)
                if (cnt[inv]) {
                    const Worker *w = &ws[best[inv]];
                    res->viol_inv = inv;
                    res->viol_depth = res->nlevels + 1;
                    res->viol_state_idx = w->out_parent[inv];
                    res->viol_outside = 1;
                    res->viol_action = w->out_action[inv];
                    memcpy(res->viol_state, w->out_state[inv], e->p.sb);
                    memcpy(res->viol_count, cnt, sizeof cnt);
                    break;
                }
            if (res->viol_inv >= 0 && cfg->stop_on_violation) {
                /* like the level-start check: the expansion that met the violation is not counted */
                res->verdict = V_INVARIANT;
                atomic_store(&e->nstates, hi);
                break;
            }
        }
        for (int t = 0; t < T; t++) {
            res->generated += ws[t].generated;
            dl += ws[t].deadlocks;
            for (int a = 0; a < KMO_MAX_ACTIONS; a++) res->action_generated[a] += ws[t].action_generated[a];
        }
        res->deadlock_states += dl;
        if (e->table_overflow) {
            res->verdict = V_ERROR;
            break;
        }
        if (cfg->check_deadlock && dl) {
            res->verdict = V_DEADLOCK;
            break;
        }
        if (g_fp_only) /* the states of the level just expanded are never read again */
            for (uint64_t c = lo >> CHUNK_BITS; c < (hi >> CHUNK_BITS); c++)
                if (e->chunks[c]) {
                    free(e->chunks[c]);
                    e->chunks[c] = NULL;
                }
        lo = hi;
        hi = atomic_load(&e->nstates);
        if (hi == lo) break;
    }
    if (res->viol_inv >= 0 && res->verdict == V_OK) res->verdict = V_INVARIANT;
    res->distinct = atomic_load(&e->nstates);
    free(ws);
    free(th);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    res->seconds = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    if (getenv("KMO_TIMING"))
        fprintf(stderr, "kmc_oracle: %d threads, %.2f s: invariants %.2f, table growth %.2f, expansion %.2f\n", T,
                res->seconds, t_inv, t_grow, t_exp);
    return e;
}

int kmo_state_bytes(void *h) { return ((Engine *)h)->p.sb; }
uint64_t kmo_num_states(void *h) { return atomic_load(&((Engine *)h)->nstates); }
/* states are stored in discovery order: level k occupies a contiguous index range */
void kmo_get_states(void *h, uint64_t first, uint64_t count, uint8_t *out) {
    Engine *e = h;
    for (uint64_t i = 0; i < count; i++) memcpy(out + i * e->p.sb, rec_ptr(e, first + i), e->p.sb);
}
int64_t kmo_parent(void *h, uint64_t idx) {
    Engine *e = h;
    uint32_t par;
    memcpy(&par, rec_ptr(e, idx) + e->p.sb, 4);
    return par == 0xFFFFFFFFu ? -1 : (int64_t)par;
}
int kmo_action(void *h, uint64_t idx) {
    Engine *e = h;
    return rec_ptr(e, idx)[e->p.sb + 4];
}
/* successors of one serialized state, for differential tests: returns count, writes
 * (action byte + state) records into out (cap records). */
typedef struct { uint8_t *out; int cap, n, sb; } collect_ctx;
static void collect_emit(void *ctx, int action, const uint8_t *succ) {
    collect_ctx *c = ctx;
    if (c->n < c->cap) {
        c->out[(size_t)c->n * (c->sb + 1)] = (uint8_t)action;
        memcpy(c->out + (size_t)c->n * (c->sb + 1) + 1, succ, c->sb);
    }
    c->n++;
}
int kmo_successors(const kmo_config *cfg, const uint8_t *state, uint8_t *out, int cap) {
    P p;
    if (!setup_params(&p, cfg)) return -1;
    collect_ctx c = {out, cap, 0, p.sb};
    model_expand(&p, state, collect_emit, &c);
    return c.n;
}
int kmo_check_invariant(const kmo_config *cfg, int inv, const uint8_t *state) {
    P p;
    if (!setup_params(&p, cfg)) return -1;
    return model_invariant(&p, inv, state);
}
void kmo_free(void *h) {
    Engine *e = h;
    if (!e) return;
    for (uint32_t c = 0; c < MAX_CHUNKS; c++) free(e->chunks[c]);
    if (e->table) munmap((void *)e->table, e->cap * sizeof(uint64_t));
    free(e);
}

#ifdef KMO_MAIN
static const char *MODEL_NAMES[] = {"IdSequence", "FiniteReplicatedLog", "KafkaTruncateToHighWatermark",
                                    "Kip101",     "Kip279",              "Kip320", "Kip320FirstTry",
                                    "AsyncIsr"};
int main(int argc, char **argv) {
    kmo_config c = {.model = M_KIP320, .N = 3, .L = 2, .R = 2, .E = 1, .K = 2, .MaxId = 10,
                    .inv_mask = 1, .check_deadlock = 0, .stop_on_violation = 1, .threads = 1, .max_states = 0};
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--model") && i + 1 < argc) {
            c.model = -1;
            for (int m = 0; m < 8; m++)
                if (!strcmp(argv[i + 1], MODEL_NAMES[m])) c.model = m;
            i++;
        } else if (!strcmp(argv[i], "--N")) c.N = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--L")) c.L = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--R")) c.R = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--E")) c.E = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--K")) c.K = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--MaxId")) c.MaxId = atoll(argv[++i]);
        else if (!strcmp(argv[i], "--inv")) c.inv_mask = (uint32_t)strtoul(argv[++i], NULL, 0);
        else if (!strcmp(argv[i], "--threads")) c.threads = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--continue")) c.stop_on_violation = 0;
        else if (!strcmp(argv[i], "--deadlock")) c.check_deadlock = 1;
        else if (!strcmp(argv[i], "--max-states")) c.max_states = strtoull(argv[++i], NULL, 0);
        else if (!strcmp(argv[i], "--fp-only")) g_fp_only = 1;
        else if (!strcmp(argv[i], "--table-log2")) g_table_log2 = atoi(argv[++i]);
        else {
            fprintf(stderr, "unknown arg %s\n", argv[i]);
            return 2;
        }
    }
    kmo_result r;
    void *h = kmo_run(&c, &r);
    printf("{\"model\": \"%s\", \"N\": %d, \"L\": %d, \"R\": %d, \"E\": %d, \"distinct\": %llu, \"generated\": %llu, "
           "\"depth\": %llu, \"verdict\": %d, \"viol_inv\": %d, \"viol_depth\": %llu, \"viol_count\": [%llu,%llu,%llu,%llu], "
           "\"deadlock_states\": %llu, \"threads\": %d, \"seconds\": %.3f, \"fp_only\": %d, \"levels\": [",
           c.model >= 0 ? MODEL_NAMES[c.model] : "?", c.N, c.L, c.R, c.E, (unsigned long long)r.distinct,
           (unsigned long long)r.generated, (unsigned long long)r.depth, r.verdict, r.viol_inv,
           (unsigned long long)r.viol_depth, (unsigned long long)r.viol_count[0], (unsigned long long)r.viol_count[1],
           (unsigned long long)r.viol_count[2], (unsigned long long)r.viol_count[3],
           (unsigned long long)r.deadlock_states, c.threads, r.seconds, g_fp_only);
    for (uint64_t i = 0; i < r.nlevels && i < KMO_MAX_LEVELS; i++)
        printf("%s%llu", i ? "," : "", (unsigned long long)r.levels[i]);
    printf("], \"action_generated\": [");
    for (int a = 0; a < 10; a++) printf("%s%llu", a ? "," : "", (unsigned long long)r.action_generated[a]);
    printf("]}\n");
    kmo_free(h);
    return 0;
}
#endif
