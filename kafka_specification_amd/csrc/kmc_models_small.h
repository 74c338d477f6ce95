// kmc_models_small.h — IdSequence, FiniteReplicatedLog and AsyncIsr lowered onto the packed state vector.
// Part of the device source (kmc_device.h lists the parts; the host engine hands their concatenation to hiprtc).
#pragma once
#include "kmc_common.h"
// ========================================================================================
// IdSequence.tla standalone
// ========================================================================================
template <long long MAXID> struct KmcIdSequence {
    static constexpr int W = 1, NKINDS = 1, NINST = 1;
    static constexpr bool HAS_EXTRA = false, HAS_CONSTRAINT = false, KIND_MAJOR = false, RUNTIME_GUARDS = false, FULL_LEAVES = false;
    struct Pre { u64 nextId; };
    static KMC_DEV void init(u64* w) { w[0] = 0; }  // IdSequence.tla:37
    static KMC_DEV Pre extract(const u64* s) { return Pre{s[0]}; }
    static KMC_DEV void launder(Pre& p) { kmc_launder(p.nextId); }
    template <int I> static KMC_DEV u32 inst(const Pre& p, const u64* s, u64* t, int& kind, u32& extra) {
        // Next == \E id \in IdSet : NextId(id)   (IdSequence.tla:39, NextId :30-33)
        kind = 0; extra = 0;
        t[0] = p.nextId + 1;
        return (long long)p.nextId <= MAXID ? 1u : 0u;
    }
    static KMC_DEV u32 violated_pre(const Pre& p, u32 inv_mask) {  // TypeOk, IdSequence.tla:43
        return (inv_mask & 1u) && !((long long)p.nextId <= MAXID + 1) ? 1u : 0u;
    }
    static KMC_DEV u32 violated(const u64* t, u32 inv_mask) { return violated_pre(extract(t), inv_mask); }
    static KMC_DEV u32 violated_stream(const u64* t, u32 inv_mask) { return violated(t, inv_mask); }   // (k_inv's entry: KmcKafka)
};

// ========================================================================================
// FiniteReplicatedLog.tla standalone
// ========================================================================================
template <int N, int L, int K> struct KmcFiniteReplicatedLog {
    static constexpr KmcLayout Y = kmc_make_layout(KMC_MODEL_FINITE_REPLICATED_LOG, N, L, 0, 0, K);
    static_assert(Y.valid, "FiniteReplicatedLog parameters cannot be packed");
    static constexpr int W = Y.W, NKINDS = 3;
    static constexpr bool HAS_EXTRA = false, HAS_CONSTRAINT = false, KIND_MAJOR = false, RUNTIME_GUARDS = false, FULL_LEAVES = false;
    static constexpr int C_APPEND = N * K, C_TRUNC = N * L, C_REPL = N * (N - 1);
    static constexpr int NINST = C_APPEND + C_TRUNC + C_REPL;
    static constexpr u64 MR = (1ull << Y.BR) - 1;
    struct Pre { u32 end[N]; u64 logv[N]; };

    static KMC_DEV void init(u64* w) {  // FiniteReplicatedLog.tla:97
        for (int k = 0; k < W; ++k) w[k] = 0;
    }
    static KMC_DEV Pre extract(const u64* s) {
        Pre p;
        kmc_static_for<0, N>([&](auto R) {
            constexpr int r = decltype(R)::value;
            p.end[r] = (u32)kmc_getbits(s, Y.end_off[r], Y.BO);
            p.logv[r] = kmc_getbits(s, Y.log_off[r], Y.BR * L);
        });
        return p;
    }
    static KMC_DEV void launder(Pre& p) {
        for (int r = 0; r < N; ++r) { kmc_launder(p.end[r]); kmc_launder(p.logv[r]); }
    }
    template <int I> static KMC_DEV u32 inst(const Pre& p, const u64* s, u64* t, int& kind, u32& extra) {
        extra = 0;
        for (int k = 0; k < W; ++k) t[k] = s[k];
        if constexpr (I < C_APPEND) {
            // \E record, offset : Append(replica, record, offset)   (:116, :99-103)
            constexpr int r = I / K, rec = I % K + 1;
            kind = 0;
            const u32 end = p.end[r];
            kmc_setbits(t, Y.log_off[r], Y.BR * L, p.logv[r] | ((u64)rec << (end * Y.BR)));
            kmc_setbits(t, Y.end_off[r], Y.BO, end + 1);
            return end < (u32)L ? 1u : 0u;
        } else if constexpr (I < C_APPEND + C_TRUNC) {
            // \E offset \in Offsets : TruncateTo(replica, offset)   (:117, :105-109)
            constexpr int J = I - C_APPEND, r = J / L, o = J % L;
            kind = 1;
            constexpr u64 keep = (o * Y.BR >= 64) ? ~0ull : ((1ull << (o * Y.BR)) - 1ull);
            kmc_setbits(t, Y.log_off[r], Y.BR * L, p.logv[r] & keep);
            kmc_setbits(t, Y.end_off[r], Y.BO, o);
            return (u32)o <= p.end[r] ? 1u : 0u;
        } else {
            // \E other # replica : ReplicateTo(replica, other)   (:118, :111-113)
            constexpr int J = I - C_APPEND - C_TRUNC, from = J / (N - 1), q = J % (N - 1), to = q + (q >= from);
            kind = 2;
            const u32 eto = p.end[to];
            const u64 rec = (p.logv[from] >> (eto * Y.BR)) & MR;
            kmc_setbits(t, Y.log_off[to], Y.BR * L, p.logv[to] | (rec << (eto * Y.BR)));
            kmc_setbits(t, Y.end_off[to], Y.BO, eto + 1);
            return (eto < p.end[from] && eto < (u32)L) ? 1u : 0u;
        }
    }
    static KMC_DEV u32 violated(const u64* t, u32 inv_mask) { return violated_pre(extract(t), inv_mask); }
    static KMC_DEV u32 violated_stream(const u64* t, u32 inv_mask) { return violated(t, inv_mask); }   // (k_inv's entry: KmcKafka)
    static KMC_DEV u32 violated_pre(const Pre& p, u32 inv_mask) {  // TypeOk, :90-95
        if (!(inv_mask & 1u)) return 0;
        bool ok = true;
        kmc_static_for<0, N>([&](auto R) {
            constexpr int r = decltype(R)::value;
            ok = ok && p.end[r] <= (u32)L;
            kmc_static_for<0, L>([&](auto O) {
                constexpr int o = decltype(O)::value;
                const u32 c = (u32)((p.logv[r] >> (o * Y.BR)) & MR);
                ok = ok && c <= (u32)K && ((u32)o < p.end[r] ? c != 0 : c == 0);
            });
        });
        return ok ? 0u : 1u;
    }
};

// ========================================================================================
// AsyncIsr.tla standalone, under the state constraint of models/MCAsyncIsr.tla (layout: kmc_layout.h)
// ========================================================================================
template <int N, int MO, int V> struct KmcAsyncIsr {
    static constexpr KmcLayout Y = kmc_make_layout(KMC_MODEL_ASYNC_ISR, N, MO, 0, V, 0);
    static_assert(Y.valid, "AsyncIsr parameters cannot be packed (need N <= 6, MaxVersion <= 7)");
    static constexpr int W = Y.W, NKINDS = 7;
    static constexpr bool HAS_EXTRA = false, HAS_CONSTRAINT = true, KIND_MAJOR = false, RUNTIME_GUARDS = false, FULL_LEAVES = false;
    static constexpr int NS = 1 << N;  // isr masks = request bits per version
    // Next (AsyncIsr.tla:152-159) flattened into instances, one per binding of each disjunct's \E
    static constexpr int B0 = 0;             // ControllerShrinkIsr        (replica # Leader)
    static constexpr int B1 = B0 + (N - 1);  // ControllerHandleRequest    (message.isr; message.version = controller's)
    static constexpr int B2 = B1 + NS;       // LeaderRequestShrinkIsr     (replica # Leader)
    static constexpr int B3 = B2 + (N - 1);  // LeaderRequestExpandIsr     (replica)
    static constexpr int B4 = B3 + N;        // LeaderWrite
    static constexpr int B5 = B4 + 1;        // LeaderHandleUpdate         (update.version 1..MaxVersion)
    static constexpr int B6 = B5 + V;        // FollowerReplicate          (replica # Leader)
    static constexpr int NINST = B6 + (N - 1);
    static constexpr u32 FULL = (1u << N) - 1;

    struct Pre {
        u32 cisr, cver, lisr, lver, pisr, pver1, hw;
        u32 off[N];
        u64 reqcur;  // the requests whose version is controllerState.version, as a bitset over isr masks
    };

    static KMC_DEV void init(u64* w) {  // Init, :137-150
        for (int k = 0; k < W; ++k) w[k] = 0;
        kmc_setbits(w, Y.a_cisr, N, FULL);
        kmc_setbits(w, Y.a_lisr, N, FULL);  // version 0, pendingIsr {}, pendingVersion Nil (-> 0), offsets 0, no messages
    }
    static KMC_DEV Pre extract(const u64* s) {
        Pre p;
        p.cisr = (u32)kmc_getbits(s, Y.a_cisr, N);
        p.cver = (u32)kmc_getbits(s, Y.a_cver, Y.BV);
        p.lisr = (u32)kmc_getbits(s, Y.a_lisr, N);
        p.lver = (u32)kmc_getbits(s, Y.a_lver, Y.BV);
        p.pisr = (u32)kmc_getbits(s, Y.a_pisr, N);
        p.pver1 = (u32)kmc_getbits(s, Y.a_pver, Y.BV);
        // HighWatermark, :58-60 (Leader never leaves leaderState.isr, so the set is never empty)
        const u32 potential = p.lisr | p.pisr;
        p.hw = ~0u;
        kmc_static_for<0, N>([&](auto R) {
            constexpr int r = decltype(R)::value;
            p.off[r] = (u32)kmc_getbits(s, Y.a_off[r], Y.BF);
            if (potential >> r & 1u) p.hw = kmc_min(p.hw, p.off[r]);
        });
        p.reqcur = p.cver <= (u32)V ? kmc_getbits(s, Y.a_req + (int)p.cver * NS, NS) : 0ull;
        return p;
    }
    static KMC_DEV void launder(Pre& p) {
        kmc_launder(p.cisr); kmc_launder(p.cver); kmc_launder(p.lisr); kmc_launder(p.lver);
        kmc_launder(p.pisr); kmc_launder(p.pver1); kmc_launder(p.hw); kmc_launder(p.reqcur);
        for (int r = 0; r < N; ++r) kmc_launder(p.off[r]);
    }
    // the state constraint (NOT in the reference): offsets[Leader] <= MaxOffset /\ controllerState.version <= MaxVersion
    static KMC_DEV bool in_model(const u64* t) {
        return (u32)kmc_getbits(t, Y.a_off[0], Y.BF) <= (u32)MO && (u32)kmc_getbits(t, Y.a_cver, Y.BV) <= (u32)V;
    }
    static KMC_DEV void controller_write(u64* t, const Pre& p, u32 isr) {  // ControllerWriteIsr :68-70 + updates' (:78, :85)
        kmc_setbits(t, Y.a_cisr, N, isr);
        kmc_setbits(t, Y.a_cver, Y.BV, p.cver + 1);
        kmc_setbits(t, Y.a_upd + (int)kmc_min(p.cver, (u32)V) * N, N, isr);  // the update of version cver+1
    }
    static KMC_DEV void leader_request(u64* t, const Pre& p, u32 isr) {  // :92-99 / :107-114
        kmc_setbits(t, Y.a_req + (int)kmc_min(p.lver, (u32)V) * NS + (int)isr, 1, 1);
        kmc_setbits(t, Y.a_pisr, N, p.pisr | isr);
        kmc_setbits(t, Y.a_pver, Y.BV, p.lver + 1);
    }
    // Guards carry `version <= MaxVersion` / `offset <= MaxOffset`: states beyond the constraint are
    // never expanded by the search, and this keeps a caller-supplied one from writing outside its fields.
    template <int I> static KMC_DEV u32 inst(const Pre& p, const u64* s, u64* t, int& kind, u32& extra) {
        extra = 0;
        for (int k = 0; k < W; ++k) t[k] = s[k];
        if constexpr (I < B1) {  // ControllerShrinkIsr :72-79
            constexpr int r = I - B0 + 1;
            kind = 0;
            controller_write(t, p, p.cisr & ~(1u << r));
            return (kmc_bit(p.cisr, r) && p.cver <= (u32)V) ? 1u : 0u;
        } else if constexpr (I < B2) {  // ControllerHandleRequest :81-86
            constexpr int m = I - B1;
            kind = 1;
            controller_write(t, p, (u32)m);
            return kmc_bit64(p.reqcur, m);
        } else if constexpr (I < B3) {  // LeaderRequestShrinkIsr :88-100
            constexpr int r = I - B2 + 1;
            kind = 2;
            leader_request(t, p, p.lisr & ~(1u << r));
            return (kmc_bit(p.lisr, r) && p.lver <= (u32)V) ? 1u : 0u;
        } else if constexpr (I < B4) {  // LeaderRequestExpandIsr :102-115
            constexpr int r = I - B3;
            kind = 3;
            leader_request(t, p, p.lisr | (1u << r));
            return (!kmc_bit(p.lisr, r) && p.off[r] >= p.hw && p.lver <= (u32)V) ? 1u : 0u;
        } else if constexpr (I < B5) {  // LeaderWrite :117-119
            kind = 4;
            kmc_setbits(t, Y.a_off[0], Y.BF, p.off[0] + 1);
            return p.off[0] <= (u32)MO ? 1u : 0u;
        } else if constexpr (I < B6) {  // LeaderHandleUpdate :121-129
            constexpr int v = I - B5 + 1;
            kind = 5;
            kmc_setbits(t, Y.a_lisr, N, kmc_getbits(s, Y.a_upd + (v - 1) * N, N));
            kmc_setbits(t, Y.a_lver, Y.BV, v);
            kmc_setbits(t, Y.a_pisr, N, 0);
            kmc_setbits(t, Y.a_pver, Y.BV, 0);
            return ((u32)v > p.lver && (u32)v <= p.cver) ? 1u : 0u;
        } else {  // FollowerReplicate :131-135
            constexpr int r = I - B6 + 1;
            kind = 6;
            kmc_setbits(t, Y.a_off[r], Y.BF, p.off[r] + 1);
            return p.off[r] < p.off[0] ? 1u : 0u;
        }
    }
    // bit 0 TypeOk :62-66 — every conjunct is a tautology of the representation except
    //   pendingVersion \in Nat (:44), false while pendingVersion = Nil (:38), e.g. in Init (:146);
    // bit 1 ValidHighWatermark :161-162;
    // bit 2 LeaderOffsetInRange (models/MCAsyncIsr.tla, not in the reference): offsets[Leader] \in Offsets (:37)
    static KMC_DEV u32 violated_pre(const Pre& p, u32 inv_mask) {
        u32 bad = 0;
        if ((inv_mask & 1u) && p.pver1 == 0) bad |= 1u;
        if (inv_mask & 2u) {
            bool ok = true;
            kmc_static_for<0, N>([&](auto R) {
                constexpr int r = decltype(R)::value;
                ok = ok && (!(p.cisr >> r & 1u) || p.off[r] >= p.hw);
            });
            if (!ok) bad |= 2u;
        }
        if ((inv_mask & 4u) && p.off[0] > (u32)MO) bad |= 4u;
        return bad;
    }
    static KMC_DEV u32 violated(const u64* t, u32 inv_mask) { return violated_pre(extract(t), inv_mask); }
    static KMC_DEV u32 violated_stream(const u64* t, u32 inv_mask) { return violated(t, inv_mask); }   // (k_inv's entry: KmcKafka)
};

