// kmc_kafka.h — KafkaReplication.tla and the five modules that give it a Next: guards (inst<I>, guard<K>), effects (apply<K>), invariants.
// Part of the device source (kmc_device.h lists the parts; the host engine hands their concatenation to hiprtc).
#pragma once
#include "kmc_common.h"
// ========================================================================================
// KafkaReplication.tla and the five modules that give it a Next
// ========================================================================================
template <int MODEL, int N, int L, int R, int E, int LM = KMC_LAYOUT_AUTO> struct KmcKafka {
    static constexpr KmcLayout Y = kmc_make_layout(MODEL, N, L, R, E, 0, LM);
    static_assert(Y.valid, "Kafka model parameters cannot be packed (need L*bits(record) <= 64, N <= 8, E <= 7)");
    static constexpr int W = Y.W;
    static constexpr bool FIRST = MODEL == KMC_MODEL_KIP320_FIRST_TRY;
    static constexpr bool K320 = MODEL == KMC_MODEL_KIP320;
    static constexpr int NKINDS = FIRST ? 10 : 9;
    // Bindings that generate the same successor twice: TLC's next-state enumeration continues from EVERY disjunct that
    // holds [TLC-recall: Tool.getNextStates, OPCODE_lor] and counts each result as "generated".  Kip279.tla:47-51
    // (an empty follower satisfies both disjuncts of BecomeFollowerTruncateKip279) and Kip320.tla:82-83 (both reasons
    // to shrink the ISR can hold at once; found by Oracle-R, which executes the module text).  One kind per model.
    static constexpr bool HAS_EXTRA = MODEL == KMC_MODEL_KIP279 || MODEL == KMC_MODEL_KIP320;
    static constexpr int EXTRA_KIND = MODEL == KMC_MODEL_KIP279 ? 7 : 4;
    static constexpr bool HAS_CONSTRAINT = false;
    static constexpr int NP = N * (N - 1);  // ordered pairs of distinct replicas
    // action instances, in the order of the Next disjuncts (the index of the disjunct is
    // the "kind"): KafkaTruncateToHighWatermark.tla:33-42, Kip101.tla:49-58, Kip279.tla:53-62,
    // Kip320.tla:150-159, Kip320FirstTry.tla:159-169
    static constexpr int B0 = 0;                    // ControllerElectLeader          (newLeader)
    static constexpr int B1 = B0 + N;               // ControllerShrinkIsr            (replica)
    static constexpr int B2 = B1 + N;               // BecomeLeader                   (request epoch, leader)
    static constexpr int B3 = B2 + (E + 1) * N;     // Leader*ExpandIsr*              (leader, replica) incl. replica = leader
    static constexpr int B4 = B3 + N * N;           // Leader*ShrinkIsr*              (leader, replica # leader)
    static constexpr int B5 = B4 + NP;              // LeaderWrite                    (replica)
    static constexpr int B6 = B5 + N;               // *LeaderIncHighWatermark        (leader)
    static constexpr int B7 = B6 + N;               // BecomeFollower*                (leader, replica # leader, request epoch)
    static constexpr int B8 = B7 + NP * (E + 1);    // FollowerReplicate / *Fetch     (leader, follower # leader)
    static constexpr int B9 = B8 + NP;              // FollowerTruncate (Kip320FirstTry only)
    static constexpr int NINST = B9 + (FIRST ? NP : 0);

    using LogT = typename KmcLogWord<(Y.BR * L <= 32)>::type;
    static constexpr LogT MR = (LogT)((1ull << Y.BR) - 1);    // one record
    static constexpr u32 MEr = (1u << Y.BEr) - 1;    // record.epoch
    static constexpr u32 FULL = (1u << N) - 1;

    // A lazy view over the packed state: fields are re-extracted on demand (one or two VALU ops
    // with compile-time offsets) instead of living in ~35 registers across the whole instance
    // loop (the instance-major kernel: 80 VGPRs = 6 waves/SIMD; with 95 VGPRs and 5 waves it was 1.8 ms slower).
    struct Pre {
        const u64* w;  // the packed state words (the caller's registers)
        KMC_DEV u32 end(int r) const { return (u32)kmc_getbits(w, Y.end_off[r], Y.BO); }
        KMC_DEV u32 hw(int r) const { return (u32)kmc_getbits(w, Y.hw_off[r], Y.BO); }
        KMC_DEV u32 ep1(int r) const { return (u32)kmc_getbits(w, Y.ep_off[r], Y.BE); }
        KMC_DEV u32 ldr1(int r) const { return (u32)kmc_getbits(w, Y.ldr_off[r], Y.BL); }
        KMC_DEV u32 isr(int r) const { return (u32)kmc_getbits(w, Y.isr_off[r], Y.BI); }
        KMC_DEV LogT logv(int r) const { return (LogT)kmc_getbits(w, Y.log_off[r], Y.BR * L); }
        KMC_DEV u32 nextRec() const { return (u32)kmc_getbits(w, Y.nextrec_off, Y.BNR); }
        KMC_DEV u32 nextEp() const { return (u32)kmc_getbits(w, Y.nextep_off, Y.BE); }
        KMC_DEV u32 qep1() const { return (u32)kmc_getbits(w, Y.qep_off, Y.BE); }
        KMC_DEV u32 qldr1() const { return (u32)kmc_getbits(w, Y.qldr_off, Y.BL); }
        KMC_DEV u32 qisr() const { return (u32)kmc_getbits(w, Y.qisr_off, Y.BI); }
        KMC_DEV u32 rldr1(int e) const { return (u32)kmc_getbits(w, Y.reqldr_off[e], Y.BL); }
        KMC_DEV u32 risr(int e) const { return (u32)kmc_getbits(w, Y.reqisr_off[e], Y.BI); }
        // shared sub-predicates of the guards, as opaque integers (see kmc_and)
        u32 one;    // 1
        u32 epok;   // nextLeaderEpoch <= MaxLeaderEpoch            (LeaderEpochSeq!NextId, IdSequence.tla:31)
        u32 pm;     // bit l: ReplicaPresumesLeadership(l)          (KafkaReplication.tla:126)
        u32 tm;     // bit l: IsTrueLeader(l)                       (:128-131)
        u32 hm;     // bit l: HasHighWatermarkReachedCurrentEpoch(l) (Kip320.tla:87-92)
        u64 fm;     // bit l*N+f: IsFollowingLeaderEpoch(l, f)      (Kip320.tla:39-42)
    };

    static KMC_DEV void init(u64* w) {  // Init, KafkaReplication.tla:109-120
        for (int k = 0; k < W; ++k) w[k] = 0;
        kmc_setbits(w, Y.qisr_off, Y.BI, FULL);  // quorumState.isr = Replicas (:119)
    }

    static KMC_DEV Pre extract(const u64* s) {
        Pre p;
        p.w = s;
        p.one = 1u;
        p.epok = p.nextEp() <= (u32)E ? 1u : 0u;
        p.pm = 0; p.tm = 0; p.hm = 0; p.fm = 0;
        kmc_static_for<0, N>([&](auto LL) {
            constexpr int l = decltype(LL)::value;
            const u32 pres = presumes<l>(p) ? 1u : 0u;
            p.pm |= pres << l;
            p.tm |= (is_true_leader<l>(p) ? 1u : 0u) << l;
            if constexpr (K320 || FIRST) p.hm |= (hw_reached_epoch<l>(p) ? 1u : 0u) << l;
            if constexpr (K320)
                kmc_static_for<0, N>([&](auto FF) {
                    constexpr int f = decltype(FF)::value;
                    p.fm |= (u64)(following_epoch<l, f>(p) ? 1u : 0u) << (l * N + f);
                });
        });
        kmc_launder(p.one); kmc_launder(p.epok); kmc_launder(p.pm); kmc_launder(p.tm); kmc_launder(p.hm);
        kmc_launder(p.fm);
        return p;
    }

    static KMC_DEV void launder(Pre& p) {  // (the state words themselves are laundered by the caller)
        kmc_launder(p.one); kmc_launder(p.epok); kmc_launder(p.pm); kmc_launder(p.tm); kmc_launder(p.hm);
        kmc_launder(p.fm);
    }

    // ---- log helpers (FiniteReplicatedLog.tla as instantiated at KafkaReplication.tla:84) ----
    static KMC_DEV u32 rec_at(LogT logv, u32 o) { return (u32)((logv >> (o * Y.BR)) & MR); }
    static KMC_DEV u32 rec_epoch(u32 rec) { return rec & MEr; }
    static KMC_DEV LogT keep_below(u32 off) {  // mask of the slots < off
        const u32 sh = off * Y.BR;
        return sh >= 8 * sizeof(LogT) ? (LogT)~(LogT)0 : (LogT)((((LogT)1) << sh) - (LogT)1);
    }
    // TruncateTo(replica, off) for off <= end (FiniteReplicatedLog.tla:105-109)
    template <int r> static KMC_DEV void truncate(u64* t, const Pre& p, u32 off) {
        kmc_setbits(t, Y.log_off[r], Y.BR * L, p.logv(r) & keep_below(off));
        kmc_setbits(t, Y.end_off[r], Y.BO, off);
    }

    // ---- predicates (KafkaReplication.tla:126-131) ----
    template <int r> static KMC_DEV bool presumes(const Pre& p) { return p.ldr1(r) == (u32)(r + 1); }
    template <int l> static KMC_DEV bool is_true_leader(const Pre& p) {
        return p.qldr1() == (u32)(l + 1) && presumes<l>(p) && p.ep1(l) == p.qep1();
    }

    // ControllerUpdateIsr(newLeader, newIsr) (:138-145); the guard nextLeaderEpoch <= E is the caller's
    static KMC_DEV void controller_update(u64* t, const Pre& p, u32 newLdr1, u32 newIsr) {
        kmc_setbits(t, Y.qep_off, Y.BE, p.nextEp() + 1);
        kmc_setbits(t, Y.qldr_off, Y.BL, newLdr1);
        kmc_setbits(t, Y.qisr_off, Y.BI, newIsr);
        kmc_static_for<0, E + 1>([&](auto EE) {
            constexpr int e = decltype(EE)::value;
            if (p.nextEp() == (u32)e) {
                kmc_setbits(t, Y.reqldr_off[e], Y.BL, newLdr1);
                kmc_setbits(t, Y.reqisr_off[e], Y.BI, newIsr);
            }
        });
        kmc_setbits(t, Y.nextep_off, Y.BE, p.nextEp() + 1);
    }
    // QuorumUpdateLeaderAndIsr(leader, newIsr) effect (:213-217)
    template <int l> static KMC_DEV void quorum_update(u64* t, u32 newIsr) {
        kmc_setbits(t, Y.qisr_off, Y.BI, newIsr);
        kmc_setbits(t, Y.isr_off[l], Y.BI, newIsr);
    }
    // IsFollowerCaughtUp(leader, follower, endOffset) (:219-225): the \E record is satisfied by
    // the leader's own record at endOffset-1 whenever that offset is below its end.
    template <int l, int f> static KMC_DEV bool caught_up(const Pre& p, u32 endOffset) {
        return p.ldr1(f) == (u32)(l + 1) && endOffset <= p.end(l) && endOffset <= p.end(f);
    }
    // Kip320.tla:39-42
    template <int l, int f> static KMC_DEV bool following_epoch(const Pre& p) {
        return presumes<l>(p) && p.ldr1(f) == (u32)(l + 1) && p.ep1(f) == p.ep1(l);
    }
    // HasHighWatermarkReachedCurrentEpoch (Kip320.tla:87-92, Kip320FirstTry.tla:122-127)
    template <int l> static KMC_DEV bool hw_reached_epoch(const Pre& p) {
        return p.hw(l) == p.end(l) ||
               (p.hw(l) < p.end(l) && rec_epoch(rec_at(p.logv(l), p.hw(l))) + 1 == p.ep1(l));
    }
    // IsFollowerCaughtUpToLeaderEpoch (Kip320FirstTry.tla:49-57), on values (`following` = the leader presumes leadership
    // and the follower names it); the <l, f> form is what the instance-major guards use, the value form the run-time ones
    static KMC_DEV bool caught_up_epoch_v(bool following, LogT log_l, LogT log_f, u32 end_l, u32 end_f, u32 endOffset) {
        if (!following) return false;
        if (endOffset == 0) return true;
        const u32 o = endOffset - 1;
        return o < end_l && o < end_f && rec_epoch(rec_at(log_f, o)) == rec_epoch(rec_at(log_l, o));
    }
    template <int l, int f> static KMC_DEV bool caught_up_epoch(const Pre& p, u32 endOffset) {
        return caught_up_epoch_v(presumes<l>(p) && p.ldr1(f) == (u32)(l + 1), p.logv(l), p.logv(f), p.end(l), p.end(f), endOffset);
    }
    // FollowerNeedsTruncation (Kip320FirstTry.tla:64-69)
    static KMC_DEV bool needs_truncation_v(LogT log_f, LogT log_l, u32 end_f, u32 end_l) {
        if (end_f > end_l) return true;
        if (end_f == 0) return false;
        const u32 o = end_f - 1;
        return o < end_l && rec_epoch(rec_at(log_l, o)) != rec_epoch(rec_at(log_f, o));
    }
    template <int f, int l> static KMC_DEV bool needs_truncation(const Pre& p) {
        return needs_truncation_v(p.logv(f), p.logv(l), p.end(f), p.end(l));
    }
    // FirstNonMatchingOffsetFromTail(leader, follower) (Kip279.tla:27-45), on the two logs and end offsets as values
    // (shared by the instance-major effects, where leader and follower are compile-time, and the kind-major ones below)
    static KMC_DEV u32 first_non_matching_v(LogT logl, LogT logf, u32 endl, u32 endf) {
        const LogT x = logl ^ logf;
        const u32 lim = kmc_min(endl, endf);  // leader empty => no match => 0
        u32 best = 0;
        kmc_static_for<0, L>([&](auto O) {
            constexpr int o = decltype(O)::value;
            if ((u32)o < lim && ((x >> (o * Y.BR)) & MR) == 0) best = o + 1;
        });
        return best;
    }
    template <int l, int f> static KMC_DEV u32 first_non_matching(const Pre& p) {
        return first_non_matching_v(p.logv(l), p.logv(f), p.end(l), p.end(f));
    }
    // LookupOffsetForEpoch(leader, follower, epoch) (Kip101.tla:27-39), on the leader's log / end and the follower's hw
    static KMC_DEV u32 lookup_offset_for_epoch_v(LogT logl, u32 el, u32 hwf, u32 epoch) {
        u32 first_larger = hwf;  // offsetWithLargerEpochs = {} -> follower hw
        bool found = false;
        kmc_static_for<0, L>([&](auto O) {
            constexpr int o = decltype(O)::value;
            if (!found && (u32)o < el && rec_epoch(rec_at(logl, o)) > epoch) { first_larger = o; found = true; }
        });
        if (el == 0) return hwf;
        if (rec_epoch(rec_at(logl, el - 1)) == epoch) return el;
        return first_larger;
    }
    template <int l, int f> static KMC_DEV u32 lookup_offset_for_epoch(const Pre& p, u32 epoch) {
        return lookup_offset_for_epoch_v(p.logv(l), p.end(l), p.hw(f), epoch);
    }

    // ---- one action instance: guard + effect ---------------------------------------------
    // Returns "enabled"; when enabled, t holds the successor.  `extra` reports additional
    // satisfying bindings that yield the same successor (TLC counts them as generated).
    template <int I> static KMC_DEV u32 inst(const Pre& p, const u64* s, u64* t, int& kind, u32& extra) {
        extra = 0;
#pragma unroll
        for (int k = 0; k < W; ++k) t[k] = s[k];
        if constexpr (I < B1) {
            // ControllerElectLeader (KafkaReplication.tla:176-179)
            constexpr int r = I - B0;
            kind = 0;
            controller_update(t, p, r + 1, p.qisr());
            u32 g = p.epok & kmc_bit(p.qisr(), r);
            g = kmc_and(g, p.qldr1() != (u32)(r + 1));
            return g;
        } else if constexpr (I < B2) {
            // ControllerShrinkIsr (:158-168), three mutually exclusive cases per replica
            constexpr int r = I - B1;
            kind = 1;
            const bool is_ldr = p.qldr1() == (u32)(r + 1);
            const bool only = p.qisr() == (1u << r);
            const u32 newLdr1 = is_ldr ? 0u : p.qldr1();
            const u32 newIsr = (is_ldr && only) ? p.qisr() : (p.qisr() & ~(1u << r));
            controller_update(t, p, newLdr1, newIsr);
            return kmc_and(p.epok, is_ldr || (p.qisr() >> r & 1u));
        } else if constexpr (I < B3) {
            // BecomeLeader (:186-195): request e names leader l
            constexpr int J = I - B2, e = J / N, l = J % N;
            kind = 2;
            kmc_setbits(t, Y.ep_off[l], Y.BE, e + 1);
            kmc_setbits(t, Y.ldr_off[l], Y.BL, l + 1);
            kmc_setbits(t, Y.isr_off[l], Y.BI, p.risr(e));
            u32 g = kmc_and(p.one, p.rldr1(e) == (u32)(l + 1));
            g = kmc_and(g, (u32)e < p.nextEp());
            g = kmc_and(g, (u32)(e + 1) > p.ep1(l));
            return g;
        } else if constexpr (I < B4) {
            constexpr int J = I - B3, l = J / N, r = J % N;
            kind = 3;
            const u32 isr = p.isr(l);
            quorum_update<l>(t, isr | (1u << r));
            u32 g = kmc_bit(p.tm, l) & kmc_bit(~isr, r);
            if constexpr (K320) {  // FencedLeaderExpandIsr (Kip320.tla:110-117)
                g &= kmc_bit64(p.fm, l * N + r) & kmc_bit(p.hm, l);
                g = kmc_and(g, p.hw(l) <= p.end(r));  // HasFollowerReachedHighWatermark :94-98
            } else if constexpr (FIRST) {  // LeaderExpandIsrBetterFencing (Kip320FirstTry.tla:134-141)
                g &= kmc_bit(p.hm, l);
                g = kmc_and(g, caught_up_epoch<l, r>(p, p.hw(l)));
            } else {  // LeaderExpandIsr (KafkaReplication.tla:248-254); IsFollowerCaughtUp :219-225
                g = kmc_and(g, p.ldr1(r) == (u32)(l + 1));
                g = kmc_and(g, p.hw(l) <= p.end(l));
                g = kmc_and(g, p.hw(l) <= p.end(r));
            }
            return g;
        } else if constexpr (I < B5) {
            constexpr int J = I - B4, l = J / (N - 1), q = J % (N - 1), r = q + (q >= l);
            kind = 4;
            const u32 isr = p.isr(l);
            quorum_update<l>(t, isr & ~(1u << r));
            u32 g = kmc_bit(p.tm, l) & kmc_bit(isr, r);
            if constexpr (K320) {  // FencedLeaderShrinkIsr (Kip320.tla:78-85)
                g = kmc_and(g, kmc_bit64(p.fm, l * N + r) == 0u || p.end(r) < p.end(l));
                extra = (kmc_bit64(p.fm, l * N + r) == 0u && p.end(r) < p.end(l)) ? 1u : 0u;  // both disjuncts of :82-83
            } else if constexpr (FIRST) {  // LeaderShrinkIsrBetterFencing (Kip320FirstTry.tla:114-120)
                g = kmc_and(g, !caught_up_epoch<l, r>(p, p.end(l)));
            } else {  // LeaderShrinkIsr (KafkaReplication.tla:233-239)
                g = kmc_and(g, !caught_up<l, r>(p, p.end(l)));
            }
            return g;
        } else if constexpr (I < B6) {
            // LeaderWrite (KafkaReplication.tla:202-207)
            constexpr int r = I - B5;
            kind = 5;
            const u32 end = p.end(r);
            const LogT rec = (LogT)(((p.nextRec() + 1) << Y.BEr) | (p.ep1(r) - 1));
            kmc_setbits(t, Y.log_off[r], Y.BR * L, (LogT)(p.logv(r) | (LogT)(rec << (end * Y.BR))));
            kmc_setbits(t, Y.end_off[r], Y.BO, end + 1);
            kmc_setbits(t, Y.nextrec_off, Y.BNR, p.nextRec() + 1);
            u32 g = kmc_bit(p.pm, r);
            g = kmc_and(g, p.nextRec() <= (u32)(R - 1));
            g = kmc_and(g, end < (u32)L);
            return g;
        } else if constexpr (I < B7) {
            constexpr int l = I - B6;
            kind = 6;
            const u32 hw = p.hw(l);
            kmc_setbits(t, Y.hw_off[l], Y.BO, hw + 1);
            u32 g;
            if constexpr (K320) {  // FencedLeaderIncHighWatermark (Kip320.tla:63-70)
                g = kmc_and(p.one, hw < p.end(l));
                kmc_static_for<0, N>([&](auto F) {
                    constexpr int f = decltype(F)::value;
                    // f \in isr  =>  IsFollowingLeaderEpoch(l, f) /\ HasOffset(f, hw)
                    const u32 in = kmc_bit(p.isr(l), f);
                    g &= (in ^ 1u) | kmc_bit64(p.fm, l * N + f);
                    g = kmc_and(g, in == 0u || hw < p.end(f));
                });
            } else if constexpr (FIRST) {  // ImprovedLeaderIncHighWatermark (Kip320FirstTry.tla:90-97)
                g = kmc_bit(p.pm, l);
                g = kmc_and(g, hw < p.end(l));
                kmc_static_for<0, N>([&](auto F) {
                    constexpr int f = decltype(F)::value;
                    g = kmc_and(g, !(p.isr(l) >> f & 1u) || caught_up_epoch<l, f>(p, hw + 1));
                });
            } else {  // LeaderIncHighWatermark (KafkaReplication.tla:264-271)
                g = kmc_bit(p.pm, l);
                g = kmc_and(g, hw <= (u32)(L - 1));
                kmc_static_for<0, N>([&](auto F) {
                    constexpr int f = decltype(F)::value;
                    g = kmc_and(g, !(p.isr(l) >> f & 1u) || (p.ldr1(f) == (u32)(l + 1) && hw < p.end(f)));
                });
            }
            return g;
        } else if constexpr (I < B8) {
            // become follower of leader l at request epoch e (leader \in Replicas in every caller,
            // so the `leader = None` branch of KafkaReplication.tla:285-286 / Kip320.tla:138-140 is dead)
            constexpr int J = I - B7, pr = J / (E + 1), e = J % (E + 1);
            constexpr int l = pr / (N - 1), q = pr % (N - 1), r = q + (q >= l);
            kind = 7;
            u32 g = kmc_and(p.one, p.rldr1(e) == (u32)(l + 1));
            g = kmc_and(g, (u32)e < p.nextEp());
            g = kmc_and(g, (u32)(e + 1) > p.ep1(r));
            kmc_setbits(t, Y.ep_off[r], Y.BE, e + 1);
            kmc_setbits(t, Y.ldr_off[r], Y.BL, l + 1);
            kmc_setbits(t, Y.isr_off[r], Y.BI, p.risr(e));
            if constexpr (FIRST) {
                // BecomeFollower (Kip320FirstTry.tla:148-157): no truncation, hw unchanged
            } else {
                u32 off;
                if constexpr (MODEL == KMC_MODEL_TRUNCATE_TO_HW) {
                    off = p.hw(r);  // KafkaTruncateToHighWatermark.tla:29-31
                } else if constexpr (MODEL == KMC_MODEL_KIP101) {
                    // BecomeFollowerTruncateKip101 (Kip101.tla:41-47)
                    const u32 er = p.end(r);
                    const u32 last_epoch = rec_epoch(rec_at(p.logv(r), er == 0 ? 0 : er - 1));
                    off = er == 0 ? 0u : lookup_offset_for_epoch<l, r>(p, last_epoch);
                } else {
                    // BecomeFollowerTruncateKip279 (Kip279.tla:47-51) / FencedBecomeFollowerAndTruncate (Kip320.tla:134-148)
                    off = first_non_matching<l, r>(p);
                    if constexpr (MODEL == KMC_MODEL_KIP279) extra = p.end(r) == 0 ? 1u : 0u;  // both disjuncts fire
                    if constexpr (K320) {
                        g &= kmc_bit(p.pm, l);
                        g = kmc_and(g, p.ep1(l) == (u32)(e + 1));
                    }
                }
                g = kmc_and(g, off <= p.end(r));  // TruncateTo is disabled, not clamped (FiniteReplicatedLog.tla:106)
                truncate<r>(t, p, off);
                kmc_setbits(t, Y.hw_off[r], Y.BO, kmc_min(off, p.hw(r)));  // BecomeFollowerAndTruncateTo (:281-294)
            }
            return g;
        } else if constexpr (I < B9) {
            // ReplicateTo(leader, follower) + follower hw (KafkaReplication.tla:302-310,
            // Kip320.tla:49-56, Kip320FirstTry.tla:103-111)
            constexpr int J = I - B8, l = J / (N - 1), q = J % (N - 1), f = q + (q >= l);
            kind = 8;
            const u32 ef = p.end(f);
            const LogT rec = (LogT)rec_at(p.logv(l), ef);
            kmc_setbits(t, Y.log_off[f], Y.BR * L, (LogT)(p.logv(f) | (LogT)(rec << (ef * Y.BR))));
            kmc_setbits(t, Y.end_off[f], Y.BO, ef + 1);
            kmc_setbits(t, Y.hw_off[f], Y.BO, kmc_min(p.hw(l), ef + 1));
            u32 g = kmc_and(p.one, ef < p.end(l));
            g = kmc_and(g, ef < (u32)L);
            if constexpr (K320) g &= kmc_bit64(p.fm, l * N + f);
            else if constexpr (FIRST) g = kmc_and(g, caught_up_epoch<l, f>(p, ef));
            else {
                g &= kmc_bit(p.pm, l);
                g = kmc_and(g, p.ldr1(f) == (u32)(l + 1));
            }
            return g;
        } else {
            // FollowerTruncate (Kip320FirstTry.tla:75-82)
            constexpr int J = I - B9, l = J / (N - 1), q = J % (N - 1), f = q + (q >= l);
            kind = 9;
            const u32 off = first_non_matching<l, f>(p);
            truncate<f>(t, p, off);
            kmc_setbits(t, Y.hw_off[f], Y.BO, kmc_min(off, p.hw(f)));
            u32 g = kmc_bit(p.pm, l);
            g = kmc_and(g, p.ldr1(f) == (u32)(l + 1));
            g = kmc_and(g, needs_truncation<f, l>(p));
            g = kmc_and(g, off <= p.end(f));
            return g;
        }
    }

    // ---- kind-major effects (replica-major layouts; k_expand's pass 2) ---------------------------
    // inst<I> above fixes the replicas / request of a binding at COMPILE time, so pass 2 must run one leaf per
    // (kind, binding) some lane enabled: 30 leaves per 64-state tile at the headline, each for ~7 busy lanes.  apply<K>
    // takes the binding of its kind at RUN time, per lane: every lane applies ITS OWN next enabled binding of kind K in
    // the same leaf, so a tile needs max-over-lanes(enabled bindings of K) leaves per kind — 12.6 per tile instead of
    // 30 (tools/locality_sim.cpp).  That needs a field of a run-time replica to be cheap: under the replica-major layouts
    // it is "select a word, shift by a multiple of a stride, extract at a compile-time offset" (no shift with one replica
    // per word, the headline's layout).  Guards are NOT re-evaluated here (pass 1 did, with
    // inst<I>); tests/host_emu.cpp holds apply<K>(b) to inst<B_K + b> on every enabled binding of every visited state.
    static constexpr int kind_base(int k) {
        return k == 0 ? B0 : k == 1 ? B1 : k == 2 ? B2 : k == 3 ? B3 : k == 4 ? B4 : k == 5 ? B5 : k == 6 ? B6
             : k == 7 ? B7 : k == 8 ? B8 : k == 9 ? B9 : NINST;
    }
    static constexpr int kind_count(int k) { return kind_base(k + 1) - kind_base(k); }
    static constexpr int max_kind_count() {
        int m = 0;
        for (int k = 0; k < NKINDS; ++k) m = kind_count(k) > m ? kind_count(k) : m;
        return m;
    }
    static constexpr bool KIND_MAJOR = Y.rm != 0;
    // Pass 2 walks SEGMENTS: a kind's bindings in windows of at most WINBITS consecutive ones (one per-lane bitset each;
    // only 6 or more replicas have kinds with more bindings than one window).
    static constexpr int WINBITS = max_kind_count() <= 32 ? 32 : 64;
    using KindBits = typename KmcLogWord<(WINBITS == 32)>::type;
    static constexpr int kind_windows(int k) { return (kind_count(k) + WINBITS - 1) / WINBITS; }
    static constexpr int n_segments() {
        int n = 0;
        for (int k = 0; k < NKINDS; ++k) n += kind_windows(k);
        return n;
    }
    static constexpr int NSEGS = n_segments();
    static constexpr int seg_kind(int sg) {
        for (int k = 0; k < NKINDS; ++k) {
            if (sg < kind_windows(k)) return k;
            sg -= kind_windows(k);
        }
        return 0;
    }
    static constexpr int seg_first(int sg) {   // first binding (within its kind) of segment sg
        for (int k = 0; k < NKINDS; ++k) {
            if (sg < kind_windows(k)) return sg * WINBITS;
            sg -= kind_windows(k);
        }
        return 0;
    }
    static constexpr int seg_count(int sg) {   // bindings in segment sg
        const int left = kind_count(seg_kind(sg)) - seg_first(sg);
        return left < WINBITS ? left : WINBITS;
    }
    // the segment's bits of the per-lane "enabled instances" bitset en32[] (32-bit words), as one value
    template <int SG> static KMC_DEV KindBits seg_bits(const u32* en32) {
        constexpr int K = seg_kind(SG), first = seg_first(SG);
        constexpr int lo = kind_base(K) + first;
        constexpr int cnt = kind_count(K) - first < WINBITS ? kind_count(K) - first : WINBITS;
        u64 v = 0;
        kmc_static_for<lo / 32, (lo + cnt + 31) / 32>([&](auto H) {
            constexpr int h = decltype(H)::value;
            if constexpr (32 * h >= lo) v |= (u64)en32[h] << (32 * h - lo);
            else v |= (u64)(en32[h] >> (lo - 32 * h));
        });
        constexpr u64 mask = cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull);
        return (KindBits)(v & mask);
    }
    // Pass 2 with FULL leaves (kmc_expand_body): a tile's (lane, binding) pairs of a kind are dealt out 64 to a leaf, every lane
    // applying one pair to the state of the pair's source lane, instead of one leaf per "next binding of every lane".  On under
    // orbit counting from KMC_FULL_LEAVES_MIN_INSTANCES action instances on (seven brokers), where the kernel is bound by its
    // vector instructions and by the length of a wave's chain of leaves (profiles/r05_full_leaves.txt, one box):
    //   BASELINE config 5 under orbit counting, 14 levels   164.9 ms -> 96.3 ms        Kip320 7/1/1/0 (1,011 stored states) 24.7 -> 1.2 ms
    //   the same without orbit counting, 10 levels            27.4 -> 27.9 ms (the memory system bounds it: off by default there)
    //   headline / config 4, plain and orbit counting         +0.9 % / -2.5 %, -0.7 % / -3.5 %: nothing to gain at 12 leaves for 9 kinds
    // (-DKMC_FULL_LEAVES_MIN_INSTANCES=0 -DKMC_FULL_LEAVES_PLAIN=1 force it everywhere: tests/test_gpu_full_leaves.py.)
    static constexpr bool FULL_LEAVES = KIND_MAJOR && NINST >= KMC_FULL_LEAVES_MIN_INSTANCES && (KMC_SYMM || KMC_FULL_LEAVES_PLAIN);
#ifndef KMC_HOST_EMU
    // what an effect reads of the state's shared sub-predicates besides its words, pulled from ANOTHER lane (src4 / 4) for an
    // effect applied on that lane's behalf: apply<4> of Kip320 reads fm (both disjuncts of Kip320.tla:82-83).  The others are
    // POISONED, so that an effect which starts reading one fails every test instead of silently using the wrong lane's.
    static KMC_DEV void pull_pre(Pre& dst, const Pre& own, int src4) {
        dst.pm = dst.tm = dst.hm = dst.epok = 0xDEADu;
        if constexpr (K320) dst.fm = kmc_pull64(src4, own.fm);
        else dst.fm = 0xDEADDEADull;
    }
#endif
    // One of `count` consecutive state words, chosen at run time: a select chain over registers, never an indexed array.
    // (Each step is an opaque v_cndmask per 32-bit half: the plain chain `i == k ? w[k] : v` was recognised as w[i], the
    // state words went to scratch memory and every leaf loaded them back with a per-lane address — 255 M more vector
    // memory instructions per run and the headline at 41.9 ms instead of 35, profiles/r03_kind_major.txt.  Halves, so that
    // a leaf which only reads a replica's small fields does not select its log.)
    template <int COUNT> static KMC_DEV u64 sel_word(const u64* w, u32 i) {
        u32 lo = (u32)w[0], hi = (u32)(w[0] >> 32);
        kmc_static_for<1, COUNT>([&](auto KK) {
            constexpr int k = decltype(KK)::value;
            const bool c = i == (u32)k;
            lo = c ? (u32)w[k] : lo;
            hi = c ? (u32)(w[k] >> 32) : hi;
            KMC_OPAQUE_PURE(lo);
            KMC_OPAQUE_PURE(hi);
        });
        return ((u64)hi << 32) | lo;
    }
    template <int COUNT> static KMC_DEV void put_word(u64* t, u32 i, u64 v) {
        kmc_static_for<0, COUNT>([&](auto KK) {
            constexpr int k = decltype(KK)::value;
            const bool c = i == (u32)k;
            u32 lo = c ? (u32)v : (u32)t[k], hi = c ? (u32)(v >> 32) : (u32)(t[k] >> 32);
            KMC_OPAQUE_PURE(lo);
            KMC_OPAQUE_PURE(hi);
            t[k] = ((u64)hi << 32) | lo;
        });
    }
    // A replica chosen at run time: its log and its group of small fields (end | hw | ep | ldr | isr from bit 0), taken from
    // the words kmc_layout.h put them in.  `raw` (one replica per word only) is the replica's whole word as the PARENT has it.
    static constexpr bool ONE_PER_WORD = Y.rm == 1;
    static constexpr u64 LOGMASK = (Y.LB >= 64) ? ~0ull : ((1ull << Y.LB) - 1ull);
    static constexpr u32 SMMASK = (Y.SB >= 32) ? ~0u : ((1u << Y.SB) - 1u);
    static constexpr int O_END = 0, O_HW = Y.BO, O_EP = 2 * Y.BO, O_LDR = 2 * Y.BO + Y.BE, O_ISR = 2 * Y.BO + Y.BE + Y.BL;
    struct Rep {
        LogT log;
        u32 sm;
        u64 raw;
        KMC_DEV u32 end() const { return (sm >> O_END) & ((1u << Y.BO) - 1u); }
        KMC_DEV u32 hw() const { return (sm >> O_HW) & ((1u << Y.BO) - 1u); }
        KMC_DEV u32 ep1() const { return (sm >> O_EP) & ((1u << Y.BE) - 1u); }
        KMC_DEV u32 ldr1() const { return (sm >> O_LDR) & ((1u << Y.BL) - 1u); }
        KMC_DEV u32 isr() const { return (sm >> O_ISR) & ((1u << Y.BI) - 1u); }
        KMC_DEV void set(int off, int bits, u32 val) {
            const u32 m = ((1u << bits) - 1u) << off;
            sm = (sm & ~m) | ((val << off) & m);
        }
    };
    static KMC_DEV LogT get_log(const u64* w, u32 r) {
        if constexpr (Y.lg_q == 1) {
            return (LogT)((sel_word<Y.lg_words>(w + Y.lg_word0, r) >> Y.lg_base) & LOGMASK);
        } else {
            const u32 sh = (r % (u32)Y.lg_q) * (u32)Y.lg_stride + (u32)Y.lg_base;
            return (LogT)((sel_word<Y.lg_words>(w + Y.lg_word0, r / (u32)Y.lg_q) >> sh) & LOGMASK);
        }
    }
    static KMC_DEV u32 get_small(const u64* w, u32 r) {
        if constexpr (Y.sm_q == 1) {
            return (u32)(sel_word<Y.sm_words>(w + Y.sm_word0, r) >> Y.sm_base) & SMMASK;
        } else {
            const u32 sh = (r % (u32)Y.sm_q) * (u32)Y.sm_stride + (u32)Y.sm_base;
            return (u32)(sel_word<Y.sm_words>(w + Y.sm_word0, r / (u32)Y.sm_q) >> sh) & SMMASK;
        }
    }
    static KMC_DEV Rep get_rep(const u64* w, u32 r) {
        if constexpr (ONE_PER_WORD) {
            const u64 x = sel_word<N>(w, r);
            return Rep{(LogT)(x & LOGMASK), (u32)(x >> Y.sm_base) & SMMASK, x};
        } else {
            return Rep{get_log(w, r), get_small(w, r), 0ull};
        }
    }
    // Writes replica r back into t.  WLOG / WSM say which part changed.  One replica per word: its word is rebuilt from the
    // PARENT's (v.raw) — so a replica is put into t BEFORE any global field of t is written (they live in the spare bits of
    // the same words).  Grouped: a read-modify-write of t's own words, in any order.
    template <bool WLOG, bool WSM> static KMC_DEV void put_rep(u64* t, u32 r, const Rep& v) {
        if constexpr (ONE_PER_WORD) {
            u64 x = v.raw;
            if constexpr (WLOG) x = (x & ~LOGMASK) | ((u64)v.log & LOGMASK);
            if constexpr (WSM) x = (x & ~((u64)SMMASK << Y.sm_base)) | ((u64)(v.sm & SMMASK) << Y.sm_base);
            put_word<N>(t, r, x);
        } else {
            if constexpr (WLOG) {
                const u32 wi = Y.lg_q == 1 ? r : r / (u32)Y.lg_q;
                const u32 sh = Y.lg_q == 1 ? (u32)Y.lg_base : (r % (u32)Y.lg_q) * (u32)Y.lg_stride + (u32)Y.lg_base;
                u64 x = sel_word<Y.lg_words>(t + Y.lg_word0, wi);
                x = (x & ~(LOGMASK << sh)) | (((u64)v.log & LOGMASK) << sh);
                put_word<Y.lg_words>(t + Y.lg_word0, wi, x);
            }
            if constexpr (WSM) {
                const u32 wi = Y.sm_q == 1 ? r : r / (u32)Y.sm_q;
                const u32 sh = Y.sm_q == 1 ? (u32)Y.sm_base : (r % (u32)Y.sm_q) * (u32)Y.sm_stride + (u32)Y.sm_base;
                u64 x = sel_word<Y.sm_words>(t + Y.sm_word0, wi);
                x = (x & ~((u64)SMMASK << sh)) | ((u64)(v.sm & SMMASK) << sh);
                put_word<Y.sm_words>(t + Y.sm_word0, wi, x);
            }
        }
    }
    // the isr of the request with leader epoch e (run-time e)
    static KMC_DEV u32 risr_rt(const Pre& p, u32 e) {
        u32 v = p.risr(0);
        kmc_static_for<1, E + 1>([&](auto EE) {
            constexpr int k = decltype(EE)::value;
            v = e == (u32)k ? p.risr(k) : v;
        });
        return v;
    }
    // (l, r) of the j-th ordered pair of distinct replicas: the enumeration inst<I> uses for its (leader, other) bindings
    static KMC_DEV void pair_of(u32 j, u32& l, u32& r) {
        l = j / (u32)(N - 1);
        const u32 q = j % (u32)(N - 1);
        r = q + (q >= l ? 1u : 0u);
    }

    // The leader named by the request with leader epoch e (run-time e), as index + 1
    static KMC_DEV u32 rldr1_rt(const Pre& p, u32 e) {
        u32 v = p.rldr1(0);
        kmc_static_for<1, E + 1>([&](auto EE) {
            constexpr int k = decltype(EE)::value;
            v = e == (u32)k ? p.rldr1(k) : v;
        });
        return v;
    }
    // The GUARD of binding b of kind K (0 / 1), b a run-time value: what inst<kind_base(K) + b> returns.  A second lowering
    // of the guards, used by KMC_VERIFY's second build (RUNTIME_GUARDS: a loop of guard<K> over a kind's bindings, b
    // wave-uniform, fused into pass 2's walk; O(kinds) code that compiles in seconds — and runs 20-110 % slower than the
    // straight-line block of every instance's guard, which shares sub-terms across instances: KMC_RT_GUARDS_MIN_INSTANCES).
    // The expressions are inst<I>'s, line by line; tests/host_emu.cpp compares the two on EVERY binding (enabled or not)
    // of every visited state.
    static constexpr bool RUNTIME_GUARDS = KIND_MAJOR && NINST > KMC_RT_GUARDS_MIN_INSTANCES;
    template <int K> static KMC_DEV u32 guard(const Pre& p, const u64* s, u32 b) {
        if constexpr (K == 0) {
            // ControllerElectLeader (KafkaReplication.tla:176-179)
            return p.epok & ((p.qisr() >> b) & 1u) & (p.qldr1() != b + 1u ? 1u : 0u);
        } else if constexpr (K == 1) {
            // ControllerShrinkIsr (:158-168)
            return p.epok & ((p.qldr1() == b + 1u || ((p.qisr() >> b) & 1u)) ? 1u : 0u);
        } else if constexpr (K == 2) {
            // BecomeLeader (:186-195)
            const u32 e = b / (u32)N, l = b % (u32)N;
            return (rldr1_rt(p, e) == l + 1u && e < p.nextEp() && e + 1u > (get_small(s, l) >> O_EP & ((1u << Y.BE) - 1u))) ? 1u : 0u;
        } else if constexpr (K == 3) {
            const u32 l = b / (u32)N, r = b % (u32)N;
            const Rep vl = get_rep(s, l), vr = get_rep(s, r);
            u32 g = ((p.tm >> l) & 1u) & ((~vl.isr() >> r) & 1u);
            if constexpr (K320) {  // FencedLeaderExpandIsr (Kip320.tla:110-117)
                g &= (u32)(p.fm >> (l * (u32)N + r)) & 1u & (p.hm >> l);
                g &= vl.hw() <= vr.end() ? 1u : 0u;
            } else if constexpr (FIRST) {  // LeaderExpandIsrBetterFencing (Kip320FirstTry.tla:134-141)
                g &= (p.hm >> l) & 1u;
                g &= caught_up_epoch_v(((p.pm >> l) & 1u) && vr.ldr1() == l + 1u, vl.log, vr.log, vl.end(), vr.end(), vl.hw()) ? 1u : 0u;
            } else {  // LeaderExpandIsr (KafkaReplication.tla:248-254)
                g &= (vr.ldr1() == l + 1u && vl.hw() <= vl.end() && vl.hw() <= vr.end()) ? 1u : 0u;
            }
            return g;
        } else if constexpr (K == 4) {
            u32 l, r;
            pair_of(b, l, r);
            const Rep vl = get_rep(s, l), vr = get_rep(s, r);
            u32 g = ((p.tm >> l) & 1u) & ((vl.isr() >> r) & 1u);
            if constexpr (K320) {  // FencedLeaderShrinkIsr (Kip320.tla:78-85)
                g &= ((((u32)(p.fm >> (l * (u32)N + r)) & 1u) == 0u) || vr.end() < vl.end()) ? 1u : 0u;
            } else if constexpr (FIRST) {  // LeaderShrinkIsrBetterFencing (Kip320FirstTry.tla:114-120)
                g &= !caught_up_epoch_v(((p.pm >> l) & 1u) && vr.ldr1() == l + 1u, vl.log, vr.log, vl.end(), vr.end(), vl.end()) ? 1u : 0u;
            } else {  // LeaderShrinkIsr (KafkaReplication.tla:233-239); IsFollowerCaughtUp :219-225
                g &= !(vr.ldr1() == l + 1u && vl.end() <= vl.end() && vl.end() <= vr.end()) ? 1u : 0u;
            }
            return g;
        } else if constexpr (K == 5) {
            // LeaderWrite (:202-207)
            const u32 end = get_small(s, b) & ((1u << Y.BO) - 1u);
            return ((p.pm >> b) & 1u) & ((p.nextRec() <= (u32)(R - 1) && end < (u32)L) ? 1u : 0u);
        } else if constexpr (K == 6) {
            const u32 l = b;
            const Rep vl = get_rep(s, l);
            const u32 hw = vl.hw(), isr = vl.isr();
            u32 g;
            if constexpr (K320) {  // FencedLeaderIncHighWatermark (Kip320.tla:63-70)
                g = hw < vl.end() ? 1u : 0u;
                kmc_static_for<0, N>([&](auto F) {
                    constexpr int f = decltype(F)::value;
                    const u32 in = (isr >> f) & 1u;
                    g &= (in ^ 1u) | ((u32)(p.fm >> (l * (u32)N + (u32)f)) & 1u);
                    g &= (in == 0u || hw < p.end(f)) ? 1u : 0u;
                });
            } else if constexpr (FIRST) {  // ImprovedLeaderIncHighWatermark (Kip320FirstTry.tla:90-97)
                g = ((p.pm >> l) & 1u) & (hw < vl.end() ? 1u : 0u);
                kmc_static_for<0, N>([&](auto F) {
                    constexpr int f = decltype(F)::value;
                    const bool following = ((p.pm >> l) & 1u) && p.ldr1(f) == l + 1u;
                    g &= (!((isr >> f) & 1u) || caught_up_epoch_v(following, vl.log, p.logv(f), vl.end(), p.end(f), hw + 1u)) ? 1u : 0u;
                });
            } else {  // LeaderIncHighWatermark (KafkaReplication.tla:264-271)
                g = ((p.pm >> l) & 1u) & (hw <= (u32)(L - 1) ? 1u : 0u);
                kmc_static_for<0, N>([&](auto F) {
                    constexpr int f = decltype(F)::value;
                    g &= (!((isr >> f) & 1u) || (p.ldr1(f) == l + 1u && hw < p.end(f))) ? 1u : 0u;
                });
            }
            return g;
        } else if constexpr (K == 7) {
            const u32 pr = b / (u32)(E + 1), e = b % (u32)(E + 1);
            u32 l, r;
            pair_of(pr, l, r);
            const Rep vr = get_rep(s, r);
            u32 g = (rldr1_rt(p, e) == l + 1u && e < p.nextEp() && e + 1u > vr.ep1()) ? 1u : 0u;
            if constexpr (!FIRST) {
                u32 off;
                if constexpr (MODEL == KMC_MODEL_TRUNCATE_TO_HW) {
                    off = vr.hw();
                } else {
                    const Rep vl = get_rep(s, l);
                    if constexpr (MODEL == KMC_MODEL_KIP101) {
                        const u32 er = vr.end();
                        const u32 last_epoch = rec_epoch(rec_at(vr.log, er == 0 ? 0u : er - 1u));
                        off = er == 0 ? 0u : lookup_offset_for_epoch_v(vl.log, vl.end(), vr.hw(), last_epoch);
                    } else {
                        off = first_non_matching_v(vl.log, vr.log, vl.end(), vr.end());
                        if constexpr (K320) g &= ((p.pm >> l) & 1u) & (vl.ep1() == e + 1u ? 1u : 0u);
                    }
                }
                g &= off <= vr.end() ? 1u : 0u;  // TruncateTo is disabled, not clamped (FiniteReplicatedLog.tla:106)
            }
            return g;
        } else if constexpr (K == 8) {
            u32 l, f;
            pair_of(b, l, f);
            const Rep vl = get_rep(s, l), vf = get_rep(s, f);
            const u32 ef = vf.end();
            u32 g = (ef < vl.end() && ef < (u32)L) ? 1u : 0u;
            if constexpr (K320) g &= (u32)(p.fm >> (l * (u32)N + f)) & 1u;
            else if constexpr (FIRST) g &= caught_up_epoch_v(((p.pm >> l) & 1u) && vf.ldr1() == l + 1u, vl.log, vf.log, vl.end(), vf.end(), ef) ? 1u : 0u;
            else g &= ((p.pm >> l) & 1u) & (vf.ldr1() == l + 1u ? 1u : 0u);
            return g;
        } else {
            // FollowerTruncate (Kip320FirstTry.tla:75-82)
            u32 l, f;
            pair_of(b, l, f);
            const Rep vl = get_rep(s, l), vf = get_rep(s, f);
            const u32 off = first_non_matching_v(vl.log, vf.log, vl.end(), vf.end());
            return (((p.pm >> l) & 1u) && vf.ldr1() == l + 1u && needs_truncation_v(vf.log, vl.log, vf.end(), vl.end()) &&
                    off <= vf.end()) ? 1u : 0u;
        }
    }

    // The effect of binding b of kind K on s -> t.  The successor and `extra` equal inst<kind_base(K) + b>'s.
    template <int K> static KMC_DEV void apply(const Pre& p, const u64* s, u64* t, u32 b, u32& extra) {
        extra = 0;
#pragma unroll
        for (int k = 0; k < W; ++k) t[k] = s[k];
        if constexpr (K == 0) {
            // ControllerElectLeader (KafkaReplication.tla:176-179)
            controller_update(t, p, b + 1u, p.qisr());
        } else if constexpr (K == 1) {
            // ControllerShrinkIsr (:158-168)
            const u32 r = b;
            const bool is_ldr = p.qldr1() == r + 1u;
            const bool only = p.qisr() == (1u << r);
            const u32 newLdr1 = is_ldr ? 0u : p.qldr1();
            const u32 newIsr = (is_ldr && only) ? p.qisr() : (p.qisr() & ~(1u << r));
            controller_update(t, p, newLdr1, newIsr);
        } else if constexpr (K == 2) {
            // BecomeLeader (:186-195): request e names leader l
            const u32 e = b / (u32)N, l = b % (u32)N;
            Rep v = get_rep(s, l);
            v.set(O_EP, Y.BE, e + 1u);
            v.set(O_LDR, Y.BL, l + 1u);
            v.set(O_ISR, Y.BI, risr_rt(p, e));
            put_rep<false, true>(t, l, v);
        } else if constexpr (K == 3) {
            // Leader*ExpandIsr* (:248-254, Kip320.tla:110-117, Kip320FirstTry.tla:134-141): QuorumUpdateLeaderAndIsr
            const u32 l = b / (u32)N, r = b % (u32)N;
            Rep v = get_rep(s, l);
            const u32 nisr = v.isr() | (1u << r);
            v.set(O_ISR, Y.BI, nisr);
            put_rep<false, true>(t, l, v);
            kmc_setbits(t, Y.qisr_off, Y.BI, nisr);
        } else if constexpr (K == 4) {
            // Leader*ShrinkIsr* (:233-239, Kip320.tla:78-85, Kip320FirstTry.tla:114-120)
            u32 l, r;
            pair_of(b, l, r);
            Rep v = get_rep(s, l);
            const u32 end_l = v.end();
            const u32 nisr = v.isr() & ~(1u << r);
            v.set(O_ISR, Y.BI, nisr);
            put_rep<false, true>(t, l, v);
            kmc_setbits(t, Y.qisr_off, Y.BI, nisr);
            if constexpr (K320) {  // both disjuncts of Kip320.tla:82-83
                const u32 following = (u32)(p.fm >> (l * (u32)N + r)) & 1u;
                const u32 end_r = get_small(s, r) & ((1u << Y.BO) - 1u);
                extra = (following == 0u && end_r < end_l) ? 1u : 0u;
            }
        } else if constexpr (K == 5) {
            // LeaderWrite (:202-207)
            const u32 r = b;
            Rep v = get_rep(s, r);
            const u32 end = v.end();
            const LogT rec = (LogT)(((p.nextRec() + 1u) << Y.BEr) | (v.ep1() - 1u));
            v.log = (LogT)(v.log | (LogT)(rec << (end * Y.BR)));
            v.set(O_END, Y.BO, end + 1u);
            put_rep<true, true>(t, r, v);
            kmc_setbits(t, Y.nextrec_off, Y.BNR, p.nextRec() + 1u);
        } else if constexpr (K == 6) {
            // *LeaderIncHighWatermark (:264-271, Kip320.tla:63-70, Kip320FirstTry.tla:90-97)
            const u32 l = b;
            Rep v = get_rep(s, l);
            v.set(O_HW, Y.BO, v.hw() + 1u);
            put_rep<false, true>(t, l, v);
        } else if constexpr (K == 7) {
            // BecomeFollower* of leader l at request epoch e (:281-294 and the five truncation rules)
            const u32 pr = b / (u32)(E + 1), e = b % (u32)(E + 1);
            u32 l, r;
            pair_of(pr, l, r);
            Rep v = get_rep(s, r);
            const u32 end_r = v.end(), hw_r = v.hw();
            v.set(O_EP, Y.BE, e + 1u);
            v.set(O_LDR, Y.BL, l + 1u);
            v.set(O_ISR, Y.BI, risr_rt(p, e));
            if constexpr (FIRST) {
                put_rep<false, true>(t, r, v);   // BecomeFollower (Kip320FirstTry.tla:148-157): no truncation, hw unchanged
            } else {
                u32 off;
                if constexpr (MODEL == KMC_MODEL_TRUNCATE_TO_HW) {
                    off = hw_r;  // KafkaTruncateToHighWatermark.tla:29-31
                } else {
                    const LogT log_l = get_log(s, l);
                    const u32 end_l = get_small(s, l) & ((1u << Y.BO) - 1u);
                    if constexpr (MODEL == KMC_MODEL_KIP101) {  // Kip101.tla:41-47
                        const u32 last_epoch = rec_epoch(rec_at(v.log, end_r == 0 ? 0u : end_r - 1u));
                        off = end_r == 0 ? 0u : lookup_offset_for_epoch_v(log_l, end_l, hw_r, last_epoch);
                    } else {  // Kip279.tla:47-51 / Kip320.tla:134-148
                        off = first_non_matching_v(log_l, v.log, end_l, end_r);
                        if constexpr (MODEL == KMC_MODEL_KIP279) extra = end_r == 0 ? 1u : 0u;
                    }
                }
                v.log = (LogT)(v.log & keep_below(off));   // TruncateTo (FiniteReplicatedLog.tla:105-109)
                v.set(O_END, Y.BO, off);
                v.set(O_HW, Y.BO, kmc_min(off, hw_r));
                put_rep<true, true>(t, r, v);
            }
        } else if constexpr (K == 8) {
            // FollowerReplicate / *Fetch (:302-310, Kip320.tla:49-56, Kip320FirstTry.tla:103-111)
            u32 l, f;
            pair_of(b, l, f);
            const Rep vl = get_rep(s, l);
            Rep v = get_rep(s, f);
            const u32 ef = v.end();
            const LogT rec = (LogT)rec_at(vl.log, ef);
            v.log = (LogT)(v.log | (LogT)(rec << (ef * Y.BR)));
            v.set(O_END, Y.BO, ef + 1u);
            v.set(O_HW, Y.BO, kmc_min(vl.hw(), ef + 1u));
            put_rep<true, true>(t, f, v);
        } else {
            // FollowerTruncate (Kip320FirstTry.tla:75-82)
            u32 l, f;
            pair_of(b, l, f);
            const Rep vl = get_rep(s, l);
            Rep v = get_rep(s, f);
            const u32 off = first_non_matching_v(vl.log, v.log, vl.end(), v.end());
            const u32 hw_f = v.hw();
            v.log = (LogT)(v.log & keep_below(off));
            v.set(O_END, Y.BO, off);
            v.set(O_HW, Y.BO, kmc_min(off, hw_f));
            put_rep<true, true>(t, f, v);
        }
    }

    // ---- invariants; bit k of the result = invariant k violated ---------------------------
    // 0 TypeOk (:101-107)  1 WeakIsr (:320-326)  2 StrongIsr (:334-340)  3 LeaderInIsr (:345)
    static KMC_DEV u32 violated(const u64* t, u32 inv_mask) { return violated_pre(extract(t), inv_mask); }
    // --- TypeOk's log part, per replica (ReplicaLog!TypeOk, FiniteReplicatedLog.tla:90-95 with LogRecords of
    // KafkaReplication.tla:82): slots below endOffset hold an element of LogRecords, slots from endOffset on are Nil.
    // Evaluated in the integer domain on the whole log word (a bool chain per slot was 400 VALU instructions per
    // tile, a fifth of k_expand's arithmetic):
    //   * fold every slot onto its lowest bit ("slot is non-Nil"); the non-Nil slots must be exactly the prefix
    //     [0, endOffset) — one compare against the prefix mask covers both "written below" and "Nil above";
    //   * every slot's code must be Nil or a member of LogRecords: a 2^BR-bit membership map indexed by the code
    //     (id+1 in 1..R, epoch in 0..E), one extract + one lookup per slot.
    static constexpr u64 valid_code_map() {  // bit c: code c is Nil or \in LogRecords  (meaningful when BR <= 6)
        u64 m = 1ull;
        for (int c = 1; c < (Y.BR <= 6 ? (1 << Y.BR) : 1); ++c) {
            const int id1 = c >> Y.BEr, ep = c & ((1 << Y.BEr) - 1);
            if (id1 >= 1 && id1 <= R && ep <= E) m |= 1ull << c;
        }
        return m;
    }
    static constexpr LogT low_bits() {  // the lowest bit of every slot
        LogT m = 0;
        for (int o = 0; o < L; ++o) m |= (LogT)((LogT)1 << (o * Y.BR));
        return m;
    }
    template <int r> static KMC_DEV u32 log_type_bad(const Pre& p) {  // 0 = ReplicaLog!TypeOk holds for replica r
        const LogT lv = p.logv(r);
        const u32 end = p.end(r);
        LogT fold = lv;
#pragma unroll
        for (int b = 1; b < Y.BR; ++b) fold |= (LogT)(lv >> b);
        fold &= low_bits();
        // end > L is rejected on its own (the caller tests end <= L); keep_below saturates there
        u32 bad = fold != (LogT)(keep_below(end) & low_bits()) ? 1u : 0u;
        if constexpr (Y.BR <= 6) {
            constexpr u64 MAP = valid_code_map();
            u32 okall = 1u;
            kmc_static_for<0, L>([&](auto O) {
                constexpr int o = decltype(O)::value;
                const u32 c = rec_at(lv, o);
                if constexpr (Y.BR <= 5) okall &= ((u32)MAP >> c);
                else okall &= (u32)(MAP >> c);
            });
            bad |= (okall & 1u) ^ 1u;
        } else {
            kmc_static_for<0, L>([&](auto O) {
                constexpr int o = decltype(O)::value;
                const u32 c = rec_at(lv, o);
                const u32 id1 = c >> Y.BEr;
                bad |= (c != 0 && !(id1 >= 1 && id1 <= (u32)R && rec_epoch(c) <= (u32)E)) ? 1u : 0u;
            });
        }
        return bad;
    }

    // The same invariants for a kernel that does nothing else (k_inv: a frontier streamed through, kmc_check_states): the view
    // holds the state's words only — the guards' shared sub-predicates (extract: 49 following-epoch pairs at seven brokers, made
    // opaque there and therefore never dead) are not the invariants' business — and the evaluation runs replica after replica
    // (SEQ: an opaque point after each, and a leader nobody in the wave presumes is skipped), so that it fits the registers of
    // four waves per SIMD instead of the 171 - 200 the interleaved form takes (profiles/r06_inv.txt).
    static KMC_DEV u32 violated_stream(const u64* t, u32 inv_mask) {
        Pre p;
        p.w = t;
        p.one = 1u; p.epok = 0; p.pm = 0; p.tm = 0; p.hm = 0; p.fm = 0;
        return violated_pre<true>(p, inv_mask);
    }
    template <bool SEQ = false> static KMC_DEV u32 violated_pre(const Pre& p, u32 inv_mask) {
        if (inv_mask == 0) return 0;
        u32 bad = 0;
        if (inv_mask & 1u) {
            // TypeOk (KafkaReplication.tla:101-107); comparisons a field's width already implies fold away
            u32 nb = (p.nextEp() > (u32)(E + 1) ? 1u : 0u) | (p.nextRec() > (u32)R ? 1u : 0u) |
                     (p.qep1() > (u32)(E + 1) ? 1u : 0u) | (p.qldr1() > (u32)N ? 1u : 0u);
            kmc_static_for<0, N>([&](auto RR) {
                constexpr int r = decltype(RR)::value;
                nb |= (p.end(r) > (u32)L ? 1u : 0u) | (p.hw(r) > (u32)L ? 1u : 0u) | (p.ep1(r) > (u32)(E + 1) ? 1u : 0u) |
                      (p.ldr1(r) > (u32)N ? 1u : 0u);
                nb |= log_type_bad<r>(p);
                if constexpr (SEQ) kmc_launder(nb);
            });
            kmc_static_for<0, E + 1>([&](auto EE) {
                constexpr int e = decltype(EE)::value;
                nb |= ((u32)e < p.nextEp() && p.rldr1(e) > (u32)N) ? 1u : 0u;
            });
            bad |= nb & 1u;
        }
        if (inv_mask & 6u) {
            // WeakIsr (:320-326) / StrongIsr (:334-340), integer domain: for a replica r1 that presumes leadership
            // with hw > 0, every r2 of its isr (weak) / of quorumState.isr (strong) must agree with it below hw:
            // \A offset < hw : \E record : HasEntry(r1, ..) /\ HasEntry(r2, ..)  <=>  hw <= end1, hw <= end2 and the two
            // logs are equal on the slots below hw.
            u32 wbad = 0, sbad = 0;
            const u32 qisr = p.qisr();
            kmc_static_for<0, N>([&](auto R1) {
                constexpr int r1 = decltype(R1)::value;
                const u32 hw = p.hw(r1);
                const u32 act = (presumes<r1>(p) && hw > 0) ? 1u : 0u;
                if constexpr (SEQ) {
                    if (!kmc_any_lane(act != 0)) return;   // (wave-uniform: no state of this tile has r1 as a leader with hw > 0)
                }
                const LogT kb = keep_below(hw);
                const LogT l1 = p.logv(r1);
                const u32 short1 = hw > p.end(r1) ? 1u : 0u;
                u32 differs = 0;  // bit r2: r2 does NOT agree with r1 below hw
                kmc_static_for<0, N>([&](auto R2) {
                    constexpr int r2 = decltype(R2)::value;
                    u32 d = short1;
                    if constexpr (r2 != r1) {
                        d |= hw > p.end(r2) ? 1u : 0u;
                        d |= ((LogT)((l1 ^ p.logv(r2)) & kb)) != 0 ? 1u : 0u;
                    }
                    differs |= d << r2;
                });
                const u32 m = act ? differs : 0u;
                wbad |= m & p.isr(r1);
                sbad |= m & qisr;
                if constexpr (SEQ) { kmc_launder(wbad); kmc_launder(sbad); }
            });
            if ((inv_mask & 2u) && wbad) bad |= 2u;
            if ((inv_mask & 4u) && sbad) bad |= 4u;
        }
        if (inv_mask & 8u) {
            const bool ok = p.qldr1() != 0 && (p.qisr() >> (p.qldr1() - 1) & 1u);
            if (!ok) bad |= 8u;
        }
        return bad;
    }
};

