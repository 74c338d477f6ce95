// kmc_sink.h — the successor sink: output stager, seen-set probe / claim, owner bucketing, enumeration.
// Part of the device source (kmc_device.h lists the parts; the host engine hands their concatenation to hiprtc).
#pragma once
#include "kmc_common.h"
// ========================================================================================
// successor sink: table probe/insert + frontier append, or owner bucketing, or enumeration
// ========================================================================================
#ifndef KMC_HOST_EMU  // ---- everything below is wave-level device code ----
// Per-wave output stager: winners wait in an LDS ring (W planes x KMC_QCAP) until 64 of them
// can be appended with ONE atomicAdd and W fully coalesced 512-byte plane stores.  (A single
// device-scope counter saturates near 90 M atomics/s; one atomic per flush of ~20 winners
// sat right on that limit.)
template <int W> struct KmcStager {
    // KMC_SYMM: a state travels with one more word — the order of its stabiliser under the permutations of Replicas, found
    // for free when its representative was chosen (KmcSymm::canon) — in plane W of the stager and of the frontiers, so that
    // the expansion does not have to walk through its N! images again to know how many states it stands for
    static constexpr int PL = W + (KMC_SYMM ? 1 : 0);
    static constexpr int QCAP = KMC_QCAP;
    u64* planes;   // LDS, [PL][QCAP]
    u32 count;     // wave-uniform; < QCAP between pushes (entries 0 .. count-1 are staged)
    u32 filtered;  // SHARDED: remote successors this wave's sender-side filter dropped (added to the level's counter once,
                   // in finish(): one atomicAdd per flush on that single line capped the sharded kernel at ~90 M flushes/s,
                   // 5.6x the time of the local kernel for the same work)
    u32 probed, won, outside;  // wave-uniform conservation counters (KmcLevelCtl), added to the level's once, in finish()
#if KMC_SYMM
    u32 corr_won;              // per lane: orbit deficits of the claims this lane won (KmcLevelCtl::corr_won)
#endif
#if KMC_CHECKSUM
    u64 csum, cxor;            // per lane: running sum and xor of the fingerprints this lane sent into the sink
#endif
#if KMC_PROFILE
    u64* prof;     // the wave's phase accumulators (5 = fingerprint, 6 = probe/claim)
#endif

    KMC_DEV void init(u64* lds) {
        planes = lds; count = 0; filtered = 0; probed = 0; won = 0; outside = 0;
#if KMC_SYMM
        corr_won = 0;
#endif
#if KMC_CHECKSUM
        csum = 0; cxor = 0;
#endif
    }
    KMC_DEV void account(bool valid, u64 fp) {  // every successor on its way into the sink
        probed += (u32)__popcll(__ballot(valid));
#if KMC_CHECKSUM
        csum += valid ? fp : 0ull;
        cxor ^= valid ? fp : 0ull;
#endif
    }

    KMC_DEV void drain(const KmcArgsLocal& a, u32 n) {  // the n <= 64 staged states -> next frontier
        const u32 lane = kmc_lane();
        const u32 seg = blockIdx.x % KMC_SEGS;
        u64 base = 0;
        if (lane == 0) base = atomicAdd(&a.ctl->next_count[seg].v, (u64)n);
        base = kmc_bcast64(base, 0);
        if (lane < n) {
            if (base + lane < a.seg_cap) {
                const u64 idx = (u64)seg * a.seg_cap + base + lane;
#pragma unroll
                for (int k = 0; k < PL; ++k) KMC_FRONTIER_STORE(&a.fout[(u64)k * a.fout_stride + idx], planes[k * QCAP + lane]);
            } else {
                atomicOr(&a.ctl->err, KMC_ERR_FRONTIER_FULL);
            }
        }
        count = 0;
    }
    // Stage the new states of a batch.  When they do not all fit, the first `room` of them complete the stager, it is
    // drained (always exactly 64: one atomicAdd, W coalesced 512-byte plane stores), and the rest start the next batch.
    KMC_DEV void push(const KmcArgsLocal& a, bool isnew, const u64* t, u32 tag = 0) {
        const u64 m = __ballot(isnew);
        if (m == 0) return;
        const u32 n = __popcll(m);
        won += n;
        const u32 rank = kmc_rank_in(m);
        const u32 room = QCAP - count;   // >= 1
        if (isnew && rank < room) {
#pragma unroll
            for (int k = 0; k < W; ++k) planes[k * QCAP + count + rank] = t[k];
            if constexpr (PL > W) planes[W * QCAP + count + rank] = tag;
        }
        if (n < room) {
            count += n;
            return;
        }
        drain(a, QCAP);
        if (isnew && rank >= room) {
#pragma unroll
            for (int k = 0; k < W; ++k) planes[k * QCAP + (rank - room)] = t[k];
            if constexpr (PL > W) planes[W * QCAP + (rank - room)] = tag;
        }
        count = n - room;
    }
    KMC_DEV void finish(const KmcArgsLocal& a, bool publish_counters = true) {
        if (count) drain(a, count);
        if (filtered && kmc_lane() == 0) atomicAdd(&a.ctl->send_filtered, (u64)filtered);   // (SHARDED only: 0 elsewhere)
        filtered = 0;
        if (!publish_counters) return;   // k_expand folds them into its per-block tail (kmc_expand_body)
#if KMC_SYMM
        {
            const u32 x = kmc_wave_sum(corr_won);
            if (x && kmc_lane() == 0) atomicAdd(&a.ctl->corr_won, (u64)x);
            corr_won = 0;
        }
#endif
        if (probed | outside) {
#if KMC_CHECKSUM
            u64 sm = csum, xr = cxor;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                sm += ((u64)(u32)__shfl_xor((int)(u32)(sm >> 32), off) << 32) | (u32)__shfl_xor((int)(u32)sm, off);
                xr ^= ((u64)(u32)__shfl_xor((int)(u32)(xr >> 32), off) << 32) | (u32)__shfl_xor((int)(u32)xr, off);
            }
            kmc_launder(sm);   // (read by lane 0 only, below: kept ahead of that branch — kmc_wave_sum)
            kmc_launder(xr);
            csum = 0; cxor = 0;
#endif
            if (kmc_lane() == 0) {
                atomicAdd(&a.ctl->probed, (u64)probed);
                if (won) atomicAdd(&a.ctl->won, (u64)won);
                if (outside) atomicAdd(&a.ctl->outside, (u64)outside);
#if KMC_CHECKSUM
                atomicAdd(&a.ctl->fp_sum, sm);
                atomicXor(&a.ctl->fp_xor, xr);
#endif
            }
        }
        probed = won = outside = 0;
    }
};

struct alignas(16) KmcSlot2 { u64 x, y; };   // a wide seen-set slot: fingerprint, check word

template <class M> struct KmcSink {
    static constexpr int W = M::W;

    // ---- probe / claim -----------------------------------------------------------------------------------------------------
    // Open addressing, linear probing.  Slots only ever change 0 -> fp, so a plain (possibly stale) load can only mis-report
    // "empty", which the compare-and-swap then corrects.  The chain is bounded: a table filled beyond ~95 % makes linear probing
    // walk millions of slots per insert (a run that looked hung), so a chain this long is reported as "table full" instead.  At
    // load <= 0.9 the chance of a 1 K chain is nil.
    //
    // ONE ROUND TRIP PER STEP FOR EVERY LANE (round 6).  The textbook loop — load; where the slot read empty, compare-and-swap;
    // walk on — is executed by a wave as the SUM of what its lanes need: the lanes whose slot read empty send their CAS while the
    // lanes that must walk on wait, then those send their load while the claimers wait: three to five dependent round trips for
    // a batch in which no lane needs more than two or three.  Here every step is one round trip in which each walking lane sends
    // what IT needs next — the CAS of the slot it saw empty, or the load of its next slot — and the wave waits once for both
    // (tests/test_abi_cpu.py::test_a_probe_step_sends_the_claims_and_the_next_loads_together reads the ISA).  v = what slot i
    // held when the lane last looked; a lane whose CAS lost takes the claimer's fingerprint as its v — its own (a duplicate) or
    // another's (walk on).  The memory traffic is the loop's, request for request.  Same box, loop -> steps (profiles/
    // r06_probe_steps.txt): headline 31.8 -> 31.5 ms, config 5 26.2 -> 25.6, the 6.45 G-state stretch 0.683 -> 0.657 s (narrow).
    //
    // A step comes in two halves so that a caller can put other work between the requests and their answers.
    // SH = 1: 16-byte slots of fingerprint + predecessor (KMC_FLAG_PAIRED): slot i's fingerprint is word 2 i, a.pred = a.table + 1.
    template <int SH = 0>
    static KMC_DEV void step_issue(const KmcArgsLocal& a, bool& active, u64 fp, u64& i, u64 v, bool& docas, u64& r_cas, u64& r_load) {
        active = active && v != fp;                 // v == fp: the state is in the table
        docas = active && v == 0;
#if KMC_TUNING
        const bool doload = active && v != 0 && !(a.flags & KMC_FLAG_X_NOWALK);
        if (a.flags & KMC_FLAG_X_NOWALK) active = docas;
#else
        const bool doload = active && v != 0;
#endif
        if (doload) i = kmc_slot_next(i, a.table_cap);
        r_cas = 1;
        r_load = 0;
        if (docas) {
#if KMC_TUNING
            if (a.flags & KMC_FLAG_X_PLAINSTORE) {
                a.table[i << SH] = fp;
                r_cas = 0;
            } else
#endif
            r_cas = atomicCAS(&a.table[i << SH], 0ull, fp);
        }
        if (doload) r_load = a.table[i << SH];
    }
    template <int SH = 0>
    static KMC_DEV bool step_resolve(const KmcArgsLocal& a, bool& active, u64 i, u64 meta, bool docas, u64 r_cas, u64 r_load, u64& v) {
        const bool won = docas && r_cas == 0;
        if (won) {
            active = false;
            if (a.pred) a.pred[i << SH] = meta;
        }
        v = docas ? r_cas : r_load;
        return won;
    }
    // Walks the chain of fp from slot i, of which the lane has seen v (called by the whole wave; the lanes with active = false
    // only take part in the ballots).  True when this lane claimed a slot: the state is new.
    template <int SH = 0>
    static KMC_DEV bool claim_steps(const KmcArgsLocal& a, bool active, u64 fp, u64 i, u64 v, u64 meta) {
        bool won = false;
        const u64 max_probes = a.table_cap - 1 < (1ull << 10) ? a.table_cap - 1 : (1ull << 10);
        for (u64 steps = 0;; ++steps) {
            if (__ballot(active && v != fp) == 0) break;
            if (steps > 2 * max_probes + 2) {           // (a lane sends at most one load and one CAS per slot)
                if (active && v != fp) atomicOr(&a.ctl->err, KMC_ERR_TABLE_FULL);
                break;
            }
            bool docas;
            u64 r_cas, r_load;
            step_issue<SH>(a, active, fp, i, v, docas, r_cas, r_load);
            won = step_resolve<SH>(a, active, i, meta, docas, r_cas, r_load, v) || won;
        }
        return won;
    }

    // The same with 16-byte slots (KMC_FLAG_FP128): word 0 is the fingerprint and is claimed exactly as above; word 1 is a
    // second, independent 64-bit hash of the state, published by the claimer right after its CAS.  Both words sit in the same
    // 128-byte line, so the probe (ONE 16-byte load) moves no more DRAM than the narrow one.  A probe that finds its
    // fingerprint compares the check word: equal -> the same state; different -> a 64-bit collision between two distinct
    // states, which the narrow table would have lost — the probe goes on to the next slot.  A check word that is still 0
    // (the claimer has not published yet, or this XCD's L2 holds the line from before it did) is re-read at the memory side
    // until it appears (rare: two lanes meeting on one new state within a microsecond; waited for inside the step).
    // The publication sits in the straight-line body of the step, ahead of every wait of the same step: written as "store;
    // return true" inside a branch it once ended up in the loop's exit block, which a wave only executes once all its lanes
    // have left the loop — and a lane of the same wave waiting for this very check word never leaves: the first -fp128 run of
    // the headline hung in exactly that way (bounded, so it reported KMC_ERR_CHECK_WORD at level 3).
    // A lane's step is the CAS of the fingerprint word of the slot it saw empty, or the 16-byte load of its next slot.
    static KMC_DEV bool claim_wide_steps(const KmcArgsLocal& a, bool active, u64 fp, u64 chk, u64 i, KmcSlot2 v, u64 meta) {
        bool won = false;
        bool walk = false;   // slot i is known to hold ANOTHER state with this fingerprint
        const u64 max_probes = a.table_cap - 1 < (1ull << 10) ? a.table_cap - 1 : (1ull << 10);
        for (u64 steps = 0;; ++steps) {
            // what the lane knows about slot i: v.x its fingerprint word, v.y its check word (0: not read yet / not published yet)
            if (__ballot(active) == 0) break;
            if (steps > 2 * max_probes + 2) {
                if (active) atomicOr(&a.ctl->err, KMC_ERR_TABLE_FULL);
                break;
            }
            const bool docas = active && !walk && v.x == 0;
            bool same = active && !walk && v.x == fp;   // the slot holds this fingerprint: the check word decides
            const bool doload = active && !docas && !same;
            walk = false;
            if (doload) i = kmc_slot_next(i, a.table_cap);
            u64* slot = a.table + 2 * i;
            u64 r_cas = 1;
            KmcSlot2 r_load = {0, 0};
            if (docas) r_cas = atomicCAS(slot, 0ull, fp);
            if (doload) r_load = *(const KmcSlot2*)slot;   // one 16-byte load
            if (docas && r_cas == 0) {                  // mine: publish the check word before anybody of this wave waits for one
                __hip_atomic_store(slot + 1, chk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.pred) a.pred[i] = meta;
                won = true;
                active = false;
            }
            if (docas && r_cas != 0) {                  // somebody else's claim: its check word must be (re)read
                v.x = r_cas;
                v.y = 0;
                same = r_cas == fp;
            }
            if (doload) v = r_load;
            if (same) {
                u64 y = v.y;
                for (u32 spins = 0; y == 0 && spins <= (1u << 16); ++spins) y = atomicOr(slot + 1, 0ull);
                if (y == 0) atomicOr(&a.ctl->err, KMC_ERR_CHECK_WORD);
                if (y == chk || y == 0) active = false;   // the same state
                else walk = true;                         // a 64-bit collision between two states: walk on
            }
        }
        return won;
    }

    // The narrow, the paired (fingerprint + predecessor) or the wide table, as the handle was opened (a wave-uniform branch); the check word is the same fingerprint
    // function under another seed.  Called by the whole wave: the lanes with valid = false only take part in the ballots.
    static KMC_DEV bool claim_any(const KmcArgsLocal& a, bool valid, const u64* t, u64 fp, u64 meta) {
        const u64 i = kmc_slot_of(fp, a.table_cap);
        if (a.flags & KMC_FLAG_FP128) {
            KmcSlot2 v = {0, 0};
            if (valid) v = *(const KmcSlot2*)(a.table + 2 * i);
            return claim_wide_steps(a, valid, fp, kmc_fingerprint<W>(t, a.seed ^ 0x6a09e667f3bcc908ull), i, v, meta);
        }
        if (a.flags & KMC_FLAG_PAIRED) return claim_steps<1>(a, valid, fp, i, valid ? a.table[2 * i] : fp, meta);
        return claim_steps(a, valid, fp, i, valid ? a.table[i] : fp, meta);
    }

    // Sender-side duplicate filter of the sharded path: true when fp was not yet in `set` (and is now).
    // BFS generates every state ~g times; without the filter all g copies cross xGMI.  A full or
    // over-long chain just answers "fresh" (the copy travels, the owner dedups): never wrong.
    static KMC_DEV bool first_time(u64* set, u64 mask, u64 fp) {
        u64 i = (fp >> 17) & mask;  // other bits than the owner's table index
        for (u32 probes = 0; probes < 64; ++probes) {
            u64 v = set[i];
            if (v == 0) {
                v = atomicCAS(&set[i], 0ull, fp);
                if (v == 0) return true;
            }
            if (v == fp) return false;
            i = (i + 1) & mask;
        }
        return true;
    }

    // Invariants are evaluated when a state is EXPANDED (kmc_expand_body), not when it is first
    // claimed: every distinct state is expanded exactly once, its fields are already extracted
    // there, and all 64 lanes hold a state to check.  (Checking winners inside the flush ran the
    // evaluation ~3x per tile with a third of the lanes useful and re-extracted every field.)
    static KMC_DEV void report_violation(const KmcArgsLocal& a, u32 bad, u64 fp, u32 deficit = 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (bad >> k & 1u) {
                atomicAdd(&a.ctl->viol_count[k], 1ull);
                atomicMax(&a.ctl->viol_fp_inv[k], ~fp);
                if (deficit) atomicAdd(&a.ctl->corr_viol[k], (u64)deficit);
            }
    }

    // A violating successor outside the state constraint (models with HAS_CONSTRAINT): it enters no
    // table and no frontier, so it is counted per generation and identified by its fingerprint; the
    // host fetches the state (and a parent) with an ENUM_MATCH pass over the expanded level.
    static KMC_DEV void report_outside_violation(const KmcArgsLocal& a, u32 bad, u64 fp) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (bad >> k & 1u) {
                atomicAdd(&a.ctl->oviol_count[k], 1ull);
                atomicMax(&a.ctl->oviol_fp_inv[k], ~fp);
            }
    }

    // Executed by the whole wave; lanes with valid=false only take part in the ballots.  MODE is a compile-time constant: a
    // kernel holds the code of its own mode only (and takes that mode's argument block, KmcArgsOf<MODE>).
    // KMC_SYMM: t is the orbit representative of the successor `raw` (what is fingerprinted, claimed, staged and shipped);
    // ENUM lists the successor itself with the representative's fingerprint, so that a trace replayed through kmc_successors
    // is a real behaviour whose states are FOUND by the fingerprints of their representatives.  stab = the order of t's
    // stabiliser: it travels with a new state (KmcStager) and gives the orbit's deficit.
    static KMC_DEV u64 fingerprint_of(const u64* t, u64 seed) {
#ifdef KMC_TEST_FP_BITS   // tests only: a fingerprint of that many bits, i.e. collisions on demand (the wide table's check
                          // word keeps its 64 bits) — tests/test_gpu_selfcheck_and_fp128.py
        return kmc_mix64((kmc_fingerprint<W>(t, seed) & ((1ull << (KMC_TEST_FP_BITS)) - 1)) + 0x9E3779B97F4A7C15ull) | 1ull;
#else
        return kmc_fingerprint<W>(t, seed);
#endif
    }
    template <u32 MODE>
    static KMC_DEV void process(const typename KmcArgsOf<MODE>::type& a, KmcStager<W>& out, bool valid, const u64* t, u64 meta,
                                const u64* raw = nullptr, u32 stab = 1) {
        const u64 fp = fingerprint_of(t, a.seed);
        out.account(valid, fp);
        if constexpr (MODE == KMC_MODE_DRY) {
            u64 acc = fp;
            if (valid && (a.flags & KMC_FLAG_DRY_RAND)) {  // ONE load from an unrelated random slot per successor
                acc ^= a.table[kmc_slot_of(kmc_mix64(fp ^ 0xABCDEF12345ull), a.table_cap)];
            } else if (valid && (a.flags & KMC_FLAG_DRY_PROBE)) {  // read-only probe sequence (the table is already full)
                u64 i = kmc_slot_of(fp, a.table_cap);
                for (u64 probes = 0; probes < a.table_cap; ++probes) {
                    const u64 v = a.table[i];
                    acc ^= v;
                    if (v == fp || v == 0) break;
                    i = kmc_slot_next(i, a.table_cap);
                }
                if (a.flags & KMC_FLAG_DRY_INV) acc ^= M::violated(t, a.inv_mask);
                // ~35 % of the probes end in a no-op CAS on the slot they found: the same atomic
                // traffic as the real claims (311 M per 888 M probes) without changing the table
                if ((a.flags & KMC_FLAG_DRY_ATOM) && (fp & 0xFF) < 90) acc ^= atomicCAS(&a.table[i], fp, fp);
            }
            if (valid && acc == 0x1234567) atomicOr(&a.ctl->err, KMC_ERR_ENUM_FULL);  // keeps the work alive
        } else if constexpr (MODE == KMC_MODE_LOCAL) {
            // once any wave has found the table full the level is lost anyway: stop probing so the
            // launch ends promptly instead of walking full chains: k_expand reads the flag once per tile (issued with the
            // frontier loads: -0.7 ms on the headline against a dependent L2 round trip in front of every probe batch) and
            // masks the batch
            // (A per-wave LDS filter of recently resolved fingerprints was tried here to skip
            // duplicate probes: only 4.9 % of the successors hit it — duplicates are not local to
            // a wave — so it was dropped.)
            // (Resolving the successors of one batch that share a fingerprint only once — a per-wave LDS lane map — and walking
            // a probe chain inside its 128-byte line before moving on were measured in round 3 and change nothing:
            // profiles/r03_probe_knobs.txt.)
            const bool isnew = claim_any(a, valid, t, fp, meta);
#if KMC_SYMM
            out.corr_won += isnew ? KmcSymm<M>::deficit(stab) : 0u;
#endif
#if KMC_TUNING
            if (a.flags & KMC_FLAG_X_NOSTAGE) return;
#endif
            out.push(a, isnew, t, stab);
        } else if constexpr (MODE == KMC_MODE_SHARDED) {
            // successors this shard owns take the local path at once (probe, claim, stage): only
            // the (P-1)/P that belong elsewhere travel
            const u32 dst = valid ? kmc_owner(fp, a.nshards) : ~0u;
            const bool isnew = claim_any(a, dst == a.shard, t, fp, meta);
#if KMC_SYMM
            out.corr_won += isnew ? KmcSymm<M>::deficit(stab) : 0u;   // (remote successors are weighed where they are claimed: k_insert)
#endif
            out.push(a, isnew, t, stab);
            // bucket the rest by owner: one wave-aggregated atomicAdd per destination present in this batch
            const u32 sub = blockIdx.x % KMC_SEGS;
            const bool remote = valid && dst != a.shard;
            const bool ship = remote && (a.sent == nullptr || first_time(a.sent, a.sent_mask, fp));
            out.filtered += (u32)__popcll(__ballot(remote && !ship));
            // One atomic round trip per batch, not one per destination: lane d reserves destination d's run.  (A loop of
            // "ballot, leader's atomicAdd, broadcast" per destination put up to P-1 dependent device-scope round trips
            // of ~2 us into every flush: k_expand per shard 8.1 ms at P = 8 for work that takes 4.3 ms locally.)
            u32 my_rank = 0, want = 0;   // this lane's rank among the batch's records for ITS destination; lane d: their number
            for (u32 d = 0; d < a.nshards; ++d) {  // wave-uniform; no memory traffic in here
                const u64 m = __ballot(ship && dst == d);
                if (ship && dst == d) my_rank = kmc_rank_in(m);
                if (kmc_lane() == d) want = (u32)__popcll(m);
            }
            u64 base = 0;
            if (want) base = atomicAdd(&a.ctl->send_count[kmc_lane()][sub].v, (u64)want);   // lanes 0..P-1, all at once
            // every record lane fetches the base of its destination's run from lane `dst`
            // (kmc_pull: the value is only USED by the lanes that ship, but lane `dst` need not be one of them — the pull must
            // run for the whole wave)
            const u32 src = ship ? dst : 0u;
            const u64 run_base = kmc_pull64((int)(src << 2), base);
            if (ship) {
                const u64 pos = run_base + my_rank;
                if (pos < a.send_cap) {
                    u64* rec = a.send + (((u64)dst * KMC_SEGS + sub) * a.send_cap + pos) * (u64)a.rec_words;
#pragma unroll
                    for (int k = 0; k < W; ++k) rec[k] = t[k];
                    if (a.rec_words > (u32)W) rec[W] = meta;
                } else {
                    atomicOr(&a.ctl->err, KMC_ERR_SEND_FULL);
                }
            }
        } else {  // KMC_MODE_ENUM
            static_assert(MODE == KMC_MODE_ENUM, "unknown mode");
            if (valid && (!(a.flags & KMC_FLAG_ENUM_MATCH) || fp == a.match_fp)) {
                const u64 pos = atomicAdd(&a.ctl->enum_count, 1ull);
                if (pos < a.send_cap) {
                    u64* rec = a.send + pos * (u64)(W + 2);
                    const u64* lst = raw ? raw : t;
#pragma unroll
                    for (int k = 0; k < W; ++k) rec[k] = lst[k];
                    rec[W] = fp;
                    rec[W + 1] = meta;
                } else {
                    atomicOr(&a.ctl->err, KMC_ERR_ENUM_FULL);
                }
            }
        }
    }
};
#endif  // !KMC_HOST_EMU
