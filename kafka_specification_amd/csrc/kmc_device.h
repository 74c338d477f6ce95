// kmc_device.h — gfx950 device code of the model checker: the Next-state relations and
// invariants of the Kafka replication specs lowered onto the bit-packed state vector of
// kmc_layout.h, and the BFS level kernels.  Written for wave64 CDNA4 only.
//
// Specialisation: everything is a template over one (model, N, L, R, E[, K]); the host
// engine instantiates exactly one configuration per code object with KMC_INSTANTIATE (via
// hiprtc), so every bit offset, loop bound and action-instance index below is a compile-time
// constant and a packed state lives in W 64-bit registers per lane.
//
// Kernel shape (k_expand): one wavefront lane per frontier state.
//   * coalesced loads of the SoA frontier planes (plane k, state i at fin[k*stride+i]); every
//     field is extracted once; the invariants of the state being expanded are checked here;
//   * pass 1: the guards of all action instances of `Next` (every binding of the specs' \E over
//     replicas / requests) in one straight-line VALU-domain block -> per-lane "enabled" bitset;
//   * pass 2, two forms.  KIND-MAJOR (Kafka models on the replica-major layout of kmc_layout.h: the headline): a
//     wave-uniform walk over the action KINDS; in every leaf each lane applies ITS OWN next enabled binding of that kind
//     (KmcKafka::apply<K>: replicas / request epoch are per-lane run-time values, a field of replica r is "select word r,
//     extract at a compile-time offset") until no lane has one left — 12 leaves per 64-state tile at the headline.
//     INSTANCE-MAJOR (every other model and layout): a walk over the instances; a scalar binary dispatch jumps to the
//     statically specialised effect of each instance some lane enabled (30 leaves per tile at the headline's constants).
//     Either way lanes never diverge on *which* action they apply, and enabled successors are compacted with
//     __ballot + mbcnt prefix ranks into a per-wave LDS ring (SoA, conflict-free): the write-combining stage;
//   * a flush drains exactly 64 successors, one per lane, so the random HBM probes of the
//     fingerprint table always run with a full wave of independent requests in flight:
//     64-bit mix hash -> open-addressed linear probe -> atomicCAS(0 -> fp) claim;
//   * winners wait in a second LDS ring and are appended to the next frontier 64 at a time:
//     one atomicAdd on a per-segment counter and W fully coalesced 512-byte plane stores.
// The seen-set replaces tlc2.tool.fp.FPSet, the frontier arrays replace
// tlc2.tool.queue.StateQueue and this loop replaces tlc2.tool.Worker.run [TLC-recall; TLC is
// not part of /root/reference].
#pragma once
#include "kmc_layout.h"

typedef unsigned long long u64;
typedef unsigned int u32;

// KMC_HOST_EMU (tests/host_emu.cpp only): the model templates below — pure integer code — are also
// compiled by g++ so that the CPU test-suite can run every guard and effect of the device models
// against the oracle without a GPU.  The kernels, the sink and everything wave-level are left out.
#ifdef KMC_HOST_EMU
#define KMC_DEV inline
#define KMC_OPAQUE(x) ((void)0)
#define KMC_OPAQUE_PURE(x) ((void)0)
#else
#define KMC_DEV __device__ __forceinline__
#define KMC_OPAQUE(x) asm volatile("" : "+v"(x))   // opaque redefinition the optimiser may not move or delete
#define KMC_OPAQUE_PURE(x) asm("" : "+v"(x))       // opaque, but deletable when the result is unused
#endif

#define KMC_MODE_LOCAL 0u    // probe/insert the local table, append winners to the next frontier
#define KMC_MODE_SHARDED 1u  // bucket successors by owner(fp) into per-destination send buffers
#define KMC_MODE_ENUM 2u     // write every successor (state, fp, kind | further bindings with this successor << 8) to a list
#define KMC_MODE_DRY 3u      // tuning aid: generate + fingerprint successors, touch no table or frontier

#define KMC_ERR_FRONTIER_FULL 1u
#define KMC_ERR_TABLE_FULL 2u
#define KMC_ERR_SEND_FULL 4u
#define KMC_ERR_ENUM_FULL 8u
#define KMC_ERR_CHECK_WORD 16u  // FP128: a claimed slot's check word never appeared (bounded wait)

#define KMC_FLAG_TRACE 1u

#define KMC_MAX_KINDS 16
#define KMC_MAX_SHARDS 8
#define KMC_QCAP 64   // per-wave output-stager capacity (winners) = the drain granularity: a push that would overflow it
                      // fills it, drains it and stages the rest (KmcStager::push) — half the LDS of a 128-entry ring,
                      // which is what lets 8 blocks (8 waves per SIMD) share a CU's 160 KB
#ifndef KMC_QCAP_WIDE
#define KMC_QCAP_WIDE 64   // ... for states of KMC_QCAP_WIDE_FROM words or more.  A 10-word state (BASELINE config 5) needs
                           // 4 x (10 x 128 + 10 x 64) x 8 = 60 KB of LDS per block: TWO blocks per CU whatever the registers
                           // allow; with 32 it is 50 KB and three fit (measured in round 4: profiles/r04_wide_kernel.txt)
#endif
#define KMC_QCAP_WIDE_FROM 8
// the stager's capacity for a state of W words (host and device agree on it: kmc_expand_lds_bytes)
constexpr int kmc_qcap(int W, int wide = KMC_QCAP_WIDE) { return W >= KMC_QCAP_WIDE_FROM ? wide : KMC_QCAP; }
#define KMC_SEGS 8    // frontier segments, each with its own append counter (block b appends to b % KMC_SEGS)

// tuning knobs (the host may override them per code object through KMC_JIT_DEFINES)
#ifndef KMC_CAS_FIRST
#define KMC_CAS_FIRST 0   // probe with atomicCAS directly instead of load-then-CAS
#endif
#ifndef KMC_PROFILE
#define KMC_PROFILE 0     // 1: per-phase s_memtime accounting into KmcLevelCtl::prof (costs ~10 %)
#endif
#if KMC_PROFILE
#define KMC_T(var) const u64 var = __builtin_amdgcn_s_memtime()
#define KMC_TADD(slot, t0, t1) prof_acc[slot] += (t1) - (t0)
#else
#define KMC_T(var)
#define KMC_TADD(slot, t0, t1)
#endif
#ifndef KMC_SC1_PROBE
#define KMC_SC1_PROBE 0   // 1: probe with agent-scope (L2-bypassing) loads: fewer stale "empty" reads -> fewer lost CASes
#endif
#ifndef KMC_NT_PROBE
#define KMC_NT_PROBE 0    // 1: the seen-set probe is a non-temporal load (`global_load_dwordx2 ... nt`).  Every random 8-byte
                          //    probe fills a whole 128-byte line (profiles/r02_request_size.txt) and the nt flavour alone
                          //    sustains 54.8 G random loads/s against 49.5 G/s — but in the kernel it is SLOWER (42.8 ms
                          //    against 35.5, profiles/r02_nt_sweep.txt): the claim's CAS wants the line the probe has just
                          //    brought into L2 (randbench: load-then-CAS runs 21.7 G pairs/s, a cold CAS 17.3 G/s)
#endif
#if KMC_NT_PROBE
#define KMC_PROBE_LOAD(p) __builtin_nontemporal_load(p)
#else
#define KMC_PROBE_LOAD(p) (*(p))
#endif
#ifndef KMC_NT_FRONTIER
#define KMC_NT_FRONTIER 1 // 1: the frontier planes (read once, written once per level) stream past the caches: nt loads / stores
                          //    (-0.5 ms on the headline, profiles/r02_nt_sweep.txt)
#endif
#if KMC_NT_FRONTIER
#define KMC_FRONTIER_LOAD(p) __builtin_nontemporal_load(p)
#define KMC_FRONTIER_STORE(p, v) __builtin_nontemporal_store((v), (p))
#else
#define KMC_FRONTIER_LOAD(p) (*(p))
#define KMC_FRONTIER_STORE(p, v) (*(p) = (v))
#endif
#ifndef KMC_ERRCHK_TILE
#define KMC_ERRCHK_TILE 1 // 1: the "table already full" early-out reads the error word once per tile (issued with the
                          //    frontier loads) instead of once per flush (a dependent L2 round trip in front of every
                          //    probe batch): -0.7 ms on the headline
#endif
#ifndef KMC_SETPRIO
#define KMC_SETPRIO 1     // 1: a wave raises its issue priority while it fingerprints and probes a batch, so memory
                          //    requests leave early and the other waves' ALU work fills the wait: -0.4 ms
#endif
#ifndef KMC_GUARD_VOLATILE
#define KMC_GUARD_VOLATILE 0  // 1: the old volatile guard asm (kept every guard chain alive in every effect leaf)
#endif
#ifndef KMC_RING_FENCE
#define KMC_RING_FENCE 0  // 1 (diagnostic): an explicit workgroup-scope fence between the LDS ring writes of a push and the
                          //    reads of a flush / drain.  A wave's LDS operations are executed in order, so this must change
                          //    nothing; it exists to rule the cross-lane ring idiom out when results differ (docs/TUNING_LOG_r1-r3.md §2)
#endif
#if KMC_RING_FENCE
#define KMC_FENCE_LDS() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup")
#else
#define KMC_FENCE_LDS() ((void)0)
#endif
#ifdef KMC_CONST_INV_MASK     // tuning / code-size experiments: the checked invariants as a compile-time constant
#define KMC_INV_MASK(a) ((u32)(KMC_CONST_INV_MASK))
#else
#define KMC_INV_MASK(a) ((a).inv_mask)
#endif
#ifndef KMC_CHECKSUM
#define KMC_CHECKSUM 0    // 1: every lane keeps a running sum and xor of the fingerprints it sends into the sink and the level's
                          //    control block receives their totals — the order-independent checksum KMC_VERIFY compares between
                          //    its two builds (both are compiled with it).  Not in the default build: four live VGPRs and a
                          //    wave reduction per launch cost the headline 3 ms when it was always on (profiles/r03_selfcheck_cost.txt)
#endif
#ifndef KMC_FAULT_DROP
#define KMC_FAULT_DROP 0  // 1 (fault injection, tests only): the first flush of block 0 / wave 0 of every LOCAL launch loses the
                          //    successor in lane 5 between the ring and the seen-set — the failure class of round 1's miscompiled
                          //    kernel.  The conservation check (generated = probed) and KMC_VERIFY's checksum must both catch it
#endif
#ifndef KMC_GROUPED_GUARDS_MIN_INSTANCES
#define KMC_GROUPED_GUARDS_MIN_INSTANCES 1000000   // Kafka configurations with MORE action instances than this evaluate pass 1
                                   // group by group (KmcKafka::group_pre): the guards of a leader's bindings only if some lane
                                   // of the tile has that replica presuming leadership, those of a (request, leader) pair only
                                   // if some lane's request of that epoch names that leader — a wave-uniform branch on a
                                   // ballot around each group, the instances' own guards (inst<I>) inside
#endif
#ifndef KMC_RT_GUARDS_MIN_INSTANCES
#define KMC_RT_GUARDS_MIN_INSTANCES 1000000   // Kafka configurations with MORE action instances than this evaluate their guards
                                          // in per-kind loops over a run-time binding (KmcKafka::guard<K>) instead of one
                                          // straight-line block of every instance's guard (inst<I>).  Off by default: the
                                          // loops compile in seconds where the block takes minutes at 7 brokers, but they run
                                          // slower everywhere (headline 38.9 ms against 31.8, config 5's first ten levels 61
                                          // against 29: profiles/r03_runtime_guards.txt) — the block shares its sub-terms
                                          // across instances, a loop cannot.  KMC_VERIFY's second build sets it to 0: its
                                          // guards are then a second, independent lowering (kmc_engine.cpp)
#endif
#ifndef KMC_SYMM
#define KMC_SYMM 0        // 1 (kmc_config.symmetry): symmetry reduction with orbit counting — every successor is replaced by the
                          // representative of its orbit under the permutations of Replicas before it is fingerprinted, and
                          // the level's counters come with the deficits (KmcLevelCtl::corr_*) that turn counts over
                          // representatives into the counts of the plain search (KmcSymm below)
#endif
#ifndef KMC_PREFETCH
#define KMC_PREFETCH 0    // 1: request the next tile's state words while the current tile is processed (measured: no gain)
#endif
// per-wave successor ring capacity: < KMC_FLUSH_N queued before a push, <= 64 pushed at once
#define KMC_FLUSH_N 64
#define KMC_RING 128
#define KMC_FLAG_DRY_PROBE 4u  // tuning: DRY mode also walks the (read-only) probe sequence
#define KMC_FLAG_DRY_INV 8u    // tuning: ... and evaluates the invariants on every successor
#define KMC_FLAG_DRY_ATOM 16u  // tuning: ... and a no-op atomicCAS on ~35 % of the probed slots
#define KMC_FLAG_X_NOSTAGE 32u    // tuning (shadow pass only): winners are not appended
#define KMC_FLAG_X_NOINV 64u      // tuning: skip invariants
#define KMC_FLAG_X_PLAINSTORE 128u  // tuning: claim with a plain store instead of atomicCAS (racy, timing only)
#define KMC_FLAG_DRY_RAND 256u  // tuning: DRY mode does one load from an uncorrelated random table slot
#define KMC_FLAG_ENUM_MATCH 512u  // ENUM lists only the successors whose fingerprint is KmcArgs::match_fp, with their parent's fp
#define KMC_FLAG_META 2u  // the ring carries a meta plane (predecessor fp for traces / kind for ENUM)
#define KMC_FLAG_INV_ONLY 2048u  // the invariant pass over a frontier that is not expanded (the last level under max_levels,
                                 // kmc_check_states): a tile ends after the invariants of its states — no guard, no effect
#define KMC_FLAG_FP128 1024u  // the seen-set's slots are 16 bytes: the fingerprint and a second, independent 64-bit hash of the
                              // state (kmc_config.wide_fingerprint): a 64-bit collision is then recognised, not lost

// A counter alone on its 128-byte line.  Device-scope atomics serialise per cache line at the
// memory side (~90 M/s): eight "separate" 8-byte append counters packed into one 64-byte line were
// still ONE hot spot — adjacent BFS levels of equal size ran 0.24 vs 0.18 ns/state depending only
// on how the two control-block slots happened to straddle a line boundary.
struct alignas(128) KmcCounterLine {
    u64 v;
    u64 pad_[15];
};

// One per BFS level; the host zeroes it before the level runs and reads it back after.
struct alignas(128) KmcLevelCtl {
    KmcCounterLine next_count[KMC_SEGS];  // states appended to each segment of the next frontier
    u64 generated[KMC_MAX_KINDS];    // successors generated per action kind (Next disjunct)
    u64 viol_count[4];               // states of the EXPANDED level violating invariant k
    u64 viol_fp_inv[4];              // max over violators of ~fp  (=> min fp), 0 = none
    u64 deadlock_count;              // expanded states without any successor
    u64 deadlock_fp_inv;
    u64 enum_count;                  // ENUM: records written
    u64 send_filtered;               // SHARDED: remote successors dropped by the sender-side filter
    u64 repeats;                     // of generated[]: successors counted a second time because another disjunct of the same
                                     // binding also holds (models with HAS_EXTRA); they are one successor, probed once
    // Conservation (checked by the host after every level, always on): what pass 2 dispatched must be what reached the sink,
    // and what the sink claimed must be what was appended:   sum(generated) - repeats - outside = probed,   won = appended.
    u64 probed;                      // successors that entered KmcSink::process (valid lanes), k_insert's records included
    u64 won;                         // claims won (new states), counted at the claim; the appends are counted by next_count[]
    u64 outside;                     // successors outside the state constraint (generated, never probed)
    u64 fp_sum, fp_xor;              // order-independent checksum of the probed successors' fingerprints (KMC_VERIFY compares
                                     // it between the two builds of the kernel)
    u64 oviol_count[4];              // successors OUTSIDE the state constraint violating invariant k (per generation)
    u64 oviol_fp_inv[4];             // max over those of ~fp
    u64 prof[8];                     // KMC_PROFILE: summed per-wave s_memtime ticks per phase (tuning aid)
    // KMC_SYMM: a counter x above counts orbit REPRESENTATIVES; the plain search's count is N! * x - corr_x, where corr_x sums
    // N! - |orbit| over the representatives counted (0 for the great majority: a state whose replicas all differ has N! images)
    u64 corr_gen[KMC_MAX_KINDS];     // of generated[k]: summed over (expanded state, enabled binding of kind k) [+ the repeats]
    u64 corr_viol[4];                // of viol_count[k]
    u64 corr_dead;                   // of deadlock_count
    u64 corr_repeats;                // of repeats
    u64 corr_won;                    // of won = the states of the produced level
    u32 err;
    u32 halt;                        // chained launches: this level was not expanded because an earlier one ended the search
    // (everything above is what a single-GPU level reports: the host copies the block only up to here)
    alignas(128) KmcCounterLine send_count[KMC_MAX_SHARDS][KMC_SEGS];  // SHARDED: records bucketed per (destination, sub-buffer)
};
#define KMC_CTL_LOCAL_BYTES (__builtin_offsetof(KmcLevelCtl, send_count))

struct KmcArgs {
    // Frontiers are SoA: word k of the state at slot i lives at f[k*stride + i].  A frontier is
    // KMC_SEGS dense segments; segment s occupies slots [s*seg_cap, s*seg_cap + seg_count[s]).
    const u64* fin;    // current frontier
    u64 fin_stride;    // plane stride in states
    u64 n_in;          // k_insert: number of records
    u64 seg_count[KMC_SEGS];  // k_expand / k_find: states per segment of the current frontier
    u64 seg_cap;       // slots per segment (both frontiers)
    u64* fout;         // next frontier
    u64 fout_stride;
    u64* table;        // open-addressed fingerprint table, 0 = empty
    u64 table_mask;    // capacity-1 (capacity is a power of two)
    u64* pred;         // optional: predecessor fingerprint per table slot (trace reconstruction)
    u64* sent;         // SHARDED, optional: fingerprints already shipped to their (remote) owner
    u64 sent_mask;
    KmcLevelCtl* ctl;
    u64 seed;
    u64* send;         // SHARDED: [shard][KMC_SEGS][send_cap] AoS records of rec_words words (state[, parent fp]);
    u64 send_cap;      //   block b fills sub-buffer b % KMC_SEGS.  ENUM: one list of W+2-word records (state, fp, kind)
    const u64* recv;   // k_insert input: AoS records of rec_words words
    u32 inv_mask;
    u32 mode;
    u32 flags;
    u32 nshards;
    u32 shard;         // this handle's shard id (SHARDED mode keeps its own successors local)
    u32 rec_words;     // exchange record size in words: W, or W+1 when predecessor fingerprints travel (trace)
    u64 match_fp;      // ENUM with KMC_FLAG_ENUM_MATCH: list only successors with this fingerprint (meta = parent fp)
    // Chained launches (kmc_run without a progress callback): the host queues several BFS levels back to back and
    // waits once per batch instead of once per level.  The level then takes its input sizes from the control block of
    // the level that produced `fin`, and does nothing when that level (or one before it) ended the search.
    const KmcLevelCtl* prev;  // null: seg_count[] above is authoritative
    u32 stop_mask;            // invariants whose violation ends the search (0 under -continue)
    u32 stop_deadlock;        // CHECK_DEADLOCK: a state without successors ends the search
};

// ----------------------------------------------------------------------------------------
// small compile-time helpers
// ----------------------------------------------------------------------------------------
template <int V> struct KmcIC { static constexpr int value = V; };
// a replica's whole log as one register value: 32-bit when it fits (integer VALU ops on 64-bit
// values cost two to four times a 32-bit one on this chip), else 64-bit
template <bool FITS32> struct KmcLogWord { using type = u64; };
template <> struct KmcLogWord<true> { using type = u32; };

template <int LO, int HI, class F> KMC_DEV void kmc_static_for(F&& f) {
    if constexpr (LO < HI) {
        f(KmcIC<LO>{});
        kmc_static_for<LO + 1, HI>(f);
    }
}
// wave-uniform binary dispatch of a runtime index onto a compile-time constant
template <int LO, int HI, class F> KMC_DEV void kmc_dispatch(int i, F&& f) {
    if constexpr (HI - LO == 1) {
        f(KmcIC<LO>{});
    } else {
        constexpr int MID = (LO + HI) / 2;
        if (i < MID) kmc_dispatch<LO, MID>(i, f);
        else kmc_dispatch<MID, HI>(i, f);
    }
}

#ifndef KMC_HOST_EMU
KMC_DEV u32 kmc_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
KMC_DEV u32 kmc_rank_in(u64 mask) {  // number of set bits of mask below this lane
    return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
}
KMC_DEV u64 kmc_bcast64(u64 v, int src) {
    u32 lo = __builtin_amdgcn_readlane((u32)v, src), hi = __builtin_amdgcn_readlane((u32)(v >> 32), src);
    return ((u64)hi << 32) | lo;
}
#endif
KMC_DEV u32 kmc_min(u32 a, u32 b) { return a < b ? a : b; }
// does any lane of the wave say so?  (the host emulation runs one state at a time: the lane itself)
KMC_DEV bool kmc_any_lane(bool x) {
#ifndef KMC_HOST_EMU
    return __ballot(x) != 0;
#else
    return x;
#endif
}
// Opaque redefinition: stops LICM from hoisting every action instance's guard/effect out of
// the instance loop (they only depend on the loop-invariant state), which would keep all of
// them live at once and cost the kernel its occupancy.
// Guards are evaluated in the VALU/VGPR domain: g stays an opaque 0/1 integer and every term is
// `cond ? g : 0` (v_cmp + v_cndmask).  Plain bool chains become 64-bit lane masks in SGPRs, and
// sixty guards sharing sub-predicates kept ~50 of those alive at once (127+ SGPR spills, and the
// scalar unit was the busiest pipe of the kernel).
KMC_DEV u32 kmc_and(u32 g, bool c) {
    u32 r = c ? g : 0u;
    // opaque: stops the fold back into select(c1 & c2, ...) = SGPR mask logic.  NOT volatile: pass 2
    // calls inst<I> for the effect only, and a volatile asm kept every (dead) guard chain alive in
    // every effect leaf — 29 % of the leaves' instructions, 1.1 ms of the headline kernel.  A plain
    // asm is just as opaque to the folder but is deleted when its result is unused.
#if KMC_GUARD_VOLATILE
    KMC_OPAQUE(r);
#else
    KMC_OPAQUE_PURE(r);
#endif
    return r;
}
KMC_DEV u32 kmc_bit(u32 m, int k) { return (m >> k) & 1u; }
KMC_DEV u32 kmc_bit64(u64 m, int k) { return (u32)(m >> k) & 1u; }
KMC_DEV void kmc_launder(u32& x) { KMC_OPAQUE(x); }
KMC_DEV void kmc_launder(u64& x) { KMC_OPAQUE(x); }

// 64-bit fingerprint of a packed state.  Never 0 (0 marks an empty table slot).
// Every state word is absorbed through a full-avalanche bijection (the splitmix64 / murmur3
// finaliser: two multiplies, three xor-shifts).  A single multiply + xor-shift per word is NOT
// enough here: packed states are highly structured, differences that survive one weak round
// line up with differences in the next word and produce systematic collisions (seen as 32
// missing states out of 75,569,791 on Kip320 3/5/5/2).
KMC_HD inline u64 kmc_mix64(u64 x) {
    x ^= x >> 30;
    x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27;
    x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}
template <int W> KMC_HD inline u64 kmc_fingerprint(const u64* w, u64 seed) {
    u64 h = kmc_mix64(seed + 0x9E3779B97F4A7C15ull * (u64)(W + 1));
#pragma unroll
    for (int k = 0; k < W; ++k) h = kmc_mix64(h ^ w[k]) + 0x9E3779B97F4A7C15ull;
    return h ? h : 1ull;
}
// owner shard of a fingerprint: its bits 40..63 scaled onto 0..nshards-1 (a multiply and a shift; a run-time `% nshards`
// on a 64-bit value is a ~100-instruction division on this ISA, once per successor)
KMC_HD inline u32 kmc_owner(u64 fp, u32 nshards) { return (u32)((((fp >> 40) & 0xFFFFFFull) * (u64)nshards) >> 24); }

// ========================================================================================
// IdSequence.tla standalone
// ========================================================================================
template <long long MAXID> struct KmcIdSequence {
    static constexpr int W = 1, NKINDS = 1, NINST = 1;
    static constexpr bool HAS_EXTRA = false, HAS_CONSTRAINT = false, KIND_MAJOR = false, RUNTIME_GUARDS = false, GROUPED_GUARDS = false;
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
};

// ========================================================================================
// FiniteReplicatedLog.tla standalone
// ========================================================================================
template <int N, int L, int K> struct KmcFiniteReplicatedLog {
    static constexpr KmcLayout Y = kmc_make_layout(KMC_MODEL_FINITE_REPLICATED_LOG, N, L, 0, 0, K);
    static_assert(Y.valid, "FiniteReplicatedLog parameters cannot be packed");
    static constexpr int W = Y.W, NKINDS = 3;
    static constexpr bool HAS_EXTRA = false, HAS_CONSTRAINT = false, KIND_MAJOR = false, RUNTIME_GUARDS = false, GROUPED_GUARDS = false;
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
    static constexpr bool HAS_EXTRA = false, HAS_CONSTRAINT = true, KIND_MAJOR = false, RUNTIME_GUARDS = false, GROUPED_GUARDS = false;
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
};

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
    // ---- guard groups (pass 1 of the wide configurations, KMC_GROUPED_GUARDS_MIN_INSTANCES) ---------------------------
    // The instances of Next fall into groups that share a cheap NECESSARY condition:
    //   group 0                 kinds 0, 1, 5, 6 (one binding per replica: 4N instances), always evaluated;
    //   group 1 + e*N + l       the bindings that act on the LeaderAndIsr request with epoch e naming leader l — BecomeLeader
    //                           (e, l) and BecomeFollower* (l, r, e) for every r # l: all need that request to exist and to name
    //                           l (KafkaReplication.tla:186-188, :281-284; Kip320.tla:134-137);
    //   group 1 + (E+1)N + l    the bindings a PRESUMED leader l takes part in — Leader*ExpandIsr (l, r), Leader*ShrinkIsr (l, r),
    //                           FollowerReplicate / *Fetch (l, f), FollowerTruncate (l, f): all need ReplicaPresumesLeadership(l)
    //                           (:126; IsTrueLeader :128-131, IsFollowingLeaderEpoch Kip320.tla:39-42, FollowerReplicate
    //                           KafkaReplication.tla:302-304, IsFollowerCaughtUpToLeaderEpoch Kip320FirstTry.tla:49-51).
    // A tile's 64 states are neighbours in the frontier (children of neighbouring parents): most groups are dead for the whole
    // wave.  tests/host_emu.cpp holds "inst<I> enabled => group_pre<group of I>" on every instance of every visited state and
    // the partition of 0..NINST-1 into the groups' lists at compile time.
    static constexpr bool GROUPED_GUARDS = KIND_MAJOR && !RUNTIME_GUARDS && NINST > KMC_GROUPED_GUARDS_MIN_INSTANCES;
    static constexpr int NGROUPS = 1 + (E + 1) * N + N;
    static constexpr int G_REQ0 = 1, G_LDR0 = 1 + (E + 1) * N;
    static constexpr int group_size(int g) {
        return g == 0 ? 4 * N : g < G_LDR0 ? N : N + (N - 1) * (FIRST ? 3 : 2);
    }
    static constexpr int group_inst(int g, int j) {   // the j-th instance of group g
        if (g == 0) return j < N ? B0 + j : j < 2 * N ? B1 + (j - N) : j < 3 * N ? B5 + (j - 2 * N) : B6 + (j - 3 * N);
        if (g < G_LDR0) {
            const int e = (g - G_REQ0) / N, l = (g - G_REQ0) % N;
            if (j == 0) return B2 + e * N + l;
            return B7 + (l * (N - 1) + (j - 1)) * (E + 1) + e;
        }
        const int l = g - G_LDR0;
        if (j < N) return B3 + l * N + j;
        j -= N;
        if (j < N - 1) return B4 + l * (N - 1) + j;
        j -= N - 1;
        if (j < N - 1) return B8 + l * (N - 1) + j;
        j -= N - 1;
        return B9 + l * (N - 1) + j;
    }
    static constexpr int group_of(int i) {
        for (int g = 0; g < NGROUPS; ++g)
            for (int j = 0; j < group_size(g); ++j)
                if (group_inst(g, j) == i) return g;
        return -1;
    }
    static constexpr bool groups_partition_the_instances() {
        int total = 0;
        for (int g = 0; g < NGROUPS; ++g) total += group_size(g);
        if (total != NINST) return false;
        for (int i = 0; i < NINST; ++i)
            if (group_of(i) < 0) return false;
        return true;
    }
    template <int G> static KMC_DEV u32 group_pre(const Pre& p) {
        if constexpr (G == 0) {
            return 1u;
        } else if constexpr (G < G_LDR0) {
            constexpr int e = (G - G_REQ0) / N, l = (G - G_REQ0) % N;
            return (p.rldr1(e) == (u32)(l + 1) && (u32)e < p.nextEp()) ? 1u : 0u;
        } else {
            return kmc_bit(p.pm, G - G_LDR0);
        }
    }
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

    static KMC_DEV u32 violated_pre(const Pre& p, u32 inv_mask) {
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

// ========================================================================================
// Symmetry reduction with orbit counting (kmc_config.symmetry; the kernels use it in KMC_SYMM builds)
// ========================================================================================
// The specs quantify over Replicas and never tell two of them apart (KafkaReplication.tla:109-120, :158-310; the five
// modules' own actions: `\E leader, replica \in Replicas`), so the N! permutations of Replicas are automorphisms of the state
// graph: they map Init to Init, successors to successors (binding by binding, so also the per-disjunct "generated" counts
// and the doubly satisfied disjuncts), and keep every invariant and the BFS depth.  The search therefore only stores and
// expands ONE state per orbit — the smallest image under the N! permutations, words compared in order — and every count it
// reports is weighted by the orbit's size N! / |stabiliser|: distinct states, states per level, generated per disjunct,
// deadlocks and violating states all come out as the numbers of the plain search (and of TLC without SYMMETRY), from
// ~1/N! of the probes.  (TLC's own SYMMETRY reports the REDUCED counts — SURVEY.md rules that out; this does not change them.)
// permute<P> is the compile-time form of kmc_permute_state (kmc_layout.h): fields move between compile-time offsets.  The
// replica ids inside a state sit in (leader, isr) PAIRS — one per replica, one in quorumState, one per LeaderAndIsr request
// — and a pair's images under ALL the permutations come from one table lookup (LDS in the kernels): entry idx = leader |
// isr << BL holds the renamed pair for PER permutations per 32-bit word (all six at three replicas), so that a permutation
// costs one bit-field extract and one insert per pair instead of two shift-indexed constant lookups each.
template <class M> struct KmcSymm {
    static constexpr KmcLayout Y = M::Y;
    static constexpr int N = Y.N, W = Y.W;
    static constexpr bool KAFKA = Y.model != KMC_MODEL_FINITE_REPLICATED_LOG;
    static constexpr int NFACT = kmc_factorial(N);
    static_assert(kmc_model_symmetric(Y.model), "this model singles out a replica: no symmetry reduction");
    static_assert(N <= KMC_SYMM_MAX_REPLICAS, "orbit counting: the walk through all images (canon_sorted's last resort) is a table of N! - 1 steps");
    static constexpr bool UNROLLED = N <= KMC_SYMM_UNROLLED_MAX;   // N! - 1 statically specialised permutations; beyond: the sorted images (canon_sorted)
    static constexpr int PB = KAFKA ? Y.BL + Y.BI : 1;       // bits of a (leader, isr) pair: 5 at N = 3, 7 at N = 4
    static constexpr int PER = 32 / PB;                      // images per table word
    static constexpr int NG = (NFACT + PER - 1) / PER;       // table words per pair value
    // (five and six replicas: the table holds, per adjacent transposition a <-> a + 1, the image of every pair value)
    static constexpr int TABLE_WORDS = !KAFKA ? 1 : UNROLLED ? (NG << PB) : ((N - 1) << PB);
    static constexpr int NPAIR = KAFKA ? N + 1 + (Y.E + 1) : 0;
    static constexpr u32 MP = (1u << PB) - 1, ML = (1u << Y.BL) - 1;

    // the pair idx with every replica in it renamed by permutation P (leader: 0 = None or index + 1; isr: a bit mask)
    static KMC_HD constexpr u32 pair_image(int P, u32 idx) {
        const u32 l = idx & ML, m = idx >> Y.BL;
        const u32 pl = (l == 0 || l > (u32)N) ? l : (u32)kmc_perm_image(N, P, (int)l - 1) + 1;
        u32 pm = 0;
        for (int i = 0; i < N; ++i)
            if (m >> i & 1u) pm |= 1u << kmc_perm_image(N, P, i);
        return pl | pm << Y.BL;
    }
    // the pair idx with the names of replicas a and a + 1 exchanged
    static KMC_HD constexpr u32 exchange_image(int a, u32 idx) {
        const u32 l = idx & ML, m = idx >> Y.BL;
        const u32 pl = l == (u32)a + 1 ? l + 1 : l == (u32)a + 2 ? l - 1 : l;
        const u32 y = ((m >> a) ^ (m >> (a + 1))) & 1u;
        return pl | ((m ^ (y << a) ^ (y << (a + 1))) << Y.BL);
    }
    // word i of the table: the images of pair (i & MP) under permutations (i >> PB) * PER ... + PER - 1, PB bits each
    static KMC_HD constexpr u32 table_word(int i) {
        const u32 idx = (u32)i & MP;
        const int g = i >> PB;
        u32 w = 0;
        for (int j = 0; j < PER; ++j)
            if (g * PER + j < NFACT) w |= pair_image(g * PER + j, idx) << (j * PB);
        return w;
    }
    struct Table { u32 w[TABLE_WORDS]; };
    static constexpr Table make_table() {
        Table t{};
        for (int i = 0; i < TABLE_WORDS; ++i) t.w[i] = !KAFKA ? 0u : UNROLLED ? table_word(i) : exchange_image(i >> PB, (u32)i & MP);
        return t;
    }
    static constexpr Table TABLE = make_table();   // (constant memory; k_expand copies it to LDS once per block)
    // pair f: replica f for f < N, quorumState for f = N, the request of epoch f - N - 1 beyond
    static constexpr int pair_ldr_off(int f) { return f < N ? Y.ldr_off[f] : f == N ? Y.qldr_off : Y.reqldr_off[f - N - 1]; }
    static constexpr int pair_isr_off(int f) { return f < N ? Y.isr_off[f] : f == N ? Y.qisr_off : Y.reqisr_off[f - N - 1]; }
    static constexpr bool pair_adjacent(int f) { return pair_isr_off(f) == pair_ldr_off(f) + Y.BL; }
    // the global fields no permutation touches (nextRecordId, nextLeaderEpoch, quorumState.leaderEpoch), as a mask of word k
    static constexpr u64 keep_mask(int k) {
        u64 m = 0;
        if (!KAFKA) return m;
        const int off[3] = {Y.nextrec_off, Y.nextep_off, Y.qep_off}, bits[3] = {Y.BNR, Y.BE, Y.BE};
        for (int f = 0; f < 3; ++f)
            for (int b = off[f]; b < off[f] + bits[f]; ++b)
                if ((b >> 6) == k) m |= 1ull << (b & 63);
        return m;
    }

    // the images of a state's pairs under every permutation: NPAIR x NG table reads, once per state
    struct Prep { u32 img[NPAIR > 0 ? NPAIR : 1][NG]; };
    static KMC_DEV void prepare(const u64* s, const u32* tab, Prep& p) {
        kmc_static_for<0, NPAIR>([&](auto FF) {
            constexpr int f = decltype(FF)::value;
            u32 idx;
            if constexpr (pair_adjacent(f)) idx = (u32)kmc_getbits(s, pair_ldr_off(f), PB);
            else idx = (u32)kmc_getbits(s, pair_ldr_off(f), Y.BL) | ((u32)kmc_getbits(s, pair_isr_off(f), Y.BI) << Y.BL);
#pragma unroll
            for (int g = 0; g < NG; ++g) p.img[f][g] = tab[(g << PB) | idx];
        });
    }
    template <int P> static KMC_DEV void permute(const u64* s, const Prep& p, u64* t) {
        kmc_static_for<0, W>([&](auto KK) {
            constexpr int k = decltype(KK)::value;
            t[k] = s[k] & keep_mask(k);
        });
        kmc_static_for<0, N>([&](auto RR) {
            constexpr int r = decltype(RR)::value, d = kmc_perm_image(N, P, r);
            kmc_orbits(t, Y.log_off[d], Y.BR * Y.L, kmc_getbits(s, Y.log_off[r], Y.BR * Y.L));
            if constexpr (!KAFKA) {
                kmc_orbits(t, Y.end_off[d], Y.BO, kmc_getbits(s, Y.end_off[r], Y.BO));
            } else {
                // end | hw | ep | ldr | isr are adjacent in every arrangement of the state vector (kmc_layout.h)
                constexpr int GB = 2 * Y.BO + Y.BE;
                static_assert(Y.hw_off[r] == Y.end_off[r] + Y.BO && Y.ep_off[r] == Y.hw_off[r] + Y.BO &&
                              Y.ldr_off[r] == Y.ep_off[r] + Y.BE && Y.isr_off[r] == Y.ldr_off[r] + Y.BL, "small group not contiguous");
                kmc_orbits(t, Y.end_off[d], GB, kmc_getbits(s, Y.end_off[r], GB));
                kmc_orbits(t, Y.ldr_off[d], PB, (p.img[r][P / PER] >> ((P % PER) * PB)) & MP);
            }
        });
        kmc_static_for<N, NPAIR>([&](auto FF) {
            constexpr int f = decltype(FF)::value;
            const u32 pi = (p.img[f][P / PER] >> ((P % PER) * PB)) & MP;
            if constexpr (pair_adjacent(f)) {
                kmc_orbits(t, pair_ldr_off(f), PB, pi);
            } else {
                kmc_orbits(t, pair_ldr_off(f), Y.BL, pi & ML);
                kmc_orbits(t, pair_isr_off(f), Y.BI, pi >> Y.BL);
            }
        });
    }
    // ---- five and six replicas: 119 / 719 statically specialised permutations are too much code, so the images are visited
    // one ADJACENT TRANSPOSITION at a time (Steinhaus-Johnson-Trotter: every permutation exactly once, consecutive ones differ
    // by exchanging two neighbouring replicas): a wave-uniform loop whose body dispatches to one of N - 1 specialised
    // "exchange replicas a and a + 1" steps working in place on the current image.
    static constexpr int NSTEPS = NFACT - 1;
    // at[k] = a: step k exchanges the replicas at positions a and a + 1; cross[k]: bit b set when the arrangement after step k
    // has moved some replica across the boundary between positions b and b + 1 (it then mixes two runs of the sorted order)
    struct Seq { unsigned char at[NSTEPS > 0 ? NSTEPS : 1], cross[NSTEPS > 0 ? NSTEPS : 1]; };
    static constexpr Seq make_sequence() {
        Seq q{};
        int perm[KMC_MAXN] = {}, dir[KMC_MAXN] = {};
        for (int i = 0; i < N; ++i) { perm[i] = i; dir[i] = -1; }
        for (int k = 0; k < NSTEPS; ++k) {
            int mp = -1, mv = -1;    // the largest element that can move in its direction past a smaller one
            for (int pos = 0; pos < N; ++pos) {
                const int v = perm[pos], np = pos + dir[v];
                if (np >= 0 && np < N && perm[np] < v && v > mv) { mv = v; mp = pos; }
            }
            const int np = mp + dir[mv];
            q.at[k] = (unsigned char)(mp < np ? mp : np);
            const int x = perm[mp]; perm[mp] = perm[np]; perm[np] = x;
            for (int v = mv + 1; v < N; ++v) dir[v] = -dir[v];
            unsigned c = 0;
            for (int b = 0; b + 1 < N; ++b)
                for (int pos = 0; pos <= b; ++pos)
                    if (perm[pos] > b) c |= 1u << b;
            q.cross[k] = (unsigned char)c;
        }
        return q;
    }
    static constexpr Seq SEQUENCE = make_sequence();
    // the image of t under the transposition of replicas A and A + 1, in place; `on` = 0 leaves this lane's t as it is
    // (everything below is an XOR of a difference: a lane that does not take part XORs zeros)
    template <int A> static KMC_DEV void exchange(u64* t, const u32* tab, u64 on = ~0ull) {
        constexpr int B = A + 1, LB = Y.BR * Y.L;
        {
            const u64 d = (kmc_getbits(t, Y.log_off[A], LB) ^ kmc_getbits(t, Y.log_off[B], LB)) & on;
            kmc_xorbits(t, Y.log_off[A], LB, d);
            kmc_xorbits(t, Y.log_off[B], LB, d);
        }
        if constexpr (!KAFKA) {
            const u64 d = (kmc_getbits(t, Y.end_off[A], Y.BO) ^ kmc_getbits(t, Y.end_off[B], Y.BO)) & on;
            kmc_xorbits(t, Y.end_off[A], Y.BO, d);
            kmc_xorbits(t, Y.end_off[B], Y.BO, d);
        } else {
            constexpr int SB = 2 * Y.BO + Y.BE + Y.BL + Y.BI;   // end | hw | ep | ldr | isr, adjacent in every arrangement
            static_assert(Y.isr_off[A] == Y.end_off[A] + SB - Y.BI && Y.isr_off[B] == Y.end_off[B] + SB - Y.BI, "small group not contiguous");
            const u64 d = (kmc_getbits(t, Y.end_off[A], SB) ^ kmc_getbits(t, Y.end_off[B], SB)) & on;
            kmc_xorbits(t, Y.end_off[A], SB, d);
            kmc_xorbits(t, Y.end_off[B], SB, d);
            // ... and the two names trade places in every (leader, isr) pair (leader values A + 1 <-> B + 1, isr bits A <-> B):
            // one table read per pair
            kmc_static_for<0, NPAIR>([&](auto FF) {
                constexpr int f = decltype(FF)::value;
                if constexpr (pair_adjacent(f)) {
                    const u32 idx = (u32)kmc_getbits(t, pair_ldr_off(f), PB);
                    kmc_xorbits(t, pair_ldr_off(f), PB, (idx ^ tab[(A << PB) | idx]) & (u32)on);
                } else {
                    const u32 idx = (u32)kmc_getbits(t, pair_ldr_off(f), Y.BL) | ((u32)kmc_getbits(t, pair_isr_off(f), Y.BI) << Y.BL);
                    const u32 x = (idx ^ tab[(A << PB) | idx]) & (u32)on;
                    kmc_xorbits(t, pair_ldr_off(f), Y.BL, x & ML);
                    kmc_xorbits(t, pair_isr_off(f), Y.BI, x >> Y.BL);
                }
            });
        }
    }
    // visits every image of s: MINIMISE keeps the smallest in c and counts how often it occurs (= the stabiliser's order);
    // otherwise c stays s and the images equal to s are counted.  `runs`: bit b set = the replicas at positions b and b + 1 of
    // s may trade places (canon_sorted: their keys are equal); an image whose arrangement crosses any other boundary is passed
    // over.  All ones: every image counts.
    template <bool MINIMISE> static KMC_DEV u32 walk(const u64* s, const u32* tab, u64* c, u32 runs = ~0u) {
        u64 t[W];
#pragma unroll
        for (int k = 0; k < W; ++k) { t[k] = s[k]; c[k] = s[k]; }
        u32 n = 1;
#pragma clang loop unroll(disable)
        for (int k = 0; k < NSTEPS; ++k) {
            int a = SEQUENCE.at[k];
            u32 crossed = SEQUENCE.cross[k];
#ifndef KMC_HOST_EMU
            a = __builtin_amdgcn_readfirstlane(a);   // the step index is wave-uniform: keep the dispatch scalar
            crossed = __builtin_amdgcn_readfirstlane(crossed);
#endif
            kmc_dispatch<0, (N > 1 ? N - 1 : 1)>(a, [&](auto AA) { exchange<decltype(AA)::value>(t, tab); });
            const bool counts = (crossed & ~runs) == 0;
            bool lt = false, eq = true;
#pragma unroll
            for (int q = 0; q < W; ++q) {
                lt = lt || (eq && t[q] < c[q]);
                eq = eq && t[q] == c[q];
            }
            lt = lt && counts;
            eq = eq && counts;
            if constexpr (MINIMISE) {
                n = lt ? 1u : n + (eq ? 1u : 0u);
#pragma unroll
                for (int q = 0; q < W; ++q) c[q] = lt ? t[q] : c[q];
            } else {
                n += eq ? 1u : 0u;
            }
        }
        return n;
    }

    // ---- five and six replicas: the representative among the SORTED images -----------------------------------------------
    // Walking through all 120 / 720 images of every successor is what the orbit-counting search spent its time on at five
    // brokers.  So the representative is chosen among far fewer: every replica gets a KEY that does not depend on how the
    // replicas are named — its log, end offset, high watermark and epoch; whether it names itself / nobody as leader, whether
    // its ISR holds itself and how many it holds; whether quorumState and each LeaderAndIsr request name it as leader / in the
    // ISR; how many OTHER replicas hold it in their ISR / name it as leader — and the representative of an orbit is its
    // smallest image (words in order, as before) AMONG THE IMAGES WHOSE KEYS ASCEND WITH THE POSITION.  Renaming permutes the
    // keys with the replicas, so every state of an orbit sees the same set of sorted images: a representative all the same.
    //   1. sort: an odd-even transposition network of N (N - 1) / 2 conditional exchanges of neighbours (exchange<A> with a
    //      lane mask; the keys trade places with the replicas);
    //   2. where neighbours' keys are equal, the sorted image is one of several.  Almost always exchanging such neighbours
    //      gives the SAME state (two followers nobody tells apart): when that holds at every tied boundary, the tied runs
    //      generate the stabiliser — a permutation that fixes the state keeps every key where it is — the sorted image is
    //      unique, and |Stab| = the product of the run lengths' factorials;
    //   3. otherwise (a tie between replicas that ARE told apart by something the key does not see: not met in 1.8 M
    //      successors of BASELINE config 4 and of the headline, tools/tie_stats.py — but nothing rests on that) the wave walks
    //      through all the images of the sorted one and keeps the smallest of those that only move replicas inside tied runs.
    struct Key { u64 a, b; };
    template <int r> static KMC_DEV Key key_at(const u64* t) {
        Key k;
        k.a = kmc_getbits(t, Y.log_off[r], Y.BR * Y.L);
        if constexpr (!KAFKA) {
            k.b = kmc_getbits(t, Y.end_off[r], Y.BO);
        } else {
            constexpr int GB = 2 * Y.BO + Y.BE;   // end | hw | ep: adjacent in every arrangement (static_assert in exchange)
            static_assert(GB + 14 + 2 * (Y.E + 1) <= 64, "replica key does not fit 64 bits");
            const u32 ldr = (u32)kmc_getbits(t, Y.ldr_off[r], Y.BL), isr = (u32)kmc_getbits(t, Y.isr_off[r], Y.BI);
            const u32 qldr = (u32)kmc_getbits(t, Y.qldr_off, Y.BL), qisr = (u32)kmc_getbits(t, Y.qisr_off, Y.BI);
            u32 f = (ldr == (u32)r + 1u ? 1u : 0u) | ((isr >> r & 1u) << 1) | ((ldr == 0u ? 1u : 0u) << 2) |
                    ((qldr == (u32)r + 1u ? 1u : 0u) << 3) | ((qisr >> r & 1u) << 4) | ((u32)__builtin_popcount(isr) << 5);
            u32 held = 0, named = 0;   // by the other replicas
            kmc_static_for<0, N>([&](auto OO) {
                constexpr int o = decltype(OO)::value;
                if constexpr (o != r) {
                    held += (u32)kmc_getbits(t, Y.isr_off[o], Y.BI) >> r & 1u;
                    named += (u32)kmc_getbits(t, Y.ldr_off[o], Y.BL) == (u32)r + 1u ? 1u : 0u;
                }
            });
            f |= held << 8 | named << 11;
            kmc_static_for<0, Y.E + 1>([&](auto EE) {
                constexpr int e = decltype(EE)::value;
                f |= ((u32)kmc_getbits(t, Y.reqldr_off[e], Y.BL) == (u32)r + 1u ? 1u : 0u) << (14 + 2 * e);
                f |= ((u32)kmc_getbits(t, Y.reqisr_off[e], Y.BI) >> r & 1u) << (15 + 2 * e);
            });
            k.b = kmc_getbits(t, Y.end_off[r], GB) | (u64)f << GB;
        }
        return k;
    }
    static KMC_DEV void canon_sorted(const u64* s, const u32* tab, u64* c, u32& stab) {
        u64 t[W];
#pragma unroll
        for (int k = 0; k < W; ++k) t[k] = s[k];
        Key key[N];
        kmc_static_for<0, N>([&](auto RR) { key[decltype(RR)::value] = key_at<decltype(RR)::value>(t); });
        kmc_static_for<0, N>([&](auto II) {
            kmc_static_for<0, (N - 1 - decltype(II)::value % 2 + 1) / 2>([&](auto JJ) {
                constexpr int a = decltype(II)::value % 2 + 2 * decltype(JJ)::value;
                if constexpr (a + 1 < N) {
                    const bool sw = key[a + 1].a < key[a].a || (key[a + 1].a == key[a].a && key[a + 1].b < key[a].b);
                    if (kmc_any_lane(sw)) {
                        exchange<a>(t, tab, sw ? ~0ull : 0ull);
                        const Key lo = key[a], hi = key[a + 1];
                        key[a] = sw ? hi : lo;
                        key[a + 1] = sw ? lo : hi;
                    }
                }
            });
        });
        // tied neighbours: is trading them the identity on t?
        u32 runs = 0, told_apart = 0;
        kmc_static_for<0, N - 1>([&](auto AA) {
            constexpr int a = decltype(AA)::value;
            const bool tie = key[a].a == key[a + 1].a && key[a].b == key[a + 1].b;
            if (kmc_any_lane(tie)) {
                u64 u[W];
#pragma unroll
                for (int k = 0; k < W; ++k) u[k] = t[k];
                exchange<a>(u, tab, tie ? ~0ull : 0ull);
                bool same = true;
#pragma unroll
                for (int k = 0; k < W; ++k) same = same && u[k] == t[k];
                runs |= tie ? 1u << a : 0u;
                told_apart |= (tie && !same) ? 1u : 0u;
            }
        });
        if (kmc_any_lane(told_apart != 0)) {   // (every lane takes the walk's answer: where nothing is told apart it is the same)
            stab = walk<true>(t, tab, c, runs);
            return;
        }
#pragma unroll
        for (int k = 0; k < W; ++k) c[k] = t[k];
        u32 n = 1, len = 1;
#pragma unroll
        for (int a = 0; a + 1 < N; ++a) {
            len = (runs >> a & 1u) ? len + 1u : 1u;
            n *= len;
        }
        stab = n;
    }

    // c = the orbit's representative (the smallest image, word 0 first; beyond KMC_SYMM_UNROLLED_MAX replicas among the sorted
    // images), stab = the permutations that fix s
    // (the unrolled forms sit in `else` branches: N! - 1 instantiations of permute<P> must not even be attempted at 7 replicas)
    static KMC_DEV void canon(const u64* s, const u32* tab, u64* c, u32& stab) {
        if constexpr (!UNROLLED) {
            canon_sorted(s, tab, c, stab);
        } else {
            Prep p;
            prepare(s, tab, p);
#pragma unroll
            for (int k = 0; k < W; ++k) c[k] = s[k];
            u32 n = 1;
            kmc_static_for<1, NFACT>([&](auto PP) {
                u64 t[W];
                permute<decltype(PP)::value>(s, p, t);
                bool lt = false, eq = true;
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    lt = lt || (eq && t[k] < c[k]);
                    eq = eq && t[k] == c[k];
                }
                n = lt ? 1u : n + (eq ? 1u : 0u);
#pragma unroll
                for (int k = 0; k < W; ++k) c[k] = lt ? t[k] : c[k];
            });
            stab = n;
        }
    }
    static KMC_DEV u32 stabiliser(const u64* s, const u32* tab) {
        if constexpr (!UNROLLED) {
            u64 c[W];
            return walk<false>(s, tab, c);
        } else {
            Prep p;
            prepare(s, tab, p);
            u32 n = 1;
            kmc_static_for<1, NFACT>([&](auto PP) {
                u64 t[W];
                permute<decltype(PP)::value>(s, p, t);
                bool eq = true;
#pragma unroll
                for (int k = 0; k < W; ++k) eq = eq && t[k] == s[k];
                n += eq ? 1u : 0u;
            });
            return n;
        }
    }
    // what a state of stabiliser order `stab` lacks to a full orbit: N! - N!/stab (0 for almost every state)
    static KMC_DEV u32 deficit(u32 stab) { return stab == 1 ? 0u : (u32)NFACT - (u32)NFACT / stab; }
};

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
    static constexpr int QCAP = kmc_qcap(W);
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

    KMC_DEV void drain(const KmcArgs& a, u32 n) {  // the n <= 64 staged states -> next frontier
        const u32 lane = kmc_lane();
        const u32 seg = blockIdx.x % KMC_SEGS;
        KMC_FENCE_LDS();
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
    KMC_DEV void push(const KmcArgs& a, bool isnew, const u64* t, u32 tag = 0) {
        const u64 m = __ballot(isnew);
        if (m == 0) return;
        const u32 n = __popcll(m);
        won += n;
        const u32 rank = kmc_rank_in(m);
        if constexpr (QCAP < 64) {
            // a batch may hold more winners than the stager: fill, drain, fill ... (wave-uniform; at most 64 / QCAP + 1 rounds)
            u32 done = 0;
            for (;;) {
                const u32 room = (u32)QCAP - count;   // >= 1
                const u32 take = n - done < room ? n - done : room;
                if (isnew && rank >= done && rank < done + take) {
#pragma unroll
                    for (int k = 0; k < W; ++k) planes[k * QCAP + count + (rank - done)] = t[k];
                    if constexpr (PL > W) planes[W * QCAP + count + (rank - done)] = tag;
                }
                count += take;
                done += take;
                if (count == (u32)QCAP) drain(a, QCAP);
                if (done == n) return;
            }
        } else {
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
    }
    KMC_DEV void finish(const KmcArgs& a, bool publish_counters = true) {
        if (count) drain(a, count);
        if (filtered && kmc_lane() == 0) atomicAdd(&a.ctl->send_filtered, (u64)filtered);
        filtered = 0;
        if (!publish_counters) return;   // k_expand folds them into its per-block tail (kmc_expand_body)
#if KMC_SYMM
        {
            u32 x = corr_won;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
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

    // probe/insert fp; returns true when this lane claimed the slot (the state is new)
    static KMC_DEV bool claim(const KmcArgs& a, u64 fp, u64 meta) { return claim_from(a, fp, fp & a.table_mask, meta); }

    // The same with 16-byte slots (KMC_FLAG_FP128): word 0 is the fingerprint and is claimed exactly as above; word 1 is a
    // second, independent 64-bit hash of the state, published by the claimer right after its CAS.  Both words sit in the same
    // 128-byte line, so the probe (ONE 16-byte load) moves no more DRAM than the narrow one.  A probe that finds its
    // fingerprint compares the check word: equal -> the same state; different -> a 64-bit collision between two distinct
    // states, which the narrow table would have lost — the probe goes on to the next slot.  A check word that is still 0
    // (the claimer has not published yet, or this XCD's L2 holds the line from before it did) is re-read at the memory side
    // (an atomic, like the claim itself) until it appears; the claimer's store precedes every wait in program order, so two
    // lanes of one wave cannot wait on each other.
    static KMC_DEV bool claim_wide(const KmcArgs& a, u64 fp, u64 chk, u64 meta) {
        u64 i = fp & a.table_mask;
        const u64 max_probes = a.table_mask < (1ull << 10) ? a.table_mask : (1ull << 10);
        for (u64 probes = 0; probes <= max_probes; ++probes) {
            u64* slot = a.table + 2 * i;
            const KmcSlot2 v = *(const KmcSlot2*)slot;   // one 16-byte load
            u64 v0 = v.x, v1 = v.y;
            bool mine = false;
            if (v0 == 0) {
                v0 = atomicCAS(slot, 0ull, fp);
                mine = v0 == 0;
                v1 = 0;  // somebody else's claim: its check word must be (re)read
            }
            // The publication sits HERE, in the straight-line body of the iteration and ahead of every wait below.  Written
            // as "store; return true" inside the branch above it ended up in the loop's exit block, which a wave only
            // executes once all its lanes have left the loop — and a lane of the same wave waiting for this very check
            // word never leaves: the first -fp128 run of the headline hung in exactly that way (bounded, so it reported
            // KMC_ERR_CHECK_WORD at level 3).
            if (mine) {
                __hip_atomic_store(slot + 1, chk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.pred) a.pred[i] = meta;
            }
            const bool wait = !mine && v0 == fp;
            if (wait) {
                for (u32 spins = 0; v1 == 0 && spins <= (1u << 16); ++spins) v1 = atomicOr(slot + 1, 0ull);
                if (v1 == 0) atomicOr(&a.ctl->err, KMC_ERR_CHECK_WORD);
            }
            if (mine) return true;
            if (wait && (v1 == chk || v1 == 0)) return false;
            i = (i + 1) & a.table_mask;
        }
        atomicOr(&a.ctl->err, KMC_ERR_TABLE_FULL);
        return false;
    }
    static KMC_DEV bool claim_from(const KmcArgs& a, u64 fp, u64 i, u64 meta) {
        // open addressing, linear probing.  Slots only ever change 0 -> fp, so a plain
        // (possibly stale) load can only mis-report "empty", which the CAS then corrects.
        // The probe chain is bounded: a table filled beyond ~95 % makes linear probing walk millions
        // of slots per insert (a run that looked hung), so a chain this long is reported as
        // "table full" instead.  At load <= 0.9 the chance of a 1 K chain is nil.
        const u64 max_probes = a.table_mask < (1ull << 10) ? a.table_mask : (1ull << 10);
        for (u64 probes = 0; probes <= max_probes; ++probes) {
#if KMC_CAS_FIRST
            u64 v = atomicCAS(&a.table[i], 0ull, fp);
            if (v == 0) {
                if (a.pred) a.pred[i] = meta;
                return true;
            }
#else
#if KMC_SC1_PROBE
            u64 v = __hip_atomic_load(&a.table[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
            u64 v = KMC_PROBE_LOAD(&a.table[i]);
#endif
            if (v == 0) {
                if (a.flags & KMC_FLAG_X_PLAINSTORE) {
                    a.table[i] = fp;
                    return true;
                }
                v = atomicCAS(&a.table[i], 0ull, fp);
                if (v == 0) {
                    if (a.pred) a.pred[i] = meta;
                    return true;
                }
            }
#endif
            if (v == fp) return false;
            i = (i + 1) & a.table_mask;
        }
        atomicOr(&a.ctl->err, KMC_ERR_TABLE_FULL);
        return false;
    }

    // the narrow or the wide table, as the handle was opened (a wave-uniform branch); the check word is the same
    // fingerprint function under another seed
    static KMC_DEV bool claim_any(const KmcArgs& a, const u64* t, u64 fp, u64 meta) {
        if (a.flags & KMC_FLAG_FP128) return claim_wide(a, fp, kmc_fingerprint<W>(t, a.seed ^ 0x6a09e667f3bcc908ull), meta);
        return claim(a, fp, meta);
    }

    // Sender-side duplicate filter of the sharded path: true when fp was not yet in `set` (and is now).
    // BFS generates every state ~g times; without the filter all g copies cross xGMI.  A full or
    // over-long chain just answers "fresh" (the copy travels, the owner dedups): never wrong.
    static KMC_DEV bool first_time(u64* set, u64 mask, u64 fp) {
        u64 i = (fp >> 17) & mask;  // other bits than the owner's table index
        for (u32 probes = 0; probes < 64; ++probes) {
            u64 v = KMC_PROBE_LOAD(&set[i]);
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
    static KMC_DEV void report_violation(const KmcArgs& a, u32 bad, u64 fp, u32 deficit = 0) {
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
    static KMC_DEV void report_outside_violation(const KmcArgs& a, u32 bad, u64 fp) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (bad >> k & 1u) {
                atomicAdd(&a.ctl->oviol_count[k], 1ull);
                atomicMax(&a.ctl->oviol_fp_inv[k], ~fp);
            }
    }

    // Executed by the whole wave; lanes with valid=false only take part in the ballots.
    // KMC_SYMM: t is the orbit representative of the successor `raw` (what is fingerprinted, claimed, staged and shipped);
    // ENUM lists the successor itself with the representative's fingerprint, so that a trace replayed through kmc_successors
    // is a real behaviour whose states are FOUND by the fingerprints of their representatives.  stab = the order of t's
    // stabiliser: it travels with a new state (KmcStager) and gives the orbit's deficit.
    static KMC_DEV void process(const KmcArgs& a, KmcStager<W>& out, bool valid, const u64* t, u64 meta, const u64* raw = nullptr,
                                u32 stab = 1) {
#ifdef KMC_TEST_FP_BITS   // tests only: a fingerprint of that many bits, i.e. collisions on demand (the wide table's check
                          // word keeps its 64 bits) — tests/test_gpu_selfcheck_and_fp128.py
        const u64 fp = kmc_mix64((kmc_fingerprint<W>(t, a.seed) & ((1ull << (KMC_TEST_FP_BITS)) - 1)) + 0x9E3779B97F4A7C15ull) | 1ull;
#else
        const u64 fp = kmc_fingerprint<W>(t, a.seed);
#endif
        out.account(valid, fp);
        if (a.mode == KMC_MODE_DRY) {
            u64 acc = fp;
            if (valid && (a.flags & KMC_FLAG_DRY_RAND)) {  // ONE load from an unrelated random slot per successor
                acc ^= a.table[kmc_mix64(fp ^ 0xABCDEF12345ull) & a.table_mask];
            } else if (valid && (a.flags & KMC_FLAG_DRY_PROBE)) {  // read-only probe sequence (the table is already full)
                u64 i = fp & a.table_mask;
                for (u64 probes = 0; probes <= a.table_mask; ++probes) {
                    const u64 v = a.table[i];
                    acc ^= v;
                    if (v == fp || v == 0) break;
                    i = (i + 1) & a.table_mask;
                }
                if (a.flags & KMC_FLAG_DRY_INV) acc ^= M::violated(t, a.inv_mask);
                // ~35 % of the probes end in a no-op CAS on the slot they found: the same atomic
                // traffic as the real claims (311 M per 888 M probes) without changing the table
                if ((a.flags & KMC_FLAG_DRY_ATOM) && (fp & 0xFF) < 90) acc ^= atomicCAS(&a.table[i], fp, fp);
            }
            if (valid && acc == 0x1234567) atomicOr(&a.ctl->err, KMC_ERR_ENUM_FULL);  // keeps the work alive
            return;
        }
        if (a.mode == KMC_MODE_LOCAL) {
            // once any wave has found the table full the level is lost anyway: stop probing so the
            // launch ends promptly instead of walking full chains (k_expand reads the flag once per
            // tile and masks the batch, see KMC_ERRCHK_TILE; this is the per-flush variant)
#if !KMC_ERRCHK_TILE
            if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&a.ctl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) &
                KMC_ERR_TABLE_FULL)
                return;
#endif
            // (A per-wave LDS filter of recently resolved fingerprints was tried here to skip
            // duplicate probes: only 4.9 % of the successors hit it — duplicates are not local to
            // a wave — so it was dropped.)
            // (Resolving the successors of one batch that share a fingerprint only once — a per-wave LDS lane map — and walking
            // a probe chain inside its 128-byte line before moving on were measured in round 3 and change nothing:
            // profiles/r03_probe_knobs.txt.)
            const bool isnew = valid && claim_any(a, t, fp, meta);
#if KMC_SYMM
            out.corr_won += isnew ? KmcSymm<M>::deficit(stab) : 0u;
#endif
            if (!(a.flags & KMC_FLAG_X_NOSTAGE)) out.push(a, isnew, t, stab);
        } else if (a.mode == KMC_MODE_SHARDED) {
            // successors this shard owns take the local path at once (probe, claim, stage): only
            // the (P-1)/P that belong elsewhere travel
            const u32 dst = valid ? kmc_owner(fp, a.nshards) : ~0u;
            const bool isnew = dst == a.shard && claim_any(a, t, fp, meta);
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
            const u32 src = ship ? dst : 0u;
            const u32 lo = (u32)__shfl((int)(u32)base, (int)src), hi = (u32)__shfl((int)(u32)(base >> 32), (int)src);
            if (ship) {
                const u64 pos = (((u64)hi << 32) | lo) + my_rank;
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

// ========================================================================================
// kernels
// ========================================================================================
#define KMC_BLOCK 256
#define KMC_WAVES (KMC_BLOCK / 64)

template <class M> KMC_DEV void kmc_expand_body(const KmcArgs& a) {
    constexpr int W = M::W;
    constexpr int NW = (M::NINST + 63) / 64;  // words of the per-lane "enabled instances" bitset
    // per-wave dynamic LDS: successor ring (W state planes of KMC_RING entries, plus one meta
    // plane — parent fp, or kind in ENUM mode — only when KMC_FLAG_META) followed by the output
    // stager's W planes of KMC_QCAP entries.  The host sizes it (kmc_expand_lds_bytes).
    extern __shared__ __attribute__((aligned(16))) u64 kmc_lds[];
    const u32 lane = kmc_lane();
    const u32 wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform, keep it scalar
    const bool has_meta = (a.flags & KMC_FLAG_META) != 0;
    const u32 ring_planes = W + (has_meta ? 1u : 0u);
    u64* q = kmc_lds + (size_t)wib * (ring_planes * KMC_RING + KmcStager<W>::PL * KmcStager<W>::QCAP);  // q[k*KMC_RING + pos]
    KmcStager<W> out;
    out.init(q + ring_planes * KMC_RING);
    u32 head = 0, count = 0;  // wave-uniform: ring read position / number of QUEUED successors
    u32 gen_lane = 0;         // lane k accumulates the successors generated by action kind k
    u32 extra_lane = 0;       // HAS_EXTRA: this lane's states' additional bindings with a repeated successor (kind EXTRA_KIND)
    u32 deadlocks = 0;
#if KMC_SYMM
    u32 extra_corr = 0;       // per lane: orbit deficits of the repeated bindings counted in extra_lane
#endif
#if KMC_PROFILE
    u64 prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // 0 load+extract+inv 1 guards 2 effects+push 3 flush 4 tail 7 total
    const u64 t_kernel0 = __builtin_amdgcn_s_memtime();
    out.prof = prof_acc;
#endif

    u32 table_full = 0;  // wave-uniform; KMC_ERRCHK_TILE: refreshed once per tile
#if KMC_SYMM
    // the images of every (leader, isr) pair under every permutation of Replicas (KmcSymm): filled here, read after the
    // block's first barrier below
    __shared__ u32 kmc_symtab[KmcSymm<M>::TABLE_WORDS];
    for (u32 i = threadIdx.x; i < (u32)KmcSymm<M>::TABLE_WORDS; i += KMC_BLOCK) kmc_symtab[i] = KmcSymm<M>::TABLE.w[i];
#endif
    const u32 nwaves = gridDim.x * KMC_WAVES;
    const u32 wave0 = blockIdx.x * KMC_WAVES + wib;
#if KMC_FAULT_DROP
    u32 nflush = 0;
#endif
    auto flush = [&](u32 nv) {  // nv <= KMC_FLUSH_N queued successors leave the ring
        u64 t0[W];
        KMC_FENCE_LDS();
        const u32 pos0 = (head + lane) & (KMC_RING - 1);
#pragma unroll
        for (int k = 0; k < W; ++k) t0[k] = q[k * KMC_RING + pos0];
        const u64 meta0 = has_meta ? q[W * KMC_RING + pos0] : 0ull;
#if KMC_SETPRIO
        __builtin_amdgcn_s_setprio(2);
#endif
#if KMC_FAULT_DROP
        const bool dropped = (a.mode == KMC_MODE_LOCAL || a.mode == KMC_MODE_SHARDED) && wave0 == 0 && nflush == 0 && lane == 5;
        ++nflush;
        KmcSink<M>::process(a, out, lane < nv && !table_full && !dropped, t0, meta0);
#elif KMC_SYMM
        {
            u64 tc[W];
            u32 stab_t;
            KmcSymm<M>::canon(t0, kmc_symtab, tc, stab_t);
            KmcSink<M>::process(a, out, lane < nv && !table_full, tc, meta0, t0, stab_t);
        }
#else
        KmcSink<M>::process(a, out, lane < nv && !table_full, t0, meta0);
#endif
#if KMC_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        head = (head + nv) & (KMC_RING - 1);
        count -= nv;
    };

    // Segment by segment; within a segment the 64-state tiles are dealt round-robin to all waves
    // of the grid.  (One flat tile index over all segments needed their prefix sums live in
    // SGPRs for the whole kernel.)
    if (a.prev) {
        // chained launch: every wave reads the finished control block of the producing level (wave-uniform values)
        const KmcLevelCtl* pv = a.prev;
        u32 stop = pv->halt | pv->err;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            stop |= ((a.stop_mask >> k & 1u) && (pv->viol_count[k] | pv->oviol_count[k])) ? 1u : 0u;
        if (a.stop_deadlock && pv->deadlock_count) stop |= 1u;
        if (__builtin_amdgcn_readfirstlane(stop)) {
            if (blockIdx.x == 0 && threadIdx.x == 0) a.ctl->halt = 1u;
            return;
        }
    }
    // per block: generated[0..15], deadlocks, probed, won, outside, repeats (see the end); KMC_SYMM: kmc_corr[0..15] corr_gen,
    // 16 corr_dead, 17 corr_repeats, 18 corr_won
    __shared__ u32 kmc_tail[64];
    if (threadIdx.x < 64) kmc_tail[threadIdx.x] = 0;
#if KMC_SYMM
    // The orbit deficits are 64-bit cells (ds_add_u64): a block's share of a level is millions of successors times up to
    // N! - 1 = 5,039 each — BASELINE config 5's 17th level (133 M stored states) wrapped 32-bit cells 256 times (round 3:
    // `generated` 2^40 too large, found by oracle/orbit_oracle.c).  What a LANE or a WAVE sums stays 32-bit (a lane sees
    // ~300 states of such a level: < 2^26 per lane even with every successor at 5,039).
    __shared__ u64 kmc_corr[32];
    if (threadIdx.x < 32) kmc_corr[threadIdx.x] = 0;
#endif
    __syncthreads();
#if KMC_SYMM
    // Sparse tiles.  A wave's time here goes into finding the representatives of its tile's successors (hundreds of vector
    // instructions each, N! images at five and six replicas), and the levels are N! times smaller than the plain search's:
    // a level of 30 K states is 470 tiles of 64 — one busy wave on every other SIMD, each crawling through ~7 flushes.
    // So a tile holds 2^tile_sh <= 64 STATES, as few as it takes to give every wave of the grid one (the host launches the
    // grid for tiles of 4, expand_grid): the successors of a level spread over eight times as many waves.
    u32 tile_sh = 6;
    {
        u64 n_all = 0;
        for (int sg = 0; sg < KMC_SEGS; ++sg) {
            u64 n = a.seg_count[sg];
            if (a.prev) {
                const u64 made = a.prev->next_count[sg].v;
                n = made < a.seg_cap ? made : a.seg_cap;
            }
            n_all += n;
        }
        const u64 per = n_all / nwaves;
        tile_sh = __builtin_amdgcn_readfirstlane(per >= 64 ? 6u : per >= 32 ? 5u : per >= 16 ? 4u : per >= 8 ? 3u : 2u);
    }
#else
    constexpr u32 tile_sh = 6;   // a tile = 64 states, one per lane
#endif
#pragma clang loop unroll(disable)
    for (int sg = 0; sg < KMC_SEGS; ++sg) {
    u64 seg_n = a.seg_count[sg];
    if (a.prev) {
        const u64 made = a.prev->next_count[sg].v;
        seg_n = made < a.seg_cap ? made : a.seg_cap;
    }
    const u64 seg_base = (u64)sg * a.seg_cap;
    const u64 seg_tiles = (seg_n + ((1u << tile_sh) - 1)) >> tile_sh;
    // rotate the starting wave per segment so that short segments do not always land on the same waves
    const u32 first = (wave0 + nwaves - (u32)((sg * 977u) % nwaves)) % nwaves;
#if KMC_PREFETCH
    // software prefetch: the next tile's state words are requested before this tile is processed,
    // so their HBM latency (a quarter of the compute-only time when exposed) hides under it
    u64 s_next[W];
    {
        const u64 j0 = ((u64)first << tile_sh) + lane;
#pragma unroll
        for (int k = 0; k < W; ++k)
            s_next[k] = (first < seg_tiles && j0 < seg_n) ? a.fin[(u64)k * a.fin_stride + seg_base + j0] : 0ull;
    }
#endif
#pragma clang loop unroll(disable)
    // (a contiguous run of tiles per wave instead of every nwaves-th tile was measured: the same 30 effect leaves per
    // tile, kernel 37.9 ms against 35.8 — the strided deal balances the tail of a level better)
    for (u64 tile = first; tile < seg_tiles; tile += nwaves) {
        const u64 j = (tile << tile_sh) + lane;
        const bool valid = (lane >> tile_sh) == 0 && j < seg_n;
#if KMC_PROFILE
        prof_acc[6] += 1;
#endif
        KMC_T(tp0);
        u64 s[W];
#if KMC_PREFETCH
#pragma unroll
        for (int k = 0; k < W; ++k) s[k] = s_next[k];
        {
            const u64 tn = tile + nwaves;
            const u64 jn = (tn << tile_sh) + lane;
            const bool vn = tn < seg_tiles && jn < seg_n;
#pragma unroll
            for (int k = 0; k < W; ++k) s_next[k] = vn ? a.fin[(u64)k * a.fin_stride + seg_base + jn] : 0ull;
        }
#else
        const u64 idx = seg_base + j;
#if KMC_ERRCHK_TILE
        const u32 errv = __hip_atomic_load(&a.ctl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
#pragma unroll
        for (int k = 0; k < W; ++k) s[k] = valid ? KMC_FRONTIER_LOAD(&a.fin[(u64)k * a.fin_stride + idx]) : 0ull;
#if KMC_ERRCHK_TILE
        table_full = __builtin_amdgcn_readfirstlane(errv) & KMC_ERR_TABLE_FULL;
#endif
#endif
        const u64 parent = (a.flags & (KMC_FLAG_TRACE | KMC_FLAG_ENUM_MATCH)) ? kmc_fingerprint<W>(s, a.seed) : 0ull;
        typename M::Pre pre = M::extract(s);
#if KMC_SYMM
        // the expanded state is its orbit's representative; everything counted for it below stands for the whole orbit, less
        // this deficit when some permutation fixes it (rare: the lanes with defl != 0 take the few extra steps)
        // (the order of its stabiliser came with it: plane W of the frontier, KmcStager)
        const u32 defl = valid ? KmcSymm<M>::deficit((u32)KMC_FRONTIER_LOAD(&a.fin[(u64)W * a.fin_stride + idx])) : 0u;
#else
        const u32 defl = 0;
#endif

        // Invariants of the states of THIS level (see KmcSink::report_violation)
        if (KMC_INV_MASK(a) && a.mode != KMC_MODE_ENUM && !(a.flags & KMC_FLAG_X_NOINV)) {
            const u32 bad = valid ? M::violated_pre(pre, KMC_INV_MASK(a)) : 0u;
            if (__ballot(bad != 0)) {
                if (bad) KmcSink<M>::report_violation(a, bad, kmc_fingerprint<W>(s, a.seed), defl);
            }
        }
        if (a.flags & KMC_FLAG_INV_ONLY) continue;   // (wave-uniform: a kernel argument)

        KMC_T(tp1);
        KMC_TADD(0, tp0, tp1);
        // Pass 1 — every guard of Next in one straight-line block: no dispatch, and the compiler
        // shares sub-terms between instances (the effects are dead code here and vanish).
        // (32-bit halves + shift-by-literal: one v_cndmask + one v_lshl_or per instance; a
        // 64-bit `g << i` made the compiler park sixty bit constants in VGPRs)
        u32 en32[2 * NW];
#pragma unroll
        for (int h = 0; h < 2 * NW; ++h) en32[h] = 0;
        const u32 valid01 = valid ? 1u : 0u;
        u32 nsucc = 0;
        if constexpr (M::GROUPED_GUARDS) {
        // (wide configurations) group by group: a group's guards are evaluated only if its necessary condition holds in some
        // lane of the tile — a scalar branch on a ballot; the blocks between the branches are what the scheduler sees at a time
        static_assert(M::groups_partition_the_instances(), "guard groups must list every action instance exactly once");
        kmc_static_for<0, M::NGROUPS>([&](auto GG) {
            constexpr int g = decltype(GG)::value;
            const u32 c = M::template group_pre<g>(pre) & valid01;
            if (g == 0 || __ballot(c != 0u)) {
                // (opaque redefinition of the state words: a group extracts the fields it reads itself — shared with the
                // other groups, the seventy fields of seven replicas stayed in registers across the whole of pass 1)
#pragma unroll
                for (int q2 = 0; q2 < W; ++q2) kmc_launder(s[q2]);
                kmc_static_for<0, M::group_size(g)>([&](auto JJ) {
                    constexpr int i = M::group_inst(g, decltype(JJ)::value);
                    u64 tt[W];
                    int kd;
                    u32 ex;
                    const u32 g01 = M::template inst<i>(pre, s, tt, kd, ex) & valid01;
                    en32[i >> 5] |= g01 << (i & 31);
                    kmc_launder(en32[i >> 5]);
                });
            }
        });
#pragma unroll
        for (int h = 0; h < 2 * NW; ++h) nsucc += __popc(en32[h]);
        } else
        if constexpr (!M::RUNTIME_GUARDS) {   // (the wide Kafka configurations evaluate their guards per kind, in pass 2's walk)
        kmc_static_for<0, M::NINST>([&](auto I) {
            constexpr int i = decltype(I)::value;
            u64 tt[W];
            int kd;
            u32 ex;
            const u32 g01 = M::template inst<i>(pre, s, tt, kd, ex) & valid01;
            en32[i >> 5] |= g01 << (i & 31);
            kmc_launder(en32[i >> 5]);  // keep the OR chain sequential (a reassociated tree keeps every leaf live)
        });
#pragma unroll
        for (int h = 0; h < 2 * NW; ++h) nsucc += __popc(en32[h]);
        }
#if KMC_SYMM
        if constexpr (M::KIND_MAJOR && !M::RUNTIME_GUARDS) {
            if (defl) {
                kmc_static_for<0, M::NSEGS>([&](auto SS) {
                    constexpr int sg = decltype(SS)::value;
                    const typename M::KindBits kb = M::template seg_bits<sg>(en32);
                    u32 c;
                    if constexpr (sizeof(kb) == 8) c = (u32)__popcll(kb);
                    else c = (u32)__popc(kb);
                    if (c) atomicAdd(&kmc_corr[M::seg_kind(sg)], (u64)(defl * c));
                });
            }
        }
#endif

        KMC_T(tp2);
        KMC_TADD(1, tp1, tp2);
        // Pass 2 — the effects.  (profiles/r03_ablation.txt, docs/TUNING_LOG_r1-r3.md §9: with the table untouched the kernel takes 20 ms in the kind-major form and
        // 26 ms in the instance-major one; the memory system needs ~31 ms for the run's probes and claims.)
        if constexpr (M::KIND_MAJOR) {
        // Kind-major walk (replica-major layouts, KmcKafka::apply<K>): for every action kind, every lane applies ITS OWN
        // next enabled binding of that kind in the same leaf — replicas and request epoch are run-time values, a field
        // of replica r is "select word r, extract at a compile-time offset" — until no lane has one left.  A tile costs
        // sum over kinds of max-over-lanes(enabled bindings) leaves: 12.6 at the headline where the instance-major walk
        // below dispatches 30 (one per (kind, binding) ANY lane enabled), each with twice the lanes busy.
        // (The walk is over SEGMENTS: a kind's bindings in windows of at most 32 / 64, one per-lane bitset each — more than
        // one window per kind only from 6 replicas on.)
#pragma clang loop unroll(disable)
        for (int sgi = 0; sgi < M::NSEGS; ++sgi) {
            typename M::KindBits km = 0;
            int k = 0;   // the segment's action kind (wave-uniform)
            kmc_dispatch<0, M::NSEGS>(sgi, [&](auto SS) {
                constexpr int sg = decltype(SS)::value;
                k = M::seg_kind(sg);
                if constexpr (M::RUNTIME_GUARDS) {
                    // Pass 1 of a wide configuration, fused into the walk: the guard of every binding of this segment, one
                    // binding per iteration (wave-uniform, the replicas it names are scalars), into the per-lane bitset the
                    // leaves below consume.  No instance bitset, no straight-line block of hundreds of guards.
#pragma clang loop unroll(disable)
                    for (u32 j = 0; j < (u32)M::seg_count(sg); ++j) {
#pragma unroll
                        for (int q2 = 0; q2 < W; ++q2) kmc_launder(s[q2]);
                        const u32 g01 = M::template guard<M::seg_kind(sg)>(pre, s, (u32)M::seg_first(sg) + j) & valid01;
                        km |= (typename M::KindBits)g01 << j;
                    }
                    if constexpr (sizeof(km) == 8) nsucc += (u32)__popcll(km);
                    else nsucc += (u32)__popc(km);
#if KMC_SYMM
                    if (defl && km) atomicAdd(&kmc_corr[M::seg_kind(sg)], (u64)(defl * (sizeof(km) == 8 ? (u32)__popcll(km) : (u32)__popc((u32)km))));
#endif
                } else {
                    km = M::template seg_bits<sg>(en32);
                }
            });
#pragma clang loop unroll(disable)
            for (;;) {
                const bool e = km != 0;
                const u64 m = __ballot(e);
                if (m == 0) break;
#if KMC_PROFILE
                prof_acc[5] += 1;  // effect leaves dispatched (per wave; prof[6] counts tiles)
#endif
                u32 b;
                if constexpr (sizeof(km) == 8) b = (u32)__builtin_ctzll(km | (1ull << 63));
                else b = (u32)__builtin_ctz(km | (1u << 31));
                km &= km - 1;
                // (opaque redefinition of the state words: the parts of an effect that do not depend on the binding stay
                // inside the loop.  The guards' shared sub-predicates in `pre` are not touched: they die with pass 1.)
#pragma unroll
                for (int q2 = 0; q2 < W; ++q2) kmc_launder(s[q2]);
                u32 extra = 0;
                u64 t[W];
                kmc_dispatch<0, M::NSEGS>(sgi, [&](auto SS) {
                    constexpr int sg = decltype(SS)::value;
                    M::template apply<M::seg_kind(sg)>(pre, s, t, b + (u32)M::seg_first(sg), extra);
                });
                const u32 n = __popcll(m);
                if constexpr (M::HAS_EXTRA) extra_lane += e ? extra : 0u;
#if KMC_SYMM
                if constexpr (M::HAS_EXTRA) extra_corr += e ? extra * defl : 0u;
#endif
                gen_lane += (lane == (u32)k) ? n : 0u;
                if (e) {
                    const u32 pos = (head + count + kmc_rank_in(m)) & (KMC_RING - 1);
#pragma unroll
                    for (int q2 = 0; q2 < W; ++q2) q[q2 * KMC_RING + pos] = t[q2];
                    if (has_meta)
                        q[W * KMC_RING + pos] = (a.mode == KMC_MODE_ENUM && !(a.flags & KMC_FLAG_ENUM_MATCH)) ? ((u64)k | ((u64)extra << 8)) : parent;
                }
                count += n;
                if (count >= KMC_FLUSH_N) {
                    KMC_T(tf0);
                    flush(KMC_FLUSH_N);
                    KMC_T(tf1);
                    KMC_TADD(3, tf0, tf1);
                }
            }
        }
        } else {
        // Instance-major walk: only the instances some lane enabled dispatch to their (statically specialised) effect;
        // a leaf runs for the whole wave although few lanes enabled it (30.1 leaves per tile and ~6.7 of 64 lanes at the
        // headline's constants on the tight layout).  Measured and dropped here: a fall-through `switch`, walking only
        // the set bits of the wave-wide OR of en32 (s_ff1), per-kind `generated` counters in scalars (the array lands in
        // scratch) or bumped with v_writelane.
        u32 cur = 0;
#pragma clang loop unroll(disable)
        for (int i = 0; i < M::NINST; ++i) {
            if ((i & 31) == 0) {
                cur = en32[0];
#pragma unroll
                for (int h = 1; h < 2 * NW; ++h)
                    if ((i >> 5) == h) cur = en32[h];
            }
            const bool e = cur & 1u;
            cur >>= 1;
            const u64 m = __ballot(e);
            if (m == 0) continue;
#if KMC_PROFILE
            prof_acc[5] += 1;  // effect leaves dispatched (per wave; prof[6] counts tiles)
#endif
            // opaque redefinition: keeps LICM from hoisting all effects out of this loop
            M::launder(pre);
#pragma unroll
            for (int k = 0; k < W; ++k) kmc_launder(s[k]);
            int kind = 0;
            u32 extra = 0;
            u64 t[W];
            kmc_dispatch<0, M::NINST>(i, [&](auto I) {
                (void)M::template inst<decltype(I)::value>(pre, s, t, kind, extra);
            });
            const u32 n = __popcll(m);
            // bindings that repeat a successor (TLC counts them as generated): summed per lane, reduced once per wave at
            // the end of the kernel (a wave reduction in every leaf cost 7 cross-lane operations per dispatched leaf)
            if constexpr (M::HAS_EXTRA) extra_lane += e ? extra : 0u;
            gen_lane += (lane == (u32)kind) ? n : 0u;
#if KMC_SYMM
            if constexpr (M::HAS_EXTRA) extra_corr += e ? extra * defl : 0u;
            if (e && defl) atomicAdd(&kmc_corr[kind], (u64)defl);
#endif
            bool keep = e;
            u64 mk = m;
            if constexpr (M::HAS_CONSTRAINT) {
                // TLC CONSTRAINT [TLC-recall, ModelChecker.doNext]: a successor outside the model counts
                // as generated but is neither fingerprinted into the seen-set nor queued; its
                // invariants ARE evaluated (each time it is generated: it is never "seen").
                // ENUM lists the raw Next relation; the level-limit check pass (DRY) looks no further.
                if (a.mode == KMC_MODE_LOCAL || a.mode == KMC_MODE_SHARDED) {
                    const bool outside = e && !M::in_model(t);
                    if (__ballot(outside)) {
                        if (outside && a.inv_mask) {
                            const u32 bad = M::violated(t, a.inv_mask);
                            if (bad) KmcSink<M>::report_outside_violation(a, bad, kmc_fingerprint<W>(t, a.seed));
                        }
                        keep = e && !outside;
                        mk = __ballot(keep);
                        out.outside += (u32)(__popcll(m) - __popcll(mk));
                    }
                }
            }
            if (keep) {
                const u32 pos = (head + count + kmc_rank_in(mk)) & (KMC_RING - 1);
#pragma unroll
                for (int k = 0; k < W; ++k) q[k * KMC_RING + pos] = t[k];
                if (has_meta)
                    q[W * KMC_RING + pos] = (a.mode == KMC_MODE_ENUM && !(a.flags & KMC_FLAG_ENUM_MATCH)) ? ((u64)kind | ((u64)extra << 8)) : parent;
            }
            count += __popcll(mk);
            if (count >= KMC_FLUSH_N) {
                KMC_T(tf0);
                flush(KMC_FLUSH_N);
                KMC_T(tf1);
                KMC_TADD(3, tf0, tf1);
            }
        }
        }   // instance-major walk
        KMC_T(tp3);
        KMC_TADD(2, tp2, tp3);
        const u64 dm = __ballot(valid && nsucc == 0);
        if (dm) {
            deadlocks += __popcll(dm);
            // one no-return atomicMax per deadlocked state (1.4 M at the headline, all on one line: fire-and-forget).
            // Reducing over the wave first was tried: its twelve cross-lane moves and their temporaries pushed k_expand over
            // the 80-VGPR budget (107 VGPRs, 4 waves per SIMD) and the headline to 38.8 ms (profiles/r03_tail.txt)
            if (valid && nsucc == 0) atomicMax(&a.ctl->deadlock_fp_inv, ~kmc_fingerprint<W>(s, a.seed));
#if KMC_SYMM
            if (valid && nsucc == 0 && defl) atomicAdd(&kmc_corr[16], (u64)defl);
#endif
        }
    }
    }
    KMC_T(tt0);
    while (count) flush(count < KMC_FLUSH_N ? count : KMC_FLUSH_N);
    out.finish(a, KMC_CHECKSUM != 0);
    KMC_T(tt1);
    KMC_TADD(4, tt0, tt1);
#if KMC_PROFILE
    prof_acc[7] = __builtin_amdgcn_s_memtime() - t_kernel0;
    if (lane == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(&a.ctl->prof[k], prof_acc[k]);
#endif
    // The level's counters leave the kernel ONCE PER BLOCK: every wave adds its share to an LDS tail, the block's last
    // barrier, and wave 0 issues one atomicAdd instruction over the 21 cells.  (Published per wave — two more atomic
    // instructions and their operands live to the end — k_expand no longer fitted 80 VGPRs: more than 8 spilled at 6 waves
    // per SIMD, so the register-budget rule of get_code_object rebuilt it for 4 waves per SIMD, 107 VGPRs, and the
    // headline took 38.9 ms instead of 35.0: profiles/r03_tail.txt.  The kernel sits on the edge of that budget; any
    // addition to this function has to be checked against `.vgpr_count` of the headline's code object.)
    u32 repeats = 0;
    if constexpr (M::HAS_EXTRA) {
        u32 x = extra_lane;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        gen_lane += (lane == (u32)M::EXTRA_KIND) ? x : 0u;
        repeats = x;
    }
    if (lane < (u32)M::NKINDS && gen_lane) atomicAdd(&kmc_tail[lane], gen_lane);
#if !KMC_CHECKSUM   // (the checksum builds publish probed / won / outside per wave, with the checksum: KmcStager::finish)
    const u32 mine = lane == 16 ? deadlocks : lane == 17 ? out.probed : lane == 18 ? out.won : lane == 19 ? out.outside : lane == 20 ? repeats : 0u;
#else
    const u32 mine = lane == 16 ? deadlocks : lane == 20 ? repeats : 0u;
#endif
    if (lane >= 16 && mine) atomicAdd(&kmc_tail[lane], mine);
#if KMC_SYMM
    {
        u32 cw = out.corr_won, cx = 0;
        if constexpr (M::HAS_EXTRA) cx = extra_corr;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            cw += __shfl_xor(cw, off);
            cx += __shfl_xor(cx, off);
        }
        out.corr_won = 0;
        if (lane == 0) {
            if (cw) atomicAdd(&kmc_corr[18], (u64)cw);
            if constexpr (M::HAS_EXTRA) {
                if (cx) {
                    atomicAdd(&kmc_corr[17], (u64)cx);
                    atomicAdd(&kmc_corr[M::EXTRA_KIND], (u64)cx);   // the repeats are part of generated[EXTRA_KIND]
                }
            }
        }
    }
#endif
    __syncthreads();
#if KMC_SYMM
    if (wib == 1 && lane < 19) {
        const u64 v = kmc_corr[lane];
        u64* dst = lane < 16 ? &a.ctl->corr_gen[lane] : lane == 16 ? &a.ctl->corr_dead : lane == 17 ? &a.ctl->corr_repeats
                 : &a.ctl->corr_won;
        if (v) atomicAdd(dst, v);
    }
#endif
    if (wib == 0 && lane < 21) {
        const u32 v = kmc_tail[lane];
        u64* dst = lane < 16 ? &a.ctl->generated[lane] : lane == 16 ? &a.ctl->deadlock_count : lane == 17 ? &a.ctl->probed
                 : lane == 18 ? &a.ctl->won : lane == 19 ? &a.ctl->outside : &a.ctl->repeats;
        if (v) atomicAdd(dst, (u64)v);
    }
}

// dynamic LDS bytes k_expand needs for a state of W words
KMC_HD inline unsigned kmc_expand_lds_bytes(int W, bool has_meta, bool symmetry = false, int qcap_wide = KMC_QCAP_WIDE) {
    return (unsigned)(KMC_WAVES * ((W + (has_meta ? 1 : 0)) * KMC_RING + (W + (symmetry ? 1 : 0)) * kmc_qcap(W, qcap_wide)) * 8);
}

// Inserts a list of AoS records (W state words + predecessor fp) into the local table:
// the initial state, and the receive side of the multi-GPU exchange.
template <class M> KMC_DEV void kmc_insert_body(const KmcArgs& a) {
    constexpr int W = M::W;
    __shared__ u64 stage[KMC_WAVES][KmcStager<W>::PL][KmcStager<W>::QCAP];
    KmcStager<W> out;
    out.init(&stage[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)][0][0]);
#if KMC_PROFILE
    u64 prof_dummy[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    out.prof = prof_dummy;
#endif
    const u64 n = a.n_in;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    const u64 rounds = (n + stride - 1) / stride;
    for (u64 r = 0; r < rounds; ++r) {
        const u64 idx = r * stride + (u64)blockIdx.x * blockDim.x + threadIdx.x;
        const bool valid = idx < n;
        u64 t[W];
#pragma unroll
        for (int k = 0; k < W; ++k) t[k] = valid ? a.recv[idx * (u64)a.rec_words + k] : 0ull;
        const u64 meta = (valid && a.rec_words > (u32)W) ? a.recv[idx * (u64)a.rec_words + W] : 0ull;
        KmcArgs b = a;
        b.mode = KMC_MODE_LOCAL;
#if KMC_SYMM
        // (records arrive as representatives — Init, kmc_engine.cpp do_begin; what they lack is the stabiliser's order)
        KmcSink<M>::process(b, out, valid, t, meta, nullptr, valid ? KmcSymm<M>::stabiliser(t, KmcSymm<M>::TABLE.w) : 1u);
#else
        KmcSink<M>::process(b, out, valid, t, meta);
#endif
    }
    out.finish(a);
}

// Writes Init as one AoS record (W words + predecessor 0) at a.send.
template <class M> KMC_DEV void kmc_init_body(const KmcArgs& a) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        u64 w[M::W];
        M::init(w);
        for (int k = 0; k < M::W; ++k) a.send[k] = w[k];
        a.send[M::W] = 0;
    }
}

// Finds the state(s) of a frontier whose fingerprint equals a.seed-keyed target (passed in
// a.table_mask) and copies the words to a.send: used to fetch a violation witness.
template <class M> KMC_DEV void kmc_find_body(const KmcArgs& a) {
    constexpr int W = M::W;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (int sg = 0; sg < KMC_SEGS; ++sg)
        for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < a.seg_count[sg]; j += stride) {
            const u64 idx = (u64)sg * a.seg_cap + j;
            u64 s[W];
#pragma unroll
            for (int k = 0; k < W; ++k) s[k] = a.fin[(u64)k * a.fin_stride + idx];
            if (kmc_fingerprint<W>(s, a.seed) == a.table_mask) {
                for (int k = 0; k < W; ++k) a.send[k] = s[k];
                a.send[W] = idx;
            }
        }
}

// SHARDED: this shard's row of the level's count exchange, built ON THE DEVICE from the control block k_expand has just
// filled (kmc_step_expand_counts): [destination][sub-buffer] records to ship (clamped to the sub-buffer's capacity, 0 for
// itself), then the caller's statistics vector, which arrives in the kernel arguments — no host round trip between the
// expansion and the all-gather.
#define KMC_ROW_STATS 64
struct KmcPackArgs {
    const KmcLevelCtl* ctl;
    long long* row;     // [nshards * KMC_SEGS + KMC_ROW_STATS]
    u64 send_cap;
    u32 nshards, shard;
    long long stats[KMC_ROW_STATS];
};
KMC_DEV void kmc_pack_row_body(const KmcPackArgs& a) {
    const u32 i = threadIdx.x;
    const u32 ncount = a.nshards * KMC_SEGS;
    if (i < ncount) {
        const u32 d = i / KMC_SEGS, sb = i % KMC_SEGS;
        const u64 c = a.ctl->send_count[d][sb].v;
        a.row[i] = d == a.shard ? 0ll : (long long)(c < a.send_cap ? c : a.send_cap);
    } else if (i < ncount + KMC_ROW_STATS) {
        a.row[i] = a.stats[i - ncount];
    }
}

#ifndef KMC_MIN_WAVES
#define KMC_MIN_WAVES 6   // __launch_bounds__ second argument for k_expand: minimum waves per SIMD (LDS admits 6 blocks/CU)
#endif
#define KMC_INSTANTIATE(NAME, ...)                                                                       \
    extern "C" __global__ __launch_bounds__(KMC_BLOCK, KMC_MIN_WAVES) void kmc_expand_##NAME(KmcArgs a) { \
        kmc_expand_body<__VA_ARGS__>(a);                                                                 \
    }                                                                                                    \
    extern "C" __global__ __launch_bounds__(KMC_BLOCK) void kmc_insert_##NAME(KmcArgs a) {               \
        kmc_insert_body<__VA_ARGS__>(a);                                                                 \
    }                                                                                                    \
    extern "C" __global__ __launch_bounds__(KMC_BLOCK) void kmc_init_##NAME(KmcArgs a) {                 \
        kmc_init_body<__VA_ARGS__>(a);                                                                   \
    }                                                                                                    \
    extern "C" __global__ __launch_bounds__(KMC_BLOCK) void kmc_find_##NAME(KmcArgs a) {                 \
        kmc_find_body<__VA_ARGS__>(a);                                                                   \
    }                                                                                                    \
    extern "C" __global__ __launch_bounds__(KMC_BLOCK) void kmc_packrow_##NAME(KmcPackArgs a) {          \
        kmc_pack_row_body(a);                                                                            \
    }
#endif  // !KMC_HOST_EMU
