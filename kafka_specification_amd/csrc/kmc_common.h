// kmc_common.h — build switches, the per-level control block, the kernel arguments, wave helpers, the fingerprint.
// Part of the device source (kmc_device.h lists the parts; the host engine hands their concatenation to hiprtc).
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
#define KMC_MODE_DRY 3u      // generate + fingerprint successors, touch no table or frontier: only compiled into a KMC_TUNING
                             // build and into KMC_VERIFY's two builds (kmc_expand_dry_*), never into what a search runs

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
                      // (a 10-word state — BASELINE config 5 — needs 4 x (10 x 128 + 10 x 64) x 8 = 60 KB of LDS per block: TWO
                      // blocks per CU whatever the registers allow.  A 32-entry stager for wide states lets three fit and is 8 %
                      // slower: profiles/r04_wide_kernel.txt; the knob is gone)
#define KMC_SEGS 8    // frontier segments, each with its own append counter (block b appends to b % KMC_SEGS)

// Build switches (the host passes them per code object through KMC_JIT_DEFINES).  The diagnostic ones only exist in a
// TUNING build (-DKMC_TUNING=1): the code objects a user runs cannot be asked for them.
#ifndef KMC_TUNING
#define KMC_TUNING 0
#endif
#if !KMC_TUNING
// (a stale script that asks for a diagnostic build without -DKMC_TUNING=1 must fail, not run a normal kernel under another
// cache key and pass vacuously: ADVICE r5)
#if defined(KMC_PROFILE) || defined(KMC_FAULT_DROP) || defined(KMC_TEST_FP_BITS)
#error "KMC_PROFILE / KMC_FAULT_DROP / KMC_TEST_FP_BITS only exist in a tuning build: add -DKMC_TUNING=1"
#endif
#endif
#ifndef KMC_PROFILE
#define KMC_PROFILE 0     // 1 (KMC_TUNING): per-phase s_memtime accounting into KmcLevelCtl::prof (costs ~10 %)
#endif
#if KMC_PROFILE
#define KMC_T(var) const u64 var = __builtin_amdgcn_s_memtime()
#define KMC_TADD(slot, t0, t1) prof_acc[slot] += (t1) - (t0)
#else
#define KMC_T(var)
#define KMC_TADD(slot, t0, t1)
#endif
// The frontier planes (read once, written once per level) stream past the caches: nt loads / stores (-0.5 ms on the headline,
// profiles/r02_nt_sweep.txt).  The seen-set's probes are plain loads: the nt flavour alone sustains more random loads per second,
// but the claim's CAS wants the line the probe has just brought into L2 (same file).
#define KMC_FRONTIER_LOAD(p) __builtin_nontemporal_load(p)
#define KMC_FRONTIER_STORE(p, v) __builtin_nontemporal_store((v), (p))
#ifndef KMC_CHECKSUM
#define KMC_CHECKSUM 0    // 1: every lane keeps a running sum and xor of the fingerprints it sends into the sink and the level's
                          //    control block receives their totals — the order-independent checksum KMC_VERIFY compares between
                          //    its two builds (both are compiled with it).  Not in the default build: four live VGPRs and a
                          //    wave reduction per launch cost the headline 3 ms when it was always on (profiles/r03_selfcheck_cost.txt)
#endif
#ifndef KMC_FAULT_DROP
#define KMC_FAULT_DROP 0  // 1 (KMC_TUNING; fault injection, tests only): the first flush of block 0 / wave 0 of every LOCAL launch loses the
                          //    successor in lane 5 between the ring and the seen-set — the failure class of round 1's miscompiled
                          //    kernel.  The conservation check (generated = probed) and KMC_VERIFY's checksum must both catch it
#endif
#ifndef KMC_RT_GUARDS_MIN_INSTANCES
#define KMC_RT_GUARDS_MIN_INSTANCES 1000000   // Kafka configurations with MORE action instances than this evaluate their guards
                                          // in per-kind loops over a run-time binding (KmcKafka::guard<K>) instead of one
                                          // straight-line block of every instance's guard (inst<I>).  Off by default: the
                                          // loops compile in seconds where the block takes minutes at 7 brokers, but they run
                                          // slower everywhere (headline 38.9 ms against 31.8, config 5's first ten levels 61
                                          // against 29: profiles/r03_runtime_guards.txt) — the block shares its sub-terms
                                          // across instances, a loop cannot.  KMC_VERIFY's second build sets it to 0: its
                                          // guards are then a second, independent lowering (kmc_engine_internal.h)
#endif
#ifndef KMC_FULL_LEAVES_MIN_INSTANCES
#define KMC_FULL_LEAVES_MIN_INSTANCES 200     // orbit counting on Kafka configurations with at least this many action instances
                                              // (seven brokers and more) runs pass 2 with FULL leaves (KmcKafka::FULL_LEAVES)
#endif
#ifndef KMC_FULL_LEAVES_PLAIN
#define KMC_FULL_LEAVES_PLAIN 0               // 1: ... and so does the plain search (measured: a 2 % loss on BASELINE config 5)
#endif
#ifndef KMC_DEFER_MIN_WORDS
#define KMC_DEFER_MIN_WORDS 8   // states of at least this many words (seven brokers with deep logs: 2 waves per SIMD by LDS) run
                                // the search's flush with a DEFERRED probe: kmc_expand_body
#endif
#ifndef KMC_SYMM
#define KMC_SYMM 0        // 1 (kmc_config.symmetry): symmetry reduction with orbit counting — every successor is replaced by the
                          // representative of its orbit under the permutations of Replicas before it is fingerprinted, and
                          // the level's counters come with the deficits (KmcLevelCtl::corr_*) that turn counts over
                          // representatives into the counts of the plain search (KmcSymm below)
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
#define KMC_FLAG_X_NOWALK 2048u   // tuning (shadow pass only, with X_PLAINSTORE): a probe chain ends at its first slot — occupied by anything = seen
#define KMC_FLAG_DRY_RAND 256u  // tuning: DRY mode does one load from an uncorrelated random table slot
#define KMC_FLAG_ENUM_MATCH 512u  // ENUM lists only the successors whose fingerprint is KmcArgs::match_fp, with their parent's fp
#define KMC_FLAG_META 2u  // the ring carries a meta plane (predecessor fp for traces / kind for ENUM)
// (the invariant pass over a frontier that is not expanded — the last level under max_levels, kmc_check_states — is its own
// kernel, kmc_inv_*: load, invariants, nothing else)
#define KMC_FLAG_FP128 1024u  // the seen-set's slots are 16 bytes: the fingerprint and a second, independent 64-bit hash of the
                              // state (kmc_config.wide_fingerprint): a 64-bit collision is then recognised, not lost

#define KMC_FLAG_PAIRED 4096u  // a run that keeps traces on 64-bit entries: the slots are 16 bytes, fingerprint + predecessor (KmcArgs::pred
                               // = table + 1, both indexed 2 i) — the claim and its predecessor dirty ONE line instead of two random ones

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

// Kernel arguments.  k_expand is one kernel PER MODE (kmc_expand_body<M, MODE>, kmc_kernels.h): the mode is a template
// parameter, so the search's own kernel (LOCAL) holds no line of the owner bucketing or of the enumeration, and its kernarg
// block is KmcArgsLocal — only what a single-GPU level reads.  Everything else (SHARDED, ENUM, k_insert, k_find, k_init,
// the tuning build's DRY) takes KmcArgs = KmcArgsLocal + the exchange / list fields; a kernel only ever loads the fields
// its mode uses.
// (Round 4: one k_expand served four modes at run time behind one 248-byte block; the headline's code object reported
// 286 spilled SGPRs and 770 v_readlane, BASELINE config 5's 750 / 2,224 — VERDICT r4 weak #3, profiles/r05_mode_split.txt.)
struct KmcArgsLocal {
    // Frontiers are SoA: word k of the state at slot i lives at f[k*stride + i].  A frontier is
    // KMC_SEGS dense segments; segment s occupies slots [s*seg_cap, s*seg_cap + seg_count[s]).
    const u64* fin;    // current frontier
    u64 fin_stride;    // plane stride in states
    u64 seg_count[KMC_SEGS];  // k_expand / k_find: states per segment of the current frontier (read per segment, never held)
    u64 seg_cap;       // slots per segment (both frontiers)
    u64* fout;         // next frontier
    u64 fout_stride;
    u64* table;        // open-addressed fingerprint table, 0 = empty
    u64 table_cap;     // slots: a multiple of 64, not necessarily a power of two (kmc_slot_of)
    u64* pred;         // optional: predecessor fingerprint per table slot (trace reconstruction)
    KmcLevelCtl* ctl;
    u64 seed;
    // Chained launches (kmc_run without a progress callback): the host queues several BFS levels back to back and
    // waits once per batch instead of once per level.  The level then takes its input sizes from the control block of
    // the level that produced `fin`, and does nothing when that level (or one before it) ended the search.
    const KmcLevelCtl* prev;  // null: seg_count[] above is authoritative
    u32 inv_mask;
    u32 flags;
    u32 stop_mask;            // invariants whose violation ends the search (0 under -continue)
    u32 stop_deadlock;        // CHECK_DEADLOCK: a state without successors ends the search
};
struct KmcArgs : KmcArgsLocal {
    u64 n_in;          // k_insert: number of records
    u64* sent;         // SHARDED, optional: fingerprints already shipped to their (remote) owner
    u64 sent_mask;
    u64* send;         // SHARDED: [shard][KMC_SEGS][send_cap] AoS records of rec_words words (state[, parent fp]);
    u64 send_cap;      //   block b fills sub-buffer b % KMC_SEGS.  ENUM: one list of W+2-word records (state, fp, kind)
    const u64* recv;   // k_insert input: AoS records of rec_words words
    u64 match_fp;      // ENUM with KMC_FLAG_ENUM_MATCH: list only successors with this fingerprint (meta = parent fp)
    u32 nshards;
    u32 shard;         // this handle's shard id (SHARDED mode keeps its own successors local)
    u32 rec_words;     // exchange record size in words: W, or W+1 when predecessor fingerprints travel (trace)
    u32 pad_;
};
// the argument block of k_expand in a given mode
template <u32 MODE> struct KmcArgsOf { using type = KmcArgs; };
template <> struct KmcArgsOf<KMC_MODE_LOCAL> { using type = KmcArgsLocal; };

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
// A register of ANOTHER lane (lane src4 / 4).  ds_bpermute only sees lanes that are active when it executes: the value is made
// opaque where it is pulled, so that the instruction stays where the whole wave runs it.  (Written plainly, `e ? x * pull(y) : 0`
// had the compiler sink the ds_bpermute into the branch of the lanes with e — where a source lane without e reads as 0: the
// first full-leaf build under orbit counting reported `generated` too large, and differently from run to run,
// profiles/r05_full_leaves.txt.)
KMC_DEV u32 kmc_pull(int src4, u32 v) {
    u32 r = (u32)__builtin_amdgcn_ds_bpermute(src4, (int)v);
    KMC_OPAQUE(r);
    return r;
}
KMC_DEV u64 kmc_pull64(int src4, u64 v) {
    const u32 lo = kmc_pull(src4, (u32)v), hi = kmc_pull(src4, (u32)(v >> 32));
    return ((u64)hi << 32) | lo;
}
// The sum of x over the wave, in every lane (a butterfly of ds_bpermute / DPP steps).  Its users read it in ONE lane
// (`if (lane == 0) atomicAdd(..., sum)`): the result is made opaque here, ahead of that branch, for the reason above — a
// butterfly step that only lane 0 executes adds up zeros.
KMC_DEV u32 kmc_wave_sum(u32 x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
    KMC_OPAQUE(x);
    return x;
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
    KMC_OPAQUE_PURE(r);
    return r;
}
KMC_DEV u32 kmc_bit(u32 m, int k) { return (m >> k) & 1u; }
KMC_DEV u32 kmc_bit64(u64 m, int k) { return (u32)(m >> k) & 1u; }
KMC_DEV void kmc_launder(u32& x) { KMC_OPAQUE(x); }
KMC_DEV void kmc_launder(u64& x) { KMC_OPAQUE(x); }

// 64-bit fingerprint of a packed state.  Never 0 (0 marks an empty table slot).
// States of up to KMC_FOLD_MIN_WORDS - 1 words: every word is absorbed through a full-avalanche bijection (the splitmix64 /
// murmur3 finaliser: two multiplies, three xor-shifts).  A single multiply + xor-shift per word is NOT enough here: packed states
// are highly structured, differences that survive one weak round line up with differences in the next word and produce
// systematic collisions (seen as 32 missing states out of 75,569,791 on Kip320 3/5/5/2).
// Wider states (seven brokers with deep logs: ten words) absorb TWO words per 64 x 64 -> 128-bit multiply, high half folded onto
// low (the multiply-fold of wyhash: the two words multiply each other, every input bit reaches both halves of the product), and
// pass through one finaliser at the end: a third of the per-word chain's multiplies.  The fingerprint of a ten-word state was ~250
// of a flush's vector instructions, 13.5 times per tile — a quarter of BASELINE config 5's kernel: 26.6 -> 25.2 ms on one box,
// every count equal to the exact fixtures (profiles/r06_fingerprint.txt).  Its quality on real states — 20 M reachable ten-word
// states, collisions in 32 / 36 / 40-bit windows of the fingerprint AND of the chain before its finaliser against the birthday
// expectation, three seeds: ratios 0.97 - 1.03 everywhere, as for the per-word chain — is tools/fp_quality/.
KMC_HD inline u64 kmc_mix64(u64 x) {
    x ^= x >> 30;
    x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27;
    x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}
// (The threshold is part of what a fingerprint MEANS — the host computes fingerprints with it for `contains`, traces, checkpoints
// (kmc_engine_step.cpp's magic) and the owner of Init — so a device build cannot be given another one behind the host's back:
// only the host-side tools (tools/fp_quality) and a tuning build's A/B, whose searches ask the host for no fingerprint, may.)
#ifdef KMC_FOLD_MIN_WORDS
#if !defined(KMC_HOST_EMU) && !KMC_TUNING
#error "KMC_FOLD_MIN_WORDS is shared by host and device: only a tuning build (-DKMC_TUNING=1) may override it"
#endif
#else
#define KMC_FOLD_MIN_WORDS 8
#endif
// the 128-bit product of a and b, high half xor low half
KMC_HD inline u64 kmc_mum(u64 a, u64 b) {
    const unsigned __int128 p = (unsigned __int128)a * (unsigned __int128)b;
    return (u64)p ^ (u64)(p >> 64);
}
template <int W> KMC_HD inline u64 kmc_fingerprint(const u64* w, u64 seed) {
    u64 h = kmc_mix64(seed + 0x9E3779B97F4A7C15ull * (u64)(W + 1));
    if constexpr (W >= KMC_FOLD_MIN_WORDS) {
        // (a word equal to the first constant would blind its partner: packed states leave their top bits clear, the constant
        // does not; the second operand carries the running hash)
#pragma unroll
        for (int k = 0; k + 1 < W; k += 2) h = kmc_mum(w[k] ^ 0xe7037ed1a0b428dbull, w[k + 1] ^ h) + 0x9E3779B97F4A7C15ull;
        if constexpr ((W & 1) != 0) h = kmc_mum(w[W - 1] ^ 0xe7037ed1a0b428dbull, h ^ 0x8ebc6af09c88c6e3ull);
        h = kmc_mix64(h);
    } else {
#pragma unroll
        for (int k = 0; k < W; ++k) h = kmc_mix64(h ^ w[k]) + 0x9E3779B97F4A7C15ull;
    }
    return h ? h : 1ull;
}
// The home slot of a fingerprint in a seen-set of `cap` slots — any multiple of 64, so that a table can be sized to the HBM it has
// instead of to the power of two below it (round 6: the 6.45 G-state stretch with 128-bit entries fits 2^33 slots = 128 GiB at load
// 0.75, where linear probing walks 4 - 8 slots per lookup and the probe rate falls to a quarter, or 12.9 G slots = 192 GiB at
// load 0.5: profiles/r06_stretch.txt).  Bits 8..39 of the fingerprint scaled onto the 64-slot groups (one 32 x 32 -> high 32
// multiply), bits 0..5 within the group; bits 40..63 choose the owner shard (kmc_owner), so a shard's fingerprints still spread
// over its whole table.  The chain continues at the next slot, wrapping at cap.
KMC_HD inline u64 kmc_slot_of(u64 fp, u64 cap) {
    return ((((u64)(u32)(fp >> 8) * (u64)(u32)(cap >> 6)) >> 32) << 6) | (fp & 63ull);
}
KMC_HD inline u64 kmc_slot_next(u64 i, u64 cap) { return i + 1 == cap ? 0ull : i + 1; }
// owner shard of a fingerprint: its bits 40..63 scaled onto 0..nshards-1 (a multiply and a shift; a run-time `% nshards`
// on a 64-bit value is a ~100-instruction division on this ISA, once per successor)
KMC_HD inline u32 kmc_owner(u64 fp, u32 nshards) { return (u32)((((fp >> 40) & 0xFFFFFFull) * (u64)nshards) >> 24); }

