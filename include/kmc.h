/* kmc.h — C ABI of the MI355X-native explicit-state model checker for the Kafka replication
 * TLA+ specs (hachikuji/kafka-specification).
 *
 * What this boundary replaces.  The reference repository is ten .tla files; the engine that
 * checks them is TLC (tla2tools.jar), which is NOT part of /root/reference, so there is no
 * reference FFI to cite line by line.  The reference-side "interface" is the set of operator
 * names a TLC .cfg binds:
 *     INIT Init                       KafkaReplication.tla:109
 *     NEXT Next                       KafkaTruncateToHighWatermark.tla:33, Kip101.tla:49,
 *                                     Kip279.tla:53, Kip320.tla:150, Kip320FirstTry.tla:159,
 *                                     FiniteReplicatedLog.tla:115, IdSequence.tla:39
 *     INVARIANTS TypeOk WeakIsr StrongIsr LeaderInIsr
 *                                     KafkaReplication.tla:101,320,334,345
 *     CONSTANTS Replicas LogSize MaxRecords MaxLeaderEpoch     KafkaReplication.tla:32-36
 *               (LogRecords, Nil: FiniteReplicatedLog.tla:22-26; MaxId: IdSequence.tla:22)
 * and the TLC classes this library stands in for are [TLC-recall, unverifiable here]:
 *     tlc2.tool.fp.FPSet.put(long) / contains / size   -> the HBM fingerprint table
 *     tlc2.tool.queue.IStateQueue.sEnqueue / sDequeue   -> the frontier arrays
 *     tlc2.tool.Worker.run()                            -> kmc_run's per-level kernels
 *     tlc2.tool.ModelChecker (extends AbstractChecker)  -> kmc_open / kmc_run / kmc_result
 * The ABI is coarse-grained on purpose: one call runs the whole BFS on the device (a JNI call
 * per fingerprint would be slower than TLC's own FPSet).  INTEGRATION.md shows the JNI stub.
 *
 * Conventions: plain C structs, caller-owned buffers, integer status returns (0 = OK), no
 * exceptions, no global state except the last-open error string; one host thread drives one
 * handle.  The library fails loudly (KMC_E_*) when no HIP device / compiler is available:
 * there is no CPU fallback.
 */
#ifndef KMC_H
#define KMC_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* models (root modules) */
#define KMC_IDSEQUENCE 0              /* IdSequence.tla            constants: max_id            */
#define KMC_FINITE_REPLICATED_LOG 1   /* FiniteReplicatedLog.tla   n_replicas, log_size, n_log_records */
#define KMC_TRUNCATE_TO_HW 2          /* KafkaTruncateToHighWatermark.tla                      */
#define KMC_KIP101 3                  /* Kip101.tla                                            */
#define KMC_KIP279 4                  /* Kip279.tla                                            */
#define KMC_KIP320 5                  /* Kip320.tla                                            */
#define KMC_KIP320_FIRST_TRY 6        /* Kip320FirstTry.tla                                    */
#define KMC_ASYNC_ISR 7               /* AsyncIsr.tla under the state constraint of models/MCAsyncIsr.tla (the spec is
                                         unbounded, AsyncIsr.tla:40-56,117): n_replicas (<= 6; replica 0 is `Leader`,
                                         :24), log_size = MaxOffset (:25), max_leader_epoch = MaxVersion (<= 7) */

/* invariant bits (invariant_mask) */
#define KMC_INV_TYPEOK 1u        /* TypeOk      KafkaReplication.tla:101 / FiniteReplicatedLog.tla:95 / IdSequence.tla:43 */
#define KMC_INV_WEAKISR 2u       /* WeakIsr     KafkaReplication.tla:320 */
#define KMC_INV_STRONGISR 4u     /* StrongIsr   KafkaReplication.tla:334 */
#define KMC_INV_LEADERINISR 8u   /* LeaderInIsr KafkaReplication.tla:345 */
/* KMC_ASYNC_ISR reuses the bit positions: 1 TypeOk (AsyncIsr.tla:62), 2 ValidHighWatermark (:161),
 * 4 LeaderOffsetInRange (models/MCAsyncIsr.tla; not in the reference) */
#define KMC_INV_VALIDHIGHWATERMARK 2u
#define KMC_INV_LEADEROFFSETINRANGE 4u

#define KMC_MAX_KINDS 16
#define KMC_MAX_SHARDS 8
#define KMC_SYMMETRY_MAX_REPLICAS 7   /* kmc_config.symmetry: |Replicas|! images per state, 5040 at most */
#define KMC_SEND_SUBS 8     /* sub-buffers per destination in the send area (spreads the append counters) */
#define KMC_COMM_ID_BYTES 128   /* an RCCL unique id (ncclUniqueId) */
#define KMC_EXCHANGE_STATS 64   /* longest statistics vector that can ride on a level's count exchange */

/* status codes */
#define KMC_OK 0
#define KMC_E_ARG 1        /* bad argument / unsupported constants           */
#define KMC_E_DEVICE 2     /* no usable HIP device, or a HIP call failed     */
#define KMC_E_COMPILE 3    /* hiprtc could not specialise the kernels        */
#define KMC_E_NOMEM 4      /* device allocation failed                       */
#define KMC_E_STATE 5      /* call out of order                              */

/* verdicts */
#define KMC_V_OK 0             /* "Model checking completed. No error has been found." */
#define KMC_V_INVARIANT 1      /* an invariant of invariant_mask is violated            */
#define KMC_V_DEADLOCK 2       /* check_deadlock and a state has no successor            */
#define KMC_V_TABLE_FULL 3     /* fingerprint table exhausted                            */
#define KMC_V_FRONTIER_FULL 4  /* a BFS level did not fit frontier_capacity              */
#define KMC_V_LEVEL_LIMIT 5    /* max_levels reached before exhaustion                   */
#define KMC_V_ERROR 6

typedef struct kmc_config {
    int32_t model;              /* KMC_* model id */
    int32_t n_replicas;         /* |Replicas|              (KafkaReplication.tla:33), <= 8 */
    int32_t log_size;           /* LogSize                 (:34) */
    int32_t max_records;        /* MaxRecords              (:35); record ids 0..MaxRecords-1 (:78) */
    int32_t max_leader_epoch;   /* MaxLeaderEpoch          (:36); epochs 0..MaxLeaderEpoch (:77), <= 7 */
    int32_t n_log_records;      /* |LogRecords| for FiniteReplicatedLog standalone (FiniteReplicatedLog.tla:24) */
    int64_t max_id;             /* MaxId for IdSequence standalone (IdSequence.tla:22) */
    uint32_t invariant_mask;    /* KMC_INV_* bits to check on every new state */
    int32_t check_deadlock;     /* TLC default is on; these bounded models need it off (-deadlock) */
    int32_t continue_on_violation; /* TLC -continue: keep exploring after the first violation */
    int32_t keep_trace;         /* keep predecessor fingerprints (8 B per table slot) for kmc_trace: the second word of the claim's own
                                   16-byte slot on 64-bit entries of states below eight words, a table of their own otherwise */
    int32_t device;             /* HIP device ordinal; -1 = host-only handle (pack/unpack/fingerprint only) */
    int32_t n_shards;           /* 1 = single GPU; P>1: this handle owns fingerprints with owner(fp)==shard_id */
    int32_t shard_id;
    uint64_t table_capacity;    /* fingerprint slots, rounded up to a multiple of 64 (any size: a table can fill the HBM there
                                   is, 288 GB hold 2^34 narrow or 12.9 G wide entries beside the frontiers); 0 = auto from free
                                   HBM (a power of two) */
    uint64_t frontier_capacity; /* states per frontier buffer; 0 = auto */
    uint64_t send_capacity;     /* n_shards>1: records per (destination, sub-buffer) per level; 0 = auto */
    uint64_t hash_seed;         /* results must not depend on it (collisions aside) */
    uint64_t max_levels;        /* 0 = unlimited (internal cap 4096) */
    const char* cache_dir;      /* compiled-kernel cache; NULL = $KMC_CACHE_DIR or <libdir>/kmc_cache */
    int32_t wide_fingerprint;   /* 1: 128-bit seen-set entries — every slot holds the 64-bit fingerprint AND a second,
                                   independent 64-bit hash of the state (16 bytes, one line fill per probe as before).  Two
                                   distinct states with the same fingerprint are then told apart instead of merged: the
                                   distinct-state count is exact up to a 128-bit collision (n^2 / 2^129).  TLC has no such
                                   switch (its FPSet is 64-bit); table_capacity still counts slots */
    int32_t symmetry;           /* 1: orbit counting.  The specs never tell two members of Replicas apart (KafkaReplication.tla
                                   :109-120, :158-310), so only the smallest image of a state under the |Replicas|!
                                   permutations is stored and expanded, and every count is weighted by the size of that
                                   state's orbit: distinct / generated / per-level / per-disjunct / deadlock / violation
                                   counts, verdict and depth are those of the plain search (and of TLC WITHOUT a SYMMETRY
                                   set — TLC's own SYMMETRY reports the reduced counts) from ~1/|Replicas|! of the probes.
                                   Kafka family and FiniteReplicatedLog, |Replicas| <= KMC_SYMMETRY_MAX_REPLICAS.  Traces are real
                                   behaviours (each step a successor of the one before), not chains of representatives.
                                   n_shards > 1: successors travel as representatives, every shard weighs the counters of
                                   kmc_step_finish / kmc_step_check_frontier itself (the states it claimed, the expansions
                                   it ran), so the sums over the shards are the plain search's numbers; predecessor links are those of the
                                   representatives, and a trace is replayed through kmc_successors like any other */
} kmc_config;

typedef struct kmc_level_info {
    uint64_t depth;             /* 1-based: the initial state is depth 1 */
    uint64_t new_states;        /* distinct states first seen at this depth (this shard); kmc_config.symmetry: as the plain
                                   search counts them (orbit sizes summed), like every count below */
    uint64_t generated_total;   /* running total, TLC's "states generated" (initial state included) */
    uint64_t distinct_total;    /* running total, TLC's "distinct states found" */
    double seconds;             /* wall time since kmc_run started */
    /* filled by kmc_step_finish only — the expansion that produced this level (this shard): */
    uint64_t generated_level[KMC_MAX_KINDS]; /* successors generated per action kind */
    uint64_t violation_count[4]; /* states of the EXPANDED level (depth-1) violating each invariant */
    uint64_t violation_fp[4];    /* smallest violating fingerprint per invariant, 0 = none */
    uint64_t outside_violation_count[4]; /* KMC_ASYNC_ISR: successors (depth) OUTSIDE the state constraint violating
                                            each invariant, counted per generation: TLC checks invariants on them
                                            although it neither fingerprints nor explores them [TLC-recall] */
    uint64_t outside_violation_fp[4];
    uint64_t deadlocks_level;    /* states of the expanded level without successors */
    uint64_t send_filtered;      /* n_shards>1: remote successors the sender-side duplicate filter did not ship */
    uint32_t error_flags;        /* 1 frontier full, 2 table full, 4 send area full */
    uint32_t pad_;
} kmc_level_info;

typedef void (*kmc_progress_cb)(const kmc_level_info* info, void* user);

typedef struct kmc_result {
    uint64_t generated;         /* states generated (every satisfying binding; initial state included) */
    uint64_t distinct;          /* distinct states found */
    uint64_t depth;             /* "The depth of the complete state graph search" */
    uint64_t queue_left;        /* states left on the frontier when the run stopped */
    int32_t verdict;            /* KMC_V_* */
    int32_t violated_invariant; /* index 0..3 (TypeOk, WeakIsr, StrongIsr, LeaderInIsr), -1 = none */
    uint64_t violation_depth;   /* depth of the first level holding a violating / deadlocked state */
    uint64_t violation_count[4];/* states at that depth violating each checked invariant (every state is
                                   checked once, when it is expanded; on a stopping violation the level
                                   produced by that expansion is not counted) */
    uint64_t violation_fp;      /* fingerprint of the reported witness (the smallest one) */
    uint64_t deadlock_states;   /* expanded states without successors (counted even when unchecked) */
    uint64_t action_generated[KMC_MAX_KINDS]; /* per Next disjunct, in the module's order */
    uint64_t n_levels;
    uint64_t table_capacity, frontier_capacity;
    double seconds_total;       /* wall time of kmc_run */
    double seconds_expand;      /* sum of k_expand kernel durations (HIP events on the engine stream) */
    uint64_t expand_launches;
    uint64_t state_words;       /* W: 64-bit words per packed state */
    uint64_t state_bits;
    uint64_t generated_repeats; /* of `generated`: successors TLC's enumeration yields a second time because two disjuncts
                                   of one binding hold at once (Kip279.tla:47-51, Kip320.tla:82-83); each is one successor
                                   and one seen-set probe, so probes = generated - 1 - generated_repeats */
    uint64_t orbit_representatives; /* kmc_config.symmetry: the states actually stored and expanded (one per orbit);
                                   without it equal to `distinct` */
    /* the rest of a run's device time, so that seconds_total is accounted for (HIP events on the engine stream): */
    double seconds_inv;         /* the invariant pass over a frontier that is not expanded (k_inv: the last level under max_levels) */
    double seconds_clear;       /* clearing the seen-set [+ predecessor table] at the start of the run */
    uint64_t inv_launches;
} kmc_result;

/* One record per expansion of the last search (kmc_run / kmc_resume, or the kmc_step_* calls of one shard): what a user tunes
 * constants and capacities by — TLC's -coverage / Progress lines give the same view per disjunct [TLC-recall].  The per-disjunct
 * counts follow the module's Next (Kip320.tla:150-159 etc.: kmc_action_name). */
typedef struct kmc_level_stat {
    uint64_t depth;             /* depth of the level this expansion PRODUCED (Init is depth 1 and has no record) */
    uint64_t frontier;          /* states expanded = the size of level depth-1, as stored (orbit representatives under symmetry) */
    uint64_t new_states;        /* distinct states first found by this expansion (the plain search's count) */
    uint64_t stored_new;        /* ... as stored */
    uint64_t generated[KMC_MAX_KINDS]; /* successors generated per disjunct of Next (the plain search's count) */
    uint64_t probes;            /* seen-set probes of this expansion */
    uint64_t deadlocks;         /* expanded states without successor */
    double table_load;          /* stored states / table slots after this expansion */
    double expand_ms;           /* duration of this expansion's k_expand launch(es), HIP events on the engine stream */
} kmc_level_stat;

typedef struct kmc_handle kmc_handle;

/* Compile (or load from cache) the kernels specialised for cfg's constants, allocate the
 * table and frontiers on cfg->device.  On failure *out is NULL and kmc_last_error() explains. */
int kmc_open(const kmc_config* cfg, kmc_handle** out);
/* Compile-and-cache only; needs no GPU (used by the build step).  arch NULL = "gfx950".
 * The kernels of one configuration live in three cached code objects — the search's own (k_expand for one GPU with the small
 * kernels around it: kmc_run needs nothing else), k_expand for the level-step interface (owner bucketing: kmc_step_*) and
 * k_expand as an enumerator (kmc_successors, trace replay) — the last two joining a handle when first asked for.
 * kmc_precompile builds all three; kmc_precompile_mode one of them (mode 0 / 1 / 2 in that order, -1 = all), so that a build
 * script can spread them over its workers. */
int kmc_precompile(const kmc_config* cfg, const char* arch);
int kmc_precompile_mode(const kmc_config* cfg, const char* arch, int32_t mode);
/* Path of the cached code object cfg's kernels are loaded from (specialised first if absent; no GPU needed).  A profile
 * records a hash of its kernels' machine code, so that a number is quoted only for the code it was measured on. */
int kmc_code_object_path(const kmc_config* cfg, const char* arch, char* out, uint64_t cap);
/* Whole breadth-first search on the device; cb (may be NULL) is called once per level. */
int kmc_run(kmc_handle* h, kmc_progress_cb cb, void* user);
int kmc_result_get(kmc_handle* h, kmc_result* out);
/* Where the wall time of a handle went BESIDES the search (what a front end's user waits for on top of kmc_result.seconds_total;
 * SURVEY 8d "wall time-to-exhaustive"): filled by kmc_open and the first kmc_run. */
typedef struct kmc_timing {
    double hip_init_s;      /* first HIP call of the process: runtime + device initialisation (0 when another handle paid it) */
    double code_object_s;   /* reading (or, cache cold, specialising) the code object and loading it into the device */
    double alloc_s;         /* hipMalloc / hipHostMalloc of the seen-set, frontiers, control blocks */
    double first_clear_s;   /* the first run's clear of the seen-set [+ predecessor table]: first touch of freshly mapped HBM */
    double open_s;          /* kmc_open, whole */
    uint64_t device_bytes;  /* allocated on the device by kmc_open */
} kmc_timing;
int kmc_timing_get(kmc_handle* h, kmc_timing* out);
/* TLC -checkpoint / -recover analogues [TLC-recall].  Save after a run stopped early (max_levels):
 * the fingerprint table, the current frontier and all counters go to `path`.  Load into a handle
 * opened with the same constants, hash seed and capacities, then kmc_resume continues the
 * search as if it had never stopped (max_levels of the new handle applies). */
int kmc_checkpoint_save(kmc_handle* h, const char* path);
int kmc_checkpoint_load(kmc_handle* h, const char* path);
int kmc_resume(kmc_handle* h, kmc_progress_cb cb, void* user);
/* Per-level sizes of the last run: fills up to cap entries, returns the number of levels. */
uint64_t kmc_level_sizes(kmc_handle* h, uint64_t* out, uint64_t cap);
/* Per-expansion records of the last run (see kmc_level_stat): fills up to cap entries, returns their number. */
uint64_t kmc_level_stats(kmc_handle* h, kmc_level_stat* out, uint64_t cap);
/* Who compiles, and whose code runs.  A box holds two hiprtc / comgr builds (the system ROCm's and the one bundled with
 * PyTorch) that emit different instructions for the same source; a process is bound to one of them.  The compiler's identity
 * (the HIP runtime build number of the bundle it belongs to, e.g. 70051831) is part of every cached code object's FILE NAME, so
 * two compilers never write the same file, and kmc_open prefers the object of the PINNED compiler (the one the build step
 * specialises with and the profiles were measured on; KMC_COMPILER_PIN overrides) when the cache holds it.
 * which = 0: this process's compiler; 1: the pinned one (0 = none pinned). */
int64_t kmc_compiler_identity(int32_t which);
void kmc_close(kmc_handle* h);
const char* kmc_last_error(void);

/* --- states as data ---------------------------------------------------------------------
 * Packed states are state_words little-endian uint64 (layout: csrc/kmc_layout.h).  The
 * "canonical bytes" form — what traces are returned in — is one byte per field:
 *   Kafka family: per replica r a block of 5+LogSize bytes at r*(5+LogSize):
 *       [0] endOffset  [1] hw  [2] leaderEpoch+1 (Nil -> 0)  [3] leader+1 ("NONE" -> 0)
 *       [4] isr bitmask  [5+o] record at offset o: 0 = Nil, else 1 + id*(MaxLeaderEpoch+1) + epoch
 *     then globals: [0] nextRecordId [1] nextLeaderEpoch [2] quorum.leaderEpoch+1 [3] quorum.leader+1
 *       [4] quorum.isr, then for e in 0..MaxLeaderEpoch: leader+1, isr of the request with that epoch
 *       (zeros while e >= nextLeaderEpoch);
 *   AsyncIsr: [0] controllerState.isr [1] .version [2] leaderState.isr [3] .version [4] .pendingIsr
 *       [5] .pendingVersion+1 (Nil -> 0) [6+r] .offsets[r]; then requests: per version 0..MaxVersion a
 *       bitset over isr masks (ceil(2^N/8) bytes); then updates: per version 1..MaxVersion+1 the isr
 *       written at that version (0 while version > controllerState.version);
 *   FiniteReplicatedLog: per replica [endOffset, record(0 = Nil | 1..K) x LogSize];
 *   IdSequence: nextId as 8 little-endian bytes. */
uint64_t kmc_state_words(kmc_handle* h);
uint64_t kmc_canon_bytes(kmc_handle* h);
int kmc_unpack_state(kmc_handle* h, const uint64_t* words, uint8_t* canon);
int kmc_pack_state(kmc_handle* h, const uint8_t* canon, uint64_t* words);
uint64_t kmc_fingerprint_of(kmc_handle* h, const uint64_t* words);
/* The representative of a packed state's orbit under the permutations of Replicas — the smallest image, words compared in
 * order as unsigned values; with five to seven replicas the smallest among the images whose replicas stand in ascending
 * order of a name-independent key (log, offsets, epoch, what the state says about the replica: kmc_layout.h,
 * kmc_replica_key_generic) — and the number of permutations that leave the state unchanged: its orbit has
 * |Replicas|! / *stabiliser members.  What a kmc_config.symmetry search stores; works on host-only handles. */
int kmc_canonical_state(kmc_handle* h, const uint64_t* words, uint64_t* representative, int32_t* stabiliser);
/* Copy the current frontier (the last completed level) to the host as packed AoS records. */
int kmc_frontier_states(kmc_handle* h, uint64_t* words, uint64_t cap_states, uint64_t* n_out);
/* All successors of one packed state, straight from the device kernels: writes up to cap
 * records of (state_words + 2) uint64: state, fingerprint, action kind — TLC's enumeration of Next on that state: one
 * record per satisfying binding AND disjunct (a successor that two disjuncts of one binding yield, Kip279.tla:47-51 /
 * Kip320.tla:82-83, is listed twice, as `generated` counts it). */
int kmc_successors(kmc_handle* h, const uint64_t* words, uint64_t* out, uint64_t cap, uint64_t* n_out);
/* The invariants (KMC_INV_* bits of `mask`, whatever cfg->invariant_mask says) each of n packed states violates, from
 * the device's own predicate — the one k_expand applies to the states it expands (TypeOk / WeakIsr / StrongIsr /
 * LeaderInIsr: KafkaReplication.tla:101,320,334,345).  With kmc_successors this is the per-state differential surface:
 * Next and the invariants on any state, reachable or not. */
int kmc_check_states(kmc_handle* h, const uint64_t* words, uint64_t n, uint32_t mask, uint32_t* violated);
/* Counterexample of the last run (needs keep_trace): canonical-byte states from the initial
 * state to the witness, with the action kind that produced each (-1 for the initial state). */
int kmc_trace(kmc_handle* h, uint8_t* canon_states, int32_t* kinds, uint64_t cap, uint64_t* n_out);
/* tlc2.tool.fp.FPSet.contains analogue [TLC-recall]: was this packed state reached by the last run
 * (on this shard)?  A host-side probe of the device table (a few 8-byte reads). */
int kmc_contains(kmc_handle* h, const uint64_t* words, int32_t* present);
/* The witness of the reported violation / deadlock as a packed state. */
int kmc_witness(kmc_handle* h, uint64_t* words);
/* Building blocks of a trace across shards (keep_trace): the predecessor fingerprint recorded for a
 * fingerprint this handle owns (*found = 0 when it is not in this shard's table; 0 = the initial state),
 * and the packed initial state.  The multi-GPU driver walks the chain owner by owner, then replays it
 * with kmc_successors on any shard. */
int kmc_pred_of(kmc_handle* h, uint64_t fp, uint64_t* pred, int32_t* found);
int kmc_init_state(kmc_handle* h, uint64_t* words);
/* The shard that owns a fingerprint among n_shards (the partition k_expand buckets by: bits 40..63 of the fingerprint
 * scaled onto 0..n_shards-1).  Pure function, no handle; -1 for n_shards outside 1..KMC_MAX_SHARDS. */
int32_t kmc_owner_of(uint64_t fp, int32_t n_shards);

const char* kmc_model_name(int32_t model);
const char* kmc_action_name(int32_t model, int32_t kind);
int32_t kmc_action_count(int32_t model);
const char* kmc_invariant_name(int32_t index);                      /* Kafka-family names */
const char* kmc_model_invariant_name(int32_t model, int32_t index); /* per model (KMC_ASYNC_ISR has its own) */

/* --- level-step interface for the multi-GPU driver (n_shards > 1) ---------------------------
 * One BFS level = kmc_step_expand (bucket successors by owner into per-destination send
 * buffers) -> the caller exchanges the buffers (RCCL all-to-all-v) -> kmc_step_insert on each
 * received buffer -> kmc_step_finish.  All pointers are device pointers on cfg->device. */
int kmc_step_begin(kmc_handle* h);                       /* reset, insert Init on its owner */
/* send_counts: KMC_MAX_SHARDS * KMC_SEND_SUBS entries, [destination][sub-buffer] */
int kmc_step_expand(kmc_handle* h, uint64_t* send_counts);
int kmc_step_send_buffer(kmc_handle* h, int32_t dst, int32_t sub, void** dev_ptr, uint64_t* record_words);
/* Let the caller own the send area: [n_shards][KMC_SEND_SUBS][records_per_sub_buffer] records of
 * (state_words+1) uint64, e.g. a torch tensor whose slices are handed to the collective. */
int kmc_step_set_send_buffer(kmc_handle* h, void* dev_ptr, uint64_t records_per_sub_buffer);
int kmc_step_insert(kmc_handle* h, const void* dev_records, uint64_t n_records);
int kmc_step_finish(kmc_handle* h, kmc_level_info* info); /* info->new_states: this shard's next frontier */
int kmc_step_set_verdict(kmc_handle* h, int32_t verdict); /* driver-decided global stop reason */
/* Multi-GPU checkpoints: between kmc_step_finish and the next kmc_step_expand every shard may save its own table /
 * frontier with kmc_checkpoint_save (after kmc_step_set_verdict(h, KMC_V_LEVEL_LIMIT)); the driver keeps the global
 * counters.  kmc_checkpoint_load into a handle with the same constants, capacities and shard id, then kmc_step_resume
 * puts it back at that level boundary. */
int kmc_step_resume(kmc_handle* h);
/* Invariant-only pass over the CURRENT (unexpanded) frontier — the last level under max_levels; fills
 * info->violation_count / violation_fp. */
int kmc_step_check_frontier(kmc_handle* h, kmc_level_info* info);
/* KMC_ASYNC_ISR: look for a violating successor outside the state constraint (it is in no table) among the
 * successors of the level the last kmc_step_finish retired; valid until the next kmc_step_expand. */
int kmc_step_find_outside(kmc_handle* h, uint64_t fp, uint64_t* words, uint64_t* parent_fp, int32_t* found);

/* --- the per-level exchange under the ABI (one process per GPU, RCCL over xGMI) -----------
 * Stands where distributed TLC's FPSet servers / state queues would stand [TLC-recall]; the reference defines
 * nothing here.  Bootstrap: rank 0 calls kmc_comm_unique_id and hands the 128 bytes to the other ranks by any
 * means (the Python driver broadcasts them through torch.distributed's store); every rank then calls
 * kmc_comm_init on its handle (rank = cfg.shard_id, size = cfg.n_shards).  librccl is bound with dlopen at that
 * point — single-GPU users never need it.  Per BFS level, after kmc_step_expand:
 *   kmc_step_exchange_counts   all-gather of the send counts + a caller-defined statistics vector (summed over
 *                              ranks into stats_sum: the caller decides termination / verdicts from it); one
 *                              stream synchronisation;
 *   kmc_step_exchange_payload  grouped ncclSend/ncclRecv of every non-empty run straight from the send area into
 *                              the receive area, then ONE k_insert over what arrived, all queued on the engine's
 *                              stream (no host wait);
 *   kmc_step_finish            as before.
 * The engine must own the send area (no kmc_step_set_send_buffer). */
int kmc_comm_unique_id(uint8_t* id /* KMC_COMM_ID_BYTES */);
int kmc_comm_init(kmc_handle* h, const uint8_t* id /* KMC_COMM_ID_BYTES */);
/* all-gather + a send/receive ring of a known pattern on the handle's communicator and stream, verified */
int kmc_comm_selftest(kmc_handle* h);
int kmc_step_exchange_counts(kmc_handle* h, const int64_t* stats, int32_t n_stats, int64_t* stats_sum,
                             uint64_t* recv_records);
int kmc_step_exchange_payload(kmc_handle* h);
/* kmc_step_expand and kmc_step_exchange_counts in one call with ONE stream synchronisation: the send counts go from
 * k_expand's control block into the all-gather's row on the device (the statistics vector rides in a kernel argument),
 * so nothing crosses the host between the expansion and the collective.  send_counts may be NULL. */
int kmc_step_expand_counts(kmc_handle* h, const int64_t* stats, int32_t n_stats, int64_t* stats_sum,
                           uint64_t* recv_records, uint64_t* send_counts);
/* A whole level of a shard — expansion, count exchange, payload, insert — as a pipeline of `parts` (2, 4 or 8) groups of
 * frontier segments: part c expands into send area c mod 2 while part c-1's counts are gathered and its records travel
 * and are inserted on a second stream, so a level's wire hides behind its own expansion.  One host wait per part; for
 * levels large enough to pay for them.  Replaces kmc_step_expand_counts + kmc_step_exchange_payload; kmc_step_finish next. */
int kmc_step_level_parts(kmc_handle* h, int32_t parts, const int64_t* stats, int32_t n_stats, int64_t* stats_sum,
                         uint64_t* recv_records);
/* The same level step for n_shards handles living in one process on one device (RCCL refuses two ranks on one
 * device): counts and statistics ([n_shards][n_stats]) are combined on the host, runs move device-to-device. */
int kmc_step_exchange_local(kmc_handle** shards, int32_t n_shards, const int64_t* stats, int32_t n_stats,
                            int64_t* stats_sum);
int kmc_step_deliver_local(kmc_handle** shards, int32_t n_shards);
/* The plan both transports execute, as a pure function (testable without a device).  counts[(s*P + d)*KMC_SEND_SUBS
 * + sub] = records shard s sends to shard d from sub-buffer sub.  Writes up to cap (peer, offset_words, words)
 * triples per list: the messages shard `me` sends (offsets into its send area) and receives (offsets into its
 * receive area), in posting order. */
int kmc_exchange_plan(const uint64_t* counts, int32_t n_shards, int32_t me, uint64_t send_cap, uint64_t rec_words,
                      uint64_t* sends, uint64_t* recvs, uint64_t cap, uint64_t* n_sends, uint64_t* n_recvs,
                      uint64_t* recv_records);

#ifdef __cplusplus
}
#endif
#endif
