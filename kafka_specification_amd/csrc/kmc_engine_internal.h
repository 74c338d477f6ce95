// kmc_engine_internal.h — what the translation units of the host engine share: the handle, the launch helpers, the code-object
// cache, a level's bookkeeping.  Nothing here is part of the C ABI (include/kmc.h); everything is hidden from the dynamic symbol
// table of libkmc.so.
//   kmc_engine_codeobj.cpp   validation of a configuration, the text handed to hiprtc, the on-disk cache, the register-budget rule
//   kmc_engine_core.cpp      kernel launches, a level's counters folded into the result (absorb), conservation, Init
//   kmc_engine_open.cpp      names, precompile, kmc_open / kmc_close, pack / unpack / fingerprint / representative of a state
//   kmc_engine_run.cpp       kmc_run's level loop (chained launches), results, kmc_successors / kmc_check_states, traces
//   kmc_engine_step.cpp      checkpoint / recover, the level-step interface of one shard
//   kmc_engine_exchange.cpp  the per-level exchange: RCCL bound with dlopen, the plan, one-shot and pipelined levels, logical shards
// The engine stands in for TLC's ModelChecker + Worker threads [TLC-recall; TLC is not part of /root/reference].  No CPU fallback
// exists: without a HIP device or compiler every entry point fails with KMC_E_DEVICE / KMC_E_COMPILE.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <rccl/rccl.h>   // types and prototypes only: librccl is bound with dlopen when a communicator is created

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/kmc.h"
// (the device headers are written for hiprtc, whose builds do not run under -Wextra: two parameters that only some builds read)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wunused-parameter"
#include "kmc_device.h"   // host-visible parts: KmcArgs, KmcLevelCtl, layout, fingerprint
#pragma clang diagnostic pop

// Levels kmc_run queues back to back before it waits (no progress callback): see run_levels.
#define KMC_CHAIN 32
#define KMC_CTL_SLOTS (3 + KMC_CHAIN)   // two alternating levels + one auxiliary + one per chained level

// KMC_VERIFY: both builds carry the fingerprint checksum (KMC_CHECKSUM, kmc_common.h); the second one differs in how it is
// compiled — optimisation level and a quarter of the occupancy target, i.e. another register allocation
#define KMC_VERIFY_PRIMARY_OPTIONS "-DKMC_CHECKSUM=1"
// The second build of the differential self-check: another optimisation level, a quarter of the occupancy target, the
// fingerprint checksum — and, for the Kafka models, ANOTHER LOWERING OF THE GUARDS: KmcKafka::guard<K> looped per kind over
// a run-time binding instead of the straight-line block of every instance's inst<I> (kmc_kafka.h, RUNTIME_GUARDS).
#define KMC_VERIFY_OPTIONS "-O1 -DKMC_MIN_WAVES=2 -DKMC_CHECKSUM=1 -DKMC_RT_GUARDS_MIN_INSTANCES=0 -DKMC_WITH_DRY=1"

// The compiler every process prefers cached objects of (kmc_engine_codeobj.cpp, compiler_id_pinned): the hiprtc / comgr bundled
// with PyTorch 2.10.0+rocm7.0 (HIP 7.0.51831) — what build() specialises with, what the driver's bench runs under, and the
// faster of the two on the four profiled kernels (profiles/r06_compiler_ab.txt).
#define KMC_PINNED_COMPILER 70051831LL

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return kmc_engine::fail(KMC_E_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#pragma GCC visibility push(hidden)

struct kmc_handle {
    kmc_config cfg{};
    KmcLayout lay{};
    int W = 0;
    std::string kname;
    std::string arch;
    kmc_timing timing{};                    // where the wall time outside the search went (kmc_timing_get)
    bool first_clear_timed = false;
    std::string cache_dir;                  // kmc_config.cache_dir, copied: the later code objects (ensure_mode) are looked up
                                            // long after kmc_open returned and the caller's string may be gone
    std::string jit_defines;                // KMC_JIT_DEFINES as it stood when the handle was opened
    int layout_mode = KMC_LAYOUT_AUTO;      // KMC_LAYOUT likewise (the later code objects are specialised for the same layout)
    bool verify = false;                    // KMC_VERIFY likewise
    hipModule_t mod = nullptr;              // the search's code object: k_expand (LOCAL), k_inv, k_insert, k_init, k_find, k_packrow
    hipFunction_t f_expand = nullptr, f_inv = nullptr, f_insert = nullptr, f_init = nullptr, f_find = nullptr, f_packrow = nullptr;
    hipFunction_t f_expand_dry = nullptr;   // only in a KMC_TUNING build of `mod` (KMC_DRYRUN / KMC_SHADOW tuning aids)
    hipModule_t mod_sh = nullptr, mod_en = nullptr;   // k_expand in SHARDED / ENUM mode: loaded when first needed (ensure_mode)
    hipFunction_t f_expand_sh = nullptr, f_expand_en = nullptr;
    hipModule_t mod_verify = nullptr;       // KMC_VERIFY=1: a second, differently compiled code object whose dry k_expand regenerates every level
    hipFunction_t f_expand_verify = nullptr;
    uint64_t verify_levels = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_chain[2 * KMC_CHAIN] = {nullptr};  // chained launches: one pair per level of a batch
    hipEvent_t ev_aux[2] = {nullptr, nullptr};       // around the clear of the seen-set (reset_run) and around k_inv
    bool clear_pending = false;                      // ev_aux brackets a clear whose duration has not been read yet
    std::vector<kmc_level_stat> level_stats;         // one record per expansion of the last search (kmc_level_stats)
    double step_expand_ms = 0;                       // stepping: k_expand time of the level being built (kmc_step_finish books it)
    int rec_words = 0;  // exchange / insert record size: W, +1 when predecessor fingerprints are kept
    int n_cus = 256;
    int blocks_per_cu = 4;  // k_expand residency, from the occupancy query at open
    u64 *table = nullptr, *pred = nullptr, *table2 = nullptr;
    std::vector<std::pair<void*, size_t>> mapped;   // buffers that are mapped ranges of chunks, not hipMalloc's (seen_set_alloc): base, bytes
    u64* sent = nullptr;      // n_shards > 1: sender-side filter of fingerprints already shipped
    uint64_t sent_cap = 0;
    uint64_t table_cap = 0;      // slots
    uint64_t slot_words = 1;     // 64-bit words per slot: 1, or 2 with kmc_config.wide_fingerprint (fingerprint + check word)
    bool paired = false;         // keep_trace on 64-bit entries: 16-byte slots of fingerprint + predecessor (pred = table + 1, KMC_FLAG_PAIRED)
    uint64_t stride_words() const { return paired ? 2 : slot_words; }        // words from one slot to the next
    uint64_t pred_stride() const { return paired ? 2 : 1; }                  // ... and from one predecessor to the next
    uint64_t inserted_level = 0; // stepping: records handed to k_insert since the last kmc_step_finish (conservation check)
    u64* frontier[2] = {nullptr, nullptr};
    uint64_t fcap = 0;
    KmcLevelCtl* ctl = nullptr;       // 3 device slots: two alternating levels + one auxiliary
    KmcLevelCtl* ctl_host = nullptr;  // pinned
    u64* scratch = nullptr;      // device: init record / find result / enum input
    uint64_t* scratch_host = nullptr; // pinned
    u64* enum_out = nullptr;     // device: ENUM records
    uint64_t enum_cap = 4096;
    u64* send = nullptr;         // SHARDED send buffers
    uint64_t send_cap = 0;
    bool send_owned = true;
    // run state
    int cur = 0;                 // frontier[cur] holds the last completed level
    uint64_t n_cur = 0;          // its size on this shard
    uint64_t seg_n[KMC_SEGS] = {0};  // ... per segment
    uint64_t prev_seg_n[KMC_SEGS] = {0};  // stepping: segments of the level kmc_step_finish just retired (in frontier[cur ^ 1])
    uint64_t seg_cap = 0;        // slots per segment
    uint64_t level = 0;          // number of completed levels
    bool stepping = false, step_expanded = false, restored = false;
    std::vector<uint64_t> levels;
    std::vector<uint64_t> init_words, witness;
    bool have_witness = false, have_deadlock = false;
    bool witness_outside = false;      // the witness is a successor outside the state constraint:
    uint64_t witness_parent_fp = 0;    //   it is in no table; this is the expanded state it was generated from
    kmc_result res{};
    // kmc_config.symmetry: the frontier / table hold one state per orbit; res.distinct and `levels` are the WEIGHTED
    // (= plain-search) numbers, raw_levels the representatives per level; nfact = |Replicas|!
    uint64_t nfact = 1;
    int planes = 0;              // words per state in a frontier: W, and under symmetry one more — the order of the state's stabiliser
    double t_start = 0;
    double dry_seconds = 0;
    uint64_t prof[8] = {0}, prof_dry[8] = {0};
    // per-level exchange under the ABI (n_shards > 1): RCCL communicator, receive area, count/statistics rows
    ncclComm_t comm = nullptr;
    u64* recv = nullptr;             // device: everything this shard receives in one level, contiguous
    uint64_t recv_cap = 0;           // records
    // the within-level pipeline (kmc_step_level_parts): a second stream for a part's collective, transfer and insert, the
    // rows of two parts in flight, and the events that order the two streams
    hipStream_t xstream = nullptr;
    hipEvent_t ev_row[2] = {nullptr, nullptr}, ev_xfer[2] = {nullptr, nullptr};
    int64_t* prow_dev[2] = {nullptr, nullptr};
    int64_t* prow_host[2] = {nullptr, nullptr};
    int64_t* xrow_dev = nullptr;     // device: this rank's row, then the gathered rows of all ranks
    int64_t* xrow_host = nullptr;    // pinned: the same
    uint64_t last_send_counts[KMC_MAX_SHARDS * KMC_SEGS] = {0};  // of the last kmc_step_expand
    std::vector<uint64_t> xcounts;   // [source][destination][sub-buffer] of the level being exchanged
    bool xcounts_valid = false;
};

namespace kmc_engine {

// ---- kmc_engine_codeobj.cpp ----
extern thread_local std::string g_err;
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
extern const char* const MODEL_NAMES[8];
extern const char* const INV_NAMES[4];
extern const char* const INV_NAMES_ASYNC[4];
extern const char* const KINDS_ASYNC[7];
extern const char* const KINDS_BASE[9];
extern const char* const KINDS_KIP320[9];
extern const char* const KINDS_FIRST[10];
extern const char* const KINDS_FRL[3];
extern const char* const MODE_SUFFIX[3];
extern const char* const MODE_FILE_TAG[3];
int64_t compiler_id_mine();
int64_t compiler_id_pinned();
bool validate(const kmc_config& c, KmcLayout* lay, std::string* name, std::string* inst, int layout_mode = -2);
int layout_mode_from_env();   // KMC_LAYOUT as it stands now (-1: not a layout's name)
int get_code_object(const kmc_config& cfg, const std::string& arch, std::vector<char>* code, std::string* kname,
                    const char* extra_options = nullptr, std::string* path_out = nullptr, unsigned mode = KMC_MODE_LOCAL,
                    const std::string* jit_defines = nullptr, int layout_mode = -2);
uint64_t pow2_floor(uint64_t x);
uint64_t pow2_ceil(uint64_t x);
double now_s();

// ---- kmc_engine_core.cpp ----
int launch(kmc_handle* h, hipFunction_t f, const KmcArgs& a, unsigned grid, hipStream_t stream = nullptr);
int ensure_mode(kmc_handle* h, unsigned mode);
int launch_expand(kmc_handle* h, unsigned mode, const KmcArgs& a, unsigned grid, hipStream_t stream = nullptr, bool verify = false);
int launch_inv(kmc_handle* h, const KmcArgs& a, uint64_t n);
// the seen-set's memory: a range of addresses mapped from 8 MiB physical chunks (hipMalloc when that cannot be had) — see the definition
#ifndef KMC_SEEN_SET_SPREAD_DEFAULT
#define KMC_SEEN_SET_SPREAD_DEFAULT 1   // KMC_SEEN_SET_SPREAD: the seen-set's chunks lie over this many times their own size of the HBM
                                        // (seen_set_alloc: spacers).  1 = as the chunks come: the spread costs seconds at open and close
#endif
u64* seen_set_alloc(kmc_handle* h, size_t bytes, bool chunks = true);
void seen_set_free(kmc_handle* h, u64* p);
KmcArgs base_args(kmc_handle* h, int ctl_slot);
uint64_t max_fanout(const kmc_handle* h);
unsigned expand_grid(kmc_handle* h, uint64_t n);
int read_ctl(kmc_handle* h, int slot);
int zero_ctl(kmc_handle* h, int slot);
uint64_t produced_segments(kmc_handle* h, const KmcLevelCtl& c, uint64_t seg[KMC_SEGS]);
int find_state(kmc_handle* h, const u64* frontier, const uint64_t seg[KMC_SEGS], uint64_t fp, std::vector<uint64_t>* out);
int find_outside_witness(kmc_handle* h, const u64* frontier, const uint64_t seg[KMC_SEGS], uint64_t fp);
int reset_run(kmc_handle* h);
uint64_t weighted(const kmc_handle* h, uint64_t raw, uint64_t corr);
uint64_t queue_now(const kmc_handle* h);
void book_level(kmc_handle* h, uint64_t produced, const KmcLevelCtl& c);
int check_conservation(kmc_handle* h, const KmcLevelCtl& c, uint64_t inserted);
bool absorb(kmc_handle* h, const KmcLevelCtl& c, const u64* parent_frontier, const uint64_t* parent_seg, int* rc);
int do_begin(kmc_handle* h);
// one expansion enters the per-level records (kmc_level_stats): `frontier` stored states were expanded, `produced` were found
void note_level(kmc_handle* h, const KmcLevelCtl& c, uint64_t frontier, uint64_t produced, double expand_ms);
// roctx ranges (one per BFS level, or per chain of levels when chained) — bound with dlopen when a profiler has
// loaded the library or KMC_ROCTX=1 asks for it; no-ops otherwise
void range_push(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
void range_pop();

// ---- kmc_engine_exchange.cpp ----
void comm_release(kmc_handle* h);

}  // namespace kmc_engine

#pragma GCC visibility pop
