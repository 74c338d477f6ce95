// kmc_engine_open.cpp — names, precompile, kmc_open / kmc_close, pack / unpack / fingerprint / representative of a state.
#include "kmc_engine_internal.h"
using namespace kmc_engine;

extern "C" {

const char* kmc_last_error(void) { return g_err.c_str(); }
const char* kmc_model_name(int32_t m) { return m >= 0 && m <= 7 ? MODEL_NAMES[m] : "?"; }
const char* kmc_invariant_name(int32_t i) { return i >= 0 && i < 4 ? INV_NAMES[i] : "?"; }
const char* kmc_model_invariant_name(int32_t model, int32_t i) {
    if (i < 0 || i >= 4) return "?";
    return model == KMC_ASYNC_ISR ? INV_NAMES_ASYNC[i] : INV_NAMES[i];
}
int32_t kmc_action_count(int32_t model) {
    switch (model) {
    case KMC_IDSEQUENCE: return 1;
    case KMC_FINITE_REPLICATED_LOG: return 3;
    case KMC_ASYNC_ISR: return 7;
    case KMC_KIP320_FIRST_TRY: return 10;
    case KMC_TRUNCATE_TO_HW: case KMC_KIP101: case KMC_KIP279: case KMC_KIP320: return 9;
    default: return 0;
    }
}
const char* kmc_action_name(int32_t model, int32_t kind) {
    if (kind < 0 || kind >= kmc_action_count(model)) return "?";
    switch (model) {
    case KMC_IDSEQUENCE: return "Next";
    case KMC_FINITE_REPLICATED_LOG: return KINDS_FRL[kind];
    case KMC_ASYNC_ISR: return KINDS_ASYNC[kind];
    case KMC_KIP320: return KINDS_KIP320[kind];
    case KMC_KIP320_FIRST_TRY: return KINDS_FIRST[kind];
    default:
        if (kind == 7)
            return model == KMC_TRUNCATE_TO_HW ? "BecomeFollowerTruncateToHighWatermark"
                   : model == KMC_KIP101       ? "BecomeFollowerTruncateKip101"
                                               : "BecomeFollowerTruncateKip279";
        return KINDS_BASE[kind];
    }
}

// mode: 0 the search's own code object (k_expand LOCAL + the small kernels), 1 k_expand SHARDED (the level-step interface),
// 2 k_expand ENUM (kmc_successors, trace replay); -1 all three.  A build script spreads the modes over its workers.
int kmc_precompile_mode(const kmc_config* cfg, const char* arch, int32_t mode) {
    if (!cfg) return fail(KMC_E_ARG, "null config");
    if (mode < -1 || mode > (int32_t)KMC_MODE_ENUM) return fail(KMC_E_ARG, "mode %d: expected -1 (all), 0 (search), 1 (sharded), 2 (enum)", mode);
    std::vector<char> code;
    std::string kname;
    const bool verify = getenv("KMC_VERIFY") && atoi(getenv("KMC_VERIFY"));
    for (unsigned m = 0; m <= KMC_MODE_ENUM; ++m) {
        if (mode >= 0 && (unsigned)mode != m) continue;
        int rc = get_code_object(*cfg, arch ? arch : "gfx950", &code, &kname, verify ? KMC_VERIFY_PRIMARY_OPTIONS : nullptr, nullptr, m);
        if (rc) return rc;
    }
    // with KMC_VERIFY set: also the second build kmc_open would load for the differential self-check
    if (verify && mode <= 0) return get_code_object(*cfg, arch ? arch : "gfx950", &code, &kname, KMC_VERIFY_OPTIONS);
    return KMC_OK;
}
int kmc_precompile(const kmc_config* cfg, const char* arch) { return kmc_precompile_mode(cfg, arch, -1); }

// Where the code object of cfg's kernels lives in the cache (compiled first if it is not there yet): the identity of the
// device code a measurement belongs to is the kernels' machine code, not the text of a header that also holds other builds.
int kmc_code_object_path(const kmc_config* cfg, const char* arch, char* out, uint64_t cap) {
    if (!cfg || !out || !cap) return fail(KMC_E_ARG, "null config / buffer");
    std::vector<char> code;
    std::string kname, path;
    int rc = get_code_object(*cfg, arch ? arch : "gfx950", &code, &kname, nullptr, &path);
    if (rc) return rc;
    if (path.size() + 1 > cap) return fail(KMC_E_ARG, "path of %zu bytes does not fit %llu", path.size(), (unsigned long long)cap);
    memcpy(out, path.c_str(), path.size() + 1);
    return KMC_OK;
}

int64_t kmc_compiler_identity(int32_t which) { return which == 0 ? compiler_id_mine() : which == 1 ? compiler_id_pinned() : -1; }

void kmc_close(kmc_handle* h) {   // (teardown: the HIP results are dropped on purpose — there is nobody to report them to)
    if (!h) return;
    if (h->cfg.device < 0 || !h->stream) {  // host-only handle, or open failed before any device work
        if (h->mod) (void)hipModuleUnload(h->mod);
        delete h;
        return;
    }
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    seen_set_free(h, h->table);
    if (!h->paired) seen_set_free(h, h->pred);   // (paired: the predecessors are the slots' second words)
    if (h->table2) (void)hipFree(h->table2);
    seen_set_free(h, h->sent);
    if (h->frontier[0]) (void)hipFree(h->frontier[0]);
    if (h->frontier[1]) (void)hipFree(h->frontier[1]);
    if (h->ctl) (void)hipFree(h->ctl);
    if (h->scratch) (void)hipFree(h->scratch);
    if (h->enum_out) (void)hipFree(h->enum_out);
    if (h->send && h->send_owned) (void)hipFree(h->send);
    comm_release(h);
    if (h->recv) (void)hipFree(h->recv);
    if (h->xstream) (void)hipStreamSynchronize(h->xstream);
    for (int i = 0; i < 2; ++i) {
        if (h->prow_dev[i]) (void)hipFree(h->prow_dev[i]);
        if (h->prow_host[i]) (void)hipHostFree(h->prow_host[i]);
        if (h->ev_row[i]) (void)hipEventDestroy(h->ev_row[i]);
        if (h->ev_xfer[i]) (void)hipEventDestroy(h->ev_xfer[i]);
    }
    if (h->xstream) (void)hipStreamDestroy(h->xstream);
    if (h->xrow_dev) (void)hipFree(h->xrow_dev);
    if (h->xrow_host) (void)hipHostFree(h->xrow_host);
    if (h->ctl_host) (void)hipHostFree(h->ctl_host);
    if (h->scratch_host) (void)hipHostFree(h->scratch_host);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    for (hipEvent_t e : h->ev_chain)
        if (e) (void)hipEventDestroy(e);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->mod_verify) (void)hipModuleUnload(h->mod_verify);
    if (h->mod_sh) (void)hipModuleUnload(h->mod_sh);
    if (h->mod_en) (void)hipModuleUnload(h->mod_en);
    if (h->mod) (void)hipModuleUnload(h->mod);
    delete h;
}

}  // extern "C"

static int open_impl(const kmc_config* cfg, kmc_handle* h) {
    h->cfg = *cfg;
    if (cfg->cache_dir) {
        h->cache_dir = cfg->cache_dir;
        h->cfg.cache_dir = h->cache_dir.c_str();
    }
    if (h->cfg.n_shards < 1) h->cfg.n_shards = 1;
    if (h->cfg.n_shards > KMC_MAX_SHARDS || h->cfg.shard_id < 0 || h->cfg.shard_id >= h->cfg.n_shards)
        return fail(KMC_E_ARG, "bad shard configuration %d/%d", h->cfg.shard_id, h->cfg.n_shards);
    std::string name, inst;
    h->layout_mode = layout_mode_from_env();   // read once: the handle's later code objects follow it (ensure_mode)
    if (!validate(h->cfg, &h->lay, &name, &inst, h->layout_mode)) {
        std::vector<char> dummy;
        return get_code_object(h->cfg, "gfx950", &dummy, &name);  // produces the KMC_E_ARG message
    }
    h->W = h->lay.W;
    h->nfact = h->cfg.symmetry ? (uint64_t)kmc_factorial(h->cfg.n_replicas) : 1;
    h->planes = h->W + (h->cfg.symmetry ? 1 : 0);
    if (h->cfg.symmetry && !kmc_model_symmetric(h->cfg.model))
        return fail(KMC_E_ARG, "symmetry (orbit counting): %s singles out a replica or has none", MODEL_NAMES[h->cfg.model]);
    h->rec_words = h->W + (cfg->keep_trace ? 1 : 0);
    if (cfg->device == -1) return KMC_OK;  // host-only handle: pack/unpack/fingerprint, no device work
    const double t_open0 = now_s();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(KMC_E_DEVICE, "no HIP device visible: this library has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(KMC_E_ARG, "device %d out of range (%d)", cfg->device, ndev);
    HIP_TRY(hipSetDevice(cfg->device));
    HIP_TRY(hipFree(nullptr));   // (the device's context is created here, not inside the first allocation: it is timed as what it is)
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
    const double t_init1 = now_s();
    h->timing.hip_init_s = t_init1 - t_open0;
    std::string arch = prop.gcnArchName;
    size_t colon = arch.find(':');
    if (colon != std::string::npos) arch = arch.substr(0, colon);
    h->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    h->arch = arch;

    std::vector<char> code;
    const bool verify = getenv("KMC_VERIFY") && atoi(getenv("KMC_VERIFY"));
    h->verify = verify;
    h->jit_defines = getenv("KMC_JIT_DEFINES") ? getenv("KMC_JIT_DEFINES") : "";
    int rc = get_code_object(h->cfg, arch, &code, &h->kname, verify ? KMC_VERIFY_PRIMARY_OPTIONS : nullptr, nullptr, KMC_MODE_LOCAL,
                             &h->jit_defines, h->layout_mode);
    if (rc) return rc;
    HIP_TRY(hipModuleLoadData(&h->mod, code.data()));
    HIP_TRY(hipModuleGetFunction(&h->f_expand, h->mod, ("kmc_expand_" + h->kname).c_str()));
    if (hipModuleGetFunction(&h->f_expand_dry, h->mod, ("kmc_expand_dry_" + h->kname).c_str()) != hipSuccess) {
        h->f_expand_dry = nullptr;   // (not a tuning build)
        (void)hipGetLastError();
    }
    HIP_TRY(hipModuleGetFunction(&h->f_inv, h->mod, ("kmc_inv_" + h->kname).c_str()));
    HIP_TRY(hipModuleGetFunction(&h->f_insert, h->mod, ("kmc_insert_" + h->kname).c_str()));
    HIP_TRY(hipModuleGetFunction(&h->f_init, h->mod, ("kmc_init_" + h->kname).c_str()));
    HIP_TRY(hipModuleGetFunction(&h->f_find, h->mod, ("kmc_find_" + h->kname).c_str()));
    HIP_TRY(hipModuleGetFunction(&h->f_packrow, h->mod, ("kmc_packrow_" + h->kname).c_str()));
    if (verify) {
        // Differential self-check for constants no oracle can reach (round 1 met a k_expand build that LOST successors
        // under heavy register spilling): a second code object of the same source, compiled at -O1 with a quarter of
        // the occupancy target and with the guards lowered the other way (KMC_VERIFY_OPTIONS), re-generates every level's
        // successors (DRY mode: no table, no frontier) and the per-action counts, deadlock counts, violation counts and
        // the checksum of the successors' fingerprints of the two builds must agree.
        std::vector<char> vcode;
        std::string vname;
        rc = get_code_object(h->cfg, arch, &vcode, &vname, KMC_VERIFY_OPTIONS, nullptr, KMC_MODE_LOCAL, &h->jit_defines, h->layout_mode);
        if (rc) return rc;
        HIP_TRY(hipModuleLoadData(&h->mod_verify, vcode.data()));
        HIP_TRY(hipModuleGetFunction(&h->f_expand_verify, h->mod_verify, ("kmc_expand_dry_" + vname).c_str()));
    }
    const double t_code1 = now_s();
    h->timing.code_object_s = t_code1 - t_init1;
    int occ = 0;
    if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&occ, h->f_expand, KMC_BLOCK,
                                                           kmc_expand_lds_bytes(h->W, cfg->keep_trace != 0, cfg->symmetry != 0)) == hipSuccess && occ > 0)
        h->blocks_per_cu = occ > 8 ? 8 : occ;
    {   // the occupancy query may admit a block more than really fits when LDS is the limit
        // (5 x 32 KiB = all 160 KiB was reported resident, ran as 4 + a queued 5th: 69 ms vs 55 ms)
        const unsigned lds = kmc_expand_lds_bytes(h->W, cfg->keep_trace != 0, cfg->symmetry != 0);
        const int by_lds = (int)((160u * 1024u - 1024u) / (lds ? lds : 1u));
        if (by_lds >= 1 && h->blocks_per_cu > by_lds) h->blocks_per_cu = by_lds;
    }
    HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&h->ev0));
    HIP_TRY(hipEventCreate(&h->ev1));

    // ---- sizing: table slots, frontier states, send records -------------------------------
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    const double budget = 0.85 * (double)free_b;
    h->slot_words = cfg->wide_fingerprint ? 2 : 1;
    // A run that keeps traces on 64-bit entries stores a claim's predecessor in the claim's own slot (16-byte slots: fingerprint,
    // predecessor) instead of a table of its own: the claim then dirties ONE random line, not two.  The second line cost a quarter
    // of the headline's kernel wherever the predecessor table lay (k_expand 28.4 ms without traces, 34.6 - 36.4 with); in the slot:
    // headline 34.9 -> 30.6 ms, BASELINE config 4 20.2 -> 18.3, same box, fresh processes, interleaved, counts exact (profiles/
    // r06_chunked_seen_set.txt, items 7 - 8).  States of KMC_DEFER_MIN_WORDS words and more keep the separate table: their kernel
    // issues a batch's first probe one flush early (kmc_kernels.h: DEFER), the paired slots take the undeferred walk, and at seven
    // brokers that costs more than the second line (config 5, ten levels: 26.9 ms separate, 27.6 paired).  KMC_PAIRED_SLOTS=0
    // restores the separate table everywhere (A/B), =2 forces the slots on wide states too; 128-bit entries keep it (their slot is full).
    static const int paired_env = getenv("KMC_PAIRED_SLOTS") ? atoi(getenv("KMC_PAIRED_SLOTS")) : 1;
    const bool paired_ok = paired_env == 2 || (paired_env == 1 && h->W < KMC_DEFER_MIN_WORDS);
    h->paired = cfg->keep_trace && !cfg->wide_fingerprint && paired_ok;
    const uint64_t slot_bytes = 8 * h->slot_words + (cfg->keep_trace ? 8 : 0);
    // auto-sizing: half of the budget for the table; a shard also keeps a sender-side filter of twice the table
    // (0.15 + 0.30), two frontiers (2 x 0.09), a send area and a receive area (0.12 each)
    const double table_share = h->cfg.n_shards > 1 ? 0.15 : 0.5;
    // An explicit capacity is taken as given (rounded up to a whole 64-slot group: kmc_slot_of): a table may be sized to the
    // memory there is, not to the power of two below it.  The automatic size stays a power of two.
    uint64_t tcap = cfg->table_capacity ? (cfg->table_capacity + 63) / 64 * 64
                                        : pow2_floor((uint64_t)(budget * table_share) / slot_bytes);
    if (tcap < 1024) tcap = 1024;
    if (tcap >> 6 > 0xFFFFFFFFull) return fail(KMC_E_ARG, "table_capacity %llu: at most 2^38 slots", (unsigned long long)tcap);
    uint64_t fcap = cfg->frontier_capacity;
    if (!fcap) {
        const double share = h->cfg.n_shards > 1 ? 0.09 : 0.20;
        fcap = (uint64_t)(budget * share) / (8ull * h->planes);
        if (fcap > tcap) fcap = tcap;
    }
    if (fcap < 64) fcap = 64;
    fcap = (fcap + 64 * KMC_SEGS - 1) / (64 * KMC_SEGS) * (64 * KMC_SEGS);  // segments start 512-byte aligned
    h->table_cap = tcap;
    h->fcap = fcap;
    h->seg_cap = fcap / KMC_SEGS;
    if (!(h->table = seen_set_alloc(h, tcap * h->stride_words() * 8))) return fail(KMC_E_NOMEM, "cannot allocate %llu table slots", (unsigned long long)tcap);
    if (getenv("KMC_VERBOSE"))
        fprintf(stderr, "[kmc] seen-set: %llu slots x %llu B at %p (%s)\n", (unsigned long long)tcap, (unsigned long long)(h->stride_words() * 8), (void*)h->table,
                h->mapped.empty() ? "one hipMalloc" : "mapped from chunks");
    if (h->paired) h->pred = h->table + 1;
    else if (cfg->keep_trace && !(h->pred = seen_set_alloc(h, tcap * 8))) return fail(KMC_E_NOMEM, "cannot allocate predecessor table");
    for (int i = 0; i < 2; ++i)
        if (hipMalloc(&h->frontier[i], fcap * 8ull * h->planes) != hipSuccess)
            return fail(KMC_E_NOMEM, "cannot allocate frontier of %llu states", (unsigned long long)fcap);
    HIP_TRY(hipMalloc(&h->ctl, KMC_CTL_SLOTS * sizeof(KmcLevelCtl)));
    HIP_TRY(hipHostMalloc(&h->ctl_host, KMC_CHAIN * sizeof(KmcLevelCtl)));
    HIP_TRY(hipMalloc(&h->scratch, 64 * 8));
    HIP_TRY(hipHostMalloc(&h->scratch_host, 64 * 8));
    HIP_TRY(hipMalloc(&h->enum_out, h->enum_cap * (h->W + 2) * 8ull));
    // Sender-side duplicate filter: a shard generates (and would ship) a remote state several times.  What it saves
    // shrinks with P (each copy of a state is generated on a different shard: 49 % of the remote successors dropped at
    // P = 2, 31 % at 4, 18 % at 8 on the headline) while every remote successor pays one more random probe for it, so it
    // is on where the wire is the bottleneck (P <= 4: one to three xGMI links per GPU carry everything) and off beyond
    // (profiles/r02_loopback_filter.jsonl: k_expand per shard 11.6 -> 8.1 ms at P = 8).  KMC_SEND_FILTER=1 / 0 forces it.
    bool want_filter = h->cfg.n_shards > 1 && h->cfg.n_shards <= 4;
    if (const char* e = getenv("KMC_SEND_FILTER")) want_filter = h->cfg.n_shards > 1 && atoi(e) != 0;
    if (getenv("KMC_NO_SEND_FILTER") && atoi(getenv("KMC_NO_SEND_FILTER"))) want_filter = false;
    // 128-bit entries: the sender-side filter remembers 64-bit fingerprints only — a second distinct remote state with the
    // same fingerprint would be dropped at the sender and never meet the owner's check-word comparison, and the conservation
    // law (probed is counted before the filter) could not see it.  No filter then, whatever the environment asks for.
    if (h->cfg.wide_fingerprint) want_filter = false;
    if (want_filter) {
        // it may meet up to ~2x as many distinct remote fingerprints as it owns
        h->sent_cap = pow2_floor(tcap * 2);   // (the filter's own index is a mask: first_time)
        if (!(h->sent = seen_set_alloc(h, h->sent_cap * 8))) h->sent_cap = 0;  // optional (random probes and claims, like the seen-set's)
    }
    if (h->cfg.n_shards > 1) {
        uint64_t scap = cfg->send_capacity;
        if (!scap) scap = (uint64_t)(budget * 0.12) / (8ull * h->rec_words * h->cfg.n_shards * KMC_SEGS);
        if (scap < 64) scap = 64;
        h->send_cap = scap;  // records per (destination, sub-buffer)
        if (hipMalloc(&h->send, scap * h->rec_words * 8ull * h->cfg.n_shards * KMC_SEGS) != hipSuccess)
            return fail(KMC_E_NOMEM, "cannot allocate send buffers");
    }
    {
        size_t free_after = 0, total_after = 0;
        if (hipMemGetInfo(&free_after, &total_after) == hipSuccess && free_b > free_after) h->timing.device_bytes = free_b - free_after;
    }
    h->timing.alloc_s = now_s() - t_code1;
    h->timing.open_s = now_s() - t_open0;
    return KMC_OK;
}

extern "C" {

int kmc_open(const kmc_config* cfg, kmc_handle** out) {
    if (!cfg || !out) return fail(KMC_E_ARG, "null argument");
    *out = nullptr;
    kmc_handle* h = new kmc_handle();
    int rc = open_impl(cfg, h);
    if (rc) {
        std::string keep = g_err;
        kmc_close(h);
        g_err = keep;
        return rc;
    }
    *out = h;
    return KMC_OK;
}

uint64_t kmc_state_words(kmc_handle* h) { return h ? h->W : 0; }

uint64_t kmc_fingerprint_of(kmc_handle* h, const uint64_t* words) {
    unsigned long long w[KMC_MAXW];
    for (int k = 0; k < h->W; ++k) w[k] = words[k];
    switch (h->W) {
    case 1: return kmc_fingerprint<1>(w, h->cfg.hash_seed);
    case 2: return kmc_fingerprint<2>(w, h->cfg.hash_seed);
    case 3: return kmc_fingerprint<3>(w, h->cfg.hash_seed);
    case 4: return kmc_fingerprint<4>(w, h->cfg.hash_seed);
    case 5: return kmc_fingerprint<5>(w, h->cfg.hash_seed);
    case 6: return kmc_fingerprint<6>(w, h->cfg.hash_seed);
    case 7: return kmc_fingerprint<7>(w, h->cfg.hash_seed);
    case 8: return kmc_fingerprint<8>(w, h->cfg.hash_seed);
    case 9: return kmc_fingerprint<9>(w, h->cfg.hash_seed);
    case 10: return kmc_fingerprint<10>(w, h->cfg.hash_seed);
    case 11: return kmc_fingerprint<11>(w, h->cfg.hash_seed);
    default: return kmc_fingerprint<12>(w, h->cfg.hash_seed);
    }
}

int kmc_canonical_state(kmc_handle* h, const uint64_t* words, uint64_t* representative, int32_t* stabiliser) {
    if (!h || !words || !representative) return fail(KMC_E_ARG, "null argument");
    if (!kmc_model_symmetric(h->lay.model)) return fail(KMC_E_ARG, "%s has no replica symmetry", MODEL_NAMES[h->lay.model]);
    unsigned long long w[KMC_MAXW] = {0}, c[KMC_MAXW] = {0};
    for (int k = 0; k < h->W; ++k) w[k] = words[k];
    int stab = 1;
    kmc_canonical_state_generic(h->lay, w, c, &stab);
    for (int k = 0; k < h->W; ++k) representative[k] = c[k];
    if (stabiliser) *stabiliser = stab;
    return KMC_OK;
}

uint64_t kmc_canon_bytes(kmc_handle* h) {
    const KmcLayout& y = h->lay;
    if (y.model == KMC_IDSEQUENCE) return 8;
    if (y.model == KMC_FINITE_REPLICATED_LOG) return (uint64_t)y.N * (1 + y.L);
    if (y.model == KMC_ASYNC_ISR) return 6 + y.N + (uint64_t)(y.E + 1) * (((1 << y.N) + 7) / 8) + (y.E + 1);
    return (uint64_t)y.N * (5 + y.L) + 5 + 2 * (y.E + 1);
}

int kmc_unpack_state(kmc_handle* h, const uint64_t* words, uint8_t* c) {
    const KmcLayout& y = h->lay;
    unsigned long long w[KMC_MAXW + 1] = {0};
    for (int k = 0; k < h->W; ++k) w[k] = words[k];
    if (y.model == KMC_IDSEQUENCE) {
        memcpy(c, &w[0], 8);
        return KMC_OK;
    }
    if (y.model == KMC_FINITE_REPLICATED_LOG) {
        for (int r = 0; r < y.N; ++r) {
            uint8_t* b = c + r * (1 + y.L);
            b[0] = (uint8_t)kmc_getbits(w, y.end_off[r], y.BO);
            for (int o = 0; o < y.L; ++o) b[1 + o] = (uint8_t)kmc_getbits(w, y.log_off[r] + o * y.BR, y.BR);
        }
        return KMC_OK;
    }
    if (y.model == KMC_ASYNC_ISR) {
        const int ns = 1 << y.N, rb = (ns + 7) / 8;
        c[0] = (uint8_t)kmc_getbits(w, y.a_cisr, y.N);
        c[1] = (uint8_t)kmc_getbits(w, y.a_cver, y.BV);
        c[2] = (uint8_t)kmc_getbits(w, y.a_lisr, y.N);
        c[3] = (uint8_t)kmc_getbits(w, y.a_lver, y.BV);
        c[4] = (uint8_t)kmc_getbits(w, y.a_pisr, y.N);
        c[5] = (uint8_t)kmc_getbits(w, y.a_pver, y.BV);
        for (int r = 0; r < y.N; ++r) c[6 + r] = (uint8_t)kmc_getbits(w, y.a_off[r], y.BF);
        uint8_t* q = c + 6 + y.N;
        memset(q, 0, (size_t)(y.E + 1) * rb);
        for (int v = 0; v <= y.E; ++v)
            for (int m = 0; m < ns; ++m)
                if (kmc_getbits(w, y.a_req + v * ns + m, 1)) q[v * rb + (m >> 3)] |= (uint8_t)(1u << (m & 7));
        uint8_t* u = q + (y.E + 1) * rb;
        for (int v = 0; v <= y.E; ++v) u[v] = (uint8_t)kmc_getbits(w, y.a_upd + v * y.N, y.N);
        return KMC_OK;
    }
    const int rs = 5 + y.L;
    for (int r = 0; r < y.N; ++r) {
        uint8_t* b = c + r * rs;
        b[0] = (uint8_t)kmc_getbits(w, y.end_off[r], y.BO);
        b[1] = (uint8_t)kmc_getbits(w, y.hw_off[r], y.BO);
        b[2] = (uint8_t)kmc_getbits(w, y.ep_off[r], y.BE);
        b[3] = (uint8_t)kmc_getbits(w, y.ldr_off[r], y.BL);
        b[4] = (uint8_t)kmc_getbits(w, y.isr_off[r], y.BI);
        for (int o = 0; o < y.L; ++o) {
            const unsigned rec = (unsigned)kmc_getbits(w, y.log_off[r] + o * y.BR, y.BR);
            // packed (id+1)<<BEr | epoch  ->  canonical 1 + id*(E+1) + epoch
            b[5 + o] = rec == 0 ? 0 : (uint8_t)(1 + ((rec >> y.BEr) - 1) * (y.E + 1) + (rec & ((1u << y.BEr) - 1)));
        }
    }
    uint8_t* g = c + y.N * rs;
    g[0] = (uint8_t)kmc_getbits(w, y.nextrec_off, y.BNR);
    g[1] = (uint8_t)kmc_getbits(w, y.nextep_off, y.BE);
    g[2] = (uint8_t)kmc_getbits(w, y.qep_off, y.BE);
    g[3] = (uint8_t)kmc_getbits(w, y.qldr_off, y.BL);
    g[4] = (uint8_t)kmc_getbits(w, y.qisr_off, y.BI);
    for (int e = 0; e <= y.E; ++e) {
        g[5 + 2 * e] = (uint8_t)kmc_getbits(w, y.reqldr_off[e], y.BL);
        g[6 + 2 * e] = (uint8_t)kmc_getbits(w, y.reqisr_off[e], y.BI);
    }
    return KMC_OK;
}

int kmc_pack_state(kmc_handle* h, const uint8_t* c, uint64_t* words) {
    const KmcLayout& y = h->lay;
    unsigned long long w[KMC_MAXW + 1] = {0};
    if (y.model == KMC_IDSEQUENCE) {
        memcpy(&w[0], c, 8);
    } else if (y.model == KMC_FINITE_REPLICATED_LOG) {
        for (int r = 0; r < y.N; ++r) {
            const uint8_t* b = c + r * (1 + y.L);
            kmc_setbits(w, y.end_off[r], y.BO, b[0]);
            for (int o = 0; o < y.L; ++o) kmc_setbits(w, y.log_off[r] + o * y.BR, y.BR, b[1 + o]);
        }
    } else if (y.model == KMC_ASYNC_ISR) {
        const int ns = 1 << y.N, rb = (ns + 7) / 8;
        kmc_setbits(w, y.a_cisr, y.N, c[0]);
        kmc_setbits(w, y.a_cver, y.BV, c[1]);
        kmc_setbits(w, y.a_lisr, y.N, c[2]);
        kmc_setbits(w, y.a_lver, y.BV, c[3]);
        kmc_setbits(w, y.a_pisr, y.N, c[4]);
        kmc_setbits(w, y.a_pver, y.BV, c[5]);
        for (int r = 0; r < y.N; ++r) kmc_setbits(w, y.a_off[r], y.BF, c[6 + r]);
        const uint8_t* q = c + 6 + y.N;
        for (int v = 0; v <= y.E; ++v)
            for (int m = 0; m < ns; ++m)
                if (q[v * rb + (m >> 3)] >> (m & 7) & 1) kmc_setbits(w, y.a_req + v * ns + m, 1, 1);
        const uint8_t* u = q + (y.E + 1) * rb;
        for (int v = 0; v <= y.E; ++v) kmc_setbits(w, y.a_upd + v * y.N, y.N, u[v]);
    } else {
        const int rs = 5 + y.L;
        for (int r = 0; r < y.N; ++r) {
            const uint8_t* b = c + r * rs;
            kmc_setbits(w, y.end_off[r], y.BO, b[0]);
            kmc_setbits(w, y.hw_off[r], y.BO, b[1]);
            kmc_setbits(w, y.ep_off[r], y.BE, b[2]);
            kmc_setbits(w, y.ldr_off[r], y.BL, b[3]);
            kmc_setbits(w, y.isr_off[r], y.BI, b[4]);
            for (int o = 0; o < y.L; ++o) {
                const unsigned code = b[5 + o];
                const unsigned rec = code == 0 ? 0 : ((((code - 1) / (y.E + 1)) + 1) << y.BEr) | ((code - 1) % (y.E + 1));
                kmc_setbits(w, y.log_off[r] + o * y.BR, y.BR, rec);
            }
        }
        const uint8_t* g = c + y.N * rs;
        kmc_setbits(w, y.nextrec_off, y.BNR, g[0]);
        kmc_setbits(w, y.nextep_off, y.BE, g[1]);
        kmc_setbits(w, y.qep_off, y.BE, g[2]);
        kmc_setbits(w, y.qldr_off, y.BL, g[3]);
        kmc_setbits(w, y.qisr_off, y.BI, g[4]);
        for (int e = 0; e <= y.E; ++e) {
            kmc_setbits(w, y.reqldr_off[e], y.BL, g[5 + 2 * e]);
            kmc_setbits(w, y.reqisr_off[e], y.BI, g[6 + 2 * e]);
        }
    }
    for (int k = 0; k < h->W; ++k) words[k] = w[k];
    return KMC_OK;
}


}  // extern "C"
