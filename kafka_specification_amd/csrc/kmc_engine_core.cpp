// kmc_engine_core.cpp — the kernel launches, a level's counters folded into the result (absorb), conservation, Init.
#include "kmc_engine_internal.h"


namespace kmc_engine {

// the small kernels (k_insert, k_init, k_find): the whole argument block
int launch(kmc_handle* h, hipFunction_t f, const KmcArgs& a, unsigned grid, hipStream_t stream) {
    KmcArgs args = a;
    size_t size = sizeof(args);
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    HIP_TRY(hipModuleLaunchKernel(f, grid, 1, 1, KMC_BLOCK, 1, 1, 0, stream ? stream : h->stream, nullptr, config));
    return KMC_OK;
}

// The code object of k_expand in SHARDED / ENUM mode joins the handle when that mode is first asked for (from the cache; a
// cold cache compiles it: kmc_precompile builds all three ahead of time).
int ensure_mode(kmc_handle* h, unsigned mode) {
    if (mode == KMC_MODE_LOCAL || mode == KMC_MODE_DRY) return KMC_OK;
    hipModule_t& mod = mode == KMC_MODE_SHARDED ? h->mod_sh : h->mod_en;
    hipFunction_t& f = mode == KMC_MODE_SHARDED ? h->f_expand_sh : h->f_expand_en;
    if (f) return KMC_OK;
    std::vector<char> code;
    std::string kname;
    int rc = get_code_object(h->cfg, h->arch, &code, &kname, h->verify ? KMC_VERIFY_PRIMARY_OPTIONS : nullptr, nullptr, mode, &h->jit_defines,
                             h->layout_mode);
    if (rc) return rc;
    HIP_TRY(hipModuleLoadData(&mod, code.data()));
    HIP_TRY(hipModuleGetFunction(&f, mod, (std::string("kmc_expand") + MODE_SUFFIX[mode] + "_" + h->kname).c_str()));
    return KMC_OK;
}

// k_expand in one of its modes (each mode is its own kernel; `verify` = the dry kernel of KMC_VERIFY's second build).  The
// search's kernel receives KmcArgsLocal — the head of the block — and nothing else.
int launch_expand(kmc_handle* h, unsigned mode, const KmcArgs& a, unsigned grid, hipStream_t stream, bool verify) {
    int rc = ensure_mode(h, mode);
    if (rc) return rc;
    hipFunction_t f = verify ? h->f_expand_verify : mode == KMC_MODE_LOCAL ? h->f_expand : mode == KMC_MODE_SHARDED ? h->f_expand_sh
                    : mode == KMC_MODE_ENUM ? h->f_expand_en : h->f_expand_dry;
    if (!f)
        return fail(KMC_E_STATE, "k_expand's dry mode is only compiled into a tuning build (KMC_JIT_DEFINES=-DKMC_TUNING=1)");
    KmcArgs args = a;
    const bool meta = (args.flags & KMC_FLAG_TRACE) || mode == KMC_MODE_ENUM;   // k_expand carves its rings out of dynamic LDS
    if (meta) args.flags |= KMC_FLAG_META;
    const unsigned lds = kmc_expand_lds_bytes(h->W, meta, h->cfg.symmetry != 0);
    size_t size = mode == KMC_MODE_LOCAL && !verify ? sizeof(KmcArgsLocal) : sizeof(KmcArgs);
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    HIP_TRY(hipModuleLaunchKernel(f, grid, 1, 1, KMC_BLOCK, 1, 1, lds, stream ? stream : h->stream, nullptr, config));
    return KMC_OK;
}

// the invariants of the n states of a frontier that is not expanded (k_inv)
int launch_inv(kmc_handle* h, const KmcArgs& a, uint64_t n) {
    KmcArgsLocal args = a;
    uint64_t blocks = (n + KMC_BLOCK - 1) / KMC_BLOCK;
    const uint64_t maxb = (uint64_t)h->n_cus * 8;
    if (blocks > maxb) blocks = maxb;
    if (blocks < 1) blocks = 1;
    size_t size = sizeof(args);
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    HIP_TRY(hipModuleLaunchKernel(h->f_inv, (unsigned)blocks, 1, 1, KMC_BLOCK, 1, 1, 0, h->stream, nullptr, config));
    return KMC_OK;
}

// The seen-set is NOT one hipMalloc.  Where a handle's table lies decides how fast its random probes are served: the headline's
// k_expand ran at one of three discrete levels — 28.4 / 30.5 / 31.5 ms with a 12 GiB table, 30.6 / 31.6 with 8 GiB — that followed
// the table's allocation and nothing else (one handle, only the table moved: all three levels; profiles/r06_table_size_and_
// placement.txt), with a hipMalloc of several GiB being a few huge physically contiguous blocks.  Mapped from 8 MiB chunks (HIP's
// virtual-memory API: one range of addresses, every chunk its own physical allocation) the same table ran at the fast level in
// every handle of every process of that box (and in most processes since - item 9 of the profile file has the exceptions: a level
// per process and box remains, never worse than hipMalloc's slow one): headline 31.3 - 31.6 -> 28.5 ms, BASELINE config 4 18.6 -> 16.8, config 4 at SURVEY's sizing
// 43.1 -> 38.7, config 5 25.6 -> 25.1 (it is bound by instructions), same box, interleaved, counts exact (profiles/r06_chunked_
// seen_set.txt).  2 / 4 / 8 MiB chunks are alike, 32 MiB less steady, 1 - 2 GiB chunks behave like hipMalloc: what matters is the
// size of the physically contiguous pieces, and what they change is the random WRITES (randbench in chunked memory: stores +30 %,
// claims +16 %, loads nothing - profiles/r06_randbench.txt); why, user space cannot see.  The frontiers gain nothing from it (streams) and stay hipMalloc's.  KMC_SEEN_SET_CHUNK_LOG2 overrides the
// chunk (0: one hipMalloc); any failure of the mapping falls back to hipMalloc.
// SCATTERED chunks (the round's last finding, calls 34 - 38): chunks created one after the other are, on an unfragmented allocator,
// physically consecutive, and a table of consecutive chunks is as fast as the place it happens to lie in - 28.5 to 34.4 G/s for the
// seen-set's load + CAS mix, 22 to 30 G/s for random stores, 5.2 to 6.5 TB/s for a streaming fill, from one 8 GiB stretch of one
// 128 GiB pool to the next (tools/membench/diversity) - which is the level a handle keeps for its life (call 34: one handle flat at
// 30.4 ms for a minute while fresh processes between its searches went 28.7 -> 31.2 -> 28.7).  The same 1,024 chunks picked from ALL
// over the pool run at 34.9 - 35.2 / 31.8 - 32.6 / 7.0 - 7.1 in every pool of every process: better than the best consecutive
// stretch.  Creating sixteen times the chunks to keep one in sixteen costs seconds, though (16,384 hipMemCreate: 3.7 s, and a
// fragmented allocator for whoever comes next: call 37), so the spread is bought with SPACERS: the table's chunks are created in
// KMC_SCATTER_CLUSTERS clusters, and after each cluster ONE physical allocation of (spread - 1) x the cluster's size is created -
// never mapped - that makes the allocator move on; the spacers go back as soon as the last chunk is mapped.  With KMC_SEEN_SET_SPREAD
// = 16 / 32 the headline's k_expand is 28.4 - 29.2 in every process where the chunks as they come give 28.4 - 30.6 (call 38, same
// box, interleaved; BASELINE config 4 16.8 - 17.0 against 16.9 - 18.6; = 4 is not enough).  It is NOT the default: owning 120 GiB
// for a moment costs 0.3 - 3 s at open, and the driver wipes what is handed back - 1 - 3 s that land in this handle's close or
// in the next process's allocations.  A search of seconds and more earns that back (KMC_SEEN_SET_SPREAD=16; bench.py's timed legs
// set it and say so); the front end answering a 30 ms search does not.
#ifndef KMC_SCATTER_CLUSTERS
#define KMC_SCATTER_CLUSTERS 64
#endif
static int spread_factor(size_t need) {
    static const int env = getenv("KMC_SEEN_SET_SPREAD") ? atoi(getenv("KMC_SEEN_SET_SPREAD")) : KMC_SEEN_SET_SPREAD_DEFAULT;
    int f = env < 1 ? 1 : env > 64 ? 64 : env;
    size_t free_b = 0, total_b = 0;
    if (f > 1 && hipMemGetInfo(&free_b, &total_b) == hipSuccess)
        while (f > 1 && (double)need * f > 0.8 * (double)free_b) --f;   // (the spacers live only until the table is mapped)
    return f;
}
u64* seen_set_alloc(kmc_handle* h, size_t bytes, bool chunks) {
    static const int lg_env = getenv("KMC_SEEN_SET_CHUNK_LOG2") ? atoi(getenv("KMC_SEEN_SET_CHUNK_LOG2")) : 23;
    const int lg = chunks ? lg_env : 0;
    void* va = nullptr;
    size_t total = 0, done = 0;
    if (lg > 0) {
        const double t0 = now_s();
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = h->cfg.device;
        size_t gran = 0;
        const char* step = "hipMemGetAllocationGranularity";   // (the call that failed, for KMC_VERBOSE)
        hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
        if (e == hipSuccess && gran == 0) e = hipErrorInvalidValue;
        size_t chunk = (size_t)1 << (lg < 40 ? lg : 40);
        if (e == hipSuccess && chunk < gran) chunk = gran;
        std::vector<hipMemGenericAllocationHandle_t> spacers;
        size_t n = 0, per_cluster = 0, spacer_bytes = 0;
        if (e == hipSuccess) {
            total = (bytes + chunk - 1) / chunk * chunk;
            n = total / chunk;
            step = "hipMemAddressReserve";
            e = hipMemAddressReserve(&va, total, chunk, nullptr, 0);
            if (e != hipSuccess) va = nullptr;
            const int spread = spread_factor(total);
            per_cluster = (n + KMC_SCATTER_CLUSTERS - 1) / KMC_SCATTER_CLUSTERS;
            spacer_bytes = spread > 1 && n >= 2 ? (size_t)(spread - 1) * per_cluster * chunk : 0;
        }
        for (size_t j = 0; e == hipSuccess && j < n; ++j) {
            hipMemGenericAllocationHandle_t piece;
            step = "hipMemCreate";
            if ((e = hipMemCreate(&piece, chunk, &prop, 0)) != hipSuccess) break;
            step = "hipMemMap";
            e = hipMemMap((char*)va + done, chunk, 0, piece, 0);
            (void)hipMemRelease(piece);   // (the mapping keeps the chunk alive; it goes with hipMemUnmap)
            if (e != hipSuccess) break;
            done += chunk;
            if (spacer_bytes && (j + 1) % per_cluster == 0 && j + 1 < n) {   // a cluster is complete: make the allocator move on
                hipMemGenericAllocationHandle_t sp;
                if (hipMemCreate(&sp, spacer_bytes, &prop, 0) == hipSuccess) spacers.push_back(sp);
                else { (void)hipGetLastError(); spacer_bytes = 0; }   // (no room: the rest of the table lies where it lies)
            }
        }
        for (hipMemGenericAllocationHandle_t sp : spacers) (void)hipMemRelease(sp);
        if (e == hipSuccess) {
            hipMemAccessDesc d{};
            d.location = prop.location;
            d.flags = hipMemAccessFlagsProtReadWrite;
            step = "hipMemSetAccess";
            e = hipMemSetAccess(va, total, &d, 1);
        }
        if (e == hipSuccess) {
            h->mapped.emplace_back(va, total);
            if (getenv("KMC_VERBOSE"))
                fprintf(stderr, "[kmc] seen-set memory: %zu chunks of %zu MiB in clusters of %zu, %zu spacers of %.2f GiB between them, %.3f s\n", n,
                        chunk >> 20, per_cluster, spacers.size(), (double)spacer_bytes / (double)(1ull << 30), now_s() - t0);
            return (u64*)va;
        }
        // undo what was mapped and fall back
        if (getenv("KMC_VERBOSE"))
            fprintf(stderr, "[kmc] seen-set memory: %s failed after %zu of %zu bytes in chunks of %zu (%s): falling back to one hipMalloc\n",
                    step, done, total, chunk, hipGetErrorString(e));
        if (va) {
            if (done) (void)hipMemUnmap(va, done);
            (void)hipMemAddressFree(va, total);
        }
        (void)hipGetLastError();
    }
    u64* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    return p;
}
void seen_set_free(kmc_handle* h, u64* p) {
    if (!p) return;
    for (size_t k = 0; k < h->mapped.size(); ++k)
        if (h->mapped[k].first == (void*)p) {
            const double t0 = now_s();
            (void)hipMemUnmap(p, h->mapped[k].second);
            const double t1 = now_s();
            (void)hipMemAddressFree(p, h->mapped[k].second);
            if (getenv("KMC_VERBOSE"))
                fprintf(stderr, "[kmc] released %.1f GiB of chunks: unmap %.3f s, address range %.3f s\n",
                        (double)h->mapped[k].second / (double)(1ull << 30), t1 - t0, now_s() - t1);
            h->mapped.erase(h->mapped.begin() + (long)k);
            return;
        }
    (void)hipFree(p);
}

KmcArgs base_args(kmc_handle* h, int ctl_slot) {
    KmcArgs a{};
    a.table = h->table;
    a.table_cap = h->table_cap;
    a.pred = h->pred;
    a.sent = h->sent;
    a.sent_mask = h->sent_cap ? h->sent_cap - 1 : 0;
    a.ctl = h->ctl + ctl_slot;
    a.seed = h->cfg.hash_seed;
    a.inv_mask = h->cfg.invariant_mask;
    a.flags = (h->cfg.keep_trace ? KMC_FLAG_TRACE : 0u) | (h->slot_words == 2 ? KMC_FLAG_FP128 : 0u) | (h->paired ? KMC_FLAG_PAIRED : 0u);
    a.nshards = (uint32_t)h->cfg.n_shards;
    a.shard = (uint32_t)h->cfg.shard_id;
    a.rec_words = (uint32_t)h->rec_words;
    a.fin_stride = a.fout_stride = h->fcap;
    a.seg_cap = h->seg_cap;
    for (int sg = 0; sg < KMC_SEGS; ++sg) a.seg_count[sg] = h->seg_n[sg];
    return a;
}

// Upper bound on the successors of one state = the number of action instances of the lowered Next (device header:
// KmcKafka::NINST etc.).  Sizes the grids of chained launches, whose input sizes only the device knows.
uint64_t max_fanout(const kmc_handle* h) {
    const uint64_t N = (uint64_t)h->cfg.n_replicas, L = (uint64_t)h->cfg.log_size, E1 = (uint64_t)h->cfg.max_leader_epoch + 1;
    switch (h->cfg.model) {
    case KMC_IDSEQUENCE: return 1;
    case KMC_FINITE_REPLICATED_LOG: return N * (uint64_t)h->cfg.n_log_records + N * L + N * (N - 1);
    case KMC_ASYNC_ISR: return (N - 1) + (1ull << N) + (N - 1) + N + 1 + (uint64_t)h->cfg.max_leader_epoch + (N - 1);
    default: {
        const uint64_t NP = N * (N - 1);
        return N + N + E1 * N + N * N + NP + N + N + NP * E1 + NP + (h->cfg.model == KMC_KIP320_FIRST_TRY ? NP : 0);
    }
    }
}

unsigned expand_grid(kmc_handle* h, uint64_t n) {
    // (orbit counting: k_expand shrinks its tiles down to 4 states when a level is small — KMC_SYMM, kmc_device.h — so the
    // grid is sized for that)
    const uint64_t per_tile = h->cfg.symmetry ? 4 : 64;
    const uint64_t tiles = (n + per_tile - 1) / per_tile;
    uint64_t blocks = (tiles + KMC_WAVES - 1) / KMC_WAVES;
    // one resident wave of blocks: more than the kernel's occupancy only queues blocks and
    // unbalances the tail (measured: 73 ms at 5 blocks/CU vs 59 ms at the resident 4)
    static const int forced = getenv("KMC_BLOCKS_PER_CU") ? atoi(getenv("KMC_BLOCKS_PER_CU")) : 0;
    const int per_cu = forced > 0 ? forced : h->blocks_per_cu;
    const uint64_t maxb = (uint64_t)h->n_cus * per_cu;
    if (blocks > maxb) blocks = maxb;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

int read_ctl(kmc_handle* h, int slot) {
    // a single-GPU level reports through the head of the block; the per-destination send counters behind it are
    // only written (and read back) in SHARDED mode
    const size_t bytes = h->cfg.n_shards > 1 || h->stepping ? sizeof(KmcLevelCtl) : KMC_CTL_LOCAL_BYTES;
    HIP_TRY(hipMemcpyAsync(h->ctl_host, h->ctl + slot, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return KMC_OK;
}

int zero_ctl(kmc_handle* h, int slot) {
    HIP_TRY(hipMemsetAsync(h->ctl + slot, 0, sizeof(KmcLevelCtl), h->stream));
    return KMC_OK;
}

// Segment sizes the device reported for the level it just produced (clipped to capacity).
uint64_t produced_segments(kmc_handle* h, const KmcLevelCtl& c, uint64_t seg[KMC_SEGS]) {
    uint64_t total = 0;
    for (int sg = 0; sg < KMC_SEGS; ++sg) {
        seg[sg] = c.next_count[sg].v < h->seg_cap ? c.next_count[sg].v : h->seg_cap;
        total += seg[sg];
    }
    return total;
}

int find_state(kmc_handle* h, const u64* frontier, const uint64_t seg[KMC_SEGS], uint64_t fp,
               std::vector<uint64_t>* out) {
    KmcArgs a = base_args(h, 2);
    a.fin = frontier;
    uint64_t n = 0;
    for (int sg = 0; sg < KMC_SEGS; ++sg) { a.seg_count[sg] = seg[sg]; n += seg[sg]; }
    a.table_cap = fp;  // kmc_find_body takes the target here
    a.send = h->scratch;
    HIP_TRY(hipMemsetAsync(h->scratch, 0xFF, (h->W + 1) * 8, h->stream));
    int rc = launch(h, h->f_find, a, expand_grid(h, n));
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(h->scratch_host, h->scratch, (h->W + 1) * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->scratch_host[h->W] == ~0ull) return fail(KMC_E_STATE, "witness fingerprint not found in frontier");
    out->assign(h->scratch_host, h->scratch_host + h->W);
    return KMC_OK;
}

// A violating successor outside the state constraint is in no table and no frontier: re-enumerate
// the successors of the expanded level, keeping those whose fingerprint is `fp` (ENUM_MATCH), to
// get its words and the parent it came from (the one with the smallest fingerprint).
int find_outside_witness(kmc_handle* h, const u64* frontier, const uint64_t seg[KMC_SEGS], uint64_t fp) {
    int rc = zero_ctl(h, 2);
    if (rc) return rc;
    KmcArgs a = base_args(h, 2);
    a.fin = frontier;
    uint64_t n = 0;
    for (int sg = 0; sg < KMC_SEGS; ++sg) { a.seg_count[sg] = seg[sg]; n += seg[sg]; }
    a.flags |= KMC_FLAG_ENUM_MATCH;
    a.match_fp = fp;
    a.send = h->enum_out;
    a.send_cap = h->enum_cap;
    a.inv_mask = 0;
    if ((rc = launch_expand(h, KMC_MODE_ENUM, a, expand_grid(h, n)))) return rc;
    if ((rc = read_ctl(h, 2))) return rc;
    const uint64_t cnt = h->ctl_host->enum_count < h->enum_cap ? h->ctl_host->enum_count : h->enum_cap;
    if (cnt == 0) return fail(KMC_E_STATE, "witness outside the constraint not found among the successors");
    std::vector<uint64_t> recs(cnt * (h->W + 2));
    HIP_TRY(hipMemcpy(recs.data(), h->enum_out, recs.size() * 8, hipMemcpyDeviceToHost));
    uint64_t best = 0;
    for (uint64_t i = 1; i < cnt; ++i)
        if (recs[i * (h->W + 2) + h->W + 1] < recs[best * (h->W + 2) + h->W + 1]) best = i;
    h->witness.assign(&recs[best * (h->W + 2)], &recs[best * (h->W + 2)] + h->W);
    h->witness_parent_fp = recs[best * (h->W + 2) + h->W + 1];
    h->witness_outside = true;
    return KMC_OK;
}

int reset_run(kmc_handle* h) {
    // (Clearing a second table on a side stream in the shadow of the run — a double-buffered seen-set — was measured in
    // round 3: the step got 0.4 ms shorter, but the memset's own kernel competes with the first, small levels and their
    // launches got 0.9 ms longer in total; dropped, profiles/r03_step_overhead.txt.)
    const double t_clear0 = now_s();
    if (!h->ev_aux[0]) {
        HIP_TRY(hipEventCreate(&h->ev_aux[0]));
        HIP_TRY(hipEventCreate(&h->ev_aux[1]));
    }
    range_push("kmc clear seen-set %s", h->kname.c_str());
    HIP_TRY(hipEventRecord(h->ev_aux[0], h->stream));
    HIP_TRY(hipMemsetAsync(h->table, 0, h->table_cap * h->stride_words() * 8, h->stream));
    if (h->pred && !h->paired) HIP_TRY(hipMemsetAsync(h->pred, 0, h->table_cap * 8, h->stream));
    HIP_TRY(hipEventRecord(h->ev_aux[1], h->stream));
    range_pop();
    h->clear_pending = true;   // (its duration is read where the stream is next waited for: do_begin)
    if (!h->first_clear_timed) {   // the first clear of a handle touches freshly mapped memory: timed once, by waiting for it
        h->first_clear_timed = true;
        HIP_TRY(hipStreamSynchronize(h->stream));
        h->timing.first_clear_s = now_s() - t_clear0;
    }
    if (h->sent) HIP_TRY(hipMemsetAsync(h->sent, 0, h->sent_cap * 8, h->stream));
    HIP_TRY(hipMemsetAsync(h->ctl, 0, KMC_CTL_SLOTS * sizeof(KmcLevelCtl), h->stream));
    h->levels.clear();
    h->level_stats.clear();
    h->step_expand_ms = 0;
    h->witness.clear();
    h->have_witness = false;
    h->have_deadlock = false;
    h->witness_outside = false;
    h->witness_parent_fp = 0;
    memset(&h->res, 0, sizeof h->res);
    h->res.violated_invariant = -1;
    h->res.table_capacity = h->table_cap;
    h->res.frontier_capacity = h->fcap;
    h->res.state_words = h->W;
    h->res.state_bits = h->lay.bits;
    h->cur = 0;
    h->n_cur = 0;
    for (int sg = 0; sg < KMC_SEGS; ++sg) h->seg_n[sg] = 0;
    h->level = 0;
    h->t_start = now_s();
    return KMC_OK;
}

// Fold the device counters of one expansion into the running result.  `c` describes the expansion
// of the frontier at depth h->level (the "parent" level): invariant violations and deadlocks refer
// to ITS states, generated / next_count to the level it produced.  Returns true when the search
// must stop.  On a stopping invariant violation the produced level is rolled back (not counted), so
// the reported numbers are those of a checker that tests each state when it is first found.
// Conservation of successors through one level's kernels (always on; two counters per wave on the device):
//   what pass 2 of k_expand dispatched, less the repeats (one successor, two bindings) and the successors outside the state
//   constraint, plus the records k_insert was handed, must be what entered the sink:  generated - repeats - outside + inserted = probed
//   and every claim the sink won must have been appended to the next frontier:       won = sum(next_count).
// Round 1 met a build of k_expand that LOST successors between dispatch and sink (docs/TUNING_LOG_r1-r3.md §2); every counter the old
// self-check compared is bumped before that point.  These two are taken on either side of it.
// kmc_config.symmetry: a device counter counts orbit representatives and comes with the summed deficits of their orbits
// (KmcLevelCtl::corr_*): the plain search's count is N! * raw - corr.
uint64_t weighted(const kmc_handle* h, uint64_t raw, uint64_t corr) { return h->cfg.symmetry ? h->nfact * raw - corr : raw; }
// states on the frontier, as the plain search counts them (the frontier is always a whole level)
uint64_t queue_now(const kmc_handle* h) { return h->cfg.symmetry && !h->levels.empty() ? h->levels.back() : h->n_cur; }
// a level of `produced` stored states enters the books
void book_level(kmc_handle* h, uint64_t produced, const KmcLevelCtl& c) {
    h->res.orbit_representatives += produced;
    const uint64_t w = weighted(h, produced, c.corr_won);
    h->res.distinct += w;
    h->levels.push_back(w);
    // (Widths.  k_expand sums the orbit deficits of a launch's counts per LANE and per WAVE in 32 bits and per BLOCK in
    // 64-bit LDS cells (kmc_device.h, kmc_corr).  Round 3 had 32-bit block cells: at 17 levels of BASELINE config 5 — 133 M
    // stored states in a level, up to 5,039 per successor — they wrapped and `generated` came out 2^40 too large, found by
    // oracle/orbit_oracle.c.  A lane sees produced / (blocks x 256) states of a level: its sums stay below 2^26 for any level
    // the frontier can hold.)
}

int check_conservation(kmc_handle* h, const KmcLevelCtl& c, uint64_t inserted) {
    if (c.err) return KMC_OK;   // a full table / frontier / send area stops probing and appending on purpose
    uint64_t gen = 0, appended = 0;
    for (int k = 0; k < KMC_MAX_KINDS; ++k) gen += c.generated[k];
    for (int sg = 0; sg < KMC_SEGS; ++sg) appended += c.next_count[sg].v;
    const uint64_t expect = gen - c.repeats - c.outside + inserted;
    if (expect != c.probed)
        return fail(KMC_E_DEVICE, "conservation violated at level %llu of kmc_expand_%s: %llu successors were dispatched "
                                  "(%llu generated - %llu repeats - %llu outside the constraint + %llu inserted) but %llu reached "
                                  "the seen-set: the kernel lost or invented successors",
                    (unsigned long long)h->level, h->kname.c_str(), (unsigned long long)expect, (unsigned long long)gen,
                    (unsigned long long)c.repeats, (unsigned long long)c.outside, (unsigned long long)inserted,
                    (unsigned long long)c.probed);
    if (c.won != appended)
        return fail(KMC_E_DEVICE, "conservation violated at level %llu of kmc_expand_%s: %llu claims were won but %llu states "
                                  "were appended to the next frontier", (unsigned long long)h->level, h->kname.c_str(),
                    (unsigned long long)c.won, (unsigned long long)appended);
    return KMC_OK;
}

bool absorb(kmc_handle* h, const KmcLevelCtl& c, const u64* parent_frontier, const uint64_t* parent_seg, int* rc) {
    kmc_result& r = h->res;
    *rc = KMC_OK;
    if (c.err & KMC_ERR_CHECK_WORD) {
        *rc = fail(KMC_E_DEVICE, "wide fingerprints: a claimed slot's check word did not appear (level %llu)", (unsigned long long)h->level);
        r.verdict = KMC_V_ERROR;
        return true;
    }
    if ((*rc = check_conservation(h, c, 0))) {
        r.verdict = KMC_V_ERROR;
        return true;
    }
    if (r.violated_invariant < 0) {
        for (int k = 0; k < 4; ++k) {
            if ((h->cfg.invariant_mask >> k & 1u) && c.viol_count[k]) {
                r.violated_invariant = k;
                r.violation_depth = h->level;
                r.violation_fp = ~c.viol_fp_inv[k];
                for (int j = 0; j < 4; ++j) r.violation_count[j] = weighted(h, c.viol_count[j], c.corr_viol[j]);
                if (parent_frontier && h->cfg.n_shards == 1) {
                    *rc = find_state(h, parent_frontier, parent_seg, r.violation_fp, &h->witness);
                    h->have_witness = *rc == KMC_OK;
                }
                break;
            }
        }
        // successors outside the state constraint that violate an invariant: one level deeper than
        // the expanded states, so a violation among those takes precedence
        for (int k = 0; k < 4 && r.violated_invariant < 0; ++k) {
            if ((h->cfg.invariant_mask >> k & 1u) && c.oviol_count[k]) {
                r.violated_invariant = k;
                r.violation_depth = h->level + 1;
                r.violation_fp = ~c.oviol_fp_inv[k];
                for (int j = 0; j < 4; ++j) r.violation_count[j] = c.oviol_count[j];
                if (parent_frontier && h->cfg.n_shards == 1) {
                    *rc = find_outside_witness(h, parent_frontier, parent_seg, r.violation_fp);
                    h->have_witness = *rc == KMC_OK;
                }
            }
        }
        if (r.violated_invariant >= 0) {
            r.verdict = KMC_V_INVARIANT;
            if (!h->cfg.continue_on_violation) return true;
        }
    }
    for (int k = 0; k < KMC_MAX_KINDS; ++k) {
        const uint64_t g = weighted(h, c.generated[k], c.corr_gen[k]);
        r.generated += g;
        r.action_generated[k] += g;
    }
    r.generated_repeats += weighted(h, c.repeats, c.corr_repeats);
    r.deadlock_states += weighted(h, c.deadlock_count, c.corr_dead);
    if (c.err & KMC_ERR_TABLE_FULL) { r.verdict = KMC_V_TABLE_FULL; return true; }
    if (c.err & (KMC_ERR_FRONTIER_FULL | KMC_ERR_SEND_FULL)) { r.verdict = KMC_V_FRONTIER_FULL; return true; }
    if (h->cfg.check_deadlock && c.deadlock_count && (r.verdict == KMC_V_OK || r.verdict == KMC_V_INVARIANT) &&
        !h->have_deadlock) {
        h->have_deadlock = true;
        if (r.verdict == KMC_V_OK) {
            r.verdict = KMC_V_DEADLOCK;
            r.violation_depth = h->level;
            r.violation_fp = ~c.deadlock_fp_inv;
            if (parent_frontier && h->cfg.n_shards == 1) {
                *rc = find_state(h, parent_frontier, parent_seg, r.violation_fp, &h->witness);
                h->have_witness = *rc == KMC_OK;
            }
            return true;
        }
    }
    return false;
}

// Produce Init and insert it on its owner.  Leaves level = 1.
int do_begin(kmc_handle* h) {
    // a stepped search that stopped on a verdict never reached kmc_step_finish: a pipelined level's last transfer and insert
    // may still be in flight on the second stream, and its records are still booked — neither belongs to the new search
    if (h->xstream) HIP_TRY(hipStreamSynchronize(h->xstream));
    h->inserted_level = 0;
    int rc = reset_run(h);
    if (rc) return rc;
    KmcArgs a = base_args(h, 0);
    a.send = h->scratch;
    if ((rc = launch(h, h->f_init, a, 1))) return rc;
    HIP_TRY(hipMemcpyAsync(h->scratch_host, h->scratch, (h->W + 1) * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->clear_pending) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, h->ev_aux[0], h->ev_aux[1]));
        h->res.seconds_clear = 1e-3 * ms;
        h->clear_pending = false;
    }
    h->init_words.assign(h->scratch_host, h->scratch_host + h->W);
    uint64_t init_orbit = 1;
    if (h->cfg.symmetry) {
        // Init is stored as its orbit's representative like every other state (the specs' Init is fixed by every
        // permutation, KafkaReplication.tla:109-120 / FiniteReplicatedLog.tla:97 — then nothing changes and the orbit is 1)
        unsigned long long c[KMC_MAXW] = {0}, w0[KMC_MAXW] = {0};
        for (int k = 0; k < h->W; ++k) w0[k] = h->init_words[k];
        int stab = 1;
        kmc_canonical_state_generic(h->lay, w0, c, &stab);
        init_orbit = h->nfact / (uint64_t)stab;
        for (int k = 0; k < h->W; ++k) h->init_words[k] = h->scratch_host[k] = c[k];
        h->scratch_host[h->W] = 0;
        HIP_TRY(hipMemcpyAsync(h->scratch, h->scratch_host, (h->W + 1) * 8, hipMemcpyHostToDevice, h->stream));
    }
    const uint64_t fp0 = kmc_fingerprint_of(h, h->init_words.data());
    const bool mine = h->cfg.n_shards <= 1 || kmc_owner(fp0, (uint32_t)h->cfg.n_shards) == (uint32_t)h->cfg.shard_id;
    if (mine) {
        KmcArgs b = base_args(h, 0);
        b.recv = h->scratch;
        b.n_in = 1;
        b.fout = h->frontier[0];
        if ((rc = launch(h, h->f_insert, b, 1))) return rc;
        h->res.generated = 1;
    }
    if ((rc = read_ctl(h, 0))) return rc;
    h->n_cur = produced_segments(h, *h->ctl_host, h->seg_n);
    h->cur = 0;
    h->res.orbit_representatives = h->n_cur;
    h->res.distinct = h->cfg.symmetry ? h->n_cur * init_orbit : h->n_cur;
    h->levels.push_back(h->res.distinct);
    h->level = 1;
    h->res.depth = 1;
    return rc;
}

void note_level(kmc_handle* h, const KmcLevelCtl& c, uint64_t frontier, uint64_t produced, double expand_ms) {
    kmc_level_stat st{};
    st.depth = h->level + 1;   // (called before the level enters the books: h->level is still the expanded level's depth)
    st.frontier = frontier;
    st.stored_new = produced;
    st.new_states = weighted(h, produced, c.corr_won);
    for (int k = 0; k < KMC_MAX_KINDS; ++k) st.generated[k] = weighted(h, c.generated[k], c.corr_gen[k]);
    st.probes = c.probed;
    st.deadlocks = weighted(h, c.deadlock_count, c.corr_dead);
    st.table_load = h->table_cap ? (double)(h->res.orbit_representatives + produced) / (double)h->table_cap : 0.0;
    st.expand_ms = expand_ms;
    h->level_stats.push_back(st);
}

// ---- roctx ranges --------------------------------------------------------------------------------------------------------
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};
Roctx* roctx() {
    static Roctx r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* want = getenv("KMC_ROCTX");
        if (want && !atoi(want)) return;   // KMC_ROCTX=0: never
        // a profiler that traces markers has the library in the process already (RTLD_NOLOAD finds it); KMC_ROCTX=1 loads it
        const char* names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};
        void* lib = nullptr;
        for (const char* n : names)
            if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!lib && want)
            for (const char* n : names)
                if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!lib) return;
        r.push = (int (*)(const char*))dlsym(lib, "roctxRangePushA");
        r.pop = (int (*)())dlsym(lib, "roctxRangePop");
        if (!r.push || !r.pop) r.push = nullptr;
    });
    return r.push ? &r : nullptr;
}
}  // namespace

void range_push(const char* fmt, ...) {
    Roctx* r = roctx();
    if (!r) return;
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    r->push(buf);
}
void range_pop() {
    if (Roctx* r = roctx()) r->pop();
}

}  // namespace kmc_engine
