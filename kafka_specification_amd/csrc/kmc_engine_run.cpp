// kmc_engine_run.cpp — kmc_run's level loop (chained launches), results, kmc_successors / kmc_check_states, witness, contains, traces.
#include "kmc_engine_internal.h"
using namespace kmc_engine;

static int run_levels(kmc_handle* h, kmc_progress_cb cb, void* user, bool fresh);

extern "C" {

int kmc_run(kmc_handle* h, kmc_progress_cb cb, void* user) {
    if (!h) return fail(KMC_E_ARG, "null handle");
    if (!h->table) return fail(KMC_E_STATE, "host-only handle (device = -1) cannot run");
    if (h->cfg.n_shards != 1) return fail(KMC_E_STATE, "kmc_run drives one GPU; use the kmc_step_* interface for shards");
    HIP_TRY(hipSetDevice(h->cfg.device));
    h->stepping = false;
    if (const char* which = getenv("KMC_DEBUG_REALLOC")) {
        // tuning aid (profiles/r06_table_size_and_placement.txt, item 3): ONE of the handle's buffers — table | f0 | f1 | ctl — is
        // moved to another place in the HBM before this search; k_expand's time follows where the seen-set lies
        HIP_TRY(hipStreamSynchronize(h->stream));
        auto move = [&](u64*& p, size_t bytes) -> int {
            u64* q = nullptr;
            void* pad = nullptr;
            (void)hipMalloc(&pad, 768ull << 20);   // (leaked: shifts where the next buffers land)
            if (hipMalloc(&q, bytes) != hipSuccess) return fail(KMC_E_NOMEM, "debug realloc");
            HIP_TRY(hipMemcpy(q, p, bytes, hipMemcpyDeviceToDevice));
            HIP_TRY(hipFree(p));
            p = q;
            return KMC_OK;
        };
        int rc0 = KMC_OK;
        if (!strcmp(which, "table")) {   // (through the seen-set's own allocator: chunks, or one hipMalloc under KMC_SEEN_SET_CHUNK_LOG2=0)
            void* pad = nullptr;
            (void)hipMalloc(&pad, 768ull << 20);
            u64* q = seen_set_alloc(h, h->table_cap * h->stride_words() * 8);
            if (!q) return fail(KMC_E_NOMEM, "debug realloc");
            seen_set_free(h, h->table);
            h->table = q;
            if (h->paired) h->pred = q + 1;
        }
        else if (!strcmp(which, "f0")) rc0 = move(h->frontier[0], h->fcap * 8ull * h->planes);
        else if (!strcmp(which, "f1")) rc0 = move(h->frontier[1], h->fcap * 8ull * h->planes);
        else if (!strcmp(which, "ctl")) { u64* c = (u64*)h->ctl; rc0 = move(c, KMC_CTL_SLOTS * sizeof(KmcLevelCtl)); h->ctl = (KmcLevelCtl*)c; }
        if (rc0) return rc0;
    }
    int rc = do_begin(h);
    if (rc) return rc;
    return run_levels(h, cb, user, true);
}

// TLC -recover analogue: continue the search of a handle restored by kmc_checkpoint_load.
int kmc_resume(kmc_handle* h, kmc_progress_cb cb, void* user) {
    if (!h) return fail(KMC_E_ARG, "null handle");
    if (!h->table || !h->restored) return fail(KMC_E_STATE, "kmc_resume needs a handle restored by kmc_checkpoint_load");
    HIP_TRY(hipSetDevice(h->cfg.device));
    h->stepping = false;
    h->restored = false;
    h->t_start = now_s() - h->res.seconds_total;
    if (h->res.verdict == KMC_V_LEVEL_LIMIT) h->res.verdict = KMC_V_OK;  // the limit that stopped the saved run is lifted
    h->res.queue_left = 0;
    return run_levels(h, cb, user, false);
}

}  // extern "C"

static int run_levels(kmc_handle* h, kmc_progress_cb cb, void* user, bool fresh) {
    int rc = KMC_OK;
    kmc_result& r = h->res;
    auto report = [&]() {
        if (!cb) return;
        kmc_level_info info{};
        info.depth = h->level;
        info.new_states = queue_now(h);
        info.generated_total = r.generated;
        info.distinct_total = r.distinct;
        info.seconds = now_s() - h->t_start;
        cb(&info, user);
    };
    if (fresh) report();
    bool stop = false;
    const uint64_t max_levels = h->cfg.max_levels ? h->cfg.max_levels : ~0ull;
    while (!stop && h->n_cur > 0) {
        if (h->level >= max_levels) {
            // the last frontier is not expanded: give its states their invariant check now
            if ((rc = zero_ctl(h, 2))) return rc;
            KmcArgs d = base_args(h, 2);
            d.fin = h->frontier[h->cur];
            range_push("kmc invariants of level %llu (%llu states, not expanded)", (unsigned long long)h->level, (unsigned long long)h->n_cur);
            HIP_TRY(hipEventRecord(h->ev_aux[0], h->stream));
            if ((rc = launch_inv(h, d, h->n_cur))) return rc;   // (a full dry expansion of BASELINE config 5's tenth level took 64 ms: twice the search)
            HIP_TRY(hipEventRecord(h->ev_aux[1], h->stream));
            range_pop();
            if ((rc = read_ctl(h, 2))) return rc;
            {
                float ims = 0;
                HIP_TRY(hipEventElapsedTime(&ims, h->ev_aux[0], h->ev_aux[1]));
                r.seconds_inv += 1e-3 * ims;
                r.inv_launches++;
            }
            KmcLevelCtl c = *h->ctl_host;
            for (int k = 0; k < KMC_MAX_KINDS; ++k) c.generated[k] = 0;
            c.deadlock_count = 0;
            c.err = 0;
            c.probed = c.won = c.outside = c.repeats = 0;   // an invariant-only pass: nothing was dispatched for the record
            for (int k = 0; k < KMC_MAX_KINDS; ++k) c.corr_gen[k] = 0;
            c.corr_dead = c.corr_repeats = c.corr_won = 0;
            absorb(h, c, h->frontier[h->cur], h->seg_n, &rc);
            if (rc) return rc;
            if (r.verdict == KMC_V_OK) r.verdict = KMC_V_LEVEL_LIMIT;
            r.queue_left = queue_now(h);
            break;
        }
        static const int shadow = getenv("KMC_SHADOW") ? atoi(getenv("KMC_SHADOW")) : 0;
        static const int dry_mode = getenv("KMC_DRYRUN") ? atoi(getenv("KMC_DRYRUN")) : 0;
        static const int no_chain = getenv("KMC_NO_CHAIN") ? atoi(getenv("KMC_NO_CHAIN")) : 0;
        // Under -continue a violation does not end the search, so levels queued behind the violating one would run and
        // overwrite its parent frontier before the host could fetch the witness (find_state / find_outside_witness):
        // such runs go level by level until the first violation has been recorded, and chain from there on.
        const bool witness_pending = h->cfg.continue_on_violation && h->cfg.invariant_mask && r.violated_invariant < 0;
        if (!cb && !shadow && !dry_mode && !no_chain && !h->f_expand_verify && !witness_pending) {
            // ---- chained launches -------------------------------------------------------------------------
            // Nobody watches the levels go by, so up to KMC_CHAIN of them are queued back to back and the host
            // waits ONCE: a level launched behind another one takes its segment sizes from that level's control
            // block on the device and does nothing if that level (or one before it) ended the search.  Per level
            // this leaves a launch and two event records on the host instead of memset + launch + copy + wait
            // (46 levels, 7 of them under 1024 states: 2.4 ms of a 38 ms check in round 1).
            uint64_t B = max_levels - h->level;
            static const uint64_t chain_max = getenv("KMC_CHAIN_MAX") ? (uint64_t)atoi(getenv("KMC_CHAIN_MAX")) : 16;
            if (B > KMC_CHAIN) B = KMC_CHAIN;
            if (chain_max >= 1 && B > chain_max) B = chain_max;
            const uint64_t fan = max_fanout(h) ? max_fanout(h) : 1;
            {
                // The load limit of the table (0.92, below) is a HOST decision, taken after a level: a batch is therefore
                // only as long as its levels provably stay under it — each level adds at most min(fan x its input,
                // frontier capacity) states.  (Without this a batch could run the table far past the limit before the
                // host looked, and after such a stop h->cur / seg_n no longer described the device's frontier: ADVICE r2.)
                const double room = 0.92 * (double)h->table_cap - (double)r.orbit_representatives;
                uint64_t in = h->n_cur, fit = 0;
                double sum = 0;
                for (; fit < B; ++fit) {
                    const uint64_t out = in > h->fcap / fan ? h->fcap : in * fan;
                    sum += (double)out;
                    if (fit > 0 && sum > room) break;   // (the first level always runs: the host checks right after it)
                    in = out;
                }
                B = fit ? fit : 1;
            }
            HIP_TRY(hipMemsetAsync(h->ctl + 3, 0, B * sizeof(KmcLevelCtl), h->stream));
            range_push("kmc %s: chain of %llu levels from depth %llu (%llu states)", h->kname.c_str(), (unsigned long long)B,
                       (unsigned long long)h->level, (unsigned long long)h->n_cur);
            uint64_t bound = h->n_cur;   // upper bound on the size of the level launch i expands
            for (uint64_t i = 0; i < B; ++i) {
                if (!h->ev_chain[2 * i]) {
                    HIP_TRY(hipEventCreate(&h->ev_chain[2 * i]));
                    HIP_TRY(hipEventCreate(&h->ev_chain[2 * i + 1]));
                }
                KmcArgs a = base_args(h, 3 + (int)i);
                const int ci = h->cur ^ (int)(i & 1);
                a.fin = h->frontier[ci];
                a.fout = h->frontier[ci ^ 1];
                a.prev = i ? h->ctl + 3 + (i - 1) : nullptr;   // the first level of a batch always runs, on host-known sizes
                a.stop_mask = h->cfg.continue_on_violation ? 0u : h->cfg.invariant_mask;
                a.stop_deadlock = (h->cfg.check_deadlock && r.verdict == KMC_V_OK) ? 1u : 0u;
                HIP_TRY(hipEventRecord(h->ev_chain[2 * i], h->stream));
                // sizes behind the first level are only known on the device: the grid is sized for the most a level can
                // grow (every state enabling every action instance), which saturates at a resident grid within two or
                // three levels but keeps the chains of tiny levels (IdSequence: 1002 one-state levels) to one block
                if ((rc = launch_expand(h, KMC_MODE_LOCAL, a, expand_grid(h, bound)))) return rc;
                bound = bound > h->fcap / fan ? h->fcap : bound * fan;
                HIP_TRY(hipEventRecord(h->ev_chain[2 * i + 1], h->stream));
            }
            HIP_TRY(hipMemcpyAsync(h->ctl_host, h->ctl + 3, B * sizeof(KmcLevelCtl), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            range_pop();
            float chain_ms[KMC_CHAIN] = {0};
            // every launch of the batch is accounted (also the ones behind the end of the search, which find nothing to
            // do and return in microseconds): the per-launch average then is what rocprofv3 --kernel-trace reports
            for (uint64_t i = 0; i < B; ++i) {
                float ms = 0;
                HIP_TRY(hipEventElapsedTime(&ms, h->ev_chain[2 * i], h->ev_chain[2 * i + 1]));
                r.seconds_expand += 1e-3 * ms;
                r.expand_launches++;
                chain_ms[i] = ms;
            }
            bool done = false;
            for (uint64_t i = 0; i < B && !done; ++i) {
                const KmcLevelCtl c = h->ctl_host[i];
                if (c.halt) break;   // the device ended the chain here; the host decides below whether the search goes on
                for (int k = 0; k < 8; ++k) h->prof[k] += c.prof[k];
                uint64_t new_seg[KMC_SEGS];
                const uint64_t produced = produced_segments(h, c, new_seg);
                stop = absorb(h, c, h->frontier[h->cur], h->seg_n, &rc);
                if (rc) return rc;
                if (stop) {
                    r.queue_left = queue_now(h);
                    done = true;
                    break;
                }
                note_level(h, c, h->n_cur, produced, chain_ms[i]);
                if (produced == 0) {
                    h->n_cur = 0;
                    done = true;
                    break;
                }
                h->cur ^= 1;
                h->n_cur = produced;
                for (int sg = 0; sg < KMC_SEGS; ++sg) h->seg_n[sg] = new_seg[sg];
                h->level++;
                r.depth = h->level;
                book_level(h, produced, c);
                if ((double)r.orbit_representatives > 0.92 * (double)h->table_cap && r.verdict == KMC_V_OK) {
                    r.verdict = KMC_V_TABLE_FULL;
                    r.queue_left = queue_now(h);
                    stop = done = true;
                }
            }
            if (done) break;
            continue;
        }
        const int slot = (int)(h->level & 1);
        const int nxt = h->cur ^ 1;
        if ((rc = zero_ctl(h, slot))) return rc;
        KmcArgs a = base_args(h, slot);
        a.fin = h->frontier[h->cur];
        a.fout = h->frontier[nxt];
        if (shadow) {  // tuning aid: the identical level first runs on a copy of the table, with KMC_XFLAGS applied
            if (!h->table2) HIP_TRY(hipMalloc(&h->table2, h->table_cap * h->stride_words() * 8));
            HIP_TRY(hipMemcpyAsync(h->table2, h->table, h->table_cap * h->stride_words() * 8, hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(hipMemsetAsync(h->ctl + 2, 0, sizeof(KmcLevelCtl), h->stream));
            KmcArgs x = a;
            x.table = h->table2;
            if (h->paired) x.pred = h->table2 + 1;
            x.ctl = h->ctl + 2;
            x.flags |= getenv("KMC_XFLAGS") ? (uint32_t)atoi(getenv("KMC_XFLAGS")) : 0u;
            HIP_TRY(hipEventRecord(h->ev0, h->stream));
            if ((rc = launch_expand(h, KMC_MODE_LOCAL, x, expand_grid(h, h->n_cur)))) return rc;
            HIP_TRY(hipEventRecord(h->ev1, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            float xms = 0;
            HIP_TRY(hipEventElapsedTime(&xms, h->ev0, h->ev1));
            h->dry_seconds += 1e-3 * xms;
        }
        range_push("kmc %s: level %llu (%llu states)", h->kname.c_str(), (unsigned long long)h->level, (unsigned long long)h->n_cur);
        HIP_TRY(hipEventRecord(h->ev0, h->stream));
        if ((rc = launch_expand(h, KMC_MODE_LOCAL, a, expand_grid(h, h->n_cur)))) return rc;
        HIP_TRY(hipEventRecord(h->ev1, h->stream));
        if ((rc = read_ctl(h, slot))) return rc;
        range_pop();
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
        r.seconds_expand += 1e-3 * ms;
        r.expand_launches++;
        const KmcLevelCtl c = *h->ctl_host;
        for (int k = 0; k < 8; ++k) h->prof[k] += c.prof[k];
        if (h->f_expand_verify) {  // KMC_VERIFY: the second build regenerates this level; the counts must agree
            KmcArgs v = a;
            v.ctl = h->ctl + 2;
            if ((rc = zero_ctl(h, 2))) return rc;
            if ((rc = launch_expand(h, KMC_MODE_DRY, v, expand_grid(h, h->n_cur), nullptr, true))) return rc;
            KmcLevelCtl vc;
            HIP_TRY(hipMemcpyAsync(&vc, h->ctl + 2, KMC_CTL_LOCAL_BYTES, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            bool same = vc.deadlock_count == c.deadlock_count;
            for (int k = 0; k < KMC_MAX_KINDS; ++k) same = same && vc.generated[k] == c.generated[k];
            for (int k = 0; k < 4; ++k) same = same && vc.viol_count[k] == c.viol_count[k];
            // (orbit counting: the deficits taken when a state is expanded — the second build finds the stabilisers again)
            for (int k = 0; k < KMC_MAX_KINDS; ++k) same = same && vc.corr_gen[k] == c.corr_gen[k];
            for (int k = 0; k < 4; ++k) same = same && vc.corr_viol[k] == c.corr_viol[k];
            same = same && vc.corr_dead == c.corr_dead && vc.corr_repeats == c.corr_repeats;
            // ... and the successors themselves: how many reached the sink, and the order-independent checksum of their
            // fingerprints (taken where a successor enters the sink — behind the ring and the flush, where round 1's
            // miscompiled kernel lost some while every count above still agreed)
            const bool same_succ = vc.probed == c.probed && vc.fp_sum == c.fp_sum && vc.fp_xor == c.fp_xor &&
                                   vc.repeats == c.repeats && vc.outside == c.outside;
            if (!same || !same_succ) {
                r.verdict = KMC_V_ERROR;
                return fail(KMC_E_DEVICE, "KMC_VERIFY: the two builds of kmc_expand_%s disagree at level %llu (%s): one of them "
                                          "is miscompiled", h->kname.c_str(), (unsigned long long)h->level,
                            same ? "the successors reaching the seen-set differ: count or fingerprint checksum"
                                 : "generated / deadlock / violation counts differ");
            }
            h->verify_levels++;
        }
        const int dry = dry_mode;
        if (dry) {  // tuning aid: time the same level again without table writes / frontier traffic
            KmcArgs d = a;  // 1: no table access at all, 2: + read-only probes, 3: + invariants on every successor
            if (dry >= 2) d.flags |= KMC_FLAG_DRY_PROBE;
            if (dry == 3) d.flags |= KMC_FLAG_DRY_INV;
            if (dry == 4) d.flags |= KMC_FLAG_DRY_ATOM;
            if (dry == 5) d.flags |= KMC_FLAG_DRY_RAND;
            d.ctl = h->ctl + 2;
            HIP_TRY(hipEventRecord(h->ev0, h->stream));
            if ((rc = launch_expand(h, KMC_MODE_DRY, d, expand_grid(h, h->n_cur)))) return rc;
            HIP_TRY(hipEventRecord(h->ev1, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            float dms = 0;
            HIP_TRY(hipEventElapsedTime(&dms, h->ev0, h->ev1));
            h->dry_seconds += 1e-3 * dms;
            KmcLevelCtl dc;
            HIP_TRY(hipMemcpy(&dc, h->ctl + 2, sizeof dc, hipMemcpyDeviceToHost));
            for (int k = 0; k < 8; ++k) h->prof_dry[k] += dc.prof[k];
            HIP_TRY(hipMemsetAsync(h->ctl + 2, 0, sizeof(KmcLevelCtl), h->stream));
        }
        uint64_t new_seg[KMC_SEGS];
        const uint64_t produced = produced_segments(h, c, new_seg);
        stop = absorb(h, c, h->frontier[h->cur], h->seg_n, &rc);
        if (rc) return rc;
        if (stop) {  // invariant (produced level rolled back), deadlock, table/frontier full
            r.queue_left = queue_now(h);
            break;
        }
        note_level(h, c, h->n_cur, produced, (double)ms);
        if (produced == 0) {
            h->n_cur = 0;
            break;
        }
        h->cur = nxt;
        h->n_cur = produced;
        for (int sg = 0; sg < KMC_SEGS; ++sg) h->seg_n[sg] = new_seg[sg];
        h->level++;
        r.depth = h->level;
        book_level(h, produced, c);
        report();
        // stop before linear probing degenerates (sized for load <= 0.5, still fine at 0.9)
        if ((double)r.orbit_representatives > 0.92 * (double)h->table_cap && r.verdict == KMC_V_OK) {
            r.verdict = KMC_V_TABLE_FULL;
            r.queue_left = queue_now(h);
            break;
        }
    }
    r.n_levels = h->levels.size();
    r.seconds_total = now_s() - h->t_start;
    if (h->prof[7]) {
        const double tot = (double)h->prof[7];
        fprintf(stderr, "[kmc] per-wave ticks: load+extract+inv %.1f%%  guards %.1f%%  effects+push(incl flush) %.1f%%  "
                        "of which flush %.1f%%  tail %.1f%%  (total %.3g ticks)\n",
                100 * h->prof[0] / tot, 100 * h->prof[1] / tot, 100 * h->prof[2] / tot, 100 * h->prof[3] / tot,
                100 * h->prof[4] / tot, tot);
        if (h->prof[6])
            fprintf(stderr, "[kmc] effect leaves dispatched per 64-state tile: %.1f (%llu tiles)\n",
                    (double)h->prof[5] / (double)h->prof[6], (unsigned long long)h->prof[6]);
        for (int k = 0; k < 8; ++k) h->prof[k] = 0;
    }
    if (h->prof_dry[7]) {
        const double tot = (double)h->prof_dry[7];
        fprintf(stderr, "[kmc] DRY per-wave ticks: load+extract+inv %.1f%%  guards %.1f%%  effects+push(incl flush) %.1f%%  "
                        "of which flush %.1f%%  tail %.1f%%  (total %.3g ticks)\n",
                100 * h->prof_dry[0] / tot, 100 * h->prof_dry[1] / tot, 100 * h->prof_dry[2] / tot,
                100 * h->prof_dry[3] / tot, 100 * h->prof_dry[4] / tot, tot);
        for (int k = 0; k < 8; ++k) h->prof_dry[k] = 0;
    }
    if (h->dry_seconds > 0) {
        fprintf(stderr, "[kmc] dry/shadow expand: %.3f ms vs real %.3f ms\n",
                1e3 * h->dry_seconds, 1e3 * r.seconds_expand);
        h->dry_seconds = 0;
    }
    return KMC_OK;
}

extern "C" {

int kmc_timing_get(kmc_handle* h, kmc_timing* out) {
    if (!h || !out) return fail(KMC_E_ARG, "null argument");
    *out = h->timing;
    return KMC_OK;
}

int kmc_result_get(kmc_handle* h, kmc_result* out) {
    if (!h || !out) return fail(KMC_E_ARG, "null argument");
    h->res.n_levels = h->levels.size();
    *out = h->res;
    return KMC_OK;
}

uint64_t kmc_level_stats(kmc_handle* h, kmc_level_stat* out, uint64_t cap) {
    if (!h) return 0;
    for (uint64_t i = 0; out && i < h->level_stats.size() && i < cap; ++i) out[i] = h->level_stats[i];
    return h->level_stats.size();
}

uint64_t kmc_level_sizes(kmc_handle* h, uint64_t* out, uint64_t cap) {
    if (!h) return 0;
    for (uint64_t i = 0; i < h->levels.size() && i < cap; ++i) out[i] = h->levels[i];
    return h->levels.size();
}

int kmc_frontier_states(kmc_handle* h, uint64_t* words, uint64_t cap_states, uint64_t* n_out) {
    if (!h || !n_out) return fail(KMC_E_ARG, "null argument");
    if (!h->table) return fail(KMC_E_STATE, "host-only handle");
    HIP_TRY(hipSetDevice(h->cfg.device));
    const uint64_t n = h->n_cur < cap_states ? h->n_cur : cap_states;
    *n_out = h->n_cur;
    if (n == 0) return KMC_OK;
    std::vector<uint64_t> plane(h->seg_cap);
    uint64_t at = 0;
    for (int sg = 0; sg < KMC_SEGS && at < n; ++sg) {
        const uint64_t m = h->seg_n[sg] < n - at ? h->seg_n[sg] : n - at;
        for (int k = 0; k < h->W && m; ++k) {
            HIP_TRY(hipMemcpy(plane.data(), h->frontier[h->cur] + (uint64_t)k * h->fcap + (uint64_t)sg * h->seg_cap,
                              m * 8, hipMemcpyDeviceToHost));
            for (uint64_t i = 0; i < m; ++i) words[(at + i) * h->W + k] = plane[i];
        }
        at += m;
    }
    return KMC_OK;
}

int kmc_successors(kmc_handle* h, const uint64_t* words, uint64_t* out, uint64_t cap, uint64_t* n_out) {
    if (!h || !words || !n_out) return fail(KMC_E_ARG, "null argument");
    if (!h->table) return fail(KMC_E_STATE, "host-only handle");
    HIP_TRY(hipSetDevice(h->cfg.device));
    // the auxiliary frontier is the scratch buffer viewed as SoA with stride 1... planes must be
    // fin[k*stride + 0], so stride 1 puts the W words back to back
    HIP_TRY(hipMemcpyAsync(h->scratch, words, h->W * 8, hipMemcpyHostToDevice, h->stream));
    if (h->cfg.symmetry) {   // plane W of this one-state frontier: the stabiliser's order (no count is taken from an ENUM pass)
        static const uint64_t one = 1;
        HIP_TRY(hipMemcpyAsync(h->scratch + h->W, &one, 8, hipMemcpyHostToDevice, h->stream));
    }
    int rc = zero_ctl(h, 2);
    if (rc) return rc;
    KmcArgs a = base_args(h, 2);
    a.fin = h->scratch;
    a.fin_stride = 1;
    for (int sg = 0; sg < KMC_SEGS; ++sg) a.seg_count[sg] = sg == 0 ? 1 : 0;
    a.send = h->enum_out;
    a.send_cap = h->enum_cap;
    a.inv_mask = 0;
    if ((rc = launch_expand(h, KMC_MODE_ENUM, a, 1))) return rc;
    if ((rc = read_ctl(h, 2))) return rc;
    const uint64_t n = h->ctl_host->enum_count < h->enum_cap ? h->ctl_host->enum_count : h->enum_cap;
    // The kind word of a record also says how many FURTHER satisfying bindings of the same disjunct yield this very successor
    // (Kip279.tla:47-51, Kip320.tla:82-83: two disjuncts of one binding hold at once): the list handed out repeats such a
    // record, so that it is TLC's enumeration of Next on this state — one entry per generated successor, as `generated` counts.
    const uint64_t rw = (uint64_t)h->W + 2;
    std::vector<uint64_t> recs(n * rw);
    if (n) HIP_TRY(hipMemcpy(recs.data(), h->enum_out, n * rw * 8, hipMemcpyDeviceToHost));
    uint64_t total = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t reps = 1 + (recs[i * rw + h->W + 1] >> 8);
        recs[i * rw + h->W + 1] &= 0xFFull;
        for (uint64_t k = 0; k < reps; ++k, ++total)
            if (out && total < cap) memcpy(out + total * rw, &recs[i * rw], rw * 8);
    }
    *n_out = total;
    return KMC_OK;
}

// The invariants of `mask` each of n packed states violates, from the device's own predicate (M::violated_pre, the one
// k_expand applies to every state it expands): one single-state pass of k_expand per state in its dry mode (successors are
// generated and dropped, no table or frontier is touched), the per-invariant violation counters of the control block
// read back.  A differential-testing entry point (states as data), not a search.
int kmc_check_states(kmc_handle* h, const uint64_t* words, uint64_t n, uint32_t mask, uint32_t* violated) {
    if (!h || !words || !violated) return fail(KMC_E_ARG, "null argument");
    if (!h->table) return fail(KMC_E_STATE, "host-only handle");
    HIP_TRY(hipSetDevice(h->cfg.device));
    for (uint64_t i = 0; i < n; ++i) {
        HIP_TRY(hipMemcpyAsync(h->scratch, words + i * h->W, h->W * 8, hipMemcpyHostToDevice, h->stream));
        if (h->cfg.symmetry) {
            static const uint64_t one = 1;
            HIP_TRY(hipMemcpyAsync(h->scratch + h->W, &one, 8, hipMemcpyHostToDevice, h->stream));
        }
        int rc = zero_ctl(h, 2);
        if (rc) return rc;
        KmcArgs a = base_args(h, 2);
        a.fin = h->scratch;
        a.fin_stride = 1;
        for (int sg = 0; sg < KMC_SEGS; ++sg) a.seg_count[sg] = sg == 0 ? 1 : 0;
        a.inv_mask = mask & 15u;
        if ((rc = launch_inv(h, a, 1))) return rc;
        if ((rc = read_ctl(h, 2))) return rc;
        uint32_t bits = 0;
        for (int k = 0; k < 4; ++k)
            if (h->ctl_host->viol_count[k]) bits |= 1u << k;
        violated[i] = bits;
    }
    return KMC_OK;
}

int kmc_witness(kmc_handle* h, uint64_t* words) {
    if (!h || !words) return fail(KMC_E_ARG, "null argument");
    if (!h->have_witness) return fail(KMC_E_STATE, "no witness recorded");
    for (int k = 0; k < h->W; ++k) words[k] = h->witness[k];
    return KMC_OK;
}

}  // extern "C"

// Looks fp up in the device table from the host (a few 8-byte reads); returns the slot.
static int table_lookup(kmc_handle* h, uint64_t fp, uint64_t* slot) {
    uint64_t i = kmc_slot_of(fp, h->table_cap);
    for (uint64_t probes = 0; probes < h->table_cap; ++probes) {
        uint64_t v = 0;
        HIP_TRY(hipMemcpy(&v, h->table + i * h->stride_words(), 8, hipMemcpyDeviceToHost));
        // (with wide slots two distinct states may carry this fingerprint; the first one is reported — the check word
        // needs the state, which the callers of this lookup do not have)
        if (v == fp) {
            *slot = i;
            return KMC_OK;
        }
        if (v == 0) break;
        i = kmc_slot_next(i, h->table_cap);
    }
    return fail(KMC_E_STATE, "fingerprint %016llx not in table", (unsigned long long)fp);
}

extern "C" {

// FPSet.contains analogue: is this packed state's fingerprint in the seen-set of the last run?
int kmc_contains(kmc_handle* h, const uint64_t* words, int32_t* present) {
    if (!h || !words || !present) return fail(KMC_E_ARG, "null argument");
    if (!h->table) return fail(KMC_E_STATE, "host-only handle");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    uint64_t slot = 0;
    uint64_t rep_words[KMC_MAXW];
    if (h->cfg.symmetry) {   // the table holds one state per orbit: ask for this state's representative
        kmc_canonical_state(h, words, rep_words, nullptr);
        words = rep_words;
    }
    const int rc = table_lookup(h, kmc_fingerprint_of(h, words), &slot);
    *present = rc == KMC_OK;
    g_err.clear();
    return KMC_OK;
}

int kmc_pred_of(kmc_handle* h, uint64_t fp, uint64_t* pred, int32_t* found) {
    if (!h || !pred || !found) return fail(KMC_E_ARG, "null argument");
    if (!h->pred) return fail(KMC_E_STATE, "kmc_pred_of needs keep_trace=1");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    uint64_t slot = 0;
    *found = table_lookup(h, fp, &slot) == KMC_OK;
    g_err.clear();
    *pred = 0;
    if (*found) HIP_TRY(hipMemcpy(pred, h->pred + slot * h->pred_stride(), 8, hipMemcpyDeviceToHost));
    return KMC_OK;
}

int32_t kmc_owner_of(uint64_t fp, int32_t n_shards) {
    if (n_shards < 1 || n_shards > KMC_MAX_SHARDS) return -1;
    return (int32_t)kmc_owner(fp, (uint32_t)n_shards);
}

int kmc_init_state(kmc_handle* h, uint64_t* words) {
    if (!h || !words) return fail(KMC_E_ARG, "null argument");
    if (h->init_words.empty()) return fail(KMC_E_STATE, "no run has started on this handle");
    for (int k = 0; k < h->W; ++k) words[k] = h->init_words[k];
    return KMC_OK;
}

int kmc_trace(kmc_handle* h, uint8_t* canon_states, int32_t* kinds, uint64_t cap, uint64_t* n_out) {
    if (!h || !n_out) return fail(KMC_E_ARG, "null argument");
    if (!h->pred) return fail(KMC_E_STATE, "kmc_trace needs keep_trace=1");
    if (!h->have_witness) return fail(KMC_E_STATE, "no violation witness recorded");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->xstream) HIP_TRY(hipStreamSynchronize(h->xstream));   // (a pipelined level's inserts write the table on that stream)
    // 1. walk predecessor fingerprints back to the initial state (pred == 0)
    std::vector<uint64_t> chain;
    uint64_t fp = h->res.violation_fp;
    if (h->witness_outside) {  // not in the table: the chain starts at the parent it was generated from
        chain.push_back(fp);
        fp = h->witness_parent_fp;
    }
    for (uint64_t guard = 0; guard < (1u << 20); ++guard) {
        chain.push_back(fp);
        uint64_t slot = 0;
        int rc = table_lookup(h, fp, &slot);
        if (rc) return rc;
        uint64_t p = 0;
        HIP_TRY(hipMemcpy(&p, h->pred + slot * h->pred_stride(), 8, hipMemcpyDeviceToHost));
        if (p == 0) break;
        fp = p;
    }
    // 2. replay forward from Init, picking at each step the successor with the next fingerprint
    const uint64_t n = chain.size();
    *n_out = n;
    const uint64_t cb = kmc_canon_bytes(h);
    std::vector<uint64_t> cur = h->init_words;
    std::vector<uint64_t> succ(h->enum_cap * (h->W + 2));
    if (kmc_fingerprint_of(h, cur.data()) != chain[n - 1]) return fail(KMC_E_STATE, "trace does not start at Init");
    for (uint64_t step = 0; step < n; ++step) {
        if (step < cap) {
            if (canon_states) kmc_unpack_state(h, cur.data(), canon_states + step * cb);
        }
        if (step + 1 == n) break;
        const uint64_t want = chain[n - 2 - step];
        uint64_t ns = 0;
        int rc = kmc_successors(h, cur.data(), succ.data(), h->enum_cap, &ns);
        if (rc) return rc;
        bool found = false;
        for (uint64_t i = 0; i < ns && i < h->enum_cap; ++i) {
            const uint64_t* rec = &succ[i * (h->W + 2)];
            if (rec[h->W] == want) {
                cur.assign(rec, rec + h->W);
                if (step + 1 < cap && kinds) kinds[step + 1] = (int32_t)rec[h->W + 1];
                found = true;
                break;
            }
        }
        if (!found) return fail(KMC_E_STATE, "trace replay lost the path at step %llu", (unsigned long long)step);
    }
    if (kinds && cap) kinds[0] = -1;
    return KMC_OK;
}


}  // extern "C"
