// kmc_engine_step.cpp — checkpoint / recover and the level-step interface (kmc_step_*) of one shard.
#include "kmc_engine_internal.h"
using namespace kmc_engine;

// ---- checkpoint / recover (TLC -checkpoint / -recover [TLC-recall]) --------------------------
// File: header, kmc_result, level sizes, segment sizes, then the fingerprint table (and the
// predecessor table when traces are kept) and the current frontier's planes, segment by segment.
namespace {
struct CkptHeader {
    char magic[8];          // "KMCCKPT7" (4: before round 5 changed the representative at four replicas; before canon_form.  5: before
                            // round 6 changed what a stored fingerprint MEANS — its home slot, kmc_slot_of, and for states of eight
                            // words and more the fingerprint itself, kmc_fingerprint: an older table would be searched in the wrong
                            // places and the resumed search would claim its states a second time.  6: before a run that keeps traces
                            // on 64-bit entries stored the predecessors in the slots, has_pred = 2)
    kmc_config cfg;         // pointers inside are not meaningful in the file
    uint64_t table_cap, fcap, seg_cap, level, n_cur, n_levels, w;
    uint64_t has_pred;      // 0: no predecessors; 1: a table of their own follows the fingerprints; 2: 16-byte slots of fingerprint +
                            // predecessor (kmc_handle::paired) — what the handle that loads the file must have too
    uint64_t layout_form;   // KmcLayout::rm of the packed states in the file (0 tight, 1 / 2 replica-major): the same constants
                            // can be packed in more than one way (KMC_LAYOUT), often into the same number of words
    uint64_t canon_form;    // kmc_config.symmetry: WHICH image of an orbit the stored states are — KMC_SYMM_UNROLLED_MAX of the build
                            // that wrote the file (up to that many replicas the smallest of all images, beyond it the smallest
                            // sorted one).  Round 5 moved it from 4 to 3 under an unchanged magic: a four-replica checkpoint of the
                            // older revision loaded cleanly and the resumed search claimed its orbits a second time
};
uint64_t pred_form(const kmc_handle* h) { return !h->pred ? 0u : h->paired ? 2u : 1u; }
bool wr(FILE* f, const void* p, size_t n) { return fwrite(p, 1, n, f) == n; }
bool rd(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n; }
// device <-> file through a bounded pinned staging buffer
int dev_to_file(FILE* f, const u64* dev, uint64_t words) {
    const uint64_t chunk = 1ull << 24;  // 128 MiB
    std::vector<uint64_t> buf(words < chunk ? words : chunk);
    for (uint64_t at = 0; at < words; at += chunk) {
        const uint64_t n = words - at < chunk ? words - at : chunk;
        HIP_TRY(hipMemcpy(buf.data(), dev + at, n * 8, hipMemcpyDeviceToHost));
        if (!wr(f, buf.data(), n * 8)) return fail(KMC_E_STATE, "checkpoint: short write");
    }
    return KMC_OK;
}
int file_to_dev(FILE* f, u64* dev, uint64_t words) {
    const uint64_t chunk = 1ull << 24;
    std::vector<uint64_t> buf(words < chunk ? words : chunk);
    for (uint64_t at = 0; at < words; at += chunk) {
        const uint64_t n = words - at < chunk ? words - at : chunk;
        if (!rd(f, buf.data(), n * 8)) return fail(KMC_E_STATE, "checkpoint: short read");
        HIP_TRY(hipMemcpy(dev + at, buf.data(), n * 8, hipMemcpyHostToDevice));
    }
    return KMC_OK;
}
}  // namespace

extern "C" {

int kmc_checkpoint_save(kmc_handle* h, const char* path) {
    if (!h || !path) return fail(KMC_E_ARG, "null argument");
    if (!h->table) return fail(KMC_E_STATE, "checkpoints are for device handles");
    // a shard of a multi-GPU search (level-step interface) saves its own table / frontier between kmc_step_finish and
    // the next kmc_step_expand; the driver keeps the global counters (sharded.py) and sets the verdict first
    if (h->cfg.n_shards != 1 && (!h->stepping || h->step_expanded))
        return fail(KMC_E_STATE, "a shard is checkpointed between kmc_step_finish and the next kmc_step_expand");
    if (h->xstream) HIP_TRY(hipStreamSynchronize(h->xstream));
    if (h->levels.empty()) return fail(KMC_E_STATE, "nothing to checkpoint: run first");
    // Only a level boundary is a consistent state: after a stop inside a level (invariant, deadlock, table / frontier
    // full) the table already holds the fingerprints of the rolled-back or partial level while the frontier is still
    // its parent — a search resumed from that would find every successor "seen" and end with states missing.
    if (h->res.verdict != KMC_V_LEVEL_LIMIT)
        return fail(KMC_E_STATE, "a checkpoint can only be taken at a level boundary: after a run that stopped at max_levels "
                                 "(verdict level_limit); this run ended with verdict %d", h->res.verdict);
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    FILE* f = fopen(path, "wb");
    if (!f) return fail(KMC_E_ARG, "cannot open %s for writing", path);
    CkptHeader hd{};
    memcpy(hd.magic, "KMCCKPT7", 8);
    hd.cfg = h->cfg;
    hd.cfg.cache_dir = nullptr;
    hd.table_cap = h->table_cap; hd.fcap = h->fcap; hd.seg_cap = h->seg_cap; hd.level = h->level;
    hd.n_cur = h->n_cur; hd.n_levels = h->levels.size(); hd.w = h->W; hd.has_pred = pred_form(h);
    hd.layout_form = (uint64_t)h->lay.rm;
    hd.canon_form = (uint64_t)KMC_SYMM_UNROLLED_MAX;
    int rc = KMC_OK;
    bool ok = wr(f, &hd, sizeof hd) && wr(f, &h->res, sizeof h->res) && wr(f, h->levels.data(), h->levels.size() * 8) &&
              wr(f, h->seg_n, sizeof h->seg_n) && wr(f, h->init_words.data(), h->W * 8);
    if (!ok) rc = fail(KMC_E_STATE, "checkpoint: short write");
    if (!rc) rc = dev_to_file(f, h->table, h->table_cap * h->stride_words());
    if (!rc && h->pred && !h->paired) rc = dev_to_file(f, h->pred, h->table_cap);
    for (int sg = 0; sg < KMC_SEGS && !rc; ++sg)
        for (int k = 0; k < h->planes && !rc; ++k)
            if (h->seg_n[sg])
                rc = dev_to_file(f, h->frontier[h->cur] + (uint64_t)k * h->fcap + (uint64_t)sg * h->seg_cap, h->seg_n[sg]);
    fclose(f);
    return rc;
}

int kmc_checkpoint_load(kmc_handle* h, const char* path) {
    if (!h || !path) return fail(KMC_E_ARG, "null argument");
    if (!h->table) return fail(KMC_E_STATE, "checkpoints are for device handles");
    HIP_TRY(hipSetDevice(h->cfg.device));
    FILE* f = fopen(path, "rb");
    if (!f) return fail(KMC_E_ARG, "cannot open %s", path);
    CkptHeader hd{};
    int rc = KMC_OK;
    if (!rd(f, &hd, sizeof hd) || memcmp(hd.magic, "KMCCKPT7", 8) != 0)
        rc = fail(KMC_E_ARG, "%s is not a checkpoint of this version", path);
    const kmc_config& a = hd.cfg;
    const kmc_config& b = h->cfg;
    if (!rc && (a.model != b.model || a.n_replicas != b.n_replicas || a.log_size != b.log_size ||
                a.max_records != b.max_records || a.max_leader_epoch != b.max_leader_epoch ||
                a.n_log_records != b.n_log_records || a.max_id != b.max_id || a.hash_seed != b.hash_seed ||
                a.n_shards != b.n_shards || a.shard_id != b.shard_id || hd.w != (uint64_t)h->W ||
                hd.layout_form != (uint64_t)h->lay.rm || (a.wide_fingerprint != 0) != (b.wide_fingerprint != 0) ||
                (a.symmetry != 0) != (b.symmetry != 0) || (b.symmetry && hd.canon_form != (uint64_t)KMC_SYMM_UNROLLED_MAX)))
        rc = fail(KMC_E_ARG, "checkpoint was taken for a different model / constants / hash seed / shard / fingerprint width / "
                             "state layout / symmetry setting / orbit representative");
    if (!rc && (hd.table_cap != h->table_cap || hd.fcap != h->fcap || hd.seg_cap != h->seg_cap ||
                hd.has_pred != pred_form(h)))
        rc = fail(KMC_E_ARG, "checkpoint capacities differ: open the handle with table_capacity=%llu frontier_capacity=%llu keep_trace=%d",
                  (unsigned long long)hd.table_cap, (unsigned long long)hd.fcap, (int)(hd.has_pred != 0));
    // the file is not trusted: every size is checked against the handle before it sizes a buffer or a device copy
    if (!rc && (hd.n_levels == 0 || hd.n_levels > 4096 || hd.level != hd.n_levels || hd.n_cur > hd.fcap))
        rc = fail(KMC_E_ARG, "checkpoint header is inconsistent (levels %llu, level %llu, frontier %llu of %llu)",
                  (unsigned long long)hd.n_levels, (unsigned long long)hd.level, (unsigned long long)hd.n_cur,
                  (unsigned long long)hd.fcap);
    // (as do_begin: a stepped search that stopped between kmc_step_expand and kmc_step_finish may still have a pipelined
    // level's transfer and insert in flight on the second stream — they must not land in the restored table — and its records
    // are still booked for a conservation check that belongs to the abandoned level)
    if (!rc && h->xstream && hipStreamSynchronize(h->xstream) != hipSuccess) rc = fail(KMC_E_DEVICE, "stream sync failed");
    if (!rc) h->inserted_level = 0;
    if (!rc) rc = reset_run(h);
    if (!rc && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(KMC_E_DEVICE, "stream sync failed");
    if (!rc) {
        kmc_result saved{};
        std::vector<uint64_t> lv(hd.n_levels);
        uint64_t segs[KMC_SEGS];
        std::vector<uint64_t> init(h->W);
        bool ok = rd(f, &saved, sizeof saved) && rd(f, lv.data(), hd.n_levels * 8) && rd(f, segs, sizeof segs) &&
                  rd(f, init.data(), h->W * 8);
        if (!ok) rc = fail(KMC_E_STATE, "checkpoint: short read");
        uint64_t seg_sum = 0, lv_sum = 0;
        for (int sg = 0; sg < KMC_SEGS && !rc; ++sg) {
            if (segs[sg] > h->seg_cap) rc = fail(KMC_E_ARG, "checkpoint: segment %d holds %llu states, capacity %llu", sg,
                                                 (unsigned long long)segs[sg], (unsigned long long)h->seg_cap);
            seg_sum += segs[sg];
        }
        for (uint64_t x : lv) lv_sum += x;
        // (under symmetry the level sizes and `distinct` are the weighted numbers; the stored states are orbit_representatives)
        if (!rc && (seg_sum != hd.n_cur || (!h->cfg.symmetry && lv.back() != hd.n_cur) || lv_sum != saved.distinct ||
                    saved.orbit_representatives > h->table_cap || saved.orbit_representatives > saved.distinct ||
                    saved.verdict != KMC_V_LEVEL_LIMIT || saved.state_words != (uint64_t)h->W ||
                    saved.table_capacity != h->table_cap || saved.frontier_capacity != h->fcap))
            rc = fail(KMC_E_ARG, "checkpoint body is inconsistent with its header / this handle");
        if (!rc && kmc_fingerprint_of(h, init.data()) == 0) rc = fail(KMC_E_ARG, "checkpoint: bad initial state");
        if (!rc) {
            h->res = saved;
            h->levels = lv;
            memcpy(h->seg_n, segs, sizeof segs);
            h->init_words = init;
        }
    }
    if (!rc) rc = file_to_dev(f, h->table, h->table_cap * h->stride_words());
    if (!rc && h->pred && !h->paired) rc = file_to_dev(f, h->pred, h->table_cap);
    h->cur = 0;
    for (int sg = 0; sg < KMC_SEGS && !rc; ++sg)
        for (int k = 0; k < h->planes && !rc; ++k)
            if (h->seg_n[sg])
                rc = file_to_dev(f, h->frontier[0] + (uint64_t)k * h->fcap + (uint64_t)sg * h->seg_cap, h->seg_n[sg]);
    fclose(f);
    if (rc) return rc;
    h->level = hd.level;
    h->n_cur = hd.n_cur;
    h->restored = true;
    return KMC_OK;
}

// ---- level-step interface ---------------------------------------------------------------
int kmc_step_begin(kmc_handle* h) {
    if (!h) return fail(KMC_E_ARG, "null handle");
    if (!h->table) return fail(KMC_E_STATE, "host-only handle (device = -1) cannot run");
    HIP_TRY(hipSetDevice(h->cfg.device));
    int rc = do_begin(h);
    h->stepping = true;
    h->step_expanded = false;
    return rc;
}

int kmc_step_expand(kmc_handle* h, uint64_t* send_counts /* [KMC_MAX_SHARDS][KMC_SEND_SUBS] */) {
    if (!h || !h->stepping) return fail(KMC_E_STATE, "kmc_step_begin first");
    // (one shard buckets nothing: every successor is its own, so it may run without a send area)
    if (!h->send && h->cfg.n_shards > 1)
        return fail(KMC_E_STATE, "no send area: open with n_shards > 1 or call kmc_step_set_send_buffer");
    HIP_TRY(hipSetDevice(h->cfg.device));
    const int slot = (int)(h->level & 1);
    int rc = zero_ctl(h, slot);
    if (rc) return rc;
    KmcArgs a = base_args(h, slot);
    a.fin = h->frontier[h->cur];
    a.fout = h->frontier[h->cur ^ 1];
    a.send = h->send;
    a.send_cap = h->send_cap;
    if ((rc = ensure_mode(h, KMC_MODE_SHARDED))) return rc;   // (a cold cache compiles here, outside the timed events)
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    if (h->n_cur) {
        if ((rc = launch_expand(h, KMC_MODE_SHARDED, a, expand_grid(h, h->n_cur)))) return rc;
    }
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    if ((rc = read_ctl(h, slot))) return rc;
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->res.seconds_expand += 1e-3 * ms;
    h->res.expand_launches++;
    h->step_expand_ms += ms;
    for (int d = 0; d < KMC_MAX_SHARDS; ++d)
        for (int sb = 0; sb < KMC_SEGS; ++sb) {
            uint64_t c = h->ctl_host->send_count[d][sb].v;
            h->last_send_counts[d * KMC_SEGS + sb] = c < h->send_cap ? c : h->send_cap;
            if (send_counts) send_counts[d * KMC_SEGS + sb] = h->last_send_counts[d * KMC_SEGS + sb];
        }
    h->xcounts_valid = false;
    h->step_expanded = true;
    return KMC_OK;
}

int kmc_step_set_send_buffer(kmc_handle* h, void* dev_ptr, uint64_t records_per_sub_buffer) {
    const uint64_t records_per_destination = records_per_sub_buffer;
    if (!h || !dev_ptr || records_per_destination == 0) return fail(KMC_E_ARG, "bad send buffer");
    if (h->send && h->send_owned) (void)hipFree(h->send);  // n_shards == 1 is allowed: one destination, itself
    h->send = (u64*)dev_ptr;
    h->send_cap = records_per_destination;
    h->send_owned = false;
    return KMC_OK;
}

int kmc_step_send_buffer(kmc_handle* h, int32_t dst, int32_t sub, void** dev_ptr, uint64_t* record_words) {
    if (!h || !h->send || dst < 0 || dst >= h->cfg.n_shards || sub < 0 || sub >= KMC_SEGS)
        return fail(KMC_E_ARG, "bad destination / sub-buffer");
    *dev_ptr = h->send + ((uint64_t)dst * KMC_SEGS + sub) * h->send_cap * h->rec_words;
    *record_words = h->rec_words;
    return KMC_OK;
}

int kmc_step_insert(kmc_handle* h, const void* dev_records, uint64_t n_records) {
    if (!h || !h->stepping || !h->step_expanded) return fail(KMC_E_STATE, "kmc_step_expand first");
    if (n_records == 0) return KMC_OK;
    HIP_TRY(hipSetDevice(h->cfg.device));
    const int slot = (int)(h->level & 1);
    KmcArgs a = base_args(h, slot);
    a.recv = (const u64*)dev_records;
    a.n_in = n_records;
    h->inserted_level += n_records;
    a.fout = h->frontier[h->cur ^ 1];
    uint64_t blocks = (n_records + KMC_BLOCK - 1) / KMC_BLOCK;
    const uint64_t maxb = (uint64_t)h->n_cus * 8;
    if (blocks > maxb) blocks = maxb;
    return launch(h, h->f_insert, a, (unsigned)blocks);
}

int kmc_step_finish(kmc_handle* h, kmc_level_info* info) {
    if (!h || !h->stepping || !h->step_expanded) return fail(KMC_E_STATE, "kmc_step_expand first");
    HIP_TRY(hipSetDevice(h->cfg.device));
    const int slot = (int)(h->level & 1);
    if (h->xstream) HIP_TRY(hipStreamSynchronize(h->xstream));   // a pipelined level's last transfer and insert
    int rc = read_ctl(h, slot);
    if (rc) return rc;
    const KmcLevelCtl c = *h->ctl_host;
    if (c.err & KMC_ERR_CHECK_WORD)
        return fail(KMC_E_DEVICE, "wide fingerprints: a claimed slot's check word did not appear (level %llu)", (unsigned long long)h->level);
    // the level's kernels on this shard: one k_expand (local successors probed at once, remote ones bucketed — both enter
    // the sink) and the k_insert launches over what the other shards sent
    if ((rc = check_conservation(h, c, h->inserted_level))) return rc;
    h->inserted_level = 0;
    const int nxt = h->cur ^ 1;
    uint64_t new_seg[KMC_SEGS];
    const uint64_t produced = produced_segments(h, c, new_seg);
    kmc_result& r = h->res;
    // kmc_config.symmetry: every count of this shard is weighed as book_level / absorb weigh kmc_run's — N! x the stored
    // states' count less the summed deficits of their orbits (KmcLevelCtl::corr_*).  A state is weighed where it is CLAIMED
    // (its owner: corr_won of k_expand's local path or of k_insert), an expansion where it is EXPANDED (this shard), so the
    // sums over the shards are the plain search's numbers.
    uint64_t gen_w[KMC_MAX_KINDS];
    for (int k = 0; k < KMC_MAX_KINDS; ++k) {
        gen_w[k] = weighted(h, c.generated[k], c.corr_gen[k]);
        r.generated += gen_w[k];
        r.action_generated[k] += gen_w[k];
    }
    r.generated_repeats += weighted(h, c.repeats, c.corr_repeats);
    const uint64_t dead_w = weighted(h, c.deadlock_count, c.corr_dead);
    r.deadlock_states += dead_w;
    const uint64_t produced_w = weighted(h, produced, c.corr_won);
    note_level(h, c, h->n_cur, produced, h->step_expand_ms);
    h->step_expand_ms = 0;
    h->cur = nxt;
    h->n_cur = produced;
    for (int sg = 0; sg < KMC_SEGS; ++sg) { h->prev_seg_n[sg] = h->seg_n[sg]; h->seg_n[sg] = new_seg[sg]; }
    h->level++;
    if (produced) r.depth = h->level;
    r.distinct += produced_w;
    r.orbit_representatives += produced;
    h->levels.push_back(produced_w);
    h->step_expanded = false;
    r.seconds_total = now_s() - h->t_start;
    if (info) {
        memset(info, 0, sizeof *info);
        info->depth = h->level;
        info->new_states = produced_w;   // (symmetry: the states of the level as the plain search counts them; 0 iff none stored)
        info->generated_total = r.generated;
        info->distinct_total = r.distinct;
        info->seconds = r.seconds_total;
        for (int k = 0; k < KMC_MAX_KINDS; ++k) info->generated_level[k] = gen_w[k];
        for (int k = 0; k < 4; ++k) {
            info->violation_count[k] = weighted(h, c.viol_count[k], c.corr_viol[k]);
            info->violation_fp[k] = c.viol_count[k] ? ~c.viol_fp_inv[k] : 0;
        }
        for (int k = 0; k < 4; ++k) {
            info->outside_violation_count[k] = c.oviol_count[k];
            info->outside_violation_fp[k] = c.oviol_count[k] ? ~c.oviol_fp_inv[k] : 0;
        }
        info->deadlocks_level = dead_w;
        info->send_filtered = c.send_filtered;
        info->error_flags = c.err;
    }
    return rc;
}


}  // extern "C"
