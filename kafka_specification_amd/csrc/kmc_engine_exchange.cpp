// kmc_engine_exchange.cpp — the per-level exchange under the ABI: RCCL bound with dlopen, the plan, one-shot and pipelined levels, logical shards.
#include "kmc_engine_internal.h"
using namespace kmc_engine;

// ---- the per-level exchange under the ABI ---------------------------------------
// After kmc_step_expand every shard holds, per (destination, sub-buffer), a dense run of records in its
// send area.  One level's exchange is
//   (1) an all-gather of one small row per rank: its KMC_SEGS send counts per destination and the caller's
//       statistics vector (the statistics of the PREVIOUS expansion ride along: one collective decides
//       termination and verdicts identically on every rank) — one stream synchronisation, because the host
//       must know the counts to post the receives;
//   (2) grouped ncclSend / ncclRecv of every non-empty (peer, sub-buffer) run, straight from the send area
//       into one contiguous receive area, on the engine's stream; and
//   (3) ONE k_insert over what arrived, queued behind the receives on the same stream — no host wait.
// The plan (who sends how many words from which offset, where each run lands) is a pure function of the
// count matrix, shared by the RCCL transport and by the in-process transport that moves the runs with
// device-to-device copies between P logical shards on one GPU (RCCL refuses two ranks on one device).
namespace {

struct KmcRccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

// librccl is bound at run time: libkmc.so must load on a box without RCCL (single-GPU use, the CPU-side
// ABI tests), and inside a PyTorch process the name resolves to the copy the wheel has already loaded
// (same SONAME), so both sides of the process talk to one RCCL.
std::string g_rccl_error;   // why librccl could not be bound (dlerror() is read ONCE, where it happens: a second call returns NULL)

void rccl_bind(KmcRccl& r) {
    const char* names[] = {getenv("KMC_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        if ((r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        const char* e = dlerror();
        g_rccl_error += std::string(g_rccl_error.empty() ? "" : "; ") + n + ": " + (e ? e : "dlopen failed");
    }
    if (!r.lib) return;
#define KMC_SYM(field, name)                                                     \
    r.field = (decltype(r.field))dlsym(r.lib, name);                             \
    if (!r.field) { const char* e = dlerror(); g_rccl_error = std::string(name) + ": " + (e ? e : "symbol not found"); r.lib = nullptr; return; }
    KMC_SYM(GetUniqueId, "ncclGetUniqueId")
    KMC_SYM(CommInitRank, "ncclCommInitRank")
    KMC_SYM(CommDestroy, "ncclCommDestroy")
    KMC_SYM(AllGather, "ncclAllGather")
    KMC_SYM(Send, "ncclSend")
    KMC_SYM(Recv, "ncclRecv")
    KMC_SYM(GroupStart, "ncclGroupStart")
    KMC_SYM(GroupEnd, "ncclGroupEnd")
    KMC_SYM(GetErrorString, "ncclGetErrorString")
#undef KMC_SYM
}

KmcRccl* rccl() {   // bound once per process, also when several host threads arrive at the same time
    static KmcRccl r;
    static std::once_flag once;
    std::call_once(once, rccl_bind, std::ref(r));
    return r.lib ? &r : nullptr;
}

#define NCCL_TRY(expr)                                                                                          \
    do {                                                                                                        \
        ncclResult_t e_ = (expr);                                                                               \
        if (e_ != ncclSuccess) return fail(KMC_E_DEVICE, "%s failed: %s", #expr, rccl()->GetErrorString(e_));  \
    } while (0)

// One message of a level's plan: `words` 64-bit words at `offset_words` of the send area (a send) or of the
// receive area (a receive), exchanged with `peer`.
struct KmcXfer {
    uint64_t peer, offset_words, words;
};
// A single message stays below 1 GiB: this RCCL build corrupted all-to-all messages above 2 GiB
// (tools/a2a_probe.py), so long runs are cut; both sides cut identically.
constexpr uint64_t KMC_XFER_MAX_WORDS = 1ull << 27;

// counts[(s * P + d) * KMC_SEGS + sub] = records shard s sends to shard d from its sub-buffer `sub`.
// Sends of `me` in (destination, sub-buffer) order; receives in (source, sub-buffer) order — RCCL matches the
// messages of a pair in posting order, and both lists enumerate a pair's runs in sub-buffer order.
void plan_level(const uint64_t* counts, int P, int me, uint64_t send_cap, uint64_t rec_words,
                std::vector<KmcXfer>* sends, std::vector<KmcXfer>* recvs, uint64_t* recv_records) {
    sends->clear();
    recvs->clear();
    auto cut = [](std::vector<KmcXfer>* out, uint64_t peer, uint64_t off, uint64_t words) {
        while (words) {
            const uint64_t n = words < KMC_XFER_MAX_WORDS ? words : KMC_XFER_MAX_WORDS;
            out->push_back(KmcXfer{peer, off, n});
            off += n;
            words -= n;
        }
    };
    for (int d = 0; d < P; ++d) {
        if (d == me) continue;
        for (int sb = 0; sb < KMC_SEGS; ++sb) {
            const uint64_t n = counts[((uint64_t)me * P + d) * KMC_SEGS + sb];
            if (n) cut(sends, (uint64_t)d, ((uint64_t)d * KMC_SEGS + sb) * send_cap * rec_words, n * rec_words);
        }
    }
    uint64_t at = 0;  // records received so far: the receive area is filled densely, source by source
    for (int s2 = 0; s2 < P; ++s2) {
        if (s2 == me) continue;
        for (int sb = 0; sb < KMC_SEGS; ++sb) {
            const uint64_t n = counts[((uint64_t)s2 * P + me) * KMC_SEGS + sb];
            if (n) cut(recvs, (uint64_t)s2, at * rec_words, n * rec_words);
            at += n;
        }
    }
    *recv_records = at;
}

int ensure_exchange_buffers(kmc_handle* h) {
    const int P = h->cfg.n_shards;
    if (!h->send || !h->send_owned)
        return fail(KMC_E_STATE, "the exchange under the ABI needs the engine-owned send area (n_shards > 1, no "
                                 "kmc_step_set_send_buffer)");
    if (!h->recv) {
        // worst case: every other shard fills all its sub-buffers for this one
        h->recv_cap = (uint64_t)(P - 1) * KMC_SEGS * h->send_cap;
        if (hipMalloc(&h->recv, h->recv_cap * h->rec_words * 8ull) != hipSuccess) {
            h->recv = nullptr;
            return fail(KMC_E_NOMEM, "cannot allocate the receive area (%llu records)", (unsigned long long)h->recv_cap);
        }
    }
    const size_t row = (size_t)P * KMC_SEGS + KMC_EXCHANGE_STATS;
    if (!h->xrow_dev) HIP_TRY(hipMalloc(&h->xrow_dev, (size_t)(P + 1) * row * 8));
    if (!h->xrow_host) HIP_TRY(hipHostMalloc(&h->xrow_host, (size_t)(P + 1) * row * 8));
    return KMC_OK;
}

int insert_received(kmc_handle* h, uint64_t n_records, hipStream_t stream = nullptr) {
    if (n_records == 0) return KMC_OK;
    const int slot = (int)(h->level & 1);
    KmcArgs a = base_args(h, slot);
    a.recv = h->recv;
    a.n_in = n_records;
    h->inserted_level += n_records;
    a.fout = h->frontier[h->cur ^ 1];
    uint64_t blocks = (n_records + KMC_BLOCK - 1) / KMC_BLOCK;
    const uint64_t maxb = (uint64_t)h->n_cus * 8;
    if (blocks > maxb) blocks = maxb;
    return launch(h, h->f_insert, a, (unsigned)blocks, stream);
}

}  // namespace

void kmc_engine::comm_release(kmc_handle* h) {
    if (h->comm && rccl()) rccl()->CommDestroy(h->comm);
    h->comm = nullptr;
}

extern "C" {

int kmc_comm_unique_id(uint8_t* id) {
    if (!id) return fail(KMC_E_ARG, "null argument");
    KmcRccl* r = rccl();
    if (!r) return fail(KMC_E_DEVICE, "librccl could not be bound: %s", g_rccl_error.empty() ? "?" : g_rccl_error.c_str());
    ncclUniqueId u;
    NCCL_TRY(r->GetUniqueId(&u));
    static_assert(sizeof u == KMC_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id, &u, sizeof u);
    return KMC_OK;
}

int kmc_comm_init(kmc_handle* h, const uint8_t* id) {
    if (!h || !id) return fail(KMC_E_ARG, "null argument");
    if (!h->table || h->cfg.n_shards < 1) return fail(KMC_E_STATE, "kmc_comm_init needs a device handle");
    KmcRccl* r = rccl();
    if (!r) return fail(KMC_E_DEVICE, "librccl not found (dlopen librccl.so.1)");
    HIP_TRY(hipSetDevice(h->cfg.device));
    comm_release(h);
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    NCCL_TRY(r->CommInitRank(&h->comm, h->cfg.n_shards, u, h->cfg.shard_id));
    if (h->cfg.n_shards > 1) return ensure_exchange_buffers(h);
    return KMC_OK;
}

// Exercises every RCCL entry point the exchange uses on this handle's communicator and stream: an all-gather of
// one row per rank and a grouped send/receive ring (rank r sends a pattern to r+1 and receives from r-1; with one
// rank that is a send to itself).  Verifies what arrived.  A world_size-1 run thereby covers the binding, the
// argument conventions and the stream ordering although a one-shard search has no remote traffic.
int kmc_comm_selftest(kmc_handle* h) {
    if (!h || !h->comm) return fail(KMC_E_STATE, "kmc_comm_init first");
    KmcRccl* r = rccl();
    HIP_TRY(hipSetDevice(h->cfg.device));
    const int P = h->cfg.n_shards, me = h->cfg.shard_id;
    const size_t n = 4096;
    u64* buf = nullptr;
    HIP_TRY(hipMalloc(&buf, (size_t)(2 + P) * n * 8));
    struct Free { u64* p; ~Free() { (void)hipFree(p); } } free_buf{buf};   // also on the error returns below
    std::vector<uint64_t> host((size_t)(2 + P) * n);
    for (size_t i = 0; i < n; ++i) host[i] = ((uint64_t)(me + 1) << 32) | i;
    HIP_TRY(hipMemcpyAsync(buf, host.data(), n * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemsetAsync(buf + n, 0, (size_t)(1 + P) * n * 8, h->stream));
    NCCL_TRY(r->GroupStart());
    NCCL_TRY(r->Send(buf, n, ncclUint64, (me + 1) % P, h->comm, h->stream));
    NCCL_TRY(r->Recv(buf + n, n, ncclUint64, (me + P - 1) % P, h->comm, h->stream));
    NCCL_TRY(r->GroupEnd());
    NCCL_TRY(r->AllGather(buf, buf + 2 * n, n, ncclUint64, h->comm, h->stream));
    HIP_TRY(hipMemcpyAsync(host.data(), buf, host.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const uint64_t from = (uint64_t)((me + P - 1) % P + 1);
    for (size_t i = 0; i < n; ++i) {
        if (host[n + i] != ((from << 32) | i)) return fail(KMC_E_DEVICE, "selftest: send/recv word %zu is wrong", i);
        for (int q = 0; q < P; ++q)
            if (host[(2 + q) * n + i] != (((uint64_t)(q + 1) << 32) | i))
                return fail(KMC_E_DEVICE, "selftest: all-gather word %zu of rank %d is wrong", i, q);
    }
    return KMC_OK;
}

int kmc_exchange_plan(const uint64_t* counts, int32_t n_shards, int32_t me, uint64_t send_cap, uint64_t rec_words,
                      uint64_t* sends, uint64_t* recvs, uint64_t cap, uint64_t* n_sends, uint64_t* n_recvs,
                      uint64_t* recv_records) {
    if (!counts || n_shards < 1 || n_shards > KMC_MAX_SHARDS || me < 0 || me >= n_shards || !n_sends || !n_recvs ||
        !recv_records)
        return fail(KMC_E_ARG, "bad argument");
    std::vector<KmcXfer> sv, rv;
    plan_level(counts, n_shards, me, send_cap, rec_words, &sv, &rv, recv_records);
    *n_sends = sv.size();
    *n_recvs = rv.size();
    for (uint64_t i = 0; i < sv.size() && i < cap && sends; ++i) {
        sends[3 * i] = sv[i].peer; sends[3 * i + 1] = sv[i].offset_words; sends[3 * i + 2] = sv[i].words;
    }
    for (uint64_t i = 0; i < rv.size() && i < cap && recvs; ++i) {
        recvs[3 * i] = rv[i].peer; recvs[3 * i + 1] = rv[i].offset_words; recvs[3 * i + 2] = rv[i].words;
    }
    return KMC_OK;
}

int kmc_step_exchange_counts(kmc_handle* h, const int64_t* stats, int32_t n_stats, int64_t* stats_sum,
                             uint64_t* recv_records) {
    if (!h || !h->stepping || !h->step_expanded) return fail(KMC_E_STATE, "kmc_step_expand first");
    if (!h->comm) return fail(KMC_E_STATE, "kmc_comm_init first");
    if (n_stats < 0 || n_stats > KMC_EXCHANGE_STATS || (n_stats && (!stats || !stats_sum)))
        return fail(KMC_E_ARG, "bad statistics vector");
    KmcRccl* r = rccl();
    HIP_TRY(hipSetDevice(h->cfg.device));
    const int P = h->cfg.n_shards, me = h->cfg.shard_id;
    const size_t row = (size_t)P * KMC_SEGS + KMC_EXCHANGE_STATS;
    int rc = P > 1 ? ensure_exchange_buffers(h) : KMC_OK;
    if (rc) return rc;
    h->xcounts.assign((size_t)P * P * KMC_SEGS, 0);
    if (P == 1) {  // nothing to gather
        for (int k = 0; k < n_stats; ++k) stats_sum[k] = stats[k];
        if (recv_records) *recv_records = 0;
        h->xcounts_valid = true;
        return KMC_OK;
    }
    int64_t* mine = h->xrow_host;
    for (int d = 0; d < P; ++d)
        for (int sb = 0; sb < KMC_SEGS; ++sb)
            mine[d * KMC_SEGS + sb] = d == me ? 0 : (int64_t)h->last_send_counts[d * KMC_SEGS + sb];
    for (int k = 0; k < KMC_EXCHANGE_STATS; ++k) mine[P * KMC_SEGS + k] = k < n_stats ? stats[k] : 0;
    HIP_TRY(hipMemcpyAsync(h->xrow_dev, mine, row * 8, hipMemcpyHostToDevice, h->stream));
    NCCL_TRY(r->AllGather(h->xrow_dev, h->xrow_dev + row, row, ncclInt64, h->comm, h->stream));
    HIP_TRY(hipMemcpyAsync(h->xrow_host + row, h->xrow_dev + row, (size_t)P * row * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int k = 0; k < n_stats; ++k) stats_sum[k] = 0;
    for (int s2 = 0; s2 < P; ++s2) {
        const int64_t* g = h->xrow_host + (size_t)(1 + s2) * row;
        for (int d = 0; d < P; ++d)
            for (int sb = 0; sb < KMC_SEGS; ++sb) {
                const int64_t c = g[d * KMC_SEGS + sb];
                if (c < 0 || (uint64_t)c > h->send_cap)
                    return fail(KMC_E_STATE, "exchange: rank %d announces %lld records for a sub-buffer of %llu", s2,
                                (long long)c, (unsigned long long)h->send_cap);
                h->xcounts[((size_t)s2 * P + d) * KMC_SEGS + sb] = (uint64_t)c;
            }
        for (int k = 0; k < n_stats; ++k) stats_sum[k] += g[P * KMC_SEGS + k];
    }
    std::vector<KmcXfer> sv, rv;
    uint64_t nrec = 0;
    plan_level(h->xcounts.data(), P, me, h->send_cap, (uint64_t)h->rec_words, &sv, &rv, &nrec);
    if (recv_records) *recv_records = nrec;
    h->xcounts_valid = true;
    return KMC_OK;
}

// kmc_step_expand + kmc_step_exchange_counts with one stream synchronisation (round 2 took two, with a host-to-device copy
// of the counts in between): k_expand fills the control block, k_packrow turns its send counters into this shard's row
// of the all-gather on the device, the collective runs behind it on the same stream, and the host reads the gathered
// rows back once.  Its own counts come out of the same rows.
int kmc_step_expand_counts(kmc_handle* h, const int64_t* stats, int32_t n_stats, int64_t* stats_sum, uint64_t* recv_records,
                           uint64_t* send_counts) {
    if (!h || !h->stepping) return fail(KMC_E_STATE, "kmc_step_begin first");
    const int P = h->cfg.n_shards, me = h->cfg.shard_id;
    if (P == 1 || !h->comm) {  // nothing to gather, or no communicator: the two-step path
        int rc = kmc_step_expand(h, send_counts);
        return rc ? rc : kmc_step_exchange_counts(h, stats, n_stats, stats_sum, recv_records);
    }
    if (!h->send) return fail(KMC_E_STATE, "no send area");
    if (n_stats < 0 || n_stats > KMC_EXCHANGE_STATS || (n_stats && (!stats || !stats_sum)))
        return fail(KMC_E_ARG, "bad statistics vector");
    static_assert(KMC_ROW_STATS == KMC_EXCHANGE_STATS, "row layout");
    KmcRccl* r = rccl();
    HIP_TRY(hipSetDevice(h->cfg.device));
    int rc = ensure_exchange_buffers(h);
    if (rc) return rc;
    const int slot = (int)(h->level & 1);
    if ((rc = zero_ctl(h, slot))) return rc;
    KmcArgs a = base_args(h, slot);
    a.fin = h->frontier[h->cur];
    a.fout = h->frontier[h->cur ^ 1];
    a.send = h->send;
    a.send_cap = h->send_cap;
    if ((rc = ensure_mode(h, KMC_MODE_SHARDED))) return rc;   // (a cold cache compiles here, outside the timed events)
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    if (h->n_cur && (rc = launch_expand(h, KMC_MODE_SHARDED, a, expand_grid(h, h->n_cur)))) return rc;
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    const size_t row = (size_t)P * KMC_SEGS + KMC_EXCHANGE_STATS;
    KmcPackArgs pa{};
    pa.ctl = h->ctl + slot;
    pa.row = (long long*)h->xrow_dev;
    pa.send_cap = h->send_cap;
    pa.nshards = (uint32_t)P;
    pa.shard = (uint32_t)me;
    for (int k = 0; k < KMC_EXCHANGE_STATS; ++k) pa.stats[k] = k < n_stats ? stats[k] : 0;
    {
        size_t size = sizeof(pa);
        void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &pa, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
        HIP_TRY(hipModuleLaunchKernel(h->f_packrow, 1, 1, 1, KMC_BLOCK, 1, 1, 0, h->stream, nullptr, config));
    }
    NCCL_TRY(r->AllGather(h->xrow_dev, h->xrow_dev + row, row, ncclInt64, h->comm, h->stream));
    HIP_TRY(hipMemcpyAsync(h->xrow_host + row, h->xrow_dev + row, (size_t)P * row * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));   // the level's only host wait before the payload is posted
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->res.seconds_expand += 1e-3 * ms;
    h->res.expand_launches++;
    h->step_expand_ms += ms;
    h->xcounts.assign((size_t)P * P * KMC_SEGS, 0);
    for (int k = 0; k < n_stats; ++k) stats_sum[k] = 0;
    for (int s2 = 0; s2 < P; ++s2) {
        const int64_t* g = h->xrow_host + (size_t)(1 + s2) * row;
        for (int d = 0; d < P; ++d)
            for (int sb = 0; sb < KMC_SEGS; ++sb) {
                const int64_t c = g[d * KMC_SEGS + sb];
                if (c < 0 || (uint64_t)c > h->send_cap)
                    return fail(KMC_E_STATE, "exchange: rank %d announces %lld records for a sub-buffer of %llu", s2,
                                (long long)c, (unsigned long long)h->send_cap);
                h->xcounts[((size_t)s2 * P + d) * KMC_SEGS + sb] = (uint64_t)c;
                if (s2 == me) {
                    h->last_send_counts[d * KMC_SEGS + sb] = (uint64_t)c;
                    if (send_counts) send_counts[d * KMC_SEGS + sb] = (uint64_t)c;
                }
            }
        for (int k = 0; k < n_stats; ++k) stats_sum[k] += g[P * KMC_SEGS + k];
    }
    std::vector<KmcXfer> sv, rv;
    uint64_t nrec = 0;
    plan_level(h->xcounts.data(), P, me, h->send_cap, (uint64_t)h->rec_words, &sv, &rv, &nrec);
    if (recv_records) *recv_records = nrec;
    h->xcounts_valid = true;
    h->step_expanded = true;
    return KMC_OK;
}

// One BFS level of a shard as a PIPELINE of `parts` parts (2, 4 or 8 groups of the frontier's KMC_SEGS segments): part c is
// expanded into send area c mod 2 on the engine's stream while part c-1's counts are gathered, its records travel and are
// inserted on a second stream — the wire of a level hides behind its own expansion (inserts append to
// the NEXT frontier and to the seen-set with atomics, so they do not disturb the expansion of the current one).  Every part
// costs the host one wait (it must know the counts to post the receives), which is why small levels keep the one-shot path
// (kmc_step_expand_counts + kmc_step_exchange_payload).  The caller's statistics ride with part 0.  Afterwards the level
// stands where kmc_step_exchange_payload leaves it: kmc_step_finish is next.
int kmc_step_level_parts(kmc_handle* h, int32_t parts, const int64_t* stats, int32_t n_stats, int64_t* stats_sum,
                         uint64_t* recv_records) {
    if (!h || !h->stepping) return fail(KMC_E_STATE, "kmc_step_begin first");
    const int P = h->cfg.n_shards, me = h->cfg.shard_id;
    if (P < 2 || !h->comm) return fail(KMC_E_STATE, "a pipelined level needs a communicator (n_shards > 1, kmc_comm_init)");
    if (parts != 2 && parts != 4 && parts != 8) return fail(KMC_E_ARG, "parts must be 2, 4 or 8");
    if (!h->send) return fail(KMC_E_STATE, "no send area");
    if (n_stats < 0 || n_stats > KMC_EXCHANGE_STATS || (n_stats && (!stats || !stats_sum)))
        return fail(KMC_E_ARG, "bad statistics vector");
    const uint64_t half_cap = h->send_cap / 2;
    if (half_cap < 64) return fail(KMC_E_STATE, "send area too small to be split for a pipelined level");
    KmcRccl* r = rccl();
    HIP_TRY(hipSetDevice(h->cfg.device));
    int rc = ensure_exchange_buffers(h);
    if (rc) return rc;
    const size_t row = (size_t)P * KMC_SEGS + KMC_EXCHANGE_STATS;
    if (!h->xstream) {
        HIP_TRY(hipStreamCreateWithFlags(&h->xstream, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            HIP_TRY(hipEventCreateWithFlags(&h->ev_row[i], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&h->ev_xfer[i], hipEventDisableTiming));
            HIP_TRY(hipMalloc(&h->prow_dev[i], (size_t)(P + 1) * row * 8));
            HIP_TRY(hipHostMalloc(&h->prow_host[i], (size_t)(P + 1) * row * 8));
        }
    }
    const int slot = (int)(h->level & 1);
    if ((rc = zero_ctl(h, slot))) return rc;
    const size_t area_words = (size_t)P * KMC_SEGS * half_cap * (size_t)h->rec_words;
    const int per = KMC_SEGS / parts;
    uint64_t total_recv = 0;
    for (int k = 0; k < n_stats; ++k) stats_sum[k] = 0;

    // stage A: part c's expansion and its row, on the engine's stream
    auto stage_a = [&](int c) -> int {
        const int a2 = c & 1;
        if (c >= 2) HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_xfer[a2], 0));   // part c-2 has left this send area
        HIP_TRY(hipMemsetAsync(&(h->ctl + slot)->send_count, 0, sizeof(KmcLevelCtl) - KMC_CTL_LOCAL_BYTES, h->stream));
        KmcArgs a = base_args(h, slot);
        a.fin = h->frontier[h->cur];
        a.fout = h->frontier[h->cur ^ 1];
        a.send = h->send + (size_t)a2 * area_words;
        a.send_cap = half_cap;
        uint64_t n_part = 0;
        for (int sg = 0; sg < KMC_SEGS; ++sg) {
            if (sg / per != c) a.seg_count[sg] = 0;
            n_part += a.seg_count[sg];
        }
        if (!h->ev_chain[2 * c]) {
            HIP_TRY(hipEventCreate(&h->ev_chain[2 * c]));
            HIP_TRY(hipEventCreate(&h->ev_chain[2 * c + 1]));
        }
        HIP_TRY(hipEventRecord(h->ev_chain[2 * c], h->stream));
        int rc2 = KMC_OK;
        if (n_part && (rc2 = launch_expand(h, KMC_MODE_SHARDED, a, expand_grid(h, n_part)))) return rc2;
        HIP_TRY(hipEventRecord(h->ev_chain[2 * c + 1], h->stream));
        KmcPackArgs pa{};
        pa.ctl = h->ctl + slot;
        pa.row = (long long*)h->prow_dev[a2];
        pa.send_cap = half_cap;
        pa.nshards = (uint32_t)P;
        pa.shard = (uint32_t)me;
        for (int k = 0; k < KMC_EXCHANGE_STATS; ++k) pa.stats[k] = (c == 0 && k < n_stats) ? stats[k] : 0;
        size_t size = sizeof(pa);
        void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &pa, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
        HIP_TRY(hipModuleLaunchKernel(h->f_packrow, 1, 1, 1, KMC_BLOCK, 1, 1, 0, h->stream, nullptr, config));
        HIP_TRY(hipEventRecord(h->ev_row[a2], h->stream));
        return KMC_OK;
    };

    if ((rc = stage_a(0))) return rc;
    for (int c = 0; c < parts; ++c) {
        const int a2 = c & 1;
        if (c + 1 < parts && (rc = stage_a(c + 1))) return rc;   // queued BEFORE the host waits for part c's counts
        // stage B: part c's counts, gathered on the second stream
        HIP_TRY(hipStreamWaitEvent(h->xstream, h->ev_row[a2], 0));
        NCCL_TRY(r->AllGather(h->prow_dev[a2], h->prow_dev[a2] + row, row, ncclInt64, h->comm, h->xstream));
        HIP_TRY(hipMemcpyAsync(h->prow_host[a2] + row, h->prow_dev[a2] + row, (size_t)P * row * 8, hipMemcpyDeviceToHost, h->xstream));
        HIP_TRY(hipStreamSynchronize(h->xstream));   // this part's host wait (the previous part's insert is behind it too)
        // stage C: the plan, the transfer and the insert of part c, on the second stream
        h->xcounts.assign((size_t)P * P * KMC_SEGS, 0);
        for (int s2 = 0; s2 < P; ++s2) {
            const int64_t* g = h->prow_host[a2] + (size_t)(1 + s2) * row;
            for (int d = 0; d < P; ++d)
                for (int sb = 0; sb < KMC_SEGS; ++sb) {
                    const int64_t cnt = g[d * KMC_SEGS + sb];
                    if (cnt < 0 || (uint64_t)cnt > half_cap)
                        return fail(KMC_E_STATE, "exchange: rank %d announces %lld records for a sub-buffer of %llu", s2,
                                    (long long)cnt, (unsigned long long)half_cap);
                    h->xcounts[((size_t)s2 * P + d) * KMC_SEGS + sb] = (uint64_t)cnt;
                    if (s2 == me) h->last_send_counts[d * KMC_SEGS + sb] = (uint64_t)cnt;
                }
            if (c == 0)
                for (int k = 0; k < n_stats; ++k) stats_sum[k] += g[P * KMC_SEGS + k];
        }
        std::vector<KmcXfer> sv, rv;
        uint64_t nrec = 0;
        plan_level(h->xcounts.data(), P, me, half_cap, (uint64_t)h->rec_words, &sv, &rv, &nrec);
        if (nrec > h->recv_cap) return fail(KMC_E_STATE, "exchange: %llu records exceed the receive area", (unsigned long long)nrec);
        const u64* area = h->send + (size_t)a2 * area_words;
        if (!sv.empty() || !rv.empty()) {
            NCCL_TRY(r->GroupStart());
            for (const KmcXfer& x : sv) NCCL_TRY(r->Send(area + x.offset_words, x.words, ncclUint64, (int)x.peer, h->comm, h->xstream));
            for (const KmcXfer& x : rv) NCCL_TRY(r->Recv(h->recv + x.offset_words, x.words, ncclUint64, (int)x.peer, h->comm, h->xstream));
            NCCL_TRY(r->GroupEnd());
        }
        HIP_TRY(hipEventRecord(h->ev_xfer[a2], h->xstream));     // the send area may be refilled (part c+2)
        if ((rc = insert_received(h, nrec, h->xstream))) return rc;   // behind the receives; the receive area is reused by
                                                                      // part c+1's transfer, which this stream orders behind it
        total_recv += nrec;
    }
    for (int c = 0; c < parts; ++c) {   // every part's expansion has completed: its row was gathered
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, h->ev_chain[2 * c], h->ev_chain[2 * c + 1]));
        h->res.seconds_expand += 1e-3 * ms;
        h->res.expand_launches++;
        h->step_expand_ms += ms;
    }
    if (recv_records) *recv_records = total_recv;
    h->xcounts_valid = false;
    h->step_expanded = true;
    return KMC_OK;
}

int kmc_step_exchange_payload(kmc_handle* h) {
    if (!h || !h->stepping || !h->step_expanded) return fail(KMC_E_STATE, "kmc_step_expand first");
    if (!h->xcounts_valid) return fail(KMC_E_STATE, "kmc_step_exchange_counts first");
    h->xcounts_valid = false;
    const int P = h->cfg.n_shards, me = h->cfg.shard_id;
    if (P == 1) return KMC_OK;
    KmcRccl* r = rccl();
    HIP_TRY(hipSetDevice(h->cfg.device));
    std::vector<KmcXfer> sv, rv;
    uint64_t nrec = 0;
    plan_level(h->xcounts.data(), P, me, h->send_cap, (uint64_t)h->rec_words, &sv, &rv, &nrec);
    if (nrec > h->recv_cap) return fail(KMC_E_STATE, "exchange: %llu records exceed the receive area", (unsigned long long)nrec);
    if (!sv.empty() || !rv.empty()) {
        NCCL_TRY(r->GroupStart());
        for (const KmcXfer& x : sv) NCCL_TRY(r->Send(h->send + x.offset_words, x.words, ncclUint64, (int)x.peer, h->comm, h->stream));
        for (const KmcXfer& x : rv) NCCL_TRY(r->Recv(h->recv + x.offset_words, x.words, ncclUint64, (int)x.peer, h->comm, h->stream));
        NCCL_TRY(r->GroupEnd());
    }
    return insert_received(h, nrec);  // queued behind the receives on the same stream
}

// The same level step for P logical shards living in ONE process on ONE device (tests, `tlc -gpus P` on a
// single GPU): counts and statistics are combined on the host, the runs move with device-to-device copies,
// every shard then inserts what it received.  stats: [n_shards][n_stats].
int kmc_step_exchange_local(kmc_handle** hs, int32_t n_shards, const int64_t* stats, int32_t n_stats, int64_t* stats_sum) {
    if (!hs || n_shards < 1 || n_shards > KMC_MAX_SHARDS) return fail(KMC_E_ARG, "bad shard list");
    const int P = n_shards;
    for (int s2 = 0; s2 < P; ++s2) {
        kmc_handle* h = hs[s2];
        if (!h || !h->stepping || !h->step_expanded) return fail(KMC_E_STATE, "kmc_step_expand first (shard %d)", s2);
        if (h->cfg.n_shards != P || h->cfg.shard_id != s2) return fail(KMC_E_ARG, "handle %d is not shard %d of %d", s2, s2, P);
        if (h->cfg.device != hs[0]->cfg.device || h->send_cap != hs[0]->send_cap || h->rec_words != hs[0]->rec_words)
            return fail(KMC_E_ARG, "local exchange: shards must share the device and the send geometry");
    }
    HIP_TRY(hipSetDevice(hs[0]->cfg.device));
    std::vector<uint64_t> counts((size_t)P * P * KMC_SEGS, 0);
    for (int s2 = 0; s2 < P; ++s2)
        for (int d = 0; d < P; ++d)
            for (int sb = 0; sb < KMC_SEGS; ++sb)
                counts[((size_t)s2 * P + d) * KMC_SEGS + sb] = d == s2 ? 0 : hs[s2]->last_send_counts[d * KMC_SEGS + sb];
    for (int k = 0; k < n_stats; ++k) {
        stats_sum[k] = 0;
        for (int s2 = 0; s2 < P; ++s2) stats_sum[k] += stats[(size_t)s2 * n_stats + k];
    }
    for (int s2 = 0; s2 < P; ++s2) {
        hs[s2]->xcounts = counts;
        hs[s2]->xcounts_valid = true;
        if (P > 1) {
            int rc = ensure_exchange_buffers(hs[s2]);
            if (rc) return rc;
        }
    }
    return KMC_OK;
}

int kmc_step_deliver_local(kmc_handle** hs, int32_t n_shards) {
    if (!hs || n_shards < 1 || n_shards > KMC_MAX_SHARDS) return fail(KMC_E_ARG, "bad shard list");
    const int P = n_shards;
    for (int s2 = 0; s2 < P; ++s2)
        if (!hs[s2] || !hs[s2]->xcounts_valid) return fail(KMC_E_STATE, "kmc_step_exchange_local first");
    HIP_TRY(hipSetDevice(hs[0]->cfg.device));
    // every shard's k_expand has completed (kmc_step_expand waits for its control block), so the send areas are final
    for (int me = 0; me < P; ++me) {
        kmc_handle* h = hs[me];
        h->xcounts_valid = false;
        std::vector<KmcXfer> sv, rv;
        uint64_t nrec = 0;
        plan_level(h->xcounts.data(), P, me, h->send_cap, (uint64_t)h->rec_words, &sv, &rv, &nrec);
        if (P > 1 && nrec > h->recv_cap) return fail(KMC_E_STATE, "exchange: receive area too small");
        // a receive from `peer` is matched by that peer's sends to `me`, in posting order, cut identically
        std::vector<size_t> cursor(P, 0);
        std::vector<std::vector<KmcXfer>> peer_sends(P);
        for (int q = 0; q < P; ++q) {
            if (q == me) continue;
            std::vector<KmcXfer> qs, qr;
            uint64_t dummy = 0;
            plan_level(h->xcounts.data(), P, q, h->send_cap, (uint64_t)h->rec_words, &qs, &qr, &dummy);
            for (const KmcXfer& x : qs)
                if ((int)x.peer == me) peer_sends[q].push_back(x);
        }
        for (const KmcXfer& x : rv) {
            const int q = (int)x.peer;
            if (cursor[q] >= peer_sends[q].size() || peer_sends[q][cursor[q]].words != x.words)
                return fail(KMC_E_STATE, "exchange plan mismatch between shards %d and %d", q, me);
            const KmcXfer& sx = peer_sends[q][cursor[q]++];
            HIP_TRY(hipMemcpyAsync(h->recv + x.offset_words, hs[q]->send + sx.offset_words, x.words * 8,
                                   hipMemcpyDeviceToDevice, h->stream));
        }
        for (int q = 0; q < P; ++q)
            if (q != me && cursor[q] != peer_sends[q].size())
                return fail(KMC_E_STATE, "exchange plan mismatch: unmatched sends from shard %d to %d", q, me);
        int rc = insert_received(h, nrec);
        if (rc) return rc;
    }
    return KMC_OK;
}

// The invariants of the CURRENT frontier without expanding it: what kmc_run does for the last level under
// max_levels (every state is normally checked when it is expanded; an unexpanded last level would otherwise go
// unchecked).  Fills violation_count / violation_fp only.
int kmc_step_check_frontier(kmc_handle* h, kmc_level_info* info) {
    if (!h || !h->stepping || !info) return fail(KMC_E_STATE, "kmc_step_begin first");
    if (h->step_expanded) return fail(KMC_E_STATE, "kmc_step_check_frontier between kmc_step_expand and kmc_step_finish");
    HIP_TRY(hipSetDevice(h->cfg.device));
    memset(info, 0, sizeof *info);
    info->depth = h->level;
    info->new_states = queue_now(h);
    if (h->n_cur == 0 || h->cfg.invariant_mask == 0) return KMC_OK;
    int rc = zero_ctl(h, 2);
    if (rc) return rc;
    KmcArgs d = base_args(h, 2);
    d.fin = h->frontier[h->cur];
    if ((rc = launch_inv(h, d, h->n_cur))) return rc;
    if ((rc = read_ctl(h, 2))) return rc;
    for (int k = 0; k < 4; ++k) {
        info->violation_count[k] = weighted(h, h->ctl_host->viol_count[k], h->ctl_host->corr_viol[k]);
        info->violation_fp[k] = h->ctl_host->viol_count[k] ? ~h->ctl_host->viol_fp_inv[k] : 0;
    }
    return KMC_OK;
}

// A violating successor OUTSIDE the state constraint is in no shard's table and no frontier.  After the
// kmc_step_finish of the expansion that generated it (and before the next kmc_step_expand overwrites that level),
// this looks for it among the successors of the retired level: *found = 1 gives its packed words and the
// fingerprint of the parent it was generated from (the smallest one).
int kmc_step_find_outside(kmc_handle* h, uint64_t fp, uint64_t* words, uint64_t* parent_fp, int32_t* found) {
    if (!h || !h->stepping || !words || !parent_fp || !found) return fail(KMC_E_ARG, "bad argument");
    if (h->step_expanded) return fail(KMC_E_STATE, "the retired level has been overwritten by kmc_step_expand");
    HIP_TRY(hipSetDevice(h->cfg.device));
    *found = 0;
    uint64_t n = 0;
    for (int sg = 0; sg < KMC_SEGS; ++sg) n += h->prev_seg_n[sg];
    if (n == 0) return KMC_OK;
    const bool had = h->have_witness;
    int rc = find_outside_witness(h, h->frontier[h->cur ^ 1], h->prev_seg_n, fp);
    if (rc) {  // "not found among the successors" is an answer here, not an error
        g_err.clear();
        h->witness_outside = false;
        h->have_witness = had;
        return KMC_OK;
    }
    for (int k = 0; k < h->W; ++k) words[k] = h->witness[k];
    *parent_fp = h->witness_parent_fp;
    *found = 1;
    return KMC_OK;
}

// Continue a sharded search from a shard checkpoint: after kmc_checkpoint_load the handle is back at the level
// boundary it was saved at; the next call is kmc_step_expand.
int kmc_step_resume(kmc_handle* h) {
    if (!h) return fail(KMC_E_ARG, "null handle");
    if (!h->table || !h->restored) return fail(KMC_E_STATE, "kmc_step_resume needs a handle restored by kmc_checkpoint_load");
    h->restored = false;
    h->stepping = true;
    h->step_expanded = false;
    h->xcounts_valid = false;
    h->t_start = now_s() - h->res.seconds_total;
    if (h->res.verdict == KMC_V_LEVEL_LIMIT) h->res.verdict = KMC_V_OK;
    h->res.queue_left = 0;
    return KMC_OK;
}

int kmc_step_set_verdict(kmc_handle* h, int32_t verdict) {
    if (!h) return fail(KMC_E_ARG, "null handle");
    h->res.verdict = verdict;
    return KMC_OK;
}


}  // extern "C"
