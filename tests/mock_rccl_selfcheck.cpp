// CPU self-check of tests/mock_rccl.cpp (built with -DMOCK_HOST: plain memory instead of device memory): P host threads
// play the ranks of one communicator and move the runs of random BFS-level count matrices exactly as
// kmc_step_exchange_payload does — the plan comes from libkmc's own kmc_exchange_plan — then every rank checks that its
// receive area holds, source by source and sub-buffer by sub-buffer, the records its peers addressed to it.  Also the
// ring + all-gather pattern of kmc_comm_selftest.  Test infrastructure (tests/test_exchange_plan_cpu.py runs it).
// usage: mock_rccl_selfcheck <path to libkmc.so> [P] [rounds]
#define MOCK_HOST 1
#define MOCK_TIMEOUT_S 20
#include "mock_rccl.cpp"

#include <dlfcn.h>
#include <atomic>
#include <cstdlib>
#include <thread>

typedef int (*plan_fn)(const uint64_t*, int32_t, int32_t, uint64_t, uint64_t, uint64_t*, uint64_t*, uint64_t, uint64_t*, uint64_t*,
                       uint64_t*);
static const int SEGS = 8;

static uint64_t rnd(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static uint64_t tag(int src, int dst, int sub, uint64_t idx, int k) {
    return ((uint64_t)(src + 1) << 56) | ((uint64_t)(dst + 1) << 48) | ((uint64_t)sub << 40) | (idx << 8) | (uint64_t)k;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s libkmc.so [P] [rounds]\n", argv[0]); return 2; }
    void* lib = dlopen(argv[1], RTLD_NOW);
    if (!lib) { fprintf(stderr, "dlopen %s: %s\n", argv[1], dlerror()); return 2; }
    plan_fn plan = (plan_fn)dlsym(lib, "kmc_exchange_plan");
    if (!plan) { fprintf(stderr, "kmc_exchange_plan not exported\n"); return 2; }
    const int P = argc > 2 ? atoi(argv[2]) : 4;
    const int rounds = argc > 3 ? atoi(argv[3]) : 20;
    const uint64_t send_cap = 700, rec_words = 4;
    ncclUniqueId id;
    ncclGetUniqueId(&id);
    std::atomic<int> failures{0};
    // the count matrices of all rounds, known to every rank (in the engine they come out of the all-gather)
    std::vector<std::vector<uint64_t>> counts(rounds, std::vector<uint64_t>((size_t)P * P * SEGS, 0));
    uint64_t seed = 0x9E3779B97F4A7C15ull;
    for (int r = 0; r < rounds; ++r)
        for (int s = 0; s < P; ++s)
            for (int d = 0; d < P; ++d)
                for (int sb = 0; sb < SEGS; ++sb) {
                    const uint64_t x = rnd(seed);
                    // a third of the sub-buffers empty, some full, rank 0 sends nothing at all in round 1 (the BFS level-1 shape)
                    uint64_t c = (x % 3 == 0) ? 0 : (x % 11 == 0 ? send_cap : (x >> 20) % send_cap);
                    if (s == d || (r == 1 && s == 0)) c = 0;
                    counts[r][((size_t)s * P + d) * SEGS + sb] = c;
                }
    auto rank_main = [&](int me) {
        ncclComm_t comm = nullptr;
        if (ncclCommInitRank(&comm, P, id, me) != ncclSuccess) { ++failures; return; }
        std::vector<uint64_t> send((size_t)P * SEGS * send_cap * rec_words), recv((size_t)(P - 1) * SEGS * send_cap * rec_words + 1);
        // kmc_comm_selftest's pattern: a ring of grouped send/recv, then an all-gather
        {
            const size_t n = 512;
            std::vector<uint64_t> buf((2 + P) * n, 0);
            for (size_t i = 0; i < n; ++i) buf[i] = ((uint64_t)(me + 1) << 32) | i;
            ncclGroupStart();
            ncclSend(buf.data(), n, ncclUint64, (me + 1) % P, comm, nullptr);
            ncclRecv(buf.data() + n, n, ncclUint64, (me + P - 1) % P, comm, nullptr);
            if (ncclGroupEnd() != ncclSuccess) { ++failures; return; }
            if (ncclAllGather(buf.data(), buf.data() + 2 * n, n, ncclUint64, comm, nullptr) != ncclSuccess) { ++failures; return; }
            const uint64_t from = (uint64_t)((me + P - 1) % P + 1);
            for (size_t i = 0; i < n; ++i) {
                if (buf[n + i] != ((from << 32) | i)) { ++failures; return; }
                for (int q = 0; q < P; ++q)
                    if (buf[(2 + q) * n + i] != (((uint64_t)(q + 1) << 32) | i)) { ++failures; return; }
            }
        }
        for (int r = 0; r < rounds; ++r) {
            const uint64_t* c = counts[r].data();
            for (int d = 0; d < P; ++d)
                for (int sb = 0; sb < SEGS; ++sb) {
                    const uint64_t n = c[((size_t)me * P + d) * SEGS + sb];
                    uint64_t* base = send.data() + (((uint64_t)d * SEGS + sb) * send_cap) * rec_words;
                    for (uint64_t i = 0; i < n; ++i)
                        for (uint64_t k = 0; k < rec_words; ++k) base[i * rec_words + k] = tag(me, d, sb, i + 1000ull * r, (int)k);
                }
            std::fill(recv.begin(), recv.end(), 0xDEADull);
            const uint64_t cap = (uint64_t)P * SEGS * 4;
            std::vector<uint64_t> sv(3 * cap), rv(3 * cap);
            uint64_t ns = 0, nr = 0, nrec = 0;
            if (plan(c, P, me, send_cap, rec_words, sv.data(), rv.data(), cap, &ns, &nr, &nrec) != 0 || ns > cap || nr > cap) { ++failures; return; }
            ncclGroupStart();
            for (uint64_t i = 0; i < ns; ++i) ncclSend(send.data() + sv[3 * i + 1], sv[3 * i + 2], ncclUint64, (int)sv[3 * i], comm, nullptr);
            for (uint64_t i = 0; i < nr; ++i) ncclRecv(recv.data() + rv[3 * i + 1], rv[3 * i + 2], ncclUint64, (int)rv[3 * i], comm, nullptr);
            if (ncclGroupEnd() != ncclSuccess) { ++failures; return; }
            // what must have arrived: source by source, sub-buffer by sub-buffer, densely
            uint64_t at = 0;
            for (int s = 0; s < P; ++s) {
                if (s == me) continue;
                for (int sb = 0; sb < SEGS; ++sb) {
                    const uint64_t n = c[((size_t)s * P + me) * SEGS + sb];
                    for (uint64_t i = 0; i < n; ++i, ++at)
                        for (uint64_t k = 0; k < rec_words; ++k)
                            if (recv[at * rec_words + k] != tag(s, me, sb, i + 1000ull * r, (int)k)) {
                                fprintf(stderr, "round %d rank %d: record %llu from rank %d sub %d word %llu is wrong\n", r, me,
                                        (unsigned long long)i, s, sb, (unsigned long long)k);
                                ++failures;
                                return;
                            }
                }
            }
            if (at != nrec || recv[at * rec_words] != 0xDEADull) { ++failures; return; }
        }
        ncclCommDestroy(comm);
    };
    std::vector<std::thread> th;
    for (int r = 0; r < P; ++r) th.emplace_back(rank_main, r);
    for (auto& t : th) t.join();
    if (failures.load()) { fprintf(stderr, "FAILED (%d ranks)\n", failures.load()); return 1; }
    // a mismatched pair must fail, not hang: rank 0 sends 8 words, rank 1 expects 16
    {
        ncclUniqueId id2;
        ncclGetUniqueId(&id2);
        std::atomic<int> errors{0};
        auto bad = [&](int me) {
            ncclComm_t comm = nullptr;
            ncclCommInitRank(&comm, 2, id2, me);
            std::vector<uint64_t> b(32, 7);
            ncclResult_t rc = me == 0 ? ncclSend(b.data(), 8, ncclUint64, 1, comm, nullptr) : ncclRecv(b.data(), 16, ncclUint64, 0, comm, nullptr);
            if (rc != ncclSuccess) ++errors;
        };
        std::thread a(bad, 0), b(bad, 1);
        a.join(); b.join();
        if (errors.load() == 0) { fprintf(stderr, "a size mismatch went unnoticed\n"); return 1; }
    }
    printf("mock rccl selfcheck ok: P=%d, %d rounds\n", P, rounds);
    return 0;
}
