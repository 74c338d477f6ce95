// A stand-in for librccl that lives inside ONE process: the "ranks" of a communicator are host threads, each driving its
// own libkmc handle on the same GPU.  Test infrastructure only (tests/test_gpu_native_exchange_threads.py and
// tests/mock_rccl_selfcheck.cpp load it through KMC_RCCL_LIB); nothing under kafka_specification_amd/ refers to it.
//
// Why it exists: RCCL refuses two ranks on one device and a gpurun box has one GPU, so the exchange under the C ABI
// (csrc/kmc_engine_exchange.cpp: kmc_comm_init / kmc_comm_selftest / kmc_step_exchange_counts / kmc_step_exchange_payload) could
// only ever run with world size 1, where it returns before any collective.  With this library in RCCL's place the very same
// code runs with P > 1 concurrent ranks: the all-gather row layout, the posting order of the grouped sends and receives,
// their offsets into the send and receive areas, the 1 GiB cuts, the k_insert queued behind the receives.  What stays
// untested is RCCL itself (its transport and its stream semantics).
//
// Semantics implemented (the subset the engine uses; same signatures as <rccl/rccl.h>):
//   ncclGetUniqueId, ncclCommInitRank (blocks until all ranks of the id have joined, like the real one), ncclCommDestroy,
//   ncclAllGather, ncclSend / ncclRecv inside ncclGroupStart / ncclGroupEnd (a bare Send / Recv is a group of one),
//   ncclGetErrorString.
// Everything is synchronous: an operation first drains the stream it was given, moves the bytes with a device-to-device
// copy (the ranks share the device) and returns when its part is done — stricter than RCCL's stream ordering, never looser.
// The messages of an ordered pair (source, destination) are matched in posting order, as RCCL matches them.  A receive
// whose message does not arrive within MOCK_TIMEOUT_S seconds, or arrives with another size, fails the call (no hang).
//
// -DMOCK_HOST builds it for plain host memory (memcpy instead of hipMemcpy): tests/mock_rccl_selfcheck.cpp checks the
// matching logic on CPU against kmc_exchange_plan.
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#ifdef MOCK_HOST
typedef void* hipStream_t;
enum ncclResult_t { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 };
enum ncclDataType_t { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5 };
struct ncclUniqueId { char internal[128]; };
typedef struct ncclComm* ncclComm_t;
static int mock_copy(void* dst, const void* src, size_t n) { memcpy(dst, src, n); return 0; }
static int mock_sync_stream(hipStream_t) { return 0; }
static int mock_sync_device() { return 0; }
#else
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
static int mock_copy(void* dst, const void* src, size_t n) { return hipMemcpy(dst, src, n, hipMemcpyDeviceToDevice) == hipSuccess ? 0 : 1; }
static int mock_sync_stream(hipStream_t s) { return hipStreamSynchronize(s) == hipSuccess ? 0 : 1; }
static int mock_sync_device() { return hipDeviceSynchronize() == hipSuccess ? 0 : 1; }
#endif

#ifndef MOCK_TIMEOUT_S
#define MOCK_TIMEOUT_S 30
#endif
#define MOCK_MAX_RANKS 16

struct Msg {
    const void* ptr;
    size_t bytes;
    bool* done;  // set by the receiver once the bytes have been copied out of the sender's buffer
};

struct Group {
    int n = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    bool broken = false;  // a rank timed out: everybody fails from here on instead of waiting for it
    const void* ag_src[MOCK_MAX_RANKS] = {nullptr};
    std::deque<Msg> q[MOCK_MAX_RANKS][MOCK_MAX_RANKS];  // [source][destination], FIFO

    // generation barrier over the n ranks; false on time-out
    bool barrier(std::unique_lock<std::mutex>& lk) {
        const uint64_t gen = generation;
        if (++arrived == n) {
            arrived = 0;
            ++generation;
            cv.notify_all();
            return !broken;
        }
        const bool ok = cv.wait_for(lk, std::chrono::seconds(MOCK_TIMEOUT_S), [&] { return generation != gen || broken; });
        if (!ok) { broken = true; cv.notify_all(); }
        return ok && !broken;
    }
};

namespace {

struct Op {
    bool send;
    void* ptr;
    size_t bytes;
    int peer;
    struct ncclComm* comm;
    hipStream_t stream;
    bool done;
};

std::mutex g_registry_mutex;
std::map<std::string, Group*> g_registry;
thread_local int t_group_depth = 0;
thread_local std::vector<Op>* t_ops = nullptr;

size_t type_bytes(ncclDataType_t t) {
    switch ((int)t) {
    case 0: case 1: return 1;
    case 2: case 3: return 4;
    case 4: case 5: return 8;
    default: return 0;
    }
}

}  // namespace

struct ncclComm {
    Group* g;
    int rank;
};

namespace {

ncclResult_t run_ops(std::vector<Op>& ops) {
    if (ops.empty()) return ncclSuccess;
    for (Op& o : ops)
        if (mock_sync_stream(o.stream)) return ncclUnhandledCudaError;
    // 1. post every send (never blocks)
    for (Op& o : ops) {
        if (!o.send) continue;
        Group* g = o.comm->g;
        std::lock_guard<std::mutex> lk(g->m);
        g->q[o.comm->rank][o.peer].push_back(Msg{o.ptr, o.bytes, &o.done});
        g->cv.notify_all();
    }
    // 2. serve the receives in posting order: the k-th receive from a peer takes that peer's k-th send to this rank
    ncclResult_t rc = ncclSuccess;
    for (Op& o : ops) {
        if (o.send) continue;
        Group* g = o.comm->g;
        Msg msg{};
        {
            std::unique_lock<std::mutex> lk(g->m);
            auto& q = g->q[o.peer][o.comm->rank];
            const bool ok = g->cv.wait_for(lk, std::chrono::seconds(MOCK_TIMEOUT_S), [&] { return !q.empty() || g->broken; });
            if (!ok || g->broken) {
                g->broken = true;
                g->cv.notify_all();
                fprintf(stderr, "[mock rccl] rank %d: no message from rank %d within %d s\n", o.comm->rank, o.peer, MOCK_TIMEOUT_S);
                return ncclSystemError;
            }
            msg = q.front();
            q.pop_front();
        }
        if (msg.bytes != o.bytes) {
            fprintf(stderr, "[mock rccl] rank %d: receive of %zu bytes from rank %d meets a send of %zu bytes\n", o.comm->rank,
                    o.bytes, o.peer, msg.bytes);
            rc = ncclInvalidArgument;
        } else if (o.bytes && mock_copy(o.ptr, msg.ptr, o.bytes)) {
            rc = ncclUnhandledCudaError;
        }
        if (mock_sync_device()) rc = ncclUnhandledCudaError;
        {
            std::lock_guard<std::mutex> lk(g->m);
            *msg.done = true;
            if (rc != ncclSuccess) g->broken = true;
            g->cv.notify_all();
        }
        if (rc != ncclSuccess) return rc;
    }
    // 3. a send completes when its receiver has copied the bytes out (the send area may then be reused)
    for (Op& o : ops) {
        if (!o.send) continue;
        Group* g = o.comm->g;
        std::unique_lock<std::mutex> lk(g->m);
        const bool ok = g->cv.wait_for(lk, std::chrono::seconds(MOCK_TIMEOUT_S), [&] { return o.done || g->broken; });
        if (!ok || g->broken) {
            g->broken = true;
            g->cv.notify_all();
            fprintf(stderr, "[mock rccl] rank %d: rank %d never received a message of %zu bytes\n", o.comm->rank, o.peer, o.bytes);
            return ncclSystemError;
        }
    }
    return ncclSuccess;
}

ncclResult_t p2p(bool send, void* ptr, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    if (!comm || peer < 0 || peer >= comm->g->n || type_bytes(type) == 0) return ncclInvalidArgument;
    Op o{send, ptr, count * type_bytes(type), peer, comm, stream, false};
    if (t_group_depth > 0) {
        if (!t_ops) t_ops = new std::vector<Op>();
        t_ops->push_back(o);
        return ncclSuccess;
    }
    std::vector<Op> one{o};
    return run_ops(one);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    static std::mutex m;
    static uint64_t counter = 0;
    if (!id) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lk(m);
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "kmc-mock-rccl-%llu-%p", (unsigned long long)++counter, (void*)&counter);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > MOCK_MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Group* g;
    {
        std::lock_guard<std::mutex> lk(g_registry_mutex);
        Group*& slot = g_registry[std::string(id.internal, sizeof id.internal)];
        if (!slot) { slot = new Group(); slot->n = nranks; }
        g = slot;
    }
    if (g->n != nranks) return ncclInvalidArgument;
    *comm = new ncclComm{g, rank};
    std::unique_lock<std::mutex> lk(g->m);
    return g->barrier(lk) ? ncclSuccess : ncclSystemError;  // like the real one: returns once every rank has joined
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    delete comm;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream) {
    if (!comm || type_bytes(datatype) == 0) return ncclInvalidArgument;
    Group* g = comm->g;
    const size_t bytes = sendcount * type_bytes(datatype);
    if (mock_sync_stream(stream)) return ncclUnhandledCudaError;
    {
        std::unique_lock<std::mutex> lk(g->m);
        g->ag_src[comm->rank] = sendbuff;
        if (!g->barrier(lk)) return ncclSystemError;  // every contribution is published
    }
    ncclResult_t rc = ncclSuccess;
    for (int r = 0; r < g->n; ++r) {
        char* dst = (char*)recvbuff + (size_t)r * bytes;
        if ((const void*)dst != g->ag_src[r] && bytes && mock_copy(dst, g->ag_src[r], bytes)) rc = ncclUnhandledCudaError;
    }
    if (mock_sync_device()) rc = ncclUnhandledCudaError;
    {
        std::unique_lock<std::mutex> lk(g->m);
        if (!g->barrier(lk)) return ncclSystemError;  // nobody overwrites its contribution before everyone has read it
    }
    return rc;
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    return p2p(true, const_cast<void*>(sendbuff), count, datatype, peer, comm, stream);
}

ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    return p2p(false, recvbuff, count, datatype, peer, comm, stream);
}

ncclResult_t ncclGroupStart() {
    ++t_group_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (t_group_depth <= 0) return ncclInvalidUsage;
    if (--t_group_depth > 0) return ncclSuccess;
    if (!t_ops) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(*t_ops);
    return run_ops(ops);
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch ((int)r) {
    case 0: return "no error (mock rccl)";
    case 1: return "a HIP call failed (mock rccl)";
    case 2: return "a rank did not show up in time (mock rccl)";
    case 4: return "invalid argument / message sizes of a matched send and receive differ (mock rccl)";
    case 5: return "invalid usage (mock rccl)";
    default: return "internal error (mock rccl)";
    }
}

}  // extern "C"
