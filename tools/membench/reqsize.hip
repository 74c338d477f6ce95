// What does ONE random 8-byte probe cost the memory side of an MI355X, and can it be made smaller?
// randbench (this directory) puts the chip's ceiling for random 8-byte loads from an 8 GiB table at ~48-50 G loads/s.
// 50 G/s is also what a 6.3 TB/s stream is in 128-byte requests, so that ceiling is EITHER "every random load fills a
// whole 128-byte L2 line and the loads are HBM-bandwidth-bound" OR "a load asks for 64 bytes and the fabric / channels
// are request-rate-bound".  FETCH_SIZE cannot tell (on gfx950 it tallies a 128-byte request as 64, MI355X_MICROARCH.md).
// This tool separates the two by timing and by the gfx950-only request-size counters (TCC_EA0_RDREQ_64B / _128B /
// _32B, TCC_EA0_RDREQ_DRAM_32B), and tries every load flavour and allocation type that could make the request smaller.
//   load flavours (MODE), all dependent random loads, one per lane-iteration:
//     0 plain global_load_dwordx2        1 nt                      2 sc1            3 sc0 sc1          4 sc0 sc1 nt
//     5 plain global_load_dword (4 B)    6 plain global_load_dwordx4 (16 B, aligned)
//     7 two loads: the slot and its partner in the OTHER 64-byte half of the same 128-byte line
//     8 two loads: the slot and its partner 32 bytes away in the SAME 64-byte half            (control for 7)
//   allocations (ALLOC): 0 hipMalloc   1 hipExtMallocWithFlags(hipDeviceMallocUncached)   2 ...(hipDeviceMallocFinegrained)
// Kernel names are k<MODE, ALLOC> so a rocprofv3 counter CSV can be read per flavour.
// Build: hipcc --offload-arch=gfx950 -O3 reqsize.hip -o reqsize        Usage: reqsize [log2_slots=30] [iters=200]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
__device__ __forceinline__ u64 mix(u64 x) { x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31; return x; }

#define LOAD64(flags) asm volatile("global_load_dwordx2 %0, %1, off " flags "\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory")

template <int MODE> __device__ __forceinline__ u64 probe(const u64* table, u64 slot) {
    const u64* p = table + slot;
    u64 v = 0;
    if (MODE == 0) { LOAD64(""); }
    else if (MODE == 1) { LOAD64("nt"); }
    else if (MODE == 2) { LOAD64("sc1"); }
    else if (MODE == 3) { LOAD64("sc0 sc1"); }
    else if (MODE == 4) { LOAD64("sc0 sc1 nt"); }
    else if (MODE == 5) {
        u32 w;
        asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(p) : "memory");
        v = w;
    } else if (MODE == 6) {
        const u64* q = table + (slot & ~1ull);
        typedef u32 u32x4 __attribute__((ext_vector_type(4)));
        u32x4 w;
        asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(q) : "memory");
        v = w.x ^ ((u64)w.w << 32);
    } else {
        const u64* q = table + (slot ^ (MODE == 7 ? 8ull : 4ull));  // +-64 bytes: other half of the line; +-32: same half
        u64 v2;
        asm volatile("global_load_dwordx2 %0, %2, off\n\tglobal_load_dwordx2 %1, %3, off\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v), "=&v"(v2) : "v"(p), "v"(q) : "memory");
        v ^= v2;
    }
    return v;
}

template <int MODE, int ALLOC> __global__ __launch_bounds__(256) void k(const u64* table, u64 mask, int iters, u64* sink) {
    u64 x = mix(blockIdx.x * 256ull + threadIdx.x + 12345 + MODE * 977 + ALLOC * 131071);
    for (int i = 0; i < iters; ++i) x = mix(x + 1 + probe<MODE>(table, x & mask));
    if (x == 0x1234) sink[0] = x;
}

static hipEvent_t e0, e1;
template <int MODE, int ALLOC> static void run(const u64* table, u64 slots, int iters, u64* sink, const char* what) {
    const int blocks = 256 * 8;
    k<MODE, ALLOC><<<blocks, 256>>>(table, slots - 1, 4, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, ALLOC><<<blocks, 256>>>(table, slots - 1, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 256 * iters;
    printf("alloc %d mode %d %-34s %7.1f M lane-iterations in %6.2f ms = %5.1f G/s\n", ALLOC, MODE, what, n / 1e6, ms, n / ms / 1e6);
    fflush(stdout);
}
template <int ALLOC> static void all_modes(const u64* t, u64 slots, int iters, u64* sink) {
    run<0, ALLOC>(t, slots, iters, sink, "plain dwordx2");
    run<1, ALLOC>(t, slots, iters, sink, "nt");
    run<2, ALLOC>(t, slots, iters, sink, "sc1");
    run<3, ALLOC>(t, slots, iters, sink, "sc0 sc1");
    run<4, ALLOC>(t, slots, iters, sink, "sc0 sc1 nt");
    run<5, ALLOC>(t, slots, iters, sink, "plain dword");
    run<6, ALLOC>(t, slots, iters, sink, "plain dwordx4");
    run<7, ALLOC>(t, slots, iters, sink, "slot + other half of its 128-B line");
    run<8, ALLOC>(t, slots, iters, sink, "slot + 32 B away, same 64-B half");
}
int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 30;
    const int iters = argc > 2 ? atoi(argv[2]) : 200;
    const u64 slots = 1ull << lg;
    u64* sink;
    if (hipMalloc(&sink, 8) != hipSuccess) return 1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int alloc = 0; alloc < 3; ++alloc) {
        u64* t = nullptr;
        hipError_t e = alloc == 0 ? hipMalloc(&t, slots * 8)
                                  : hipExtMallocWithFlags((void**)&t, slots * 8, alloc == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
        if (e != hipSuccess) { printf("alloc %d failed: %s\n", alloc, hipGetErrorString(e)); continue; }
        hipMemset(t, 0, slots * 8);
        hipDeviceSynchronize();
        printf("# alloc %d = %s, table of 2^%d slots\n", alloc, alloc == 0 ? "hipMalloc" : alloc == 1 ? "uncached" : "fine-grained", lg);
        if (alloc == 0) all_modes<0>(t, slots, iters, sink);
        if (alloc == 1) all_modes<1>(t, slots, iters, sink);
        if (alloc == 2) all_modes<2>(t, slots, iters, sink);
        hipFree(t);
    }
    return 0;
}
