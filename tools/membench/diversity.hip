// Is it the DIVERSITY of a table's physical chunks that makes it fast?  (round 6, call 36; after tools/membench/placement, call 35:
// 1 GiB groups of sequentially created chunks run random stores at 27.9 G/s in long runs - and at 36 - 40 G/s exactly where a group
// straddles two of the allocator's blocks.)  A pool of P GiB is mapped from 8 MiB chunks created one after the other; an 8 GiB
// "table" is then any 1,024 of the pool's chunks, chosen IN THE KERNEL'S ADDRESS ARITHMETIC (no remapping): slot s lies in chunk
// f(s >> 20).  f = consecutive chunks from position q; every k-th chunk; a multiplicative permutation of the pool's chunks.
// Measured per choice: randbench's mode 7 (load, CAS for 35 %), mode 4 (plain stores) and a streaming fill of the same chunks.
//   hipcc --offload-arch=gfx950 -O3 diversity.hip -o diversity ; ./diversity [pool GiB, a power of two, default 128]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
__device__ __forceinline__ u64 mix(u64 x) { x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31; return x; }
struct Pick { u64 start, stride, mul, nchunks_mask; };   // chunk of table-chunk c = ((start + c * stride) * mul) & nchunks_mask
__device__ __forceinline__ u64 slot_of(const Pick& p, u64 s) {   // s < 2^30 table slots; 2^20 slots per 8 MiB chunk
    const u64 c = s >> 20, off = s & ((1ull << 20) - 1);
    return ((((p.start + c * p.stride) * p.mul) & p.nchunks_mask) << 20) | off;
}
__global__ __launch_bounds__(256) void k(u64* pool, Pick p, int iters, int mode, u64* sink) {
    u64 x = mix(blockIdx.x * 256ull + threadIdx.x + 4242), acc = 0;
    const u64 mask = (1ull << 30) - 1;
    if (mode == 7) {
        for (int i = 0; i < iters; ++i) {
            x = mix(x + 1);
            u64* a = pool + slot_of(p, x & mask);
            u64 v = *a;
            if (((x >> 40) & 0xFF) < 90) v = atomicCAS(a, v, x | 1);
            acc ^= v;
        }
    } else if (mode == 4) {
        for (int i = 0; i < iters; ++i) { x = mix(x + 1); pool[slot_of(p, x & mask)] = x | 1; }
    } else if (mode == 1) {
        for (int i = 0; i < iters; ++i) { x = mix(x + 1); acc ^= pool[slot_of(p, x & mask)]; }
    } else {   // streaming fill of the table's 2^30 slots, 16 bytes per lane and step
        const u64 n2 = 1ull << 29, step = (u64)gridDim.x * 256;
        for (u64 s2 = blockIdx.x * 256ull + threadIdx.x; s2 < n2; s2 += step)
            *(ulonglong2*)(pool + slot_of(p, 2 * s2)) = make_ulonglong2(0, 0);
    }
    if (acc == 0x1234) sink[0] = acc;
}
int main(int argc, char** argv) {
    const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 128;
    const size_t chunk = 8ull << 20, total = gib << 30, nchunks = total / chunk;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    void* va = nullptr;
    if (hipMemAddressReserve(&va, total, chunk, nullptr, 0) != hipSuccess) { printf("reserve failed\n"); return 1; }
    for (size_t done = 0; done < total; done += chunk) {
        hipMemGenericAllocationHandle_t piece;
        if (hipMemCreate(&piece, chunk, &prop, 0) != hipSuccess || hipMemMap((char*)va + done, chunk, 0, piece, 0) != hipSuccess) { printf("chunk at %zu failed\n", done); return 1; }
        (void)hipMemRelease(piece);
    }
    hipMemAccessDesc d{}; d.location = prop.location; d.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(va, total, &d, 1) != hipSuccess) { printf("access failed\n"); return 1; }
    u64* sink; (void)hipMalloc(&sink, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipMemset(va, 0, total);
    printf("# pool of %zu GiB = %zu chunks of 8 MiB, created one after the other; every table below is 1,024 of them (8 GiB)\n", gib, nchunks);
    printf("# table's chunks                                   load+CAS mix G/s   stores G/s   loads G/s   fill TB/s\n");
    auto run = [&](const char* name, Pick p) {
        double r[4];
        const int modes[4] = {7, 4, 1, 0};
        for (int m = 0; m < 4; ++m) {
            float best = 1e9f;
            const int iters = 400;
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0); k<<<2048, 256>>>((u64*)va, p, iters, modes[m], sink); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            r[m] = modes[m] ? 2048.0 * 256 * iters / best / 1e6 : 8.0 * (1ull << 30) / best / 1e9;
        }
        printf("%-50s %8.2f        %8.2f    %8.2f    %6.2f\n", name, r[0], r[1], r[2], r[3]);
    };
    const u64 m = nchunks - 1;
    char nm[128];
    for (u64 q = 0; q + 1024 <= nchunks; q += (nchunks - 1024) / 7) { snprintf(nm, sizeof nm, "consecutive from chunk %llu", q); run(nm, Pick{q, 1, 1, m}); }
    for (u64 kk = 2; kk * 1024 <= nchunks; kk *= 2) { snprintf(nm, sizeof nm, "every %llu-th chunk (over %llu GiB)", kk, kk * 8); run(nm, Pick{0, kk, 1, m}); }
    run("permuted: (c * 0x9E3779B1) mod pool", Pick{0, 1, 0x9E3779B1ull, m});
    run("permuted: ((5000 + c) * 0x85EBCA6B) mod pool", Pick{5000, 1, 0x85EBCA6Bull, m});
    return 0;
}
