// Which PHYSICAL chunks are fast?  (round 6, call 35.)  The headline's k_expand sits at a level that is fixed when a handle's seen-set
// is mapped and that drifts, for fresh processes, with what the driver's allocator hands out at that moment (call 34: one long-lived
// handle flat at 30.4 ms for a minute while fresh processes between its searches went 28.7 -> 30.3 - 31.2 -> 28.7), the streaming
// clear of the table moving in step.  So some of the HBM's physical pages are slower than others.  This maps N chunks of 8 MiB
// into one range and measures GROUPS of them (1 GiB each): a streaming memset (the min of 12) and random 8-byte stores (randbench's
// mode 4 over the group: 2^25 stores), group by group, in allocation order.
//   hipcc --offload-arch=gfx950 -O3 placement.hip -o placement ; ./placement [GiB to map, default 64]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef unsigned long long u64;
__device__ __forceinline__ u64 mix(u64 x) { x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31; return x; }
__global__ __launch_bounds__(256) void rstore(u64* t, u64 mask, int iters) {
    u64 x = mix(blockIdx.x * 256ull + threadIdx.x + 12345);
    for (int i = 0; i < iters; ++i) { x = mix(x + 1); t[x & mask] = x | 1; }
}
__global__ __launch_bounds__(256) void rmix(u64* t, u64 mask, int iters, u64* sink) {   // randbench's mode 7: load, CAS for 35 %
    u64 x = mix(blockIdx.x * 256ull + threadIdx.x + 999), acc = 0;
    for (int i = 0; i < iters; ++i) {
        x = mix(x + 1);
        u64 v = t[x & mask];
        if (((x >> 40) & 0xFF) < 90) v = atomicCAS(&t[x & mask], v, x | 1);
        acc ^= v;
    }
    if (acc == 0x1234) sink[0] = acc;
}
int main(int argc, char** argv) {
    const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 64;
    const size_t chunk = 8ull << 20, group = 1ull << 30, total = gib << 30, per_group = group / chunk;
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
    printf("# %zu GiB mapped from %zu chunks of 8 MiB at %p; groups of %zu chunks (1 GiB), in allocation order\n", gib, total / chunk, va, per_group);
    printf("# group  memset GB/s (best of 12)  random 8-byte stores G/s  load+CAS mix G/s\n");
    std::vector<double> ms_all;
    for (size_t g = 0; g < total / group; ++g) {
        char* p = (char*)va + g * group;
        float best = 1e9f;
        for (int r = 0; r < 12; ++r) {
            (void)hipEventRecord(e0); (void)hipMemsetAsync(p, r & 1, group); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        (void)hipMemset(p, 0, group);
        float st = 1e9f, mx = 1e9f;
        for (int r = 0; r < 3; ++r) {
            (void)hipEventRecord(e0); rstore<<<2048, 256>>>((u64*)p, group / 8 - 1, 64); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < st) st = ms;
        }
        (void)hipMemset(p, 0, group);
        for (int r = 0; r < 3; ++r) {
            (void)hipEventRecord(e0); rmix<<<2048, 256>>>((u64*)p, group / 8 - 1, 64, sink); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < mx) mx = ms;
        }
        const double n = 2048.0 * 256 * 64;
        printf("%5zu  %8.0f  %8.2f  %8.2f\n", g, group / best / 1e6, n / st / 1e6, n / mx / 1e6);
        ms_all.push_back(group / best / 1e6);
    }
    std::sort(ms_all.begin(), ms_all.end());
    printf("# memset GB/s over the groups: min %.0f, median %.0f, max %.0f\n", ms_all.front(), ms_all[ms_all.size() / 2], ms_all.back());
    // the whole range at once, and its first / last 8 GiB: what a table lying there would see
    for (int part = 0; part < 3; ++part) {
        const size_t bytes = part == 0 ? total : (8ull << 30);
        char* p = (char*)va + (part == 2 ? total - bytes : 0);
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            (void)hipEventRecord(e0); (void)hipMemsetAsync(p, 0, bytes); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        float mx = 1e9f;
        for (int r = 0; r < 2; ++r) {
            (void)hipEventRecord(e0); rmix<<<2048, 256>>>((u64*)p, bytes / 8 - 1, 400, sink); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < mx) mx = ms;
        }
        printf("# %s: memset %.0f GB/s, load+CAS mix %.2f G/s\n", part == 0 ? "the whole range" : part == 1 ? "its first 8 GiB" : "its last 8 GiB", bytes / best / 1e6, 2048.0 * 256 * 400 / mx / 1e6);
    }
    return 0;
}
