// Random-access roofline of one MI355X for the seen-set's access pattern: 8-byte accesses at
// uniformly random slots of an 8 GiB table (each moves one 64-byte sector).
//   mode 0: dependent random loads            (latency-bound per lane, many lanes)
//   mode 1: 4 independent random loads / lane (memory-level parallelism x4)
//   mode 2: random atomicCAS(0 -> x) on a zeroed table, one per lane-iteration
//   mode 3: load, then atomicCAS when the slot was empty (the claim sequence)
//   mode 4: random plain 8-byte stores
//   mode 5: random atomicMax without using the result (no-return atomic)
//   mode 6: random agent-scope (sc1, L2-bypassing) loads
//   mode 7: the seen-set's own mix: a random load, and for 35 % of the accesses a CAS on the slot just read
//           (the headline run claims 312 M of its 888 M probes)
//   mode 8..11: random ALIGNED plain stores of 16 / 32 / 64 / 128 bytes (one lane writes the whole unit with dwordx4 stores):
//           does a write that covers a whole 32-byte sector / 64-byte half line / 128-byte line escape the read-modify-write
//           that an 8-byte store into an untouched DRAM line pays?  (round 4: what a claim protocol with wider slots could hope for)
//   mode 12: load, then a plain 8-byte store when the slot was empty (a claim without the atomic)
//   mode 13: the WIDE seen-set's mix (kmc_config.wide_fingerprint): one 16-byte load of a 16-byte slot, and for 35 % of the
//           accesses a CAS on its first word and an agent-scope store of its second (KmcSink::claim_wide)
// Round 6: footprints beyond 2^30 slots (RANDBENCH_MAX_LOG2, up to 2^34 = 128 GiB), for the regime of the 6.45 G-state stretch;
// RANDBENCH_ALIGN_GIB=1 places the table at a 1 GiB-aligned address inside a larger allocation (does the driver map it with
// larger fragments then?); RANDBENCH_MODES=1,3,7 runs only those modes.
// RANDBENCH_CHUNK_LOG2=23 (round 6, after the seen-set's own change): the table is a range of addresses mapped from physical chunks of
// 2^23 bytes (hipMemAddressReserve + hipMemCreate / hipMemMap per chunk, as KmcEngine's seen_set_alloc does) instead of one hipMalloc:
// the ceilings of the memory the product's seen-set now lies in.
// Prints G accesses/s.  Build: hipcc --offload-arch=gfx950 -O3 randbench.hip -o randbench
// Usage: randbench [first_mode [log2_slots ...]]  — with sizes given, every mode runs on a table of
// each size (footprint sweep: does a seen-set partition that fits L2 / Infinity Cache probe faster?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
__device__ __forceinline__ u64 mix(u64 x) { x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31; return x; }
__global__ __launch_bounds__(256) void k(u64* table, u64 mask, int iters, int mode, u64* sink) {
    u64 x = mix(blockIdx.x * 256ull + threadIdx.x + 12345);
    u64 acc = 0;
    if (mode == 0) {
        for (int i = 0; i < iters; ++i) { x = mix(x + table[x & mask]); }
        acc = x;
    } else if (mode == 1) {
        for (int i = 0; i < iters; i += 4) {
            u64 a = mix(x + 1), b = mix(x + 2), c = mix(x + 3), d = mix(x + 4);
            u64 va = table[a & mask], vb = table[b & mask], vc = table[c & mask], vd = table[d & mask];
            x = mix(x ^ va ^ vb ^ vc ^ vd ^ a);
        }
        acc = x;
    } else if (mode == 2) {
        for (int i = 0; i < iters; ++i) { x = mix(x + 1); acc ^= atomicCAS(&table[x & mask], 0ull, x | 1); }
    } else if (mode == 3) {
        for (int i = 0; i < iters; ++i) {
            x = mix(x + 1);
            u64 v = table[x & mask];
            if (v == 0) v = atomicCAS(&table[x & mask], 0ull, x | 1);
            acc ^= v;
        }
    } else if (mode == 4) {
        for (int i = 0; i < iters; ++i) { x = mix(x + 1); table[x & mask] = x | 1; }
    } else if (mode == 5) {
        for (int i = 0; i < iters; ++i) { x = mix(x + 1); atomicMax(&table[x & mask], x | 1); }
    } else if (mode == 6) {
        for (int i = 0; i < iters; ++i) { x = mix(x + __hip_atomic_load(&table[x & mask], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
        acc = x;
    } else if (mode == 7) {
        for (int i = 0; i < iters; ++i) {
            x = mix(x + 1);
            u64 v = table[x & mask];
            if (((x >> 40) & 0xFF) < 90) v = atomicCAS(&table[x & mask], v, x | 1);  // 35 %: claim whatever is there
            acc ^= v;
        }
    } else if (mode >= 8 && mode <= 11) {
        const int quads = 1 << (mode - 8);              // 16-byte pieces per store unit: 1, 2, 4, 8
        const u64 unit_mask = mask & ~(u64)(2 * quads - 1);   // slot index aligned to the unit
        for (int i = 0; i < iters; ++i) {
            x = mix(x + 1);
            uint4* p = (uint4*)&table[x & unit_mask];
            const uint4 v = make_uint4((unsigned)x, (unsigned)(x >> 32), (unsigned)i, 1u);
            for (int q = 0; q < quads; ++q) p[q] = v;
        }
    } else if (mode == 13) {
        const u64 smask = mask >> 1;   // 16-byte slots over the same footprint
        for (int i = 0; i < iters; ++i) {
            x = mix(x + 1);
            u64* slot = table + 2 * (x & smask);
            const ulonglong2 v = *(const ulonglong2*)slot;
            u64 r = v.x ^ v.y;
            if (((x >> 40) & 0xFF) < 90) {
                r ^= atomicCAS(slot, v.x, x | 1);
                __hip_atomic_store(slot + 1, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            acc ^= r;
        }
    } else {
        for (int i = 0; i < iters; ++i) {
            x = mix(x + 1);
            u64 v = table[x & mask];
            if (v == 0) table[x & mask] = x | 1;
            acc ^= v;
        }
    }
    if (acc == 0x1234) sink[0] = acc;
}
int main(int argc, char** argv) {
    const int max_log2 = getenv("RANDBENCH_MAX_LOG2") ? atoi(getenv("RANDBENCH_MAX_LOG2")) : 30;   // 30: 8 GiB
    const u64 max_slots = 1ull << max_log2;
    const bool align_gib = getenv("RANDBENCH_ALIGN_GIB") && atoi(getenv("RANDBENCH_ALIGN_GIB"));
    bool want[14];
    for (int m = 0; m < 14; ++m) want[m] = getenv("RANDBENCH_MODES") == nullptr;
    if (const char* ms = getenv("RANDBENCH_MODES"))
        for (const char* p = ms; *p;) { const int m = atoi(p); if (m >= 0 && m < 14) want[m] = true; while (*p && *p != ',') ++p; if (*p) ++p; }
    u64 *raw, *table, *sink;
    // RANDBENCH_SKIP=K: K allocations of the same size are made (and kept) first, so that the table lands elsewhere in the HBM
    // (round 6: does WHERE an 8 GiB table lies change its rate?  The headline's kernel differs by 5 - 8 % from process to process)
    for (int k = 0; k < (getenv("RANDBENCH_SKIP") ? atoi(getenv("RANDBENCH_SKIP")) : 0); ++k) {
        u64* other;
        if (hipMalloc(&other, max_slots * 8) != hipSuccess) { printf("hipMalloc %d failed\n", k); return 1; }
    }
    const int chunk_lg = getenv("RANDBENCH_CHUNK_LOG2") ? atoi(getenv("RANDBENCH_CHUNK_LOG2")) : 0;
    if (chunk_lg > 0) {
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || !gran) { printf("no granularity\n"); return 1; }
        size_t chunk = (size_t)1 << chunk_lg;
        if (chunk < gran) chunk = gran;
        const size_t total = (max_slots * 8 + chunk - 1) / chunk * chunk;
        void* va = nullptr;
        if (hipMemAddressReserve(&va, total, chunk, nullptr, 0) != hipSuccess) { printf("hipMemAddressReserve failed\n"); return 1; }
        // RANDBENCH_SPREAD=k: k chunks are created per chunk kept, the others released after the last one is mapped - the kept ones
        // then lie over k times the physical extent (does a table spread over more of the HBM take random writes faster still?)
        const int spread = getenv("RANDBENCH_SPREAD") ? atoi(getenv("RANDBENCH_SPREAD")) : 1;
        static hipMemGenericAllocationHandle_t extra[1 << 16];
        size_t n_extra = 0;
        for (size_t done = 0; done < total; done += chunk) {
            hipMemGenericAllocationHandle_t piece;
            if (hipMemCreate(&piece, chunk, &prop, 0) != hipSuccess || hipMemMap((char*)va + done, chunk, 0, piece, 0) != hipSuccess) { printf("chunk at %zu failed\n", done); return 1; }
            (void)hipMemRelease(piece);
            for (int s = 1; s < spread && n_extra < (1 << 16); ++s)
                if (hipMemCreate(&extra[n_extra], chunk, &prop, 0) == hipSuccess) ++n_extra; else { printf("# spread: out of memory after %zu extra chunks\n", n_extra); break; }
        }
        for (size_t k = 0; k < n_extra; ++k) (void)hipMemRelease(extra[k]);
        if (spread > 1) printf("# spread %d: %zu chunks created beside the table's and released\n", spread, n_extra);
        hipMemAccessDesc d{};
        d.location = prop.location;
        d.flags = hipMemAccessFlagsProtReadWrite;
        if (hipMemSetAccess(va, total, &d, 1) != hipSuccess) { printf("hipMemSetAccess failed\n"); return 1; }
        raw = (u64*)va;
        printf("# table of 2^%d slots mapped from %zu chunks of %zu MiB (granularity %zu KiB)\n", max_log2, total / chunk, chunk >> 20, gran >> 10);
    } else
    if (hipMalloc(&raw, max_slots * 8 + (align_gib ? (1ull << 30) : 0)) != hipSuccess) { printf("hipMalloc of 2^%d slots failed\n", max_log2); return 1; }
    table = align_gib ? (u64*)(((unsigned long long)raw + (1ull << 30) - 1) & ~((1ull << 30) - 1)) : raw;
    hipMalloc(&sink, 8);
    if (getenv("RANDBENCH_MAX_LOG2") || align_gib)
        printf("# allocation of 2^%d slots at %p (table at %p: aligned to 2^%d bytes)\n", max_log2, (void*)raw, (void*)table,
               __builtin_ctzll((unsigned long long)table));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int sizes[16] = {30}, nsizes = 1;
    if (argc > 2) { nsizes = 0; for (int i = 2; i < argc && nsizes < 16; ++i) sizes[nsizes++] = atoi(argv[i]); }
    for (int si = 0; si < nsizes; ++si) {
    const u64 slots = 1ull << (sizes[si] < 10 ? 10 : sizes[si] > max_log2 ? max_log2 : sizes[si]);
    if (nsizes > 1 || argc > 2) printf("# table of 2^%d slots = %.1f MiB\n", sizes[si], slots * 8 / 1048576.0);
    for (int mode = (argc > 1 ? atoi(argv[1]) : 0); mode < 14; ++mode)
        for (int bpc : {8}) {
            if (!want[mode]) continue;
            hipMemset(table, 0, slots * 8);
            const int blocks = 256 * bpc, iters = (mode == 0 || mode == 6) ? 400 : 800;
            k<<<blocks, 256>>>(table, slots - 1, 8, mode, sink);  // warm-up
            hipDeviceSynchronize();
            hipEventRecord(e0);
            k<<<blocks, 256>>>(table, slots - 1, iters, mode, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double n = (double)blocks * 256 * iters;
            printf("mode %d blocks/CU %d: %.1f M accesses in %.2f ms = %.1f G/s (%.2f TB/s of 64-B sectors)\n", mode, bpc,
                   n / 1e6, ms, n / ms / 1e6, n * 64 / ms / 1e9);
        }
    }
    return 0;
}
