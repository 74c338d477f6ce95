// ddd_prims — go / no-go measurement for "delayed duplicate detection" (VERDICT r02, next-round item 2a): the three
// primitives a sort-merge seen-set would be made of, on one synthetic BFS level, so that their measured cost can be put
// beside the hash-probe path's (one 128-byte line fill per generated successor, kmc_device.h KmcSink::claim).
//
//   1. PARTITION  n records of 32 bytes (8-B fingerprint + 24-B packed state) into B fingerprint-range buckets:
//      per block of 4096 records an LDS histogram, ONE global atomicAdd per (block, bucket) to reserve the bucket's run,
//      then the records are written to their runs (a run is 4096/B records: 512 B at B = 256, 128 B at B = 1024).
//      This is the write-combining radix pass; a level whose buckets must fit an LDS sort (<= 2 K records) needs
//      log_B(n / 2K) of them.
//   2. LDS SORT   buckets of 2048 (fingerprint, index) pairs, bitonic in LDS, one bucket per 256-thread block.
//   3. MERGE      every sorted bucket against a sorted, compact known set (8 B per known state) streamed once:
//      binary-search-free two-pointer merge per block over its fingerprint range; counts survivors.
// Reports GB/s (bytes read + written / time) and ms for the given n, and the sum priced for one headline-sized level.
//
// Build: hipcc --offload-arch=gfx950 -O3 ddd_prims.hip -o ddd_prims      Usage: ddd_prims [log2_n=26] [known_millions=250]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
__device__ __forceinline__ u64 mix(u64 x) { x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31; return x; }

struct Rec { u64 fp, w0, w1, w2; };

__global__ __launch_bounds__(256) void k_gen(Rec* r, u64 n) {
    for (u64 i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
        const u64 f = mix(i * 0x9E3779B97F4A7C15ull + 7);
        r[i] = Rec{f, f ^ 1, f ^ 2, f ^ 3};
    }
}

// ---- 1. partition ----------------------------------------------------------------------------------------------
#define TILE 4096
template <int LOGB> __global__ __launch_bounds__(256) void k_partition(const Rec* in, Rec* out, u64 n, u64* bucket_fill, u64 bucket_cap, int shift) {
    constexpr int B = 1 << LOGB;
    __shared__ u32 hist[B];
    __shared__ u64 base[B];
    const u64 tiles = (n + TILE - 1) / TILE;
    for (u64 t = blockIdx.x; t < tiles; t += gridDim.x) {
        for (int b = threadIdx.x; b < B; b += 256) hist[b] = 0;
        __syncthreads();
        Rec r[TILE / 256];
        u32 rank[TILE / 256];
#pragma unroll
        for (int k = 0; k < TILE / 256; ++k) {
            const u64 i = t * TILE + k * 256 + threadIdx.x;
            if (i < n) {
                r[k] = in[i];
                rank[k] = atomicAdd(&hist[(r[k].fp >> shift) & (B - 1)], 1u);
            }
        }
        __syncthreads();
        for (int b = threadIdx.x; b < B; b += 256) base[b] = hist[b] ? atomicAdd(&bucket_fill[b * 16], (u64)hist[b]) : 0;  // one counter per 128-B line
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TILE / 256; ++k) {
            const u64 i = t * TILE + k * 256 + threadIdx.x;
            if (i < n) {
                const u32 b = (r[k].fp >> shift) & (B - 1);
                const u64 pos = base[b] + rank[k];
                if (pos < bucket_cap) out[(u64)b * bucket_cap + pos] = r[k];
            }
        }
        __syncthreads();
    }
}

// ---- 2. LDS bitonic sort of 2048 (fp, idx) pairs per block -----------------------------------------------------------
#define SORTN 2048
__global__ __launch_bounds__(256) void k_lds_sort(const Rec* in, u64* out_fp, u32* out_idx, u64 nbuckets) {
    __shared__ u64 key[SORTN];
    __shared__ u32 val[SORTN];
    for (u64 b = blockIdx.x; b < nbuckets; b += gridDim.x) {
        for (int i = threadIdx.x; i < SORTN; i += 256) { key[i] = in[b * SORTN + i].fp; val[i] = (u32)i; }
        __syncthreads();
        for (int k = 2; k <= SORTN; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = threadIdx.x; i < SORTN; i += 256) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const bool up = (i & k) == 0;
                        const u64 a = key[i], c = key[ixj];
                        if ((a > c) == up) { key[i] = c; key[ixj] = a; const u32 t = val[i]; val[i] = val[ixj]; val[ixj] = t; }
                    }
                }
                __syncthreads();
            }
        for (int i = threadIdx.x; i < SORTN; i += 256) { out_fp[b * SORTN + i] = key[i]; out_idx[b * SORTN + i] = val[i]; }
        __syncthreads();
    }
}

// ---- 3. merge: a sorted run of candidates against the sorted known set of its fingerprint range ------------------------
// (both sides streamed once; the known side is what costs: 8 B per known state per level)
__global__ __launch_bounds__(256) void k_merge(const u64* cand, u64 ncand_per_block, const u64* known, u64 nknown_per_block, u64 nblocks, u64* survivors) {
    __shared__ u64 kn[2048];
    for (u64 b = blockIdx.x; b < nblocks; b += gridDim.x) {
        u64 cnt = 0;
        // stream the block's slice of the known set through LDS in 2048-entry chunks; each thread binary-searches its candidates in the chunk
        for (u64 off = 0; off < nknown_per_block; off += 2048) {
            for (int i = threadIdx.x; i < 2048; i += 256) kn[i] = off + i < nknown_per_block ? known[b * nknown_per_block + off + i] : ~0ull;
            __syncthreads();
            for (u64 c = threadIdx.x; c < ncand_per_block; c += 256) {
                const u64 f = cand[b * ncand_per_block + c];
                int lo = 0, hi = 2047;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (kn[mid] < f) lo = mid + 1; else hi = mid; }
                cnt += kn[lo] == f;
            }
            __syncthreads();
        }
        if (cnt) atomicAdd(survivors, cnt);
    }
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CHECK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main(int argc, char** argv) {
    const int log2n = argc > 1 ? atoi(argv[1]) : 26;
    const u64 known_m = argc > 2 ? strtoull(argv[2], 0, 10) : 250;
    const u64 n = 1ull << log2n;
    Rec *in, *out;
    CHECK(hipMalloc(&in, n * sizeof(Rec)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_gen<<<2048, 256>>>(in, n);
    CHECK(hipDeviceSynchronize());
    printf("# ddd_prims: n = 2^%d = %llu records of 32 B (%.2f GB), known set %llu M fingerprints\n", log2n, n, n * 32e-9, known_m);
    double part_ms_256 = 0;
    for (int logb : {6, 8, 10, 12}) {
        const u64 B = 1ull << logb, cap = (n / B) * 5 / 4 + 4096;
        u64* fill;
        CHECK(hipMalloc(&out, B * cap * sizeof(Rec)));
        CHECK(hipMalloc(&fill, B * 128));
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemset(fill, 0, B * 128));
            CHECK(hipEventRecord(e0));
            const int grid = 256 * 6;
            if (logb == 6) k_partition<6><<<grid, 256>>>(in, out, n, fill, cap, 64 - 6);
            if (logb == 8) k_partition<8><<<grid, 256>>>(in, out, n, fill, cap, 64 - 8);
            if (logb == 10) k_partition<10><<<grid, 256>>>(in, out, n, fill, cap, 64 - 10);
            if (logb == 12) k_partition<12><<<grid, 256>>>(in, out, n, fill, cap, 64 - 12);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            best = fminf(best, time_ms(e0, e1));
        }
        printf("partition  B = %4llu buckets (runs of %4llu B): %7.3f ms  %7.1f GB/s (read + write)  %6.2f G records/s\n", B,
               TILE / B * 32ull, best, 2.0 * n * 32 / best * 1e-6, n / best * 1e-6);
        if (logb == 8) part_ms_256 = best;
        CHECK(hipFree(out)); CHECK(hipFree(fill));
    }
    // LDS sort
    u64* sfp; u32* sidx;
    CHECK(hipMalloc(&sfp, n * 8)); CHECK(hipMalloc(&sidx, n * 4));
    float sort_ms = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        k_lds_sort<<<256 * 8, 256>>>(in, sfp, sidx, n / SORTN);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        sort_ms = fminf(sort_ms, time_ms(e0, e1));
    }
    printf("lds sort   %llu buckets of %d (fp, idx): %7.3f ms  %6.2f G keys/s\n", n / SORTN, SORTN, sort_ms, n / sort_ms * 1e-6);
    // merge: n candidates (sorted within blocks) against known_m M known fingerprints, both cut into the same nblocks ranges
    const u64 nknown = known_m * 1000000ull, nblocks = n / SORTN;
    u64* known; u64* surv;
    CHECK(hipMalloc(&known, (nknown + nblocks * 2048) * 8)); CHECK(hipMalloc(&surv, 8));
    CHECK(hipMemset(known, 0x7f, (nknown + nblocks * 2048) * 8)); CHECK(hipMemset(surv, 0, 8));
    float merge_ms = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        k_merge<<<256 * 8, 256>>>(sfp, SORTN, known, nknown / nblocks, nblocks, surv);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        merge_ms = fminf(merge_ms, time_ms(e0, e1));
    }
    printf("merge      %llu sorted runs of %d against %llu M known fingerprints: %7.3f ms  %7.1f GB/s of known set\n", nblocks, SORTN,
           known_m, merge_ms, nknown * 8 / merge_ms * 1e-6);
    // price one wide level of the headline: 60 M successors -> 2 partition passes at B = 256 (to reach <= 2 K per bucket), sort, merge
    const double scale = 60e6 / (double)n;
    printf("# one wide headline level (60 M successors, 250 M known): 2 x partition(256) %.2f ms + sort %.2f ms + merge %.2f ms = %.2f ms;"
           " the hash-probe kernel spends ~2.4 ms on such a level (60 M probes at ~25 G/s incl. claims)\n",
           2 * part_ms_256 * scale, sort_ms * scale, merge_ms, 2 * part_ms_256 * scale + sort_ms * scale + merge_ms);
    return 0;
}
