// fp_quality.cpp — how good is kmc_fingerprint<W> on REAL states?  A breadth-first search over the device model templates compiled
// for the host (tests/_host_emu.so) collects the first n distinct packed states of a configuration exactly (full states in a set),
// then the fingerprint function of csrc/kmc_common.h — as compiled: -DKMC_FOLD_MIN_WORDS=... selects the form — is truncated to b
// bits at several positions and its collisions among the n states are counted against the birthday expectation n^2 / 2^(b+1).
// A systematic weakness on structured states (round 1's one-multiply absorb lost 32 of 75 M states) shows up as a window whose
// collision count is far above expectation.  TEST / TUNING INFRASTRUCTURE.
//   g++ -O2 -std=c++17 -DKMC_HOST_EMU -I kafka_specification_amd/csrc [-DKMC_FOLD_MIN_WORDS=1000: the per-word chain] [-DFPQ_RAW: the folded chain before its finaliser] tools/fp_quality/fp_quality.cpp tests/_host_emu.so -o /tmp/fpq
//   /tmp/fpq MODEL N L R E K MAX_STATES
#include "kmc_common.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <unordered_set>
#include <vector>

extern "C" {
int emu_words(int, int, int, int, int, int);
int emu_init(int, int, int, int, int, int, u64*);
int emu_successors(int, int, int, int, int, int, const u64*, u64*, int);
}

#ifdef FPQ_RAW
// -DFPQ_RAW: the folded chain WITHOUT its final mix64 — the finaliser is a bijection that would make any 64-bit value look uniform
// in every window; what is looked at here is what the multiplies themselves leave in each window (restated from kmc_fingerprint)
template <int W> static u64 fp_of(const u64* w, u64 seed) {
    u64 h = kmc_mix64(seed + 0x9E3779B97F4A7C15ull * (u64)(W + 1));
    for (int k = 0; k + 1 < W; k += 2) h = kmc_mum(w[k] ^ 0xe7037ed1a0b428dbull, w[k + 1] ^ h) + 0x9E3779B97F4A7C15ull;
    if (W & 1) h = kmc_mum(w[W - 1] ^ 0xe7037ed1a0b428dbull, h ^ 0x8ebc6af09c88c6e3ull);
    return h;
}
#else
template <int W> static u64 fp_of(const u64* w, u64 seed) { return kmc_fingerprint<W>(w, seed); }
#endif
static u64 fp_any(int W, const u64* w, u64 seed) {
    switch (W) {
    case 3: return fp_of<3>(w, seed);
    case 4: return fp_of<4>(w, seed);
    case 8: return fp_of<8>(w, seed);
    case 9: return fp_of<9>(w, seed);
    case 10: return fp_of<10>(w, seed);
    case 11: return fp_of<11>(w, seed);
    default: fprintf(stderr, "W = %d not instantiated\n", W); exit(2);
    }
}

int main(int argc, char** argv) {
    if (argc < 8) return 2;
    const int model = atoi(argv[1]), N = atoi(argv[2]), L = atoi(argv[3]), R = atoi(argv[4]), E = atoi(argv[5]), K = atoi(argv[6]);
    const size_t max_states = (size_t)atof(argv[7]);
    const int W = emu_words(model, N, L, R, E, K);
    if (W <= 0) return 2;
    std::vector<u64> init(W);
    emu_init(model, N, L, R, E, K, init.data());
    auto key = [&](const u64* w) { return std::string((const char*)w, (size_t)W * 8); };
    std::unordered_set<std::string> seen{key(init.data())};
    std::vector<u64> all(init), frontier(init), out((size_t)4096 * (W + 1));
    all.reserve((max_states + 4096) * W);
    seen.reserve(max_states + 4096);
    int depth = 0;
    while (!frontier.empty() && seen.size() < max_states) {
        ++depth;
        std::vector<u64> next;
        for (size_t s = 0; s < frontier.size() && seen.size() < max_states; s += W) {
            const int n = emu_successors(model, N, L, R, E, K, &frontier[s], out.data(), 4096);
            for (int i = 0; i < n; ++i) {
                const u64* t = &out[(size_t)i * (W + 1)];
                if (seen.insert(key(t)).second) { next.insert(next.end(), t, t + W); all.insert(all.end(), t, t + W); }
            }
        }
        frontier.swap(next);
    }
    const size_t n = all.size() / W;
    printf("W = %d, %zu distinct states over %d levels, KMC_FOLD_MIN_WORDS = %d (%s)\n", W, n, depth, (int)KMC_FOLD_MIN_WORDS,
           W >= KMC_FOLD_MIN_WORDS ? "two words per 64 x 64 -> 128 multiply" : "one mix64 per word");
    for (u64 seed : {0ull, 0x5EED2ull, 0x6a09e667f3bcc908ull}) {
        std::vector<u64> fp(n);
        for (size_t i = 0; i < n; ++i) fp[i] = fp_any(W, &all[i * W], seed);
        {   // the full 64 bits: any collision at all among n <= 1e8 states would be a defect (expectation n^2 / 2^65 < 3e-4)
            std::vector<u64> s(fp);
            std::sort(s.begin(), s.end());
            size_t c = 0;
            for (size_t i = 1; i < n; ++i) c += s[i] == s[i - 1];
            printf("seed %llx: 64-bit collisions %zu\n", seed, c);
        }
        for (int bits : {32, 36, 40}) {
            for (int shift : {0, 8, 12, 24, 64 - bits}) {
                if (shift + bits > 64) continue;
                std::vector<u64> s(n);
                const u64 m = bits == 64 ? ~0ull : ((1ull << bits) - 1);
                for (size_t i = 0; i < n; ++i) s[i] = (fp[i] >> shift) & m;
                std::sort(s.begin(), s.end());
                size_t c = 0;
                for (size_t i = 1; i < n; ++i) c += s[i] == s[i - 1];
                const double expect = (double)n * (double)n / std::ldexp(1.0, bits + 1);
                printf("  bits %2d..%2d: collisions %9zu  expected %11.1f  ratio %.3f\n", shift, shift + bits - 1, c, expect, c / expect);
            }
        }
        // the seen-set's view: occupancy of the 64-slot groups of a table at load ~0.5 (kmc_slot_of) - a chi-square over 4096 buckets
        {
            const u64 cap = ((u64)(2 * n) + 63) / 64 * 64;
            std::vector<double> h(4096, 0.0);
            for (size_t i = 0; i < n; ++i) h[(size_t)((__uint128_t)(kmc_slot_of(fp[i], cap) >> 6) * 4096 / (cap >> 6))] += 1;
            double chi = 0, e = (double)n / 4096;
            for (double x : h) chi += (x - e) * (x - e) / e;
            printf("  slot groups: chi-square over 4096 buckets %.0f (expected ~4095 +- 90)\n", chi);
        }
    }
    return 0;
}
