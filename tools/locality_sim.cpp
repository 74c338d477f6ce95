// locality_sim — a CPU emulation of how k_expand's WAVES walk a BFS level, to price changes before building them:
//   (0) how many effect leaves a 64-state tile needs under the KIND-MAJOR walk (sum over action kinds of the largest
//       number of enabled bindings any lane has: every lane applies its own next binding per leaf) — 12.6 where the
//       instance-major walk dispatches 31: the number that started round 3's rewrite of pass 2,
//   (1) how many effect leaves a 64-state tile dispatches (distinct enabled action instances per tile), and
//   (2) how many seen-set probes a small per-wave filter of recently resolved fingerprints would answer,
// under today's frontier order (64-winner batches of thousands of concurrent waves interleaved into 8 segments, tiles
// dealt round-robin) and under a family-preserving order (every block appends to its OWN segment, waves take chunks of
// consecutive tiles).  The model templates are the device's own (kmc_device.h compiled for the host, as tests/host_emu.cpp
// does); the scheduling is emulated flush by flush: all waves advance round-robin, one 64-successor flush at a time.
//
//   g++ -O2 -std=c++17 -DKMC_HOST_EMU -o /tmp/locality_sim tools/locality_sim.cpp && /tmp/locality_sim
//
// TOOL, not product and not test: nothing links it.
#define KMC_HOST_EMU 1
#include "../kafka_specification_amd/csrc/kmc_device.h"

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unordered_set>
#include <vector>

#ifndef SIM_L
#define SIM_L 4
#endif
#ifndef SIM_R
#define SIM_R 4
#endif
using M = KmcKafka<KMC_MODEL_KIP320, 3, SIM_L, SIM_R, 2>;
constexpr int W = M::W;
using State = std::array<u64, W>;
struct SHash {
    size_t operator()(const State& s) const { return (size_t)kmc_fingerprint<W>(s.data(), 0); }
};

static double g_kind_major_leaves = 0;
struct Succ {
    State t;
    u64 fp;
};

// successors of a tile in the kernel's order: instance-major (pass 2 walks the instances, each dispatched leaf queues
// the successors of the lanes that enabled it)
static int expand_tile(const State* s, int n, std::vector<Succ>& out) {
    out.clear();
    int leaves = 0;
    int cnt[64][16];
    memset(cnt, 0, sizeof cnt);
    typename M::Pre pre[64];
    for (int l = 0; l < n; ++l) pre[l] = M::extract(s[l].data());
    kmc_static_for<0, M::NINST>([&](auto I) {
        bool any = false;
        for (int l = 0; l < n; ++l) {
            State t;
            int kind;
            u32 extra;
            if (M::template inst<decltype(I)::value>(pre[l], s[l].data(), t.data(), kind, extra)) {
                out.push_back({t, kmc_fingerprint<W>(t.data(), 0)});
                any = true;
                cnt[l][kind]++;
            }
        }
        leaves += any;
    });
    for (int k = 0; k < 16; ++k) {   // kind-major: a kind costs as many leaves as its busiest lane has enabled bindings
        int mx = 0;
        for (int l = 0; l < n; ++l) mx = std::max(mx, cnt[l][k]);
        g_kind_major_leaves += mx;
    }
    return leaves;
}

struct Wave {
    std::vector<long> tiles;  // tile indices this wave processes, in order
    size_t next_tile = 0;
    std::vector<Succ> succ;   // successors of the current tile still to flush
    size_t pos = 0;
    std::vector<State> stager;
    std::vector<u64> filter;  // direct-mapped recency filter (0 = empty)
    bool done() const { return next_tile >= tiles.size() && pos >= succ.size(); }
};

struct Stats {
    double tiles = 0, leaves = 0, probes = 0, filter_hits = 0, states = 0;
};

int main(int argc, char** argv) {
    const int NW = argc > 1 ? atoi(argv[1]) : 96;          // concurrent waves (4 per block)
    const int mode = argc > 2 ? atoi(argv[2]) : 0;         // 0 today's order, 1 per-block segments + chunked tiles,
                                                           // 2 / 3 / 4: the level SORTED by replica word 2 / 1 / 0 first (then the others), chunked tiles
    const int CH = argc > 3 ? atoi(argv[3]) : 4;           // tiles per chunk (mode 1)
    const int FS = argc > 4 ? atoi(argv[4]) : 256;         // filter entries per wave (0 = none)
    const double MAXS = argc > 5 ? atof(argv[5]) : 1e18;   // stop after the level that takes the total past this many states
    const int NSEG = mode == 0 ? 8 : mode == 1 ? NW / 4 : 1;
    if (mode >= 2 && (W != 3 || M::Y.rm != 1)) { fprintf(stderr, "modes 2-4 need one replica per word (build with -DSIM_L=6 -DSIM_R=6)\n"); return 1; }
    std::unordered_set<State, SHash> seen;
    std::vector<std::vector<State>> cur(NSEG), nxt(NSEG);
    State init;
    M::init(init.data());
    seen.insert(init);
    cur[0].push_back(init);
    Stats tot;
    for (int level = 1;; ++level) {
        // tiles: (segment, offset) in segment order
        std::vector<std::pair<int, long>> tiles;
        for (int sg = 0; sg < NSEG; ++sg)
            for (long o = 0; o < (long)cur[sg].size(); o += 64) tiles.push_back({sg, o});
        if (tiles.empty()) break;
        if (mode >= 2) {
            // states that differ in ONE replica only — the parents of a common successor differ in two — end up near each other
            // when the level is ordered by the other replicas' words (replica-major layout: word r = replica r)
            const int a = 4 - mode, b = (a + 1) % 3, c = (a + 2) % 3;
            std::sort(cur[0].begin(), cur[0].end(), [&](const State& x, const State& y) {
                return x[a] != y[a] ? x[a] < y[a] : x[b] != y[b] ? x[b] < y[b] : x[c] < y[c];
            });
        }
        std::vector<Wave> waves(NW);
        if (mode == 0) {
            // per segment, tiles dealt round-robin to all waves (rotated start per segment, like the kernel)
            long t = 0;
            for (auto& tl : tiles) { waves[t % NW].tiles.push_back(&tl - tiles.data()); ++t; }
        } else {
            for (long c = 0; c * CH < (long)tiles.size(); ++c)
                for (long k = c * CH; k < std::min<long>((c + 1) * CH, tiles.size()); ++k) waves[c % NW].tiles.push_back(k);
        }
        for (auto& w : waves) w.filter.assign(FS ? FS : 1, 0);
        Stats lv;
        bool active = true;
        while (active) {
            active = false;
            for (int wi = 0; wi < NW; ++wi) {
                Wave& w = waves[wi];
                if (w.done()) continue;
                active = true;
                if (w.pos >= w.succ.size()) {  // fetch the next tile
                    auto [sg, off] = tiles[w.tiles[w.next_tile++]];
                    const int n = (int)std::min<long>(64, (long)cur[sg].size() - off);
                    lv.leaves += expand_tile(&cur[sg][off], n, w.succ);
                    lv.tiles += 1;
                    lv.states += n;
                    w.pos = 0;
                    if (w.succ.empty()) continue;
                }
                // one flush: up to 64 successors
                const size_t end = std::min(w.pos + 64, w.succ.size());
                for (; w.pos < end; ++w.pos) {
                    const Succ& s = w.succ[w.pos];
                    lv.probes += 1;
                    if (FS) {
                        u64& slot = w.filter[(s.fp >> 20) % FS];
                        if (slot == s.fp) { lv.filter_hits += 1; continue; }
                        slot = s.fp;
                    }
                    if (seen.insert(s.t).second) w.stager.push_back(s.t);
                }
                const int seg = mode == 0 ? (wi / 4) % 8 : mode == 1 ? wi / 4 : 0;
                while (w.stager.size() >= 64 || (w.done() && !w.stager.empty())) {
                    const size_t n = std::min<size_t>(64, w.stager.size());
                    nxt[seg].insert(nxt[seg].end(), w.stager.begin(), w.stager.begin() + n);
                    w.stager.erase(w.stager.begin(), w.stager.begin() + n);
                }
            }
        }
        long produced = 0;
        for (int sg = 0; sg < NSEG; ++sg) { produced += nxt[sg].size(); cur[sg].swap(nxt[sg]); nxt[sg].clear(); }
        if (lv.states >= 20000)
            printf("level %2d: %8.0f states %6.0f tiles  leaves/tile %5.1f  filter hits %4.1f %% of %9.0f probes\n", level, lv.states,
                   lv.tiles, lv.leaves / lv.tiles, 100.0 * lv.filter_hits / std::max(1.0, lv.probes), lv.probes);
        tot.tiles += lv.tiles; tot.leaves += lv.leaves; tot.probes += lv.probes; tot.filter_hits += lv.filter_hits; tot.states += lv.states;
        if (!produced || tot.states > MAXS) break;
    }
    printf("kind-major walk: %.2f effect leaves per tile (instance-major: %.2f)\n", g_kind_major_leaves / tot.tiles, tot.leaves / tot.tiles);
    printf("TOTAL mode %d NW %d CH %d FS %d: %.0f states, %.0f tiles (%.1f states/tile), leaves/tile %.2f, filter hits %.2f %% of %.0f probes\n",
           mode, NW, CH, FS, tot.states, tot.tiles, tot.states / tot.tiles, tot.leaves / tot.tiles, 100.0 * tot.filter_hits / tot.probes, tot.probes);
    return 0;
}
