// host_emu_bfs.cpp — a whole breadth-first search over the DEVICE model templates compiled for the host (tests/host_emu.cpp), as a
// stand-alone program, so that it can run under -fsanitize=address,undefined (tests/test_sanitizers_cpu.py: SURVEY section 5 asks
// for sanitizer runs of everything that can run on a CPU; an ASan library cannot be loaded into an uninstrumented Python).
// TEST INFRASTRUCTURE ONLY.   usage: host_emu_bfs MODEL N L R E K INV_MASK [LAYOUT_MODE]   ->   one JSON line
#include "host_emu.cpp"

#include <cstdio>
#include <cstdlib>
#include <string>
#include <unordered_set>
#include <vector>

int main(int argc, char** argv) {
    if (argc < 8) { fprintf(stderr, "usage: host_emu_bfs MODEL N L R E K INV_MASK\n"); return 2; }
    const int model = atoi(argv[1]), N = atoi(argv[2]), L = atoi(argv[3]), R = atoi(argv[4]), E = atoi(argv[5]), K = atoi(argv[6]);
    const unsigned mask = (unsigned)strtoul(argv[7], nullptr, 0);
    if (argc > 8) emu_layout(atoi(argv[8]));   // KMC_LAYOUT_* of the entry meant (0 = the automatic choice)
    const int W = emu_words(model, N, L, R, E, K);
    if (W <= 0) { fprintf(stderr, "configuration not compiled into tests/host_emu.cpp\n"); return 2; }
    std::vector<u64> init(W);
    emu_init(model, N, L, R, E, K, init.data());
    auto key = [&](const u64* w) { return std::string((const char*)w, (size_t)W * 8); };
    std::unordered_set<std::string> seen{key(init.data())};
    std::vector<std::vector<u64>> frontier{init};
    unsigned long long generated = 1, violating = 0, deadlocks = 0, depth = 0, kind_checked = 0, kind_bad = 0;
    std::vector<u64> out((size_t)4096 * (W + 1));
    std::string levels;
    while (!frontier.empty()) {
        ++depth;
        levels += (levels.empty() ? "" : ",") + std::to_string(frontier.size());
        std::vector<std::vector<u64>> next;
        for (const std::vector<u64>& s : frontier) {
            if (emu_violated(model, N, L, R, E, K, s.data(), mask) != 0) ++violating;
            int chk = 0, bad = 0;   // pass 2's two lowerings (apply<K> against inst<I>) on every enabled binding of every state
            if (emu_kind_major_check(model, N, L, R, E, K, s.data(), &chk, &bad) == 1) { kind_checked += chk; kind_bad += bad; }
            const int n = emu_successors(model, N, L, R, E, K, s.data(), out.data(), 4096);
            if (n < 0 || n > 4096) { fprintf(stderr, "successor list overflow\n"); return 3; }
            if (n == 0) ++deadlocks;
            generated += (unsigned long long)n;
            for (int i = 0; i < n; ++i) {
                const u64* t = &out[(size_t)i * (W + 1)];
                if (seen.insert(key(t)).second) next.emplace_back(t, t + W);
            }
        }
        frontier.swap(next);
    }
    printf("{\"distinct\": %zu, \"generated\": %llu, \"depth\": %llu, \"violating_states\": %llu, \"deadlock_states\": %llu, "
           "\"kind_major_bindings_checked\": %llu, \"kind_major_bindings_bad\": %llu, \"levels\": [%s]}\n",
           seen.size(), generated, depth, violating, deadlocks, kind_checked, kind_bad, levels.c_str());
    return 0;
}
