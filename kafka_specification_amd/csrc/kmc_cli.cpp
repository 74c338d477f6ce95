// kmc_cli.cpp — native `tlc`-shaped front end over the C ABI (include/kmc.h); the C++ twin of
// kafka_specification_amd/tlc.py, for hosts without Python/torch.
//
//   tlc [-config X.cfg] [-deadlock] [-continue] [-workers N] [-fp SEED] [-fp128] [-symmetry] [-fpcheck] [-verify] [-force] [-levels-csv FILE] [-table SLOTS]
//       [-frontier STATES] [-device D] [-notrace] [-v] Spec.tla
//
// [TLC-recall] flag names and output lines follow tlc2.TLC; TLC itself is not part of the
// reference repository.  The module name selects one of the lowered models; constants
// and invariants come from the .cfg (CONSTANT(S), INIT, NEXT, SPECIFICATION, INVARIANT(S),
// CHECK_DEADLOCK).  No TLA+ is parsed — therefore the spec file given (and every module it EXTENDS / INSTANCEs that
// is found beside it) is hashed against the revision the kernels were lowered from; an edited spec is refused unless
// -force (see kafka_specification_amd/spec_revision.py, the same table).  -fpcheck repeats the search with a second
// fingerprint seed and compares the counts; the summary prints TLC's collision-probability estimate.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <cmath>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../include/kmc.h"

static double wall_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

namespace {

// ---- spec identity: sha256 of the reference's modules (sha256sum /root/reference/*.tla) ------------------------
const char* const SPEC_SHA256[][2] = {
    {"AsyncIsr", "6c29fa0de4174c9a36b1b933b43df115f53254f727cdf6b05b1c6306e77f3e46"},
    {"FiniteReplicatedLog", "cc6193fcc9bc75be6ce2b4ab9719d9bb645c0e0441e53fda630097e13edde747"},
    {"IdSequence", "bf8a9f74c8ccad8aa2ee4139df4297942f9a8a429afffb2e1f032833b6912654"},
    {"KafkaReplication", "80fffc6c9e430106cb0be059f542239fa85079ffe5834493f96cbadfe122ed9a"},
    {"KafkaTruncateToHighWatermark", "97b38e8c1d41113173784cee59c884b54554a9608ac7ddad442f80eed7af6d35"},
    {"Kip101", "77a8ea39557ae7e19c975eb62cc1b4253761a22426c562c1e891fc361f918f6f"},
    {"Kip279", "9568ffd395b77367722757b8a3aab999b682ab657e4fb13d41629c3696b4e39f"},
    {"Kip320", "4f2dccca13414af26dbebeffbdb105b46973f55c79a46c04be76cde73aa963ca"},
    {"Kip320FirstTry", "b2985bc5dad379d15cc88e838f0e3bce8e232a52a197c7f5dcd941821467d705"},
    {"Util", "56a9514e28ac684657a91456f1f8985351bf23c9ec29d65e69a65987cfc9b385"},
};

std::string sha256_hex(const std::string& data) {  // FIPS 180-4
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
        0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
        0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
        0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
        0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
        0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    std::string m = data;
    const uint64_t bits = (uint64_t)data.size() * 8;
    m += (char)0x80;
    while (m.size() % 64 != 56) m += (char)0;
    for (int i = 7; i >= 0; --i) m += (char)(bits >> (8 * i));
    auto rotr = [](uint32_t x, int n) { return (x >> n) | (x << (32 - n)); };
    for (size_t off = 0; off < m.size(); off += 64) {
        uint32_t w[64];
        for (int i = 0; i < 16; ++i)
            w[i] = ((uint32_t)(unsigned char)m[off + 4 * i] << 24) | ((uint32_t)(unsigned char)m[off + 4 * i + 1] << 16) |
                   ((uint32_t)(unsigned char)m[off + 4 * i + 2] << 8) | (uint32_t)(unsigned char)m[off + 4 * i + 3];
        for (int i = 16; i < 64; ++i) {
            const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
            const uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; ++i) {
            const uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g);
            const uint32_t t1 = hh + S1 + ch + K[i] + w[i];
            const uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), maj = (a & b) ^ (a & c) ^ (b & c);
            const uint32_t t2 = S0 + maj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    char out[65];
    for (int i = 0; i < 8; ++i) snprintf(out + 8 * i, 9, "%08x", h[i]);
    return out;
}

bool slurp(const std::string& path, std::string* out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[4096];
    size_t n;
    out->clear();
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out->append(buf, n);
    fclose(f);
    return true;
}

std::string strip_comments(const std::string& in);

// 0 = every module of the closure that exists matches, 1 = the root module file is absent, 2 = mismatch
int check_spec(const std::string& path, const std::string& self_exe) {
    std::string text;
    if (!slurp(path, &text)) {
        fprintf(stderr, "Warning: %s does not exist: checking the built-in lowering (reference revision, sha256 table in "
                        "the front end)\n", path.c_str());
        return 1;
    }
    const size_t slash = path.find_last_of('/');
    const std::string dir = slash == std::string::npos ? "." : path.substr(0, slash);
    std::string root = path.substr(slash == std::string::npos ? 0 : slash + 1);
    if (root.size() > 4) root.resize(root.size() - 4);
    static const std::set<std::string> standard = {"Integers", "Naturals", "Sequences", "FiniteSets", "TLC", "Reals", "Bags"};
    std::vector<std::string> todo = {root};
    std::set<std::string> seen;
    bool bad = false;
    while (!todo.empty()) {
        const std::string mod = todo.back();
        todo.pop_back();
        if (seen.count(mod) || standard.count(mod)) continue;
        seen.insert(mod);
        std::string data;
        const std::string file = dir + "/" + mod + ".tla";
        if (!slurp(file, &data)) {
            fprintf(stderr, "Warning: module %s: %s not found (not verified)\n", mod.c_str(), file.c_str());
            continue;
        }
        std::string want;
        for (const auto& e : SPEC_SHA256)
            if (mod == e[0]) want = e[1];
        if (want.empty() && mod == "MCAsyncIsr") {  // authored in this repository: compare with the copy beside the binary
            std::string mine;
            const size_t k = self_exe.find_last_of('/');
            const std::string base = k == std::string::npos ? "." : self_exe.substr(0, k);
            if (slurp(base + "/../models/MCAsyncIsr.tla", &mine)) want = sha256_hex(mine);
        }
        const std::string got = sha256_hex(data);
        if (want.empty()) {
            fprintf(stderr, "Error: module %s: not a module of the lowered revision\n", mod.c_str());
            bad = true;
        } else if (want != got) {
            fprintf(stderr, "Error: module %s: %s differs from the revision the kernels were lowered from (sha256 %.16s..., "
                            "expected %.16s...)\n", mod.c_str(), file.c_str(), got.c_str(), want.c_str());
            bad = true;
        }
        const std::string t = strip_comments(data);
        for (size_t at = t.find("EXTENDS"); at != std::string::npos; at = t.find("EXTENDS", at + 7)) {
            size_t e = t.find('\n', at);
            std::string line = t.substr(at + 7, (e == std::string::npos ? t.size() : e) - at - 7), name;
            for (char ch : line + ",") {
                if (isalnum((unsigned char)ch) || ch == '_') name += ch;
                else { if (!name.empty()) todo.push_back(name); name.clear(); }
            }
        }
        for (size_t at = t.find("INSTANCE"); at != std::string::npos; at = t.find("INSTANCE", at + 8)) {
            size_t i = at + 8;
            while (i < t.size() && isspace((unsigned char)t[i])) ++i;
            std::string name;
            while (i < t.size() && (isalnum((unsigned char)t[i]) || t[i] == '_')) name += t[i++];
            if (!name.empty()) todo.push_back(name);
        }
    }
    return bad ? 2 : 0;
}

struct Cfg {
    std::map<std::string, std::string> constants;  // raw value text ("{b1, b2}" or "6")
    std::vector<std::string> invariants, constraints;
    int check_deadlock = 1;  // TLC's default
    bool has_init_next = false, has_specification = false;
    std::string error;
};

std::string strip_comments(const std::string& in) {
    std::string out;
    for (size_t i = 0; i < in.size();) {
        if (in.compare(i, 2, "(*") == 0) {
            size_t e = in.find("*)", i + 2);
            i = e == std::string::npos ? in.size() : e + 2;
            out += ' ';
        } else if (in.compare(i, 2, "\\*") == 0) {
            while (i < in.size() && in[i] != '\n') ++i;
        } else {
            out += in[i++];
        }
    }
    return out;
}

std::vector<std::string> tokenize(const std::string& s) {
    std::vector<std::string> t;
    for (size_t i = 0; i < s.size();) {
        if (isspace((unsigned char)s[i])) { ++i; continue; }
        if (s[i] == '{') {
            size_t e = s.find('}', i);
            if (e == std::string::npos) e = s.size() - 1;
            t.push_back(s.substr(i, e - i + 1));
            i = e + 1;
        } else if (s[i] == '=') {
            t.push_back("=");
            ++i;
        } else if (s.compare(i, 2, "<-") == 0) {
            t.push_back("<-");
            i += 2;
        } else {
            size_t e = i;
            while (e < s.size() && !isspace((unsigned char)s[e]) && s[e] != '=' && s[e] != '{' && s.compare(e, 2, "<-") != 0) ++e;
            t.push_back(s.substr(i, e - i));
            i = e;
        }
    }
    return t;
}

bool is_keyword(const std::string& w) {
    static const char* kw[] = {"CONSTANT", "CONSTANTS", "INIT", "NEXT", "SPECIFICATION", "INVARIANT", "INVARIANTS",
                               "CHECK_DEADLOCK", "SYMMETRY", "CONSTRAINT", "CONSTRAINTS", "ACTION_CONSTRAINT", "VIEW",
                               "PROPERTY", "PROPERTIES", "ALIAS", "POSTCONDITION"};
    for (const char* k : kw)
        if (w == k) return true;
    return false;
}

Cfg parse_cfg(const std::string& text) {
    Cfg c;
    const std::vector<std::string> t = tokenize(strip_comments(text));
    std::string section;
    for (size_t i = 0; i < t.size(); ++i) {
        const std::string& w = t[i];
        if (is_keyword(w)) {
            if (w == "SYMMETRY" || w == "VIEW" || w == "ACTION_CONSTRAINT" ||
                w.rfind("PROPERT", 0) == 0 || w == "ALIAS" || w == "POSTCONDITION") {
                c.error = w + " is not supported (it changes the distinct-state count or asks for liveness)";
                return c;
            }
            section = w;
            continue;
        }
        if (section == "CONSTANT" || section == "CONSTANTS") {
            if (i + 1 < t.size() && t[i + 1] == "<-") {
                c.error = "`" + w + " <- ...` (substitution by an operator of the spec) is not supported: no TLA+ is parsed, so a "
                          "replaced definition cannot be honoured; assign a value with `=`";
                return c;
            }
            if (i + 2 < t.size() + 0 && t[i + 1] == "=") {
                c.constants[w] = t[i + 2];
                i += 2;
            } else {
                c.error = "expected `" + w + " = value` in CONSTANTS";
                return c;
            }
        } else if (section == "INIT") {
            if (w != "Init") c.error = "only INIT Init is known";
            c.has_init_next = true;
        } else if (section == "NEXT") {
            if (w != "Next") c.error = "only NEXT Next is known";
            c.has_init_next = true;
        } else if (section == "SPECIFICATION") {
            if (w != "Spec") c.error = "only SPECIFICATION Spec is known";
            c.has_specification = true;
        } else if (section == "INVARIANT" || section == "INVARIANTS") {
            c.invariants.push_back(w);
        } else if (section == "CONSTRAINT" || section == "CONSTRAINTS") {
            c.constraints.push_back(w);  // only MCAsyncIsr's StateConstraint is lowered (checked by main)
        } else if (section == "CHECK_DEADLOCK") {
            c.check_deadlock = w == "TRUE";
        } else {
            c.error = "unexpected token " + w;
        }
        if (!c.error.empty()) return c;
    }
    // [TLC-recall] TLC refuses the combination too (EC.TLC_CONFIG_NOT_BOTH_SPEC_AND_INIT)
    if (c.has_init_next && c.has_specification) c.error = "a .cfg gives either SPECIFICATION or INIT / NEXT, not both";
    return c;
}

int set_size(const std::string& v) {  // "{a, b, c}" -> 3
    if (v.size() < 2 || v[0] != '{') return -1;
    int n = 0;
    bool in = false;
    for (char ch : v.substr(1, v.size() - 2)) {
        if (ch == ',') in = false;
        else if (!isspace((unsigned char)ch) && !in) { in = true; ++n; }
    }
    return n;
}

std::string now() {
    char buf[64];
    time_t t = time(nullptr);
    strftime(buf, sizeof buf, "%Y-%m-%d %H:%M:%S", localtime(&t));
    return buf;
}

std::string bitset_names(unsigned mask, int n, const char* prefix) {
    std::string s = "{";
    bool first = true;
    for (int r = 0; r < n; ++r)
        if (mask >> r & 1u) {
            if (!first) s += ", ";
            s += prefix + std::to_string(r + 1);
            first = false;
        }
    return s + "}";
}

// canonical bytes -> TLA+ text, the way TLC prints a trace state
void print_state(const kmc_config& c, const uint8_t* b) {
    if (c.model == KMC_IDSEQUENCE) {
        uint64_t v;
        memcpy(&v, b, 8);
        printf("nextId = %llu\n", (unsigned long long)v);
        return;
    }
    const int N = c.n_replicas, L = c.log_size, E = c.max_leader_epoch;
    if (c.model == KMC_ASYNC_ISR) {  // AsyncIsr.tla:31-35; replica 1 is `Leader`
        const int rb = ((1 << N) + 7) / 8;
        const uint8_t* q = b + 6 + N;
        const uint8_t* u = q + (E + 1) * rb;
        printf("/\\ controllerState = [isr |-> %s, version |-> %d]\n", bitset_names(b[0], N, "r").c_str(), b[1]);
        printf("/\\ leaderState = [isr |-> %s, version |-> %d, pendingIsr |-> %s, pendingVersion |-> %d, offsets |-> (",
               bitset_names(b[2], N, "r").c_str(), b[3], bitset_names(b[4], N, "r").c_str(), b[5] - 1);
        for (int r = 0; r < N; ++r) printf("%sr%d :> %d", r ? " @@ " : "", r + 1, b[6 + r]);
        printf(")]\n/\\ requests = {");
        bool first = true;
        for (int v = 0; v <= E; ++v)
            for (int m = 0; m < (1 << N); ++m)
                if (q[v * rb + (m >> 3)] >> (m & 7) & 1) {
                    printf("%s[isr |-> %s, version |-> %d]", first ? "" : ", ", bitset_names(m, N, "r").c_str(), v);
                    first = false;
                }
        printf("}\n/\\ updates = {");
        for (int v = 1; v <= b[1] && v <= E + 1; ++v)
            printf("%s[isr |-> %s, version |-> %d]", v > 1 ? ", " : "", bitset_names(u[v - 1], N, "r").c_str(), v);
        printf("}\n");
        return;
    }
    if (c.model == KMC_FINITE_REPLICATED_LOG) {
        printf("logs = (");
        for (int r = 0; r < N; ++r) {
            const uint8_t* k = b + r * (1 + L);
            printf("%sr%d :> [endOffset |-> %d, records |-> <<", r ? " @@ " : "", r + 1, k[0]);
            for (int o = 0; o < L; ++o) printf("%s%s", o ? ", " : "", k[1 + o] ? ("x" + std::to_string(k[1 + o])).c_str() : "Nil");
            printf(">>]");
        }
        printf(")\n");
        return;
    }
    const int rs = 5 + L;
    auto ldr = [&](int x) { return x == 0 ? std::string("\"NONE\"") : "b" + std::to_string(x); };
    printf("/\\ replicaLog = (");
    for (int r = 0; r < N; ++r) {
        const uint8_t* k = b + r * rs;
        printf("%sb%d :> [endOffset |-> %d, records |-> <<", r ? " @@ " : "", r + 1, k[0]);
        for (int o = 0; o < L; ++o) {
            const int code = k[5 + o];
            if (o) printf(", ");
            if (code == 0) printf("-1");
            else printf("[id |-> %d, epoch |-> %d]", (code - 1) / (E + 1), (code - 1) % (E + 1));
        }
        printf(">>]");
    }
    printf(")\n/\\ replicaState = (");
    for (int r = 0; r < N; ++r) {
        const uint8_t* k = b + r * rs;
        printf("%sb%d :> [hw |-> %d, leaderEpoch |-> %d, leader |-> %s, isr |-> %s]", r ? " @@ " : "", r + 1, k[1],
               k[2] - 1, ldr(k[3]).c_str(), bitset_names(k[4], N, "b").c_str());
    }
    const uint8_t* g = b + N * rs;
    printf(")\n/\\ nextRecordId = %d\n/\\ nextLeaderEpoch = %d\n/\\ leaderAndIsrRequests = {", g[0], g[1]);
    for (int e = 0; e < g[1]; ++e)
        printf("%s[leaderEpoch |-> %d, leader |-> %s, isr |-> %s]", e ? ", " : "", e, ldr(g[5 + 2 * e]).c_str(),
               bitset_names(g[6 + 2 * e], N, "b").c_str());
    printf("}\n/\\ quorumState = [leaderEpoch |-> %d, leader |-> %s, isr |-> %s]\n", g[2] - 1, ldr(g[3]).c_str(),
           bitset_names(g[4], N, "b").c_str());
}

void on_level(const kmc_level_info* i, void*) {
    if (i->depth == 1)
        printf("Finished computing initial states: %llu distinct state generated at %s.\n",
               (unsigned long long)i->distinct_total, now().c_str());
    else
        printf("Progress(%llu) at %s: %llu states generated, %llu distinct states found, %llu states left on queue.\n",
               (unsigned long long)i->depth, now().c_str(), (unsigned long long)i->generated_total,
               (unsigned long long)i->distinct_total, (unsigned long long)i->new_states);
}

}  // namespace

// Stock TLC's other switches [TLC-recall] (kafka_specification_amd/tlc.py holds the same tables): what does not change the
// question is accepted and ignored with a note, what asks for another mode of operation is refused.
static const char* lookup_flag(const std::string& a, const char* const (*table)[2], size_t n) {
    for (size_t k = 0; k < n; ++k)
        if (a == table[k][0]) return table[k][1];
    return nullptr;
}
static const char* tlc_ignored_flag(const std::string& a) {
    static const char* const t[][2] = {
        {"-modelcheck", "model checking is the only mode"}, {"-cleanup", "no states directory is written"},
        {"-nowarning", "no TLA+ is evaluated, so no evaluation warnings exist"}, {"-terse", "values are printed in full"},
        {"-tool", "no tool-mode message codes"}, {"-gzip", "checkpoints are not compressed"}, {"-debug", "no debug output"},
        {"-noGenerateSpecTE", "no trace-expression spec is generated"},
        {"-difftrace", "traces print every variable of every state"}};
    return lookup_flag(a, t, sizeof t / sizeof t[0]);
}
static const char* tlc_ignored_with_value(const std::string& a) {
    static const char* const t[][2] = {
        {"-metadir", "nothing is written there"}, {"-userFile", "the lowered models print nothing"},
        {"-fpmem", "the fingerprint table lives in HBM: -table SLOTS sizes it"}, {"-fpbits", "one table, no partitioning by bits"},
        {"-maxSetSize", "no set is enumerated at run time"}, {"-coverage", "action coverage is not collected"},
        {"-lncheck", "no liveness checking"}};
    return lookup_flag(a, t, sizeof t / sizeof t[0]);
}
static const char* tlc_refused_flag(const std::string& a) {
    static const char* const t[][2] = {
        {"-simulate", "random simulation is another mode of TLC; only exhaustive breadth-first model checking is implemented"},
        {"-depth", "it belongs to -simulate"}, {"-seed", "it belongs to -simulate"}, {"-aril", "it belongs to -simulate"},
        {"-dump", "the reachable states stay on the GPU (kmc_frontier_states gives a level's states through the C ABI)"},
        {"-view", "a VIEW changes the distinct-state count"}, {"-dfid", "depth-first iterative deepening is another search order"},
        {"-generateSpecTE", "no trace-expression spec is generated"}};
    return lookup_flag(a, t, sizeof t / sizeof t[0]);
}

int main(int argc, char** argv) {
    std::string spec, cfg_path;
    kmc_config c;
    memset(&c, 0, sizeof c);
    c.n_shards = 1;
    c.keep_trace = 1;
    bool no_deadlock = false, fpcheck = false, force = false, verbose = false;
    std::string levels_csv;
    const double t_main0 = wall_s();
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&](const char* what) -> const char* {
            if (i + 1 >= argc) { fprintf(stderr, "Error: %s needs a value\n", what); exit(2); }
            return argv[++i];
        };
        if (a == "-abi-sizes") {   // tests/test_abi_cpu.py: the structs of include/kmc.h as this translation unit sees them
            printf("%zu %zu %zu %zu %zu\n", sizeof(kmc_config), sizeof(kmc_level_info), sizeof(kmc_result), sizeof(kmc_level_stat), sizeof(kmc_timing));
            return 0;
        }
        if (a == "-config") cfg_path = val("-config");
        else if (a == "-deadlock") no_deadlock = true;
        else if (a == "-continue") c.continue_on_violation = 1;
        else if (a == "-workers") val("-workers");  // accepted, ignored: the GPU's waves are the workers
        else if (a == "-fp") c.hash_seed = strtoull(val("-fp"), nullptr, 0);
        else if (a == "-table") c.table_capacity = strtoull(val("-table"), nullptr, 0);
        else if (a == "-frontier") c.frontier_capacity = strtoull(val("-frontier"), nullptr, 0);
        else if (a == "-device") c.device = atoi(val("-device"));
        else if (a == "-notrace") c.keep_trace = 0;
        else if (a == "-fpcheck") fpcheck = true;
        else if (a == "-fp128") c.wide_fingerprint = 1;  // 128-bit seen-set entries: fingerprint + independent check word
        else if (a == "-symmetry") c.symmetry = 1;  // orbit counting: one stored state per orbit of the permutations of Replicas
        else if (a == "-force") force = true;
        else if (a == "-levels-csv") levels_csv = val("-levels-csv");  // one line per BFS level: frontier, generated per disjunct, new, table load, kernel ms
        else if (a == "-v") verbose = true;   // where the wall time went: start-up (HIP, code object, allocation, first touch), search, teardown
        else if (a == "-verify") setenv("KMC_VERIFY", "1", 1);  // every level is regenerated by a second build of the kernels
        else if (const char* why = tlc_ignored_flag(a)) fprintf(stderr, "Note: %s is accepted for compatibility and ignored (%s)\n", a.c_str(), why);
        else if (const char* why = tlc_ignored_with_value(a)) {
            const char* v = val(a.c_str());
            fprintf(stderr, "Note: %s %s is accepted for compatibility and ignored (%s)\n", a.c_str(), v, why);
        } else if (const char* why = tlc_refused_flag(a)) {
            fprintf(stderr, "Error: %s is not supported: %s\n", a.c_str(), why);
            return 2;
        } else if (a == "-checkpoint") {
            const char* v = val("-checkpoint");   // TLC's interval in minutes; this front end has no sharded checkpoints
            fprintf(stderr, "Note: -checkpoint %s is accepted and ignored (a search takes milliseconds to seconds; the Python front "
                            "end's -checkpoint DIR saves a level-limited sharded search)\n", v);
        }
        else if (!a.empty() && a[0] == '-') { fprintf(stderr, "Error: unknown option %s\n", a.c_str()); return 2; }
        else spec = a;
    }
    if (spec.empty()) { fprintf(stderr, "usage: tlc [-config X.cfg] [-deadlock] [-continue] [-fp N] Spec.tla\n"); return 2; }
    size_t slash = spec.find_last_of('/');
    std::string module = spec.substr(slash == std::string::npos ? 0 : slash + 1);
    if (module.size() > 4 && module.substr(module.size() - 4) == ".tla") module.resize(module.size() - 4);
    if (cfg_path.empty()) cfg_path = spec.substr(0, spec.size() - (spec.size() > 4 && spec.substr(spec.size() - 4) == ".tla" ? 4 : 0)) + ".cfg";
    FILE* f = fopen(cfg_path.c_str(), "rb");
    if (!f) { fprintf(stderr, "Error: configuration file %s not found\n", cfg_path.c_str()); return 2; }
    std::string text;
    char buf[4096];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
    fclose(f);
    Cfg cfg = parse_cfg(text);
    if (!cfg.error.empty()) { fprintf(stderr, "Error: %s\n", cfg.error.c_str()); return 2; }

    c.model = -1;
    for (int m = 0; m <= 6; ++m)
        if (module == kmc_model_name(m)) c.model = m;
    if (module == "MCAsyncIsr") c.model = KMC_ASYNC_ISR;  // models/MCAsyncIsr.tla = AsyncIsr + the state constraint
    if (module == "AsyncIsr") {
        fprintf(stderr, "Error: AsyncIsr.tla is unbounded; check it through models/MCAsyncIsr.tla (CONSTRAINT StateConstraint)\n");
        return 2;
    }
    if (!cfg.constraints.empty() && c.model != KMC_ASYNC_ISR) {
        fprintf(stderr, "Error: CONSTRAINT is not supported for this module (it changes the distinct-state count)\n");
        return 2;
    }
    if (c.model < 0) { fprintf(stderr, "Error: module %s has no lowered model\n", module.c_str()); return 2; }
    auto need = [&](const char* name) -> std::string {
        auto it = cfg.constants.find(name);
        if (it == cfg.constants.end()) { fprintf(stderr, "Error: constant %s is not assigned in the .cfg\n", name); exit(2); }
        return it->second;
    };
    auto need_int = [&](const char* name) -> long long {
        const std::string v = need(name);
        char* end = nullptr;
        const long long x = strtoll(v.c_str(), &end, 10);
        if (v.empty() || end == v.c_str() || *end != 0) {
            fprintf(stderr, "Error: constant %s must be an integer, got %s\n", name, v.c_str());
            exit(2);
        }
        return x;
    };
    if (c.model == KMC_IDSEQUENCE) {
        c.max_id = need_int("MaxId");
    } else if (c.model == KMC_ASYNC_ISR) {
        const std::string reps = need("Replicas"), leader = need("Leader");
        if (reps.find(leader) == std::string::npos) {
            fprintf(stderr, "Error: Leader must be an element of Replicas (AsyncIsr.tla:29)\n");
            return 2;
        }
        if (cfg.constraints.size() != 1 || cfg.constraints[0] != "StateConstraint") {
            fprintf(stderr, "Error: MCAsyncIsr needs exactly `CONSTRAINT StateConstraint`: AsyncIsr is unbounded without it\n");
            return 2;
        }
        c.n_replicas = set_size(reps);
        c.log_size = (int32_t)need_int("MaxOffset");
        c.max_leader_epoch = (int32_t)need_int("MaxVersion");
    } else if (c.model == KMC_FINITE_REPLICATED_LOG) {
        c.n_replicas = set_size(need("Replicas"));
        c.n_log_records = set_size(need("LogRecords"));
        need("Nil");
        c.log_size = (int32_t)need_int("LogSize");
    } else {
        const std::string reps = need("Replicas");
        if (reps.find("NONE") != std::string::npos) {
            fprintf(stderr, "Error: Replicas must not contain \"NONE\" (KafkaReplication.tla:42)\n");
            return 2;
        }
        c.n_replicas = set_size(reps);
        c.log_size = (int32_t)need_int("LogSize");
        c.max_records = (int32_t)need_int("MaxRecords");
        c.max_leader_epoch = (int32_t)need_int("MaxLeaderEpoch");
    }
    for (const std::string& inv : cfg.invariants) {
        int bit = -1;
        for (int k = 0; k < 4; ++k)
            if (inv == kmc_model_invariant_name(c.model, k)) bit = k;
        if (bit < 0 || (c.model <= KMC_FINITE_REPLICATED_LOG && bit != 0)) {
            fprintf(stderr, "Error: unknown invariant %s for module %s\n", inv.c_str(), module.c_str());
            return 2;
        }
        c.invariant_mask |= 1u << bit;
    }
    c.check_deadlock = no_deadlock ? 0 : cfg.check_deadlock;

    if (check_spec(spec, argv[0]) == 2) {
        if (!force) {
            fprintf(stderr, "Error: the spec differs from the revision the GPU kernels were lowered from; nothing was checked "
                            "(-force checks the built-in lowering anyway)\n");
            return 2;
        }
        fprintf(stderr, "Warning: -force: checking the BUILT-IN lowering, not the text of the spec given\n");
    }

    printf("kafka_specification_amd model checker (MI355X, native CLI) — module %s, config %s\n", module.c_str(), cfg_path.c_str());
    printf("Running breadth-first search Model-Checking with fp seed %llu on GPU %d.\n", (unsigned long long)c.hash_seed, c.device);
    printf("Computing initial states...\n");
    kmc_handle* h = nullptr;
    const double t_open0 = wall_s();
    if (kmc_open(&c, &h) != KMC_OK) { fprintf(stderr, "Error: %s\n", kmc_last_error()); return 3; }
    const double t_run0 = wall_s();
    if (kmc_run(h, on_level, nullptr) != KMC_OK) { fprintf(stderr, "Error: %s\n", kmc_last_error()); kmc_close(h); return 3; }
    const double t_run1 = wall_s();
    kmc_result r;
    kmc_result_get(h, &r);
    kmc_timing tm;
    memset(&tm, 0, sizeof tm);
    kmc_timing_get(h, &tm);
    int rc = 0;
    if (r.verdict == KMC_V_OK) {
        printf("Model checking completed. No error has been found.\n");
    } else if (r.verdict == KMC_V_INVARIANT) {
        printf("Error: Invariant %s is violated%s.\n", kmc_model_invariant_name(c.model, r.violated_invariant),
               r.violation_depth == 1 ? " by the initial state" : "");
        rc = 12;
    } else if (r.verdict == KMC_V_DEADLOCK) {
        printf("Error: Deadlock reached.\n");
        rc = 11;
    } else {
        printf("Error: search stopped with verdict %d (table %llu slots, frontier %llu states)\n", r.verdict,
               (unsigned long long)r.table_capacity, (unsigned long long)r.frontier_capacity);
        rc = 1;
    }
    const uint64_t cb = kmc_canon_bytes(h);
    if (r.verdict == KMC_V_INVARIANT && c.keep_trace) {
        const uint64_t cap = r.violation_depth + 1;
        std::vector<uint8_t> states(cap * cb);
        std::vector<int32_t> kinds(cap);
        uint64_t nt = 0;
        if (kmc_trace(h, states.data(), kinds.data(), cap, &nt) == KMC_OK) {
            printf("Error: The behavior up to this point is:\n");
            for (uint64_t k = 0; k < nt && k < cap; ++k) {
                if (k == 0) printf("State 1: <Initial predicate>\n");
                else printf("State %llu: <%s of module %s>\n", (unsigned long long)(k + 1), kmc_action_name(c.model, kinds[k]), module.c_str());
                print_state(c, states.data() + k * cb);
                printf("\n");
            }
        } else {
            fprintf(stderr, "Error: %s\n", kmc_last_error());
        }
    } else if (r.verdict == KMC_V_DEADLOCK) {
        std::vector<uint64_t> w(kmc_state_words(h));
        std::vector<uint8_t> st(cb);
        if (kmc_witness(h, w.data()) == KMC_OK) {
            kmc_unpack_state(h, w.data(), st.data());
            printf("Error: The deadlocked state is:\n");
            print_state(c, st.data());
        }
    }
    if (!levels_csv.empty()) {
        // the per-level view a user tunes constants and capacities by (kmc_level_stats: one record per expansion)
        std::vector<kmc_level_stat> st(kmc_level_stats(h, nullptr, 0));
        const uint64_t ns = kmc_level_stats(h, st.data(), st.size());
        if (FILE* lf = fopen(levels_csv.c_str(), "w")) {
            fprintf(lf, "depth,frontier,new_states,stored_new,probes,deadlocks,table_load,expand_ms");
            const int nk = kmc_action_count(c.model);
            for (int k = 0; k < nk; ++k) fprintf(lf, ",%s", kmc_action_name(c.model, k));
            fprintf(lf, "\n");
            for (uint64_t i = 0; i < ns && i < st.size(); ++i) {
                fprintf(lf, "%llu,%llu,%llu,%llu,%llu,%llu,%.6f,%.4f", (unsigned long long)st[i].depth, (unsigned long long)st[i].frontier,
                        (unsigned long long)st[i].new_states, (unsigned long long)st[i].stored_new, (unsigned long long)st[i].probes,
                        (unsigned long long)st[i].deadlocks, st[i].table_load, st[i].expand_ms);
                for (int k = 0; k < nk; ++k) fprintf(lf, ",%llu", (unsigned long long)st[i].generated[k]);
                fprintf(lf, "\n");
            }
            fclose(lf);
        } else {
            fprintf(stderr, "Error: cannot write %s\n", levels_csv.c_str());
        }
    }
    printf("%llu states generated, %llu distinct states found, %llu states left on queue.\n", (unsigned long long)r.generated,
           (unsigned long long)r.distinct, (unsigned long long)r.queue_left);
    printf("The depth of the complete state graph search is %llu.\n", (unsigned long long)r.depth);
    // TLC's closing estimate [TLC-recall: "calculated (optimistic)" = distinct x (generated - distinct) / 2^64]
    uint64_t col_distinct = r.distinct, col_generated = r.generated;   // what the seen-set holds / was probed with
    if (c.symmetry) {
        printf("Symmetry reduction (orbit counting over the permutations of Replicas): %llu states were stored and expanded, one "
               "per orbit; every count above is the plain search's.\n", (unsigned long long)r.orbit_representatives);
        // the seen-set holds the representatives: they are what can collide
        col_generated = r.distinct ? (uint64_t)((double)r.generated * (double)r.orbit_representatives / (double)r.distinct) : r.generated;
        col_distinct = r.orbit_representatives;
    }
    if (c.wide_fingerprint) {
        printf("The seen-set stores 128 bits per state (the fingerprint and an independent check word); probability that two "
               "distinct states were merged:\n");
        printf("  birthday bound on the stored entries:  val = %.2E\n", (double)col_distinct * (double)col_distinct / std::pow(2.0, 129));
    } else {
        printf("The seen-set stores 64-bit fingerprints; estimates of the probability that not all reachable states were checked "
               "because two distinct states had the same fingerprint:\n");
        printf("  calculated (optimistic):  val = %.2E\n",
               (double)col_distinct * (double)(col_generated > col_distinct ? col_generated - col_distinct : 0) / std::pow(2.0, 64));
        const double birthday = (double)col_distinct * (double)col_distinct / std::pow(2.0, 65);
        printf("  birthday bound on the stored fingerprints:  val = %.2E\n", birthday);
        if (birthday > 0.1) {   // (Kip320 3/6/6/3: 6,452,700,520 states, bound 1.1 — the 64-bit search returns one state fewer)
            printf("  Recommendation: that bound is above 0.1: counts of this size are only bit-exact with 128-bit entries - re-run with "
                   "-fp128 (or, on the Kafka modules, -symmetry: a sixth of the stored fingerprints at three brokers).\n");
            // A FINISHED search whose count is more likely wrong than not must not look like a success to a script: BASELINE's
            // target is the bit-identical count (exit 14; a verdict that already failed keeps its own code).
            if (rc == 0) {
                printf("Warning: the distinct-state count of this run is NOT certified (64-bit fingerprints, birthday bound %.2E > 0.1): "
                       "exit code 14.\n", birthday);
                rc = 14;
            }
        }
    }
    if (fpcheck) {  // a collision moves with the seed: equal counts under two seeds make a silent loss very unlikely
        kmc_close(h);
        h = nullptr;
        kmc_config c2 = c;
        c2.hash_seed = c.hash_seed * 0x9E3779B97F4A7C15ull + 0x5851F42D4C957F2Dull;
        c2.keep_trace = 0;
        kmc_result r2;
        if (kmc_open(&c2, &h) != KMC_OK || kmc_run(h, nullptr, nullptr) != KMC_OK || kmc_result_get(h, &r2) != KMC_OK) {
            fprintf(stderr, "Error: %s\n", kmc_last_error());
            if (h) kmc_close(h);
            return 3;
        }
        const bool same = r2.verdict == r.verdict && r2.distinct == r.distinct && r2.generated == r.generated && r2.depth == r.depth;
        printf("Fingerprint check: second run with fp seed %llu: %llu states generated, %llu distinct states found, depth %llu - %s\n",
               (unsigned long long)c2.hash_seed, (unsigned long long)r2.generated, (unsigned long long)r2.distinct,
               (unsigned long long)r2.depth,
               same ? "identical to the first run." : "DIFFERENT from the first run: a fingerprint collision dropped states in at least one of them "
                      "(a collision can only lose states: the larger count is the better lower bound).");
        if (!same && rc == 0) rc = 13;
    }
    printf("Finished in %.3fs (%.0f distinct states/s; %.3fs in the expand kernel) at (%s)\n", r.seconds_total,
           r.distinct / (r.seconds_total > 1e-9 ? r.seconds_total : 1e-9), r.seconds_expand, now().c_str());
    const double t_close0 = wall_s();
    kmc_close(h);
    if (verbose) {
        // (printed after the teardown it accounts for; kmc_timing: include/kmc.h)
        const double t_end = wall_s();
        printf("Wall time: %.3fs in this process = %.3fs before kmc_open (arguments, .cfg, spec revision) + %.3fs kmc_open "
               "(HIP initialisation %.3fs, code object %.3fs, allocation of %.1f GiB %.3fs) + %.3fs kmc_run (first clear of the "
               "seen-set %.3fs, search %.3fs) + %.3fs verdict / trace + %.3fs teardown\n",
               t_end - t_main0, t_open0 - t_main0, t_run0 - t_open0, tm.hip_init_s, tm.code_object_s,
               (double)tm.device_bytes / (double)(1ull << 30), tm.alloc_s, t_run1 - t_run0, tm.first_clear_s, r.seconds_total,
               t_close0 - t_run1, t_end - t_close0);
    }
    return rc;
}
