// kmc_cli.cpp — native `tlc`-shaped front end over the C ABI (include/kmc.h); the C++ twin of
// kafka_specification_amd/tlc.py, for hosts without Python/torch.
//
//   tlc [-config X.cfg] [-deadlock] [-continue] [-workers N] [-fp SEED] [-table SLOTS]
//       [-frontier STATES] [-device D] [-notrace] Spec.tla
//
// [TLC-recall] flag names and output lines follow tlc2.TLC; TLC itself is not part of the
// reference repository.  The module name selects one of the lowered models; constants
// and invariants come from the .cfg (CONSTANT(S), INIT, NEXT, SPECIFICATION, INVARIANT(S),
// CHECK_DEADLOCK).  No TLA+ is parsed.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <string>
#include <vector>

#include "../../include/kmc.h"

namespace {

struct Cfg {
    std::map<std::string, std::string> constants;  // raw value text ("{b1, b2}" or "6")
    std::vector<std::string> invariants, constraints;
    int check_deadlock = 1;  // TLC's default
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
            t.push_back("=");
            i += 2;
        } else {
            size_t e = i;
            while (e < s.size() && !isspace((unsigned char)s[e]) && s[e] != '=' && s[e] != '{') ++e;
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
            if (i + 2 < t.size() + 0 && t[i + 1] == "=") {
                c.constants[w] = t[i + 2];
                i += 2;
            } else {
                c.error = "expected `" + w + " = value` in CONSTANTS";
                return c;
            }
        } else if (section == "INIT") {
            if (w != "Init") c.error = "only INIT Init is known";
        } else if (section == "NEXT") {
            if (w != "Next") c.error = "only NEXT Next is known";
        } else if (section == "SPECIFICATION") {
            if (w != "Spec") c.error = "only SPECIFICATION Spec is known";
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

int main(int argc, char** argv) {
    std::string spec, cfg_path;
    kmc_config c;
    memset(&c, 0, sizeof c);
    c.n_shards = 1;
    c.keep_trace = 1;
    bool no_deadlock = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&](const char* what) -> const char* {
            if (i + 1 >= argc) { fprintf(stderr, "Error: %s needs a value\n", what); exit(2); }
            return argv[++i];
        };
        if (a == "-config") cfg_path = val("-config");
        else if (a == "-deadlock") no_deadlock = true;
        else if (a == "-continue") c.continue_on_violation = 1;
        else if (a == "-workers") val("-workers");  // accepted, ignored: the GPU's waves are the workers
        else if (a == "-fp") c.hash_seed = strtoull(val("-fp"), nullptr, 0);
        else if (a == "-table") c.table_capacity = strtoull(val("-table"), nullptr, 0);
        else if (a == "-frontier") c.frontier_capacity = strtoull(val("-frontier"), nullptr, 0);
        else if (a == "-device") c.device = atoi(val("-device"));
        else if (a == "-notrace") c.keep_trace = 0;
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
    if (c.model == KMC_IDSEQUENCE) {
        c.max_id = atoll(need("MaxId").c_str());
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
        c.log_size = atoi(need("MaxOffset").c_str());
        c.max_leader_epoch = atoi(need("MaxVersion").c_str());
    } else if (c.model == KMC_FINITE_REPLICATED_LOG) {
        c.n_replicas = set_size(need("Replicas"));
        c.n_log_records = set_size(need("LogRecords"));
        need("Nil");
        c.log_size = atoi(need("LogSize").c_str());
    } else {
        const std::string reps = need("Replicas");
        if (reps.find("NONE") != std::string::npos) {
            fprintf(stderr, "Error: Replicas must not contain \"NONE\" (KafkaReplication.tla:42)\n");
            return 2;
        }
        c.n_replicas = set_size(reps);
        c.log_size = atoi(need("LogSize").c_str());
        c.max_records = atoi(need("MaxRecords").c_str());
        c.max_leader_epoch = atoi(need("MaxLeaderEpoch").c_str());
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

    printf("kafka_specification_amd model checker (MI355X, native CLI) — module %s, config %s\n", module.c_str(), cfg_path.c_str());
    printf("Running breadth-first search Model-Checking with fp seed %llu on GPU %d.\n", (unsigned long long)c.hash_seed, c.device);
    printf("Computing initial states...\n");
    kmc_handle* h = nullptr;
    if (kmc_open(&c, &h) != KMC_OK) { fprintf(stderr, "Error: %s\n", kmc_last_error()); return 3; }
    if (kmc_run(h, on_level, nullptr) != KMC_OK) { fprintf(stderr, "Error: %s\n", kmc_last_error()); kmc_close(h); return 3; }
    kmc_result r;
    kmc_result_get(h, &r);
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
    printf("%llu states generated, %llu distinct states found, %llu states left on queue.\n", (unsigned long long)r.generated,
           (unsigned long long)r.distinct, (unsigned long long)r.queue_left);
    printf("The depth of the complete state graph search is %llu.\n", (unsigned long long)r.depth);
    printf("Finished in %.3fs (%.0f distinct states/s; %.3fs in the expand kernel) at (%s)\n", r.seconds_total,
           r.distinct / (r.seconds_total > 1e-9 ? r.seconds_total : 1e-9), r.seconds_expand, now().c_str());
    kmc_close(h);
    return rc;
}
