// kmc_engine.cpp — host side of the MI355X model checker behind the C ABI of include/kmc.h.
//
// Owns: specialising the device code for one (model, constants) through hiprtc (with an
// on-disk code-object cache), the HBM fingerprint table / frontier buffers, and the
// level-synchronous BFS loop that stands in for TLC's ModelChecker + Worker threads
// [TLC-recall; TLC is not part of /root/reference].  No CPU fallback exists: without a HIP
// device or compiler every entry point fails with KMC_E_DEVICE / KMC_E_COMPILE.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <rccl/rccl.h>   // types and prototypes only: librccl is bound with dlopen when a communicator is created

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/kmc.h"
#include "kmc_device.h"   // host-visible parts: KmcArgs, KmcLevelCtl, layout, fingerprint
#include "kmc_sources.inc"  // generated: KMC_SRC_DEVICE (kmc_layout.h and the parts of kmc_device.h, as text)

// Levels kmc_run queues back to back before it waits (no progress callback): see run_levels.
#define KMC_CHAIN 32
#define KMC_CTL_SLOTS (3 + KMC_CHAIN)   // two alternating levels + one auxiliary + one per chained level

// KMC_VERIFY: both builds carry the fingerprint checksum (KMC_CHECKSUM, kmc_device.h); the second one differs in how it is
// compiled — optimisation level and a quarter of the occupancy target, i.e. another register allocation
#define KMC_VERIFY_PRIMARY_OPTIONS "-DKMC_CHECKSUM=1"
// The second build of the differential self-check: another optimisation level, a quarter of the occupancy target, the
// fingerprint checksum — and, for the Kafka models, ANOTHER LOWERING OF THE GUARDS: KmcKafka::guard<K> looped per kind over
// a run-time binding instead of the straight-line block of every instance's inst<I> (kmc_device.h, RUNTIME_GUARDS).
#define KMC_VERIFY_OPTIONS "-O1 -DKMC_MIN_WAVES=2 -DKMC_CHECKSUM=1 -DKMC_RT_GUARDS_MIN_INSTANCES=0 -DKMC_WITH_DRY=1"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[2048];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(KMC_E_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

const char* const MODEL_NAMES[] = {"IdSequence", "FiniteReplicatedLog", "KafkaTruncateToHighWatermark",
                                   "Kip101",     "Kip279",              "Kip320",
                                   "Kip320FirstTry", "AsyncIsr"};
const char* const INV_NAMES[] = {"TypeOk", "WeakIsr", "StrongIsr", "LeaderInIsr"};
// AsyncIsr.tla:62,161; LeaderOffsetInRange is defined in models/MCAsyncIsr.tla (not in the reference)
const char* const INV_NAMES_ASYNC[] = {"TypeOk", "ValidHighWatermark", "LeaderOffsetInRange", "?"};
const char* const KINDS_ASYNC[] = {"ControllerShrinkIsr", "ControllerHandleRequest", "LeaderRequestShrinkIsr",
                                   "LeaderRequestExpandIsr", "LeaderWrite", "LeaderHandleUpdate", "FollowerReplicate"};

const char* const KINDS_BASE[] = {"ControllerElectLeader", "ControllerShrinkIsr", "BecomeLeader",
                                  "LeaderExpandIsr",       "LeaderShrinkIsr",     "LeaderWrite",
                                  "LeaderIncHighWatermark", nullptr,              "FollowerReplicate"};
const char* const KINDS_KIP320[] = {"ControllerElectLeader",        "ControllerShrinkIsr",
                                    "BecomeLeader",                 "FencedLeaderExpandIsr",
                                    "FencedLeaderShrinkIsr",        "LeaderWrite",
                                    "FencedLeaderIncHighWatermark", "FencedBecomeFollowerAndTruncate",
                                    "FencedFollowerFetch"};
const char* const KINDS_FIRST[] = {"ControllerElectLeader",
                                   "ControllerShrinkIsr",
                                   "BecomeLeader",
                                   "LeaderExpandIsrBetterFencing",
                                   "LeaderShrinkIsrBetterFencing",
                                   "LeaderWrite",
                                   "ImprovedLeaderIncHighWatermark",
                                   "BecomeFollower",
                                   "FollowerFetch",
                                   "FollowerTruncate"};
const char* const KINDS_FRL[] = {"Append", "TruncateTo", "ReplicateTo"};

uint64_t fnv1a(const std::string& s, uint64_t h = 1469598103934665603ull) {
    for (unsigned char c : s) {
        h ^= c;
        h *= 1099511628211ull;
    }
    return h;
}

bool validate(const kmc_config& c, KmcLayout* lay, std::string* name, std::string* inst) {
    char buf[256];
    switch (c.model) {
    case KMC_IDSEQUENCE:
        if (c.max_id < 0) return false;
        *lay = kmc_make_layout(c.model, 0, 0, 0, 0, 0);
        snprintf(buf, sizeof buf, "IdSequence_M%lld", (long long)c.max_id);
        *name = buf;
        snprintf(buf, sizeof buf, "KmcIdSequence<%lldLL>", (long long)c.max_id);
        *inst = buf;
        return true;
    case KMC_FINITE_REPLICATED_LOG:
        *lay = kmc_make_layout(c.model, c.n_replicas, c.log_size, 0, 0, c.n_log_records);
        if (!lay->valid || c.n_replicas < 2) return false;
        snprintf(buf, sizeof buf, "FiniteReplicatedLog_N%d_L%d_K%d%s", c.n_replicas, c.log_size, c.n_log_records,
                 c.symmetry ? "_sym" : "");
        *name = buf;
        snprintf(buf, sizeof buf, "KmcFiniteReplicatedLog<%d,%d,%d>", c.n_replicas, c.log_size, c.n_log_records);
        *inst = buf;
        return true;
    case KMC_ASYNC_ISR:  // log_size = MaxOffset, max_leader_epoch = MaxVersion (the constraint's bounds)
        *lay = kmc_make_layout(c.model, c.n_replicas, c.log_size, 0, c.max_leader_epoch, 0);
        if (!lay->valid || c.n_replicas < 1 || c.log_size < 1) return false;  // ASSUME MaxOffset > 0, AsyncIsr.tla:28
        snprintf(buf, sizeof buf, "AsyncIsr_N%d_O%d_V%d", c.n_replicas, c.log_size, c.max_leader_epoch);
        *name = buf;
        snprintf(buf, sizeof buf, "KmcAsyncIsr<%d,%d,%d>", c.n_replicas, c.log_size, c.max_leader_epoch);
        *inst = buf;
        return true;
    case KMC_TRUNCATE_TO_HW:
    case KMC_KIP101:
    case KMC_KIP279:
    case KMC_KIP320:
    case KMC_KIP320_FIRST_TRY: {
        // KMC_LAYOUT=tight|rm (tests, A/B measurements) overrides the automatic choice between the two arrangements of
        // the state vector (kmc_layout.h); host and device evaluate the same constexpr function with the same mode
        const char* lenv = getenv("KMC_LAYOUT");
        const int lm = !lenv || !*lenv || !strcmp(lenv, "auto") ? KMC_LAYOUT_AUTO
                       : !strcmp(lenv, "tight") ? KMC_LAYOUT_TIGHT : !strcmp(lenv, "rm") ? KMC_LAYOUT_RM
                       : !strcmp(lenv, "rmg") ? KMC_LAYOUT_RMG : -1;
        if (lm < 0) return false;
        *lay = kmc_make_layout(c.model, c.n_replicas, c.log_size, c.max_records, c.max_leader_epoch, 0, lm);
        if (!lay->valid || c.n_replicas < 2) return false;
        snprintf(buf, sizeof buf, "%s_N%d_L%d_R%d_E%d%s%s", MODEL_NAMES[c.model], c.n_replicas, c.log_size,
                 c.max_records, c.max_leader_epoch,
                 lm == KMC_LAYOUT_TIGHT ? "_tight" : lm == KMC_LAYOUT_RM ? "_rm" : lm == KMC_LAYOUT_RMG ? "_rmg" : "",
                 c.symmetry ? "_sym" : "");
        *name = buf;
        snprintf(buf, sizeof buf, "KmcKafka<%d,%d,%d,%d,%d,%d>", c.model, c.n_replicas, c.log_size, c.max_records,
                 c.max_leader_epoch, lm);
        *inst = buf;
        return true;
    }
    default: return false;
    }
}

std::string strip_for_concat(const char* src) {
    // drop '#pragma once' and the local includes so the parts can be fed to hiprtc as one file; drop `//` comments
    // (line structure kept) so that the text — and with it the key of the code-object cache — only changes with the code
    std::string out, line;
    for (const char* p = src;; ++p) {
        if (*p == '\n' || *p == 0) {
            bool in_str = false;
            for (size_t k = 0; k + 1 < line.size(); ++k) {
                if (line[k] == '"' && (k == 0 || line[k - 1] != '\\')) in_str = !in_str;
                if (!in_str && line[k] == '/' && line[k + 1] == '/') {
                    line.erase(k);
                    while (!line.empty() && (line.back() == ' ' || line.back() == '\t')) line.pop_back();
                    break;
                }
            }
            if (line.rfind("#pragma once", 0) != 0 && line.rfind("#include \"kmc_", 0) != 0) {
                out += line;
            }
            out += '\n';
            line.clear();
            if (*p == 0) break;
        } else {
            line += *p;
        }
    }
    return out;
}

std::string default_cache_dir() {
    if (const char* e = getenv("KMC_CACHE_DIR")) return e;
    Dl_info info;
    if (dladdr((void*)&default_cache_dir, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        size_t k = p.find_last_of('/');
        if (k != std::string::npos) return p.substr(0, k) + "/kmc_cache";
    }
    return "./kmc_cache";
}

bool read_file(const std::string& path, std::vector<char>* out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out->resize(n > 0 ? n : 0);
    bool ok = n > 0 && fread(out->data(), 1, n, f) == (size_t)n;
    fclose(f);
    return ok;
}

// .vgpr_spill_count of one kernel, read from the code object's AMDGPU metadata note (msgpack: the
// keys of a kernel's map are sorted, so the count follows the kernel's ".name" value).  -1 = not found.
#define KMC_MAX_VGPR_SPILLS 8
long expand_vgpr_spills(const std::vector<char>& code, const std::string& kernel) {
    const std::string blob(code.begin(), code.end());
    size_t at = blob.find(kernel);
    while (at != std::string::npos) {  // the name also occurs in the symbol table: take the one inside the metadata
        const size_t key = blob.find(".vgpr_spill_count", at);
        const size_t next_name = blob.find(".name", at + kernel.size());
        if (key != std::string::npos && (next_name == std::string::npos || key < next_name || key - at < 2048)) {
            const unsigned char* q = (const unsigned char*)blob.data() + key + 17;
            if (q[0] <= 0x7f) return q[0];
            if (q[0] == 0xcc) return q[1];
            if (q[0] == 0xcd) return (q[1] << 8) | q[2];
            if (q[0] == 0xce) return ((long)q[1] << 24) | (q[2] << 16) | (q[3] << 8) | q[4];
            return -1;
        }
        at = blob.find(kernel, at + 1);
    }
    return -1;
}

// Compile (or fetch from the cache) the code object specialised for cfg.
// `mode` = which k_expand the object holds (kmc_kernels.h, KMC_ONLY_MODE): KMC_MODE_LOCAL — the search's own kernel with the
// small kernels around it — KMC_MODE_SHARDED or KMC_MODE_ENUM; one cached file each, so that a front end which never steps or
// enumerates never pays for those kernels, and the search's kernel is not recompiled (minutes at seven brokers) for them.
const char* const MODE_SUFFIX[3] = {"", "_sh", "_en"};
const char* const MODE_FILE_TAG[3] = {"", "-sharded", "-enum"};
int get_code_object(const kmc_config& cfg, const std::string& arch, std::vector<char>* code, std::string* kname,
                    const char* extra_options = nullptr, std::string* path_out = nullptr, unsigned mode = KMC_MODE_LOCAL,
                    const std::string* jit_defines = nullptr) {
    if (mode > KMC_MODE_ENUM) return fail(KMC_E_ARG, "no code object for mode %u", mode);
    KmcLayout lay;
    std::string name, inst;
    if (!validate(cfg, &lay, &name, &inst))
        return fail(KMC_E_ARG, "unsupported model/constants (model=%d N=%d L=%d R=%d E=%d K=%d): need 2<=N<=8, "
                               "L*bits(record)<=64, E<=7 (AsyncIsr: N<=6, MaxVersion<=7)", cfg.model, cfg.n_replicas, cfg.log_size,
                    cfg.max_records, cfg.max_leader_epoch, cfg.n_log_records);
    *kname = name;
    if (cfg.symmetry && (!kmc_model_symmetric(cfg.model) || cfg.n_replicas > KMC_SYMMETRY_MAX_REPLICAS))
        return fail(KMC_E_ARG, "symmetry (orbit counting) is for the Kafka family and FiniteReplicatedLog with at most 7 replicas: "
                               "%s singles out a replica, or N = %d > %d",
                    MODEL_NAMES[cfg.model], cfg.n_replicas, KMC_SYMMETRY_MAX_REPLICAS);

    // optional tuning overrides, e.g. KMC_JIT_DEFINES="-DKMC_MIN_WAVES=5 -DKMC_PROFILE=1"
    std::vector<std::string> defines;
    std::string defines_key;
    // (a handle's later code objects — ensure_mode — are built with the defines its first one was opened under: jit_defines)
    std::string all_defines = jit_defines ? *jit_defines : getenv("KMC_JIT_DEFINES") ? getenv("KMC_JIT_DEFINES") : "";
    if (extra_options) all_defines += std::string(" ") + extra_options;
    if (cfg.symmetry) all_defines += " -DKMC_SYMM=1";
    if (mode != KMC_MODE_LOCAL) all_defines += " -DKMC_ONLY_MODE=" + std::to_string(mode);
    if (!all_defines.empty()) {
        const char* d = all_defines.c_str();
        std::string tok;
        for (const char* q = d;; ++q) {
            if (*q == ' ' || *q == 0) {
                if (!tok.empty()) { defines.push_back(tok); defines_key += tok + " "; }
                tok.clear();
                if (*q == 0) break;
            } else {
                tok += *q;
            }
        }
    }
    std::string src = strip_for_concat(KMC_SRC_DEVICE) + "\nKMC_INSTANTIATE(" +
                      name + ", " + inst + ")\n";
    // ONE code object per (source, architecture, defines), whoever compiled it.  The PyTorch wheel bundles its own
    // hiprtc / comgr next to the system ROCm's (same hiprtcVersion, different LLVM builds: from round 4's source on they emit
    // different instructions for the same text), and a process binds to one or the other (_native.py).  Rounds 1-3 keyed the
    // cache by the HIP runtime's build number too, so the bench (torch's runtime) and a rocprofv3 run (system ROCm) each
    // compiled and ran their own object — a profile then described other machine code than the line it is quoted beside.
    // A gfx950 code object loads under either runtime: the cache is keyed by what is compiled, not by who asks.
    int rtc_major = 0, rtc_minor = 0;
    hiprtcVersion(&rtc_major, &rtc_minor);
    char key[64];
    snprintf(key, sizeof key, "%016llx",
             (unsigned long long)fnv1a(src + "|" + arch + "|" + std::to_string(rtc_major) + "." + std::to_string(rtc_minor) +
                                       "|" + defines_key));
    const std::string dir = cfg.cache_dir ? std::string(cfg.cache_dir) : default_cache_dir();
    const std::string path = dir + "/" + name + "-" + arch + "-" + key + MODE_FILE_TAG[mode] + ".hsaco";
    if (path_out) *path_out = path;
    if (read_file(path, code)) return KMC_OK;
    if (getenv("KMC_VERBOSE"))
        fprintf(stderr, "[kmc] specialising kernels for %s (first use; wide configurations take minutes)\n", name.c_str());

    // k_expand is compiled for 6 waves/SIMD (80 VGPRs).  Wide configurations (7-8 replicas: hundreds of
    // action instances, several words of state) do not fit: at 184 spilled VGPRs on top of 466 spilled
    // SGPRs, Kip320 with 7 replicas lost successors (six missing states at BFS level 3; the same code is
    // right at -O1, at -O0 and with a larger register budget, and the model templates are right when
    // compiled for the host — tests/test_device_models_on_host.py).  So the register budget follows the
    // kernel: recompile with fewer waves per SIMD until k_expand spills (almost) no VGPRs.  An explicit
    // -DKMC_MIN_WAVES in KMC_JIT_DEFINES is respected as given.
    const bool waves_forced = defines_key.find("KMC_MIN_WAVES") != std::string::npos;
    size_t n = 0;
    for (int waves = 6; waves >= 1;) {
        hiprtcProgram prog;
        if (hiprtcCreateProgram(&prog, src.c_str(), "kmc_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
            return fail(KMC_E_COMPILE, "hiprtcCreateProgram failed");
        const std::string archopt = "--offload-arch=" + arch;
        const std::string wavesopt = "-DKMC_MIN_WAVES=" + std::to_string(waves);
        std::vector<const char*> opts = {archopt.c_str(), "-O3", "-std=c++17"};
        if (!waves_forced) opts.push_back(wavesopt.c_str());
        for (const std::string& d : defines) opts.push_back(d.c_str());
        hiprtcResult r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
        if (r != HIPRTC_SUCCESS) {
            size_t ln = 0;
            hiprtcGetProgramLogSize(prog, &ln);
            std::string log(ln, 0);
            if (ln) hiprtcGetProgramLog(prog, &log[0]);
            hiprtcDestroyProgram(&prog);
            return fail(KMC_E_COMPILE, "hiprtc failed for %s: %s\n%.1500s", name.c_str(), hiprtcGetErrorString(r), log.c_str());
        }
        hiprtcGetCodeSize(prog, &n);
        code->resize(n);
        hiprtcGetCode(prog, code->data());
        hiprtcDestroyProgram(&prog);
        const long spills = expand_vgpr_spills(*code, std::string("kmc_expand") + MODE_SUFFIX[mode] + "_" + name);
        if (waves_forced) break;  // an explicit -DKMC_MIN_WAVES (tuning, bug hunts) is taken as given and never cached as default
        // The guard must not pass by accident: an unreadable spill count, or a kernel that still spills at
        // one wave per SIMD, is a failed specialisation — not a kernel to run and cache.
        if (spills < 0)
            return fail(KMC_E_COMPILE, "cannot read .vgpr_spill_count of kmc_expand%s_%s from the code object's metadata: "
                                       "the register-budget rule cannot be checked", MODE_SUFFIX[mode], name.c_str());
        if (spills <= KMC_MAX_VGPR_SPILLS) break;
        if (waves == 1)
            return fail(KMC_E_COMPILE, "kmc_expand%s_%s spills %ld vector registers even at one wave per SIMD: constants too "
                                       "wide for this kernel shape", MODE_SUFFIX[mode], name.c_str(), spills);
        // (these kernels take up to minutes to compile: jump by the size of the overflow, do not crawl)
        // (a near miss at 6 waves gets 5 — 96 registers: the kind-major headline kernel spills 9 at 80 and 2 at 96 and runs
        // equally fast at either, profiles/r03_kind_major.txt)
        // (the orbit-counting headline kernel spills 23 at 80: at 5 waves — 96 registers, 8 spilled — it runs 7.4 ms, at 4 waves
        // — 108, none — 7.7 ms, profiles/r03_symmetry.txt: a miss of up to 24 tries 5 first and falls to 4 from there)
        const int next = spills > 64 ? 2 : spills > 40 ? 3 : spills > 24 ? 4 : 5;
        waves = next < waves ? next : waves - 1;
    }
    // best-effort cache write (atomic rename)
    mkdir(dir.c_str(), 0755);
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    if (FILE* f = fopen(tmp.c_str(), "wb")) {
        bool ok = fwrite(code->data(), 1, n, f) == n;
        fclose(f);
        if (ok) rename(tmp.c_str(), path.c_str());
        else unlink(tmp.c_str());
        // Who compiled it.  The key above holds hiprtc's major.minor only (so that the bench under torch's runtime and a
        // rocprofv3 run under the system's load the SAME object), but the two bundled compilers emit different instructions for
        // the same text: which of them filled this slot of the cache is recorded beside it — COMPILERS.jsonl, one appended line
        // per object written (bench.py reports it next to kernel_code_sha256; an object from a compiler found to be bad can be
        // told from its neighbours and deleted).
        if (ok) {
            int rt = 0;
            (void)hipRuntimeGetVersion(&rt);
            if (FILE* idx = fopen((dir + "/COMPILERS.jsonl").c_str(), "ab")) {
                const size_t slash = path.find_last_of('/');
                fprintf(idx, "{\"file\": \"%s\", \"hiprtc\": \"%d.%d\", \"hip_runtime_version\": %d, \"waves\": \"%s\"}\n",
                        path.substr(slash == std::string::npos ? 0 : slash + 1).c_str(), rtc_major, rtc_minor, rt,
                        waves_forced ? "as given" : "rule");
                fclose(idx);
            }
        }
    }
    return KMC_OK;
}

uint64_t pow2_floor(uint64_t x) {
    uint64_t p = 1;
    while (p * 2 <= x) p *= 2;
    return p;
}
uint64_t pow2_ceil(uint64_t x) {
    uint64_t p = 1;
    while (p < x) p *= 2;
    return p;
}

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

struct kmc_handle {
    kmc_config cfg{};
    KmcLayout lay{};
    int W = 0;
    std::string kname;
    std::string arch;
    kmc_timing timing{};                    // where the wall time outside the search went (kmc_timing_get)
    bool first_clear_timed = false;
    std::string cache_dir;                  // kmc_config.cache_dir, copied: the later code objects (ensure_mode) are looked up
                                            // long after kmc_open returned and the caller's string may be gone
    std::string jit_defines;                // KMC_JIT_DEFINES as it stood when the handle was opened
    bool verify = false;                    // KMC_VERIFY likewise
    hipModule_t mod = nullptr;              // the search's code object: k_expand (LOCAL), k_inv, k_insert, k_init, k_find, k_packrow
    hipFunction_t f_expand = nullptr, f_inv = nullptr, f_insert = nullptr, f_init = nullptr, f_find = nullptr, f_packrow = nullptr;
    hipFunction_t f_expand_dry = nullptr;   // only in a KMC_TUNING build of `mod` (KMC_DRYRUN / KMC_SHADOW tuning aids)
    hipModule_t mod_sh = nullptr, mod_en = nullptr;   // k_expand in SHARDED / ENUM mode: loaded when first needed (ensure_mode)
    hipFunction_t f_expand_sh = nullptr, f_expand_en = nullptr;
    hipModule_t mod_verify = nullptr;       // KMC_VERIFY=1: a second, differently compiled code object whose dry k_expand regenerates every level
    hipFunction_t f_expand_verify = nullptr;
    uint64_t verify_levels = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_chain[2 * KMC_CHAIN] = {nullptr};  // chained launches: one pair per level of a batch
    int rec_words = 0;  // exchange / insert record size: W, +1 when predecessor fingerprints are kept
    int n_cus = 256;
    int blocks_per_cu = 4;  // k_expand residency, from the occupancy query at open
    u64 *table = nullptr, *pred = nullptr, *table2 = nullptr;
    u64* sent = nullptr;      // n_shards > 1: sender-side filter of fingerprints already shipped
    uint64_t sent_cap = 0;
    uint64_t table_cap = 0;      // slots
    uint64_t slot_words = 1;     // 64-bit words per slot: 1, or 2 with kmc_config.wide_fingerprint (fingerprint + check word)
    uint64_t inserted_level = 0; // stepping: records handed to k_insert since the last kmc_step_finish (conservation check)
    u64* frontier[2] = {nullptr, nullptr};
    uint64_t fcap = 0;
    KmcLevelCtl* ctl = nullptr;       // 3 device slots: two alternating levels + one auxiliary
    KmcLevelCtl* ctl_host = nullptr;  // pinned
    u64* scratch = nullptr;      // device: init record / find result / enum input
    uint64_t* scratch_host = nullptr; // pinned
    u64* enum_out = nullptr;     // device: ENUM records
    uint64_t enum_cap = 4096;
    u64* send = nullptr;         // SHARDED send buffers
    uint64_t send_cap = 0;
    bool send_owned = true;
    // run state
    int cur = 0;                 // frontier[cur] holds the last completed level
    uint64_t n_cur = 0;          // its size on this shard
    uint64_t seg_n[KMC_SEGS] = {0};  // ... per segment
    uint64_t prev_seg_n[KMC_SEGS] = {0};  // stepping: segments of the level kmc_step_finish just retired (in frontier[cur ^ 1])
    uint64_t seg_cap = 0;        // slots per segment
    uint64_t level = 0;          // number of completed levels
    bool stepping = false, step_expanded = false, restored = false;
    std::vector<uint64_t> levels;
    std::vector<uint64_t> init_words, witness;
    bool have_witness = false, have_deadlock = false;
    bool witness_outside = false;      // the witness is a successor outside the state constraint:
    uint64_t witness_parent_fp = 0;    //   it is in no table; this is the expanded state it was generated from
    kmc_result res{};
    // kmc_config.symmetry: the frontier / table hold one state per orbit; res.distinct and `levels` are the WEIGHTED
    // (= plain-search) numbers, raw_levels the representatives per level; nfact = |Replicas|!
    uint64_t nfact = 1;
    int planes = 0;              // words per state in a frontier: W, and under symmetry one more — the order of the state's stabiliser
    double t_start = 0;
    double dry_seconds = 0;
    uint64_t prof[8] = {0}, prof_dry[8] = {0};
    // per-level exchange under the ABI (n_shards > 1): RCCL communicator, receive area, count/statistics rows
    ncclComm_t comm = nullptr;
    u64* recv = nullptr;             // device: everything this shard receives in one level, contiguous
    uint64_t recv_cap = 0;           // records
    // the within-level pipeline (kmc_step_level_parts): a second stream for a part's collective, transfer and insert, the
    // rows of two parts in flight, and the events that order the two streams
    hipStream_t xstream = nullptr;
    hipEvent_t ev_row[2] = {nullptr, nullptr}, ev_xfer[2] = {nullptr, nullptr};
    int64_t* prow_dev[2] = {nullptr, nullptr};
    int64_t* prow_host[2] = {nullptr, nullptr};
    int64_t* xrow_dev = nullptr;     // device: this rank's row, then the gathered rows of all ranks
    int64_t* xrow_host = nullptr;    // pinned: the same
    uint64_t last_send_counts[KMC_MAX_SHARDS * KMC_SEGS] = {0};  // of the last kmc_step_expand
    std::vector<uint64_t> xcounts;   // [source][destination][sub-buffer] of the level being exchanged
    bool xcounts_valid = false;
};

namespace {

// the small kernels (k_insert, k_init, k_find): the whole argument block
int launch(kmc_handle* h, hipFunction_t f, const KmcArgs& a, unsigned grid, hipStream_t stream = nullptr) {
    KmcArgs args = a;
    size_t size = sizeof(args);
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    HIP_TRY(hipModuleLaunchKernel(f, grid, 1, 1, KMC_BLOCK, 1, 1, 0, stream ? stream : h->stream, nullptr, config));
    return KMC_OK;
}

// The code object of k_expand in SHARDED / ENUM mode joins the handle when that mode is first asked for (from the cache; a
// cold cache compiles it: kmc_precompile builds all three ahead of time).
int ensure_mode(kmc_handle* h, unsigned mode) {
    if (mode == KMC_MODE_LOCAL || mode == KMC_MODE_DRY) return KMC_OK;
    hipModule_t& mod = mode == KMC_MODE_SHARDED ? h->mod_sh : h->mod_en;
    hipFunction_t& f = mode == KMC_MODE_SHARDED ? h->f_expand_sh : h->f_expand_en;
    if (f) return KMC_OK;
    std::vector<char> code;
    std::string kname;
    int rc = get_code_object(h->cfg, h->arch, &code, &kname, h->verify ? KMC_VERIFY_PRIMARY_OPTIONS : nullptr, nullptr, mode, &h->jit_defines);
    if (rc) return rc;
    HIP_TRY(hipModuleLoadData(&mod, code.data()));
    HIP_TRY(hipModuleGetFunction(&f, mod, (std::string("kmc_expand") + MODE_SUFFIX[mode] + "_" + h->kname).c_str()));
    return KMC_OK;
}

// k_expand in one of its modes (each mode is its own kernel; `verify` = the dry kernel of KMC_VERIFY's second build).  The
// search's kernel receives KmcArgsLocal — the head of the block — and nothing else.
int launch_expand(kmc_handle* h, unsigned mode, const KmcArgs& a, unsigned grid, hipStream_t stream = nullptr, bool verify = false) {
    int rc = ensure_mode(h, mode);
    if (rc) return rc;
    hipFunction_t f = verify ? h->f_expand_verify : mode == KMC_MODE_LOCAL ? h->f_expand : mode == KMC_MODE_SHARDED ? h->f_expand_sh
                    : mode == KMC_MODE_ENUM ? h->f_expand_en : h->f_expand_dry;
    if (!f)
        return fail(KMC_E_STATE, "k_expand's dry mode is only compiled into a tuning build (KMC_JIT_DEFINES=-DKMC_TUNING=1)");
    KmcArgs args = a;
    const bool meta = (args.flags & KMC_FLAG_TRACE) || mode == KMC_MODE_ENUM;   // k_expand carves its rings out of dynamic LDS
    if (meta) args.flags |= KMC_FLAG_META;
    const unsigned lds = kmc_expand_lds_bytes(h->W, meta, h->cfg.symmetry != 0);
    size_t size = mode == KMC_MODE_LOCAL && !verify ? sizeof(KmcArgsLocal) : sizeof(KmcArgs);
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    HIP_TRY(hipModuleLaunchKernel(f, grid, 1, 1, KMC_BLOCK, 1, 1, lds, stream ? stream : h->stream, nullptr, config));
    return KMC_OK;
}

// the invariants of the n states of a frontier that is not expanded (k_inv)
int launch_inv(kmc_handle* h, const KmcArgs& a, uint64_t n) {
    KmcArgsLocal args = a;
    uint64_t blocks = (n + KMC_BLOCK - 1) / KMC_BLOCK;
    const uint64_t maxb = (uint64_t)h->n_cus * 8;
    if (blocks > maxb) blocks = maxb;
    if (blocks < 1) blocks = 1;
    size_t size = sizeof(args);
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    HIP_TRY(hipModuleLaunchKernel(h->f_inv, (unsigned)blocks, 1, 1, KMC_BLOCK, 1, 1, 0, h->stream, nullptr, config));
    return KMC_OK;
}

KmcArgs base_args(kmc_handle* h, int ctl_slot) {
    KmcArgs a{};
    a.table = h->table;
    a.table_mask = h->table_cap - 1;
    a.pred = h->pred;
    a.sent = h->sent;
    a.sent_mask = h->sent_cap ? h->sent_cap - 1 : 0;
    a.ctl = h->ctl + ctl_slot;
    a.seed = h->cfg.hash_seed;
    a.inv_mask = h->cfg.invariant_mask;
    a.flags = (h->cfg.keep_trace ? KMC_FLAG_TRACE : 0u) | (h->slot_words == 2 ? KMC_FLAG_FP128 : 0u);
    a.nshards = (uint32_t)h->cfg.n_shards;
    a.shard = (uint32_t)h->cfg.shard_id;
    a.rec_words = (uint32_t)h->rec_words;
    a.fin_stride = a.fout_stride = h->fcap;
    a.seg_cap = h->seg_cap;
    for (int sg = 0; sg < KMC_SEGS; ++sg) a.seg_count[sg] = h->seg_n[sg];
    return a;
}

// Upper bound on the successors of one state = the number of action instances of the lowered Next (device header:
// KmcKafka::NINST etc.).  Sizes the grids of chained launches, whose input sizes only the device knows.
uint64_t max_fanout(const kmc_handle* h) {
    const uint64_t N = (uint64_t)h->cfg.n_replicas, L = (uint64_t)h->cfg.log_size, E1 = (uint64_t)h->cfg.max_leader_epoch + 1;
    switch (h->cfg.model) {
    case KMC_IDSEQUENCE: return 1;
    case KMC_FINITE_REPLICATED_LOG: return N * (uint64_t)h->cfg.n_log_records + N * L + N * (N - 1);
    case KMC_ASYNC_ISR: return (N - 1) + (1ull << N) + (N - 1) + N + 1 + (uint64_t)h->cfg.max_leader_epoch + (N - 1);
    default: {
        const uint64_t NP = N * (N - 1);
        return N + N + E1 * N + N * N + NP + N + N + NP * E1 + NP + (h->cfg.model == KMC_KIP320_FIRST_TRY ? NP : 0);
    }
    }
}

unsigned expand_grid(kmc_handle* h, uint64_t n) {
    // (orbit counting: k_expand shrinks its tiles down to 4 states when a level is small — KMC_SYMM, kmc_device.h — so the
    // grid is sized for that)
    const uint64_t per_tile = h->cfg.symmetry ? 4 : 64;
    const uint64_t tiles = (n + per_tile - 1) / per_tile;
    uint64_t blocks = (tiles + KMC_WAVES - 1) / KMC_WAVES;
    // one resident wave of blocks: more than the kernel's occupancy only queues blocks and
    // unbalances the tail (measured: 73 ms at 5 blocks/CU vs 59 ms at the resident 4)
    static const int forced = getenv("KMC_BLOCKS_PER_CU") ? atoi(getenv("KMC_BLOCKS_PER_CU")) : 0;
    const int per_cu = forced > 0 ? forced : h->blocks_per_cu;
    const uint64_t maxb = (uint64_t)h->n_cus * per_cu;
    if (blocks > maxb) blocks = maxb;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

int read_ctl(kmc_handle* h, int slot) {
    // a single-GPU level reports through the head of the block; the per-destination send counters behind it are
    // only written (and read back) in SHARDED mode
    const size_t bytes = h->cfg.n_shards > 1 || h->stepping ? sizeof(KmcLevelCtl) : KMC_CTL_LOCAL_BYTES;
    HIP_TRY(hipMemcpyAsync(h->ctl_host, h->ctl + slot, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return KMC_OK;
}

int zero_ctl(kmc_handle* h, int slot) {
    HIP_TRY(hipMemsetAsync(h->ctl + slot, 0, sizeof(KmcLevelCtl), h->stream));
    return KMC_OK;
}

// Segment sizes the device reported for the level it just produced (clipped to capacity).
uint64_t produced_segments(kmc_handle* h, const KmcLevelCtl& c, uint64_t seg[KMC_SEGS]) {
    uint64_t total = 0;
    for (int sg = 0; sg < KMC_SEGS; ++sg) {
        seg[sg] = c.next_count[sg].v < h->seg_cap ? c.next_count[sg].v : h->seg_cap;
        total += seg[sg];
    }
    return total;
}

int find_state(kmc_handle* h, const u64* frontier, const uint64_t seg[KMC_SEGS], uint64_t fp,
               std::vector<uint64_t>* out) {
    KmcArgs a = base_args(h, 2);
    a.fin = frontier;
    uint64_t n = 0;
    for (int sg = 0; sg < KMC_SEGS; ++sg) { a.seg_count[sg] = seg[sg]; n += seg[sg]; }
    a.table_mask = fp;  // kmc_find_body takes the target here
    a.send = h->scratch;
    HIP_TRY(hipMemsetAsync(h->scratch, 0xFF, (h->W + 1) * 8, h->stream));
    int rc = launch(h, h->f_find, a, expand_grid(h, n));
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(h->scratch_host, h->scratch, (h->W + 1) * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->scratch_host[h->W] == ~0ull) return fail(KMC_E_STATE, "witness fingerprint not found in frontier");
    out->assign(h->scratch_host, h->scratch_host + h->W);
    return KMC_OK;
}

// A violating successor outside the state constraint is in no table and no frontier: re-enumerate
// the successors of the expanded level, keeping those whose fingerprint is `fp` (ENUM_MATCH), to
// get its words and the parent it came from (the one with the smallest fingerprint).
int find_outside_witness(kmc_handle* h, const u64* frontier, const uint64_t seg[KMC_SEGS], uint64_t fp) {
    int rc = zero_ctl(h, 2);
    if (rc) return rc;
    KmcArgs a = base_args(h, 2);
    a.fin = frontier;
    uint64_t n = 0;
    for (int sg = 0; sg < KMC_SEGS; ++sg) { a.seg_count[sg] = seg[sg]; n += seg[sg]; }
    a.flags |= KMC_FLAG_ENUM_MATCH;
    a.match_fp = fp;
    a.send = h->enum_out;
    a.send_cap = h->enum_cap;
    a.inv_mask = 0;
    if ((rc = launch_expand(h, KMC_MODE_ENUM, a, expand_grid(h, n)))) return rc;
    if ((rc = read_ctl(h, 2))) return rc;
    const uint64_t cnt = h->ctl_host->enum_count < h->enum_cap ? h->ctl_host->enum_count : h->enum_cap;
    if (cnt == 0) return fail(KMC_E_STATE, "witness outside the constraint not found among the successors");
    std::vector<uint64_t> recs(cnt * (h->W + 2));
    HIP_TRY(hipMemcpy(recs.data(), h->enum_out, recs.size() * 8, hipMemcpyDeviceToHost));
    uint64_t best = 0;
    for (uint64_t i = 1; i < cnt; ++i)
        if (recs[i * (h->W + 2) + h->W + 1] < recs[best * (h->W + 2) + h->W + 1]) best = i;
    h->witness.assign(&recs[best * (h->W + 2)], &recs[best * (h->W + 2)] + h->W);
    h->witness_parent_fp = recs[best * (h->W + 2) + h->W + 1];
    h->witness_outside = true;
    return KMC_OK;
}

int reset_run(kmc_handle* h) {
    // (Clearing a second table on a side stream in the shadow of the run — a double-buffered seen-set — was measured in
    // round 3: the step got 0.4 ms shorter, but the memset's own kernel competes with the first, small levels and their
    // launches got 0.9 ms longer in total; dropped, profiles/r03_step_overhead.txt.)
    const double t_clear0 = now_s();
    HIP_TRY(hipMemsetAsync(h->table, 0, h->table_cap * h->slot_words * 8, h->stream));
    if (h->pred) HIP_TRY(hipMemsetAsync(h->pred, 0, h->table_cap * 8, h->stream));
    if (!h->first_clear_timed) {   // the first clear of a handle touches freshly mapped memory: timed once, by waiting for it
        h->first_clear_timed = true;
        HIP_TRY(hipStreamSynchronize(h->stream));
        h->timing.first_clear_s = now_s() - t_clear0;
    }
    if (h->sent) HIP_TRY(hipMemsetAsync(h->sent, 0, h->sent_cap * 8, h->stream));
    HIP_TRY(hipMemsetAsync(h->ctl, 0, KMC_CTL_SLOTS * sizeof(KmcLevelCtl), h->stream));
    h->levels.clear();
    h->witness.clear();
    h->have_witness = false;
    h->have_deadlock = false;
    h->witness_outside = false;
    h->witness_parent_fp = 0;
    memset(&h->res, 0, sizeof h->res);
    h->res.violated_invariant = -1;
    h->res.table_capacity = h->table_cap;
    h->res.frontier_capacity = h->fcap;
    h->res.state_words = h->W;
    h->res.state_bits = h->lay.bits;
    h->cur = 0;
    h->n_cur = 0;
    for (int sg = 0; sg < KMC_SEGS; ++sg) h->seg_n[sg] = 0;
    h->level = 0;
    h->t_start = now_s();
    return KMC_OK;
}

// Fold the device counters of one expansion into the running result.  `c` describes the expansion
// of the frontier at depth h->level (the "parent" level): invariant violations and deadlocks refer
// to ITS states, generated / next_count to the level it produced.  Returns true when the search
// must stop.  On a stopping invariant violation the produced level is rolled back (not counted), so
// the reported numbers are those of a checker that tests each state when it is first found.
// Conservation of successors through one level's kernels (always on; two counters per wave on the device):
//   what pass 2 of k_expand dispatched, less the repeats (one successor, two bindings) and the successors outside the state
//   constraint, plus the records k_insert was handed, must be what entered the sink:  generated - repeats - outside + inserted = probed
//   and every claim the sink won must have been appended to the next frontier:       won = sum(next_count).
// Round 1 met a build of k_expand that LOST successors between dispatch and sink (docs/TUNING_LOG_r1-r3.md §2); every counter the old
// self-check compared is bumped before that point.  These two are taken on either side of it.
// kmc_config.symmetry: a device counter counts orbit representatives and comes with the summed deficits of their orbits
// (KmcLevelCtl::corr_*): the plain search's count is N! * raw - corr.
uint64_t weighted(const kmc_handle* h, uint64_t raw, uint64_t corr) { return h->cfg.symmetry ? h->nfact * raw - corr : raw; }
// states on the frontier, as the plain search counts them (the frontier is always a whole level)
uint64_t queue_now(const kmc_handle* h) { return h->cfg.symmetry && !h->levels.empty() ? h->levels.back() : h->n_cur; }
// a level of `produced` stored states enters the books
void book_level(kmc_handle* h, uint64_t produced, const KmcLevelCtl& c) {
    h->res.orbit_representatives += produced;
    const uint64_t w = weighted(h, produced, c.corr_won);
    h->res.distinct += w;
    h->levels.push_back(w);
    // (Widths.  k_expand sums the orbit deficits of a launch's counts per LANE and per WAVE in 32 bits and per BLOCK in
    // 64-bit LDS cells (kmc_device.h, kmc_corr).  Round 3 had 32-bit block cells: at 17 levels of BASELINE config 5 — 133 M
    // stored states in a level, up to 5,039 per successor — they wrapped and `generated` came out 2^40 too large, found by
    // oracle/orbit_oracle.c.  A lane sees produced / (blocks x 256) states of a level: its sums stay below 2^26 for any level
    // the frontier can hold.)
}

int check_conservation(kmc_handle* h, const KmcLevelCtl& c, uint64_t inserted) {
    if (c.err) return KMC_OK;   // a full table / frontier / send area stops probing and appending on purpose
    uint64_t gen = 0, appended = 0;
    for (int k = 0; k < KMC_MAX_KINDS; ++k) gen += c.generated[k];
    for (int sg = 0; sg < KMC_SEGS; ++sg) appended += c.next_count[sg].v;
    const uint64_t expect = gen - c.repeats - c.outside + inserted;
    if (expect != c.probed)
        return fail(KMC_E_DEVICE, "conservation violated at level %llu of kmc_expand_%s: %llu successors were dispatched "
                                  "(%llu generated - %llu repeats - %llu outside the constraint + %llu inserted) but %llu reached "
                                  "the seen-set: the kernel lost or invented successors",
                    (unsigned long long)h->level, h->kname.c_str(), (unsigned long long)expect, (unsigned long long)gen,
                    (unsigned long long)c.repeats, (unsigned long long)c.outside, (unsigned long long)inserted,
                    (unsigned long long)c.probed);
    if (c.won != appended)
        return fail(KMC_E_DEVICE, "conservation violated at level %llu of kmc_expand_%s: %llu claims were won but %llu states "
                                  "were appended to the next frontier", (unsigned long long)h->level, h->kname.c_str(),
                    (unsigned long long)c.won, (unsigned long long)appended);
    return KMC_OK;
}

bool absorb(kmc_handle* h, const KmcLevelCtl& c, const u64* parent_frontier, const uint64_t* parent_seg, int* rc) {
    kmc_result& r = h->res;
    *rc = KMC_OK;
    if (c.err & KMC_ERR_CHECK_WORD) {
        *rc = fail(KMC_E_DEVICE, "wide fingerprints: a claimed slot's check word did not appear (level %llu)", (unsigned long long)h->level);
        r.verdict = KMC_V_ERROR;
        return true;
    }
    if ((*rc = check_conservation(h, c, 0))) {
        r.verdict = KMC_V_ERROR;
        return true;
    }
    if (r.violated_invariant < 0) {
        for (int k = 0; k < 4; ++k) {
            if ((h->cfg.invariant_mask >> k & 1u) && c.viol_count[k]) {
                r.violated_invariant = k;
                r.violation_depth = h->level;
                r.violation_fp = ~c.viol_fp_inv[k];
                for (int j = 0; j < 4; ++j) r.violation_count[j] = weighted(h, c.viol_count[j], c.corr_viol[j]);
                if (parent_frontier && h->cfg.n_shards == 1) {
                    *rc = find_state(h, parent_frontier, parent_seg, r.violation_fp, &h->witness);
                    h->have_witness = *rc == KMC_OK;
                }
                break;
            }
        }
        // successors outside the state constraint that violate an invariant: one level deeper than
        // the expanded states, so a violation among those takes precedence
        for (int k = 0; k < 4 && r.violated_invariant < 0; ++k) {
            if ((h->cfg.invariant_mask >> k & 1u) && c.oviol_count[k]) {
                r.violated_invariant = k;
                r.violation_depth = h->level + 1;
                r.violation_fp = ~c.oviol_fp_inv[k];
                for (int j = 0; j < 4; ++j) r.violation_count[j] = c.oviol_count[j];
                if (parent_frontier && h->cfg.n_shards == 1) {
                    *rc = find_outside_witness(h, parent_frontier, parent_seg, r.violation_fp);
                    h->have_witness = *rc == KMC_OK;
                }
            }
        }
        if (r.violated_invariant >= 0) {
            r.verdict = KMC_V_INVARIANT;
            if (!h->cfg.continue_on_violation) return true;
        }
    }
    for (int k = 0; k < KMC_MAX_KINDS; ++k) {
        const uint64_t g = weighted(h, c.generated[k], c.corr_gen[k]);
        r.generated += g;
        r.action_generated[k] += g;
    }
    r.generated_repeats += weighted(h, c.repeats, c.corr_repeats);
    r.deadlock_states += weighted(h, c.deadlock_count, c.corr_dead);
    if (c.err & KMC_ERR_TABLE_FULL) { r.verdict = KMC_V_TABLE_FULL; return true; }
    if (c.err & (KMC_ERR_FRONTIER_FULL | KMC_ERR_SEND_FULL)) { r.verdict = KMC_V_FRONTIER_FULL; return true; }
    if (h->cfg.check_deadlock && c.deadlock_count && (r.verdict == KMC_V_OK || r.verdict == KMC_V_INVARIANT) &&
        !h->have_deadlock) {
        h->have_deadlock = true;
        if (r.verdict == KMC_V_OK) {
            r.verdict = KMC_V_DEADLOCK;
            r.violation_depth = h->level;
            r.violation_fp = ~c.deadlock_fp_inv;
            if (parent_frontier && h->cfg.n_shards == 1) {
                *rc = find_state(h, parent_frontier, parent_seg, r.violation_fp, &h->witness);
                h->have_witness = *rc == KMC_OK;
            }
            return true;
        }
    }
    return false;
}

// Produce Init and insert it on its owner.  Leaves level = 1.
int do_begin(kmc_handle* h) {
    // a stepped search that stopped on a verdict never reached kmc_step_finish: a pipelined level's last transfer and insert
    // may still be in flight on the second stream, and its records are still booked — neither belongs to the new search
    if (h->xstream) HIP_TRY(hipStreamSynchronize(h->xstream));
    h->inserted_level = 0;
    int rc = reset_run(h);
    if (rc) return rc;
    KmcArgs a = base_args(h, 0);
    a.send = h->scratch;
    if ((rc = launch(h, h->f_init, a, 1))) return rc;
    HIP_TRY(hipMemcpyAsync(h->scratch_host, h->scratch, (h->W + 1) * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->init_words.assign(h->scratch_host, h->scratch_host + h->W);
    uint64_t init_orbit = 1;
    if (h->cfg.symmetry) {
        // Init is stored as its orbit's representative like every other state (the specs' Init is fixed by every
        // permutation, KafkaReplication.tla:109-120 / FiniteReplicatedLog.tla:97 — then nothing changes and the orbit is 1)
        unsigned long long c[KMC_MAXW] = {0}, w0[KMC_MAXW] = {0};
        for (int k = 0; k < h->W; ++k) w0[k] = h->init_words[k];
        int stab = 1;
        kmc_canonical_state_generic(h->lay, w0, c, &stab);
        init_orbit = h->nfact / (uint64_t)stab;
        for (int k = 0; k < h->W; ++k) h->init_words[k] = h->scratch_host[k] = c[k];
        h->scratch_host[h->W] = 0;
        HIP_TRY(hipMemcpyAsync(h->scratch, h->scratch_host, (h->W + 1) * 8, hipMemcpyHostToDevice, h->stream));
    }
    const uint64_t fp0 = kmc_fingerprint_of(h, h->init_words.data());
    const bool mine = h->cfg.n_shards <= 1 || kmc_owner(fp0, (uint32_t)h->cfg.n_shards) == (uint32_t)h->cfg.shard_id;
    if (mine) {
        KmcArgs b = base_args(h, 0);
        b.recv = h->scratch;
        b.n_in = 1;
        b.fout = h->frontier[0];
        if ((rc = launch(h, h->f_insert, b, 1))) return rc;
        h->res.generated = 1;
    }
    if ((rc = read_ctl(h, 0))) return rc;
    h->n_cur = produced_segments(h, *h->ctl_host, h->seg_n);
    h->cur = 0;
    h->res.orbit_representatives = h->n_cur;
    h->res.distinct = h->cfg.symmetry ? h->n_cur * init_orbit : h->n_cur;
    h->levels.push_back(h->res.distinct);
    h->level = 1;
    h->res.depth = 1;
    return rc;
}

}  // namespace

extern "C" {

const char* kmc_last_error(void) { return g_err.c_str(); }
const char* kmc_model_name(int32_t m) { return m >= 0 && m <= 7 ? MODEL_NAMES[m] : "?"; }
const char* kmc_invariant_name(int32_t i) { return i >= 0 && i < 4 ? INV_NAMES[i] : "?"; }
const char* kmc_model_invariant_name(int32_t model, int32_t i) {
    if (i < 0 || i >= 4) return "?";
    return model == KMC_ASYNC_ISR ? INV_NAMES_ASYNC[i] : INV_NAMES[i];
}
int32_t kmc_action_count(int32_t model) {
    switch (model) {
    case KMC_IDSEQUENCE: return 1;
    case KMC_FINITE_REPLICATED_LOG: return 3;
    case KMC_ASYNC_ISR: return 7;
    case KMC_KIP320_FIRST_TRY: return 10;
    case KMC_TRUNCATE_TO_HW: case KMC_KIP101: case KMC_KIP279: case KMC_KIP320: return 9;
    default: return 0;
    }
}
const char* kmc_action_name(int32_t model, int32_t kind) {
    if (kind < 0 || kind >= kmc_action_count(model)) return "?";
    switch (model) {
    case KMC_IDSEQUENCE: return "Next";
    case KMC_FINITE_REPLICATED_LOG: return KINDS_FRL[kind];
    case KMC_ASYNC_ISR: return KINDS_ASYNC[kind];
    case KMC_KIP320: return KINDS_KIP320[kind];
    case KMC_KIP320_FIRST_TRY: return KINDS_FIRST[kind];
    default:
        if (kind == 7)
            return model == KMC_TRUNCATE_TO_HW ? "BecomeFollowerTruncateToHighWatermark"
                   : model == KMC_KIP101       ? "BecomeFollowerTruncateKip101"
                                               : "BecomeFollowerTruncateKip279";
        return KINDS_BASE[kind];
    }
}

// mode: 0 the search's own code object (k_expand LOCAL + the small kernels), 1 k_expand SHARDED (the level-step interface),
// 2 k_expand ENUM (kmc_successors, trace replay); -1 all three.  A build script spreads the modes over its workers.
int kmc_precompile_mode(const kmc_config* cfg, const char* arch, int32_t mode) {
    if (!cfg) return fail(KMC_E_ARG, "null config");
    if (mode < -1 || mode > (int32_t)KMC_MODE_ENUM) return fail(KMC_E_ARG, "mode %d: expected -1 (all), 0 (search), 1 (sharded), 2 (enum)", mode);
    std::vector<char> code;
    std::string kname;
    const bool verify = getenv("KMC_VERIFY") && atoi(getenv("KMC_VERIFY"));
    for (unsigned m = 0; m <= KMC_MODE_ENUM; ++m) {
        if (mode >= 0 && (unsigned)mode != m) continue;
        int rc = get_code_object(*cfg, arch ? arch : "gfx950", &code, &kname, verify ? KMC_VERIFY_PRIMARY_OPTIONS : nullptr, nullptr, m);
        if (rc) return rc;
    }
    // with KMC_VERIFY set: also the second build kmc_open would load for the differential self-check
    if (verify && mode <= 0) return get_code_object(*cfg, arch ? arch : "gfx950", &code, &kname, KMC_VERIFY_OPTIONS);
    return KMC_OK;
}
int kmc_precompile(const kmc_config* cfg, const char* arch) { return kmc_precompile_mode(cfg, arch, -1); }

// Where the code object of cfg's kernels lives in the cache (compiled first if it is not there yet): the identity of the
// device code a measurement belongs to is the kernels' machine code, not the text of a header that also holds other builds.
int kmc_code_object_path(const kmc_config* cfg, const char* arch, char* out, uint64_t cap) {
    if (!cfg || !out || !cap) return fail(KMC_E_ARG, "null config / buffer");
    std::vector<char> code;
    std::string kname, path;
    int rc = get_code_object(*cfg, arch ? arch : "gfx950", &code, &kname, nullptr, &path);
    if (rc) return rc;
    if (path.size() + 1 > cap) return fail(KMC_E_ARG, "path of %zu bytes does not fit %llu", path.size(), (unsigned long long)cap);
    memcpy(out, path.c_str(), path.size() + 1);
    return KMC_OK;
}

static void comm_release(kmc_handle* h);

void kmc_close(kmc_handle* h) {
    if (!h) return;
    if (h->cfg.device < 0 || !h->stream) {  // host-only handle, or open failed before any device work
        if (h->mod) hipModuleUnload(h->mod);
        delete h;
        return;
    }
    hipSetDevice(h->cfg.device);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->table) hipFree(h->table);
    if (h->pred) hipFree(h->pred);
    if (h->table2) hipFree(h->table2);
    if (h->sent) hipFree(h->sent);
    if (h->frontier[0]) hipFree(h->frontier[0]);
    if (h->frontier[1]) hipFree(h->frontier[1]);
    if (h->ctl) hipFree(h->ctl);
    if (h->scratch) hipFree(h->scratch);
    if (h->enum_out) hipFree(h->enum_out);
    if (h->send && h->send_owned) hipFree(h->send);
    comm_release(h);
    if (h->recv) hipFree(h->recv);
    if (h->xstream) hipStreamSynchronize(h->xstream);
    for (int i = 0; i < 2; ++i) {
        if (h->prow_dev[i]) hipFree(h->prow_dev[i]);
        if (h->prow_host[i]) hipHostFree(h->prow_host[i]);
        if (h->ev_row[i]) hipEventDestroy(h->ev_row[i]);
        if (h->ev_xfer[i]) hipEventDestroy(h->ev_xfer[i]);
    }
    if (h->xstream) hipStreamDestroy(h->xstream);
    if (h->xrow_dev) hipFree(h->xrow_dev);
    if (h->xrow_host) hipHostFree(h->xrow_host);
    if (h->ctl_host) hipHostFree(h->ctl_host);
    if (h->scratch_host) hipHostFree(h->scratch_host);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    for (hipEvent_t e : h->ev_chain)
        if (e) hipEventDestroy(e);
    if (h->stream) hipStreamDestroy(h->stream);
    if (h->mod_verify) hipModuleUnload(h->mod_verify);
    if (h->mod_sh) hipModuleUnload(h->mod_sh);
    if (h->mod_en) hipModuleUnload(h->mod_en);
    if (h->mod) hipModuleUnload(h->mod);
    delete h;
}

static int open_impl(const kmc_config* cfg, kmc_handle* h) {
    h->cfg = *cfg;
    if (cfg->cache_dir) {
        h->cache_dir = cfg->cache_dir;
        h->cfg.cache_dir = h->cache_dir.c_str();
    }
    if (h->cfg.n_shards < 1) h->cfg.n_shards = 1;
    if (h->cfg.n_shards > KMC_MAX_SHARDS || h->cfg.shard_id < 0 || h->cfg.shard_id >= h->cfg.n_shards)
        return fail(KMC_E_ARG, "bad shard configuration %d/%d", h->cfg.shard_id, h->cfg.n_shards);
    std::string name, inst;
    if (!validate(h->cfg, &h->lay, &name, &inst)) {
        std::vector<char> dummy;
        return get_code_object(h->cfg, "gfx950", &dummy, &name);  // produces the KMC_E_ARG message
    }
    h->W = h->lay.W;
    h->nfact = h->cfg.symmetry ? (uint64_t)kmc_factorial(h->cfg.n_replicas) : 1;
    h->planes = h->W + (h->cfg.symmetry ? 1 : 0);
    if (h->cfg.symmetry && !kmc_model_symmetric(h->cfg.model))
        return fail(KMC_E_ARG, "symmetry (orbit counting): %s singles out a replica or has none", MODEL_NAMES[h->cfg.model]);
    h->rec_words = h->W + (cfg->keep_trace ? 1 : 0);
    if (cfg->device == -1) return KMC_OK;  // host-only handle: pack/unpack/fingerprint, no device work
    const double t_open0 = now_s();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(KMC_E_DEVICE, "no HIP device visible: this library has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(KMC_E_ARG, "device %d out of range (%d)", cfg->device, ndev);
    HIP_TRY(hipSetDevice(cfg->device));
    HIP_TRY(hipFree(nullptr));   // (the device's context is created here, not inside the first allocation: it is timed as what it is)
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
    const double t_init1 = now_s();
    h->timing.hip_init_s = t_init1 - t_open0;
    std::string arch = prop.gcnArchName;
    size_t colon = arch.find(':');
    if (colon != std::string::npos) arch = arch.substr(0, colon);
    h->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    h->arch = arch;

    std::vector<char> code;
    const bool verify = getenv("KMC_VERIFY") && atoi(getenv("KMC_VERIFY"));
    h->verify = verify;
    h->jit_defines = getenv("KMC_JIT_DEFINES") ? getenv("KMC_JIT_DEFINES") : "";
    int rc = get_code_object(h->cfg, arch, &code, &h->kname, verify ? KMC_VERIFY_PRIMARY_OPTIONS : nullptr);
    if (rc) return rc;
    HIP_TRY(hipModuleLoadData(&h->mod, code.data()));
    HIP_TRY(hipModuleGetFunction(&h->f_expand, h->mod, ("kmc_expand_" + h->kname).c_str()));
    if (hipModuleGetFunction(&h->f_expand_dry, h->mod, ("kmc_expand_dry_" + h->kname).c_str()) != hipSuccess) {
        h->f_expand_dry = nullptr;   // (not a tuning build)
        (void)hipGetLastError();
    }
    HIP_TRY(hipModuleGetFunction(&h->f_inv, h->mod, ("kmc_inv_" + h->kname).c_str()));
    HIP_TRY(hipModuleGetFunction(&h->f_insert, h->mod, ("kmc_insert_" + h->kname).c_str()));
    HIP_TRY(hipModuleGetFunction(&h->f_init, h->mod, ("kmc_init_" + h->kname).c_str()));
    HIP_TRY(hipModuleGetFunction(&h->f_find, h->mod, ("kmc_find_" + h->kname).c_str()));
    HIP_TRY(hipModuleGetFunction(&h->f_packrow, h->mod, ("kmc_packrow_" + h->kname).c_str()));
    if (verify) {
        // Differential self-check for constants no oracle can reach (round 1 met a k_expand build that LOST successors
        // under heavy register spilling): a second code object of the same source, compiled at -O1 with a quarter of
        // the occupancy target and with the guards lowered the other way (KMC_VERIFY_OPTIONS), re-generates every level's
        // successors (DRY mode: no table, no frontier) and the per-action counts, deadlock counts, violation counts and
        // the checksum of the successors' fingerprints of the two builds must agree.
        std::vector<char> vcode;
        std::string vname;
        rc = get_code_object(h->cfg, arch, &vcode, &vname, KMC_VERIFY_OPTIONS);
        if (rc) return rc;
        HIP_TRY(hipModuleLoadData(&h->mod_verify, vcode.data()));
        HIP_TRY(hipModuleGetFunction(&h->f_expand_verify, h->mod_verify, ("kmc_expand_dry_" + vname).c_str()));
    }
    const double t_code1 = now_s();
    h->timing.code_object_s = t_code1 - t_init1;
    int occ = 0;
    if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&occ, h->f_expand, KMC_BLOCK,
                                                           kmc_expand_lds_bytes(h->W, cfg->keep_trace != 0, cfg->symmetry != 0)) == hipSuccess && occ > 0)
        h->blocks_per_cu = occ > 8 ? 8 : occ;
    {   // the occupancy query may admit a block more than really fits when LDS is the limit
        // (5 x 32 KiB = all 160 KiB was reported resident, ran as 4 + a queued 5th: 69 ms vs 55 ms)
        const unsigned lds = kmc_expand_lds_bytes(h->W, cfg->keep_trace != 0, cfg->symmetry != 0);
        const int by_lds = (int)((160u * 1024u - 1024u) / (lds ? lds : 1u));
        if (by_lds >= 1 && h->blocks_per_cu > by_lds) h->blocks_per_cu = by_lds;
    }
    HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&h->ev0));
    HIP_TRY(hipEventCreate(&h->ev1));

    // ---- sizing: table slots, frontier states, send records -------------------------------
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    const double budget = 0.85 * (double)free_b;
    h->slot_words = cfg->wide_fingerprint ? 2 : 1;
    const uint64_t slot_bytes = 8 * h->slot_words + (cfg->keep_trace ? 8 : 0);
    // auto-sizing: half of the budget for the table; a shard also keeps a sender-side filter of twice the table
    // (0.15 + 0.30), two frontiers (2 x 0.09), a send area and a receive area (0.12 each)
    const double table_share = h->cfg.n_shards > 1 ? 0.15 : 0.5;
    uint64_t tcap = cfg->table_capacity ? pow2_ceil(cfg->table_capacity)
                                        : pow2_floor((uint64_t)(budget * table_share) / slot_bytes);
    if (tcap < 1024) tcap = 1024;
    uint64_t fcap = cfg->frontier_capacity;
    if (!fcap) {
        const double share = h->cfg.n_shards > 1 ? 0.09 : 0.20;
        fcap = (uint64_t)(budget * share) / (8ull * h->planes);
        if (fcap > tcap) fcap = tcap;
    }
    if (fcap < 64) fcap = 64;
    fcap = (fcap + 64 * KMC_SEGS - 1) / (64 * KMC_SEGS) * (64 * KMC_SEGS);  // segments start 512-byte aligned
    h->table_cap = tcap;
    h->fcap = fcap;
    h->seg_cap = fcap / KMC_SEGS;
    if (hipMalloc(&h->table, tcap * h->slot_words * 8) != hipSuccess) return fail(KMC_E_NOMEM, "cannot allocate %llu table slots", (unsigned long long)tcap);
    if (cfg->keep_trace && hipMalloc(&h->pred, tcap * 8) != hipSuccess) return fail(KMC_E_NOMEM, "cannot allocate predecessor table");
    for (int i = 0; i < 2; ++i)
        if (hipMalloc(&h->frontier[i], fcap * 8ull * h->planes) != hipSuccess)
            return fail(KMC_E_NOMEM, "cannot allocate frontier of %llu states", (unsigned long long)fcap);
    HIP_TRY(hipMalloc(&h->ctl, KMC_CTL_SLOTS * sizeof(KmcLevelCtl)));
    HIP_TRY(hipHostMalloc(&h->ctl_host, KMC_CHAIN * sizeof(KmcLevelCtl)));
    HIP_TRY(hipMalloc(&h->scratch, 64 * 8));
    HIP_TRY(hipHostMalloc(&h->scratch_host, 64 * 8));
    HIP_TRY(hipMalloc(&h->enum_out, h->enum_cap * (h->W + 2) * 8ull));
    // Sender-side duplicate filter: a shard generates (and would ship) a remote state several times.  What it saves
    // shrinks with P (each copy of a state is generated on a different shard: 49 % of the remote successors dropped at
    // P = 2, 31 % at 4, 18 % at 8 on the headline) while every remote successor pays one more random probe for it, so it
    // is on where the wire is the bottleneck (P <= 4: one to three xGMI links per GPU carry everything) and off beyond
    // (profiles/r02_loopback_filter.jsonl: k_expand per shard 11.6 -> 8.1 ms at P = 8).  KMC_SEND_FILTER=1 / 0 forces it.
    bool want_filter = h->cfg.n_shards > 1 && h->cfg.n_shards <= 4;
    if (const char* e = getenv("KMC_SEND_FILTER")) want_filter = h->cfg.n_shards > 1 && atoi(e) != 0;
    if (getenv("KMC_NO_SEND_FILTER") && atoi(getenv("KMC_NO_SEND_FILTER"))) want_filter = false;
    // 128-bit entries: the sender-side filter remembers 64-bit fingerprints only — a second distinct remote state with the
    // same fingerprint would be dropped at the sender and never meet the owner's check-word comparison, and the conservation
    // law (probed is counted before the filter) could not see it.  No filter then, whatever the environment asks for.
    if (h->cfg.wide_fingerprint) want_filter = false;
    if (want_filter) {
        // it may meet up to ~2x as many distinct remote fingerprints as it owns
        h->sent_cap = tcap * 2;
        if (hipMalloc(&h->sent, h->sent_cap * 8) != hipSuccess) { h->sent = nullptr; h->sent_cap = 0; }  // optional
    }
    if (h->cfg.n_shards > 1) {
        uint64_t scap = cfg->send_capacity;
        if (!scap) scap = (uint64_t)(budget * 0.12) / (8ull * h->rec_words * h->cfg.n_shards * KMC_SEGS);
        if (scap < 64) scap = 64;
        h->send_cap = scap;  // records per (destination, sub-buffer)
        if (hipMalloc(&h->send, scap * h->rec_words * 8ull * h->cfg.n_shards * KMC_SEGS) != hipSuccess)
            return fail(KMC_E_NOMEM, "cannot allocate send buffers");
    }
    {
        size_t free_after = 0, total_after = 0;
        if (hipMemGetInfo(&free_after, &total_after) == hipSuccess && free_b > free_after) h->timing.device_bytes = free_b - free_after;
    }
    h->timing.alloc_s = now_s() - t_code1;
    h->timing.open_s = now_s() - t_open0;
    return KMC_OK;
}

int kmc_open(const kmc_config* cfg, kmc_handle** out) {
    if (!cfg || !out) return fail(KMC_E_ARG, "null argument");
    *out = nullptr;
    kmc_handle* h = new kmc_handle();
    int rc = open_impl(cfg, h);
    if (rc) {
        std::string keep = g_err;
        kmc_close(h);
        g_err = keep;
        return rc;
    }
    *out = h;
    return KMC_OK;
}

uint64_t kmc_state_words(kmc_handle* h) { return h ? h->W : 0; }

uint64_t kmc_fingerprint_of(kmc_handle* h, const uint64_t* words) {
    unsigned long long w[KMC_MAXW];
    for (int k = 0; k < h->W; ++k) w[k] = words[k];
    switch (h->W) {
    case 1: return kmc_fingerprint<1>(w, h->cfg.hash_seed);
    case 2: return kmc_fingerprint<2>(w, h->cfg.hash_seed);
    case 3: return kmc_fingerprint<3>(w, h->cfg.hash_seed);
    case 4: return kmc_fingerprint<4>(w, h->cfg.hash_seed);
    case 5: return kmc_fingerprint<5>(w, h->cfg.hash_seed);
    case 6: return kmc_fingerprint<6>(w, h->cfg.hash_seed);
    case 7: return kmc_fingerprint<7>(w, h->cfg.hash_seed);
    case 8: return kmc_fingerprint<8>(w, h->cfg.hash_seed);
    case 9: return kmc_fingerprint<9>(w, h->cfg.hash_seed);
    case 10: return kmc_fingerprint<10>(w, h->cfg.hash_seed);
    case 11: return kmc_fingerprint<11>(w, h->cfg.hash_seed);
    default: return kmc_fingerprint<12>(w, h->cfg.hash_seed);
    }
}

int kmc_canonical_state(kmc_handle* h, const uint64_t* words, uint64_t* representative, int32_t* stabiliser) {
    if (!h || !words || !representative) return fail(KMC_E_ARG, "null argument");
    if (!kmc_model_symmetric(h->lay.model)) return fail(KMC_E_ARG, "%s has no replica symmetry", MODEL_NAMES[h->lay.model]);
    unsigned long long w[KMC_MAXW] = {0}, c[KMC_MAXW] = {0};
    for (int k = 0; k < h->W; ++k) w[k] = words[k];
    int stab = 1;
    kmc_canonical_state_generic(h->lay, w, c, &stab);
    for (int k = 0; k < h->W; ++k) representative[k] = c[k];
    if (stabiliser) *stabiliser = stab;
    return KMC_OK;
}

uint64_t kmc_canon_bytes(kmc_handle* h) {
    const KmcLayout& y = h->lay;
    if (y.model == KMC_IDSEQUENCE) return 8;
    if (y.model == KMC_FINITE_REPLICATED_LOG) return (uint64_t)y.N * (1 + y.L);
    if (y.model == KMC_ASYNC_ISR) return 6 + y.N + (uint64_t)(y.E + 1) * (((1 << y.N) + 7) / 8) + (y.E + 1);
    return (uint64_t)y.N * (5 + y.L) + 5 + 2 * (y.E + 1);
}

int kmc_unpack_state(kmc_handle* h, const uint64_t* words, uint8_t* c) {
    const KmcLayout& y = h->lay;
    unsigned long long w[KMC_MAXW + 1] = {0};
    for (int k = 0; k < h->W; ++k) w[k] = words[k];
    if (y.model == KMC_IDSEQUENCE) {
        memcpy(c, &w[0], 8);
        return KMC_OK;
    }
    if (y.model == KMC_FINITE_REPLICATED_LOG) {
        for (int r = 0; r < y.N; ++r) {
            uint8_t* b = c + r * (1 + y.L);
            b[0] = (uint8_t)kmc_getbits(w, y.end_off[r], y.BO);
            for (int o = 0; o < y.L; ++o) b[1 + o] = (uint8_t)kmc_getbits(w, y.log_off[r] + o * y.BR, y.BR);
        }
        return KMC_OK;
    }
    if (y.model == KMC_ASYNC_ISR) {
        const int ns = 1 << y.N, rb = (ns + 7) / 8;
        c[0] = (uint8_t)kmc_getbits(w, y.a_cisr, y.N);
        c[1] = (uint8_t)kmc_getbits(w, y.a_cver, y.BV);
        c[2] = (uint8_t)kmc_getbits(w, y.a_lisr, y.N);
        c[3] = (uint8_t)kmc_getbits(w, y.a_lver, y.BV);
        c[4] = (uint8_t)kmc_getbits(w, y.a_pisr, y.N);
        c[5] = (uint8_t)kmc_getbits(w, y.a_pver, y.BV);
        for (int r = 0; r < y.N; ++r) c[6 + r] = (uint8_t)kmc_getbits(w, y.a_off[r], y.BF);
        uint8_t* q = c + 6 + y.N;
        memset(q, 0, (size_t)(y.E + 1) * rb);
        for (int v = 0; v <= y.E; ++v)
            for (int m = 0; m < ns; ++m)
                if (kmc_getbits(w, y.a_req + v * ns + m, 1)) q[v * rb + (m >> 3)] |= (uint8_t)(1u << (m & 7));
        uint8_t* u = q + (y.E + 1) * rb;
        for (int v = 0; v <= y.E; ++v) u[v] = (uint8_t)kmc_getbits(w, y.a_upd + v * y.N, y.N);
        return KMC_OK;
    }
    const int rs = 5 + y.L;
    for (int r = 0; r < y.N; ++r) {
        uint8_t* b = c + r * rs;
        b[0] = (uint8_t)kmc_getbits(w, y.end_off[r], y.BO);
        b[1] = (uint8_t)kmc_getbits(w, y.hw_off[r], y.BO);
        b[2] = (uint8_t)kmc_getbits(w, y.ep_off[r], y.BE);
        b[3] = (uint8_t)kmc_getbits(w, y.ldr_off[r], y.BL);
        b[4] = (uint8_t)kmc_getbits(w, y.isr_off[r], y.BI);
        for (int o = 0; o < y.L; ++o) {
            const unsigned rec = (unsigned)kmc_getbits(w, y.log_off[r] + o * y.BR, y.BR);
            // packed (id+1)<<BEr | epoch  ->  canonical 1 + id*(E+1) + epoch
            b[5 + o] = rec == 0 ? 0 : (uint8_t)(1 + ((rec >> y.BEr) - 1) * (y.E + 1) + (rec & ((1u << y.BEr) - 1)));
        }
    }
    uint8_t* g = c + y.N * rs;
    g[0] = (uint8_t)kmc_getbits(w, y.nextrec_off, y.BNR);
    g[1] = (uint8_t)kmc_getbits(w, y.nextep_off, y.BE);
    g[2] = (uint8_t)kmc_getbits(w, y.qep_off, y.BE);
    g[3] = (uint8_t)kmc_getbits(w, y.qldr_off, y.BL);
    g[4] = (uint8_t)kmc_getbits(w, y.qisr_off, y.BI);
    for (int e = 0; e <= y.E; ++e) {
        g[5 + 2 * e] = (uint8_t)kmc_getbits(w, y.reqldr_off[e], y.BL);
        g[6 + 2 * e] = (uint8_t)kmc_getbits(w, y.reqisr_off[e], y.BI);
    }
    return KMC_OK;
}

int kmc_pack_state(kmc_handle* h, const uint8_t* c, uint64_t* words) {
    const KmcLayout& y = h->lay;
    unsigned long long w[KMC_MAXW + 1] = {0};
    if (y.model == KMC_IDSEQUENCE) {
        memcpy(&w[0], c, 8);
    } else if (y.model == KMC_FINITE_REPLICATED_LOG) {
        for (int r = 0; r < y.N; ++r) {
            const uint8_t* b = c + r * (1 + y.L);
            kmc_setbits(w, y.end_off[r], y.BO, b[0]);
            for (int o = 0; o < y.L; ++o) kmc_setbits(w, y.log_off[r] + o * y.BR, y.BR, b[1 + o]);
        }
    } else if (y.model == KMC_ASYNC_ISR) {
        const int ns = 1 << y.N, rb = (ns + 7) / 8;
        kmc_setbits(w, y.a_cisr, y.N, c[0]);
        kmc_setbits(w, y.a_cver, y.BV, c[1]);
        kmc_setbits(w, y.a_lisr, y.N, c[2]);
        kmc_setbits(w, y.a_lver, y.BV, c[3]);
        kmc_setbits(w, y.a_pisr, y.N, c[4]);
        kmc_setbits(w, y.a_pver, y.BV, c[5]);
        for (int r = 0; r < y.N; ++r) kmc_setbits(w, y.a_off[r], y.BF, c[6 + r]);
        const uint8_t* q = c + 6 + y.N;
        for (int v = 0; v <= y.E; ++v)
            for (int m = 0; m < ns; ++m)
                if (q[v * rb + (m >> 3)] >> (m & 7) & 1) kmc_setbits(w, y.a_req + v * ns + m, 1, 1);
        const uint8_t* u = q + (y.E + 1) * rb;
        for (int v = 0; v <= y.E; ++v) kmc_setbits(w, y.a_upd + v * y.N, y.N, u[v]);
    } else {
        const int rs = 5 + y.L;
        for (int r = 0; r < y.N; ++r) {
            const uint8_t* b = c + r * rs;
            kmc_setbits(w, y.end_off[r], y.BO, b[0]);
            kmc_setbits(w, y.hw_off[r], y.BO, b[1]);
            kmc_setbits(w, y.ep_off[r], y.BE, b[2]);
            kmc_setbits(w, y.ldr_off[r], y.BL, b[3]);
            kmc_setbits(w, y.isr_off[r], y.BI, b[4]);
            for (int o = 0; o < y.L; ++o) {
                const unsigned code = b[5 + o];
                const unsigned rec = code == 0 ? 0 : ((((code - 1) / (y.E + 1)) + 1) << y.BEr) | ((code - 1) % (y.E + 1));
                kmc_setbits(w, y.log_off[r] + o * y.BR, y.BR, rec);
            }
        }
        const uint8_t* g = c + y.N * rs;
        kmc_setbits(w, y.nextrec_off, y.BNR, g[0]);
        kmc_setbits(w, y.nextep_off, y.BE, g[1]);
        kmc_setbits(w, y.qep_off, y.BE, g[2]);
        kmc_setbits(w, y.qldr_off, y.BL, g[3]);
        kmc_setbits(w, y.qisr_off, y.BI, g[4]);
        for (int e = 0; e <= y.E; ++e) {
            kmc_setbits(w, y.reqldr_off[e], y.BL, g[5 + 2 * e]);
            kmc_setbits(w, y.reqisr_off[e], y.BI, g[6 + 2 * e]);
        }
    }
    for (int k = 0; k < h->W; ++k) words[k] = w[k];
    return KMC_OK;
}

static int run_levels(kmc_handle* h, kmc_progress_cb cb, void* user, bool fresh);


int kmc_run(kmc_handle* h, kmc_progress_cb cb, void* user) {
    if (!h) return fail(KMC_E_ARG, "null handle");
    if (!h->table) return fail(KMC_E_STATE, "host-only handle (device = -1) cannot run");
    if (h->cfg.n_shards != 1) return fail(KMC_E_STATE, "kmc_run drives one GPU; use the kmc_step_* interface for shards");
    HIP_TRY(hipSetDevice(h->cfg.device));
    h->stepping = false;
    int rc = do_begin(h);
    if (rc) return rc;
    return run_levels(h, cb, user, true);
}

// TLC -recover analogue: continue the search of a handle restored by kmc_checkpoint_load.
int kmc_resume(kmc_handle* h, kmc_progress_cb cb, void* user) {
    if (!h) return fail(KMC_E_ARG, "null handle");
    if (!h->table || !h->restored) return fail(KMC_E_STATE, "kmc_resume needs a handle restored by kmc_checkpoint_load");
    HIP_TRY(hipSetDevice(h->cfg.device));
    h->stepping = false;
    h->restored = false;
    h->t_start = now_s() - h->res.seconds_total;
    if (h->res.verdict == KMC_V_LEVEL_LIMIT) h->res.verdict = KMC_V_OK;  // the limit that stopped the saved run is lifted
    h->res.queue_left = 0;
    return run_levels(h, cb, user, false);
}

static int run_levels(kmc_handle* h, kmc_progress_cb cb, void* user, bool fresh) {
    int rc = KMC_OK;
    kmc_result& r = h->res;
    auto report = [&]() {
        if (!cb) return;
        kmc_level_info info{};
        info.depth = h->level;
        info.new_states = queue_now(h);
        info.generated_total = r.generated;
        info.distinct_total = r.distinct;
        info.seconds = now_s() - h->t_start;
        cb(&info, user);
    };
    if (fresh) report();
    bool stop = false;
    const uint64_t max_levels = h->cfg.max_levels ? h->cfg.max_levels : ~0ull;
    while (!stop && h->n_cur > 0) {
        if (h->level >= max_levels) {
            // the last frontier is not expanded: give its states their invariant check now
            if ((rc = zero_ctl(h, 2))) return rc;
            KmcArgs d = base_args(h, 2);
            d.fin = h->frontier[h->cur];
            if ((rc = launch_inv(h, d, h->n_cur))) return rc;   // (a full dry expansion of BASELINE config 5's tenth level took 64 ms: twice the search)
            if ((rc = read_ctl(h, 2))) return rc;
            KmcLevelCtl c = *h->ctl_host;
            for (int k = 0; k < KMC_MAX_KINDS; ++k) c.generated[k] = 0;
            c.deadlock_count = 0;
            c.err = 0;
            c.probed = c.won = c.outside = c.repeats = 0;   // an invariant-only pass: nothing was dispatched for the record
            for (int k = 0; k < KMC_MAX_KINDS; ++k) c.corr_gen[k] = 0;
            c.corr_dead = c.corr_repeats = c.corr_won = 0;
            absorb(h, c, h->frontier[h->cur], h->seg_n, &rc);
            if (rc) return rc;
            if (r.verdict == KMC_V_OK) r.verdict = KMC_V_LEVEL_LIMIT;
            r.queue_left = queue_now(h);
            break;
        }
        static const int shadow = getenv("KMC_SHADOW") ? atoi(getenv("KMC_SHADOW")) : 0;
        static const int dry_mode = getenv("KMC_DRYRUN") ? atoi(getenv("KMC_DRYRUN")) : 0;
        static const int no_chain = getenv("KMC_NO_CHAIN") ? atoi(getenv("KMC_NO_CHAIN")) : 0;
        // Under -continue a violation does not end the search, so levels queued behind the violating one would run and
        // overwrite its parent frontier before the host could fetch the witness (find_state / find_outside_witness):
        // such runs go level by level until the first violation has been recorded, and chain from there on.
        const bool witness_pending = h->cfg.continue_on_violation && h->cfg.invariant_mask && r.violated_invariant < 0;
        if (!cb && !shadow && !dry_mode && !no_chain && !h->f_expand_verify && !witness_pending) {
            // ---- chained launches -------------------------------------------------------------------------
            // Nobody watches the levels go by, so up to KMC_CHAIN of them are queued back to back and the host
            // waits ONCE: a level launched behind another one takes its segment sizes from that level's control
            // block on the device and does nothing if that level (or one before it) ended the search.  Per level
            // this leaves a launch and two event records on the host instead of memset + launch + copy + wait
            // (46 levels, 7 of them under 1024 states: 2.4 ms of a 38 ms check in round 1).
            uint64_t B = max_levels - h->level;
            static const uint64_t chain_max = getenv("KMC_CHAIN_MAX") ? (uint64_t)atoi(getenv("KMC_CHAIN_MAX")) : 16;
            if (B > KMC_CHAIN) B = KMC_CHAIN;
            if (chain_max >= 1 && B > chain_max) B = chain_max;
            const uint64_t fan = max_fanout(h) ? max_fanout(h) : 1;
            {
                // The load limit of the table (0.92, below) is a HOST decision, taken after a level: a batch is therefore
                // only as long as its levels provably stay under it — each level adds at most min(fan x its input,
                // frontier capacity) states.  (Without this a batch could run the table far past the limit before the
                // host looked, and after such a stop h->cur / seg_n no longer described the device's frontier: ADVICE r2.)
                const double room = 0.92 * (double)h->table_cap - (double)r.orbit_representatives;
                uint64_t in = h->n_cur, fit = 0;
                double sum = 0;
                for (; fit < B; ++fit) {
                    const uint64_t out = in > h->fcap / fan ? h->fcap : in * fan;
                    sum += (double)out;
                    if (fit > 0 && sum > room) break;   // (the first level always runs: the host checks right after it)
                    in = out;
                }
                B = fit ? fit : 1;
            }
            HIP_TRY(hipMemsetAsync(h->ctl + 3, 0, B * sizeof(KmcLevelCtl), h->stream));
            uint64_t bound = h->n_cur;   // upper bound on the size of the level launch i expands
            for (uint64_t i = 0; i < B; ++i) {
                if (!h->ev_chain[2 * i]) {
                    HIP_TRY(hipEventCreate(&h->ev_chain[2 * i]));
                    HIP_TRY(hipEventCreate(&h->ev_chain[2 * i + 1]));
                }
                KmcArgs a = base_args(h, 3 + (int)i);
                const int ci = h->cur ^ (int)(i & 1);
                a.fin = h->frontier[ci];
                a.fout = h->frontier[ci ^ 1];
                a.prev = i ? h->ctl + 3 + (i - 1) : nullptr;   // the first level of a batch always runs, on host-known sizes
                a.stop_mask = h->cfg.continue_on_violation ? 0u : h->cfg.invariant_mask;
                a.stop_deadlock = (h->cfg.check_deadlock && r.verdict == KMC_V_OK) ? 1u : 0u;
                HIP_TRY(hipEventRecord(h->ev_chain[2 * i], h->stream));
                // sizes behind the first level are only known on the device: the grid is sized for the most a level can
                // grow (every state enabling every action instance), which saturates at a resident grid within two or
                // three levels but keeps the chains of tiny levels (IdSequence: 1002 one-state levels) to one block
                if ((rc = launch_expand(h, KMC_MODE_LOCAL, a, expand_grid(h, bound)))) return rc;
                bound = bound > h->fcap / fan ? h->fcap : bound * fan;
                HIP_TRY(hipEventRecord(h->ev_chain[2 * i + 1], h->stream));
            }
            HIP_TRY(hipMemcpyAsync(h->ctl_host, h->ctl + 3, B * sizeof(KmcLevelCtl), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            // every launch of the batch is accounted (also the ones behind the end of the search, which find nothing to
            // do and return in microseconds): the per-launch average then is what rocprofv3 --kernel-trace reports
            for (uint64_t i = 0; i < B; ++i) {
                float ms = 0;
                HIP_TRY(hipEventElapsedTime(&ms, h->ev_chain[2 * i], h->ev_chain[2 * i + 1]));
                r.seconds_expand += 1e-3 * ms;
                r.expand_launches++;
            }
            bool done = false;
            for (uint64_t i = 0; i < B && !done; ++i) {
                const KmcLevelCtl c = h->ctl_host[i];
                if (c.halt) break;   // the device ended the chain here; the host decides below whether the search goes on
                for (int k = 0; k < 8; ++k) h->prof[k] += c.prof[k];
                uint64_t new_seg[KMC_SEGS];
                const uint64_t produced = produced_segments(h, c, new_seg);
                stop = absorb(h, c, h->frontier[h->cur], h->seg_n, &rc);
                if (rc) return rc;
                if (stop) {
                    r.queue_left = queue_now(h);
                    done = true;
                    break;
                }
                if (produced == 0) {
                    h->n_cur = 0;
                    done = true;
                    break;
                }
                h->cur ^= 1;
                h->n_cur = produced;
                for (int sg = 0; sg < KMC_SEGS; ++sg) h->seg_n[sg] = new_seg[sg];
                h->level++;
                r.depth = h->level;
                book_level(h, produced, c);
                if ((double)r.orbit_representatives > 0.92 * (double)h->table_cap && r.verdict == KMC_V_OK) {
                    r.verdict = KMC_V_TABLE_FULL;
                    r.queue_left = queue_now(h);
                    stop = done = true;
                }
            }
            if (done) break;
            continue;
        }
        const int slot = (int)(h->level & 1);
        const int nxt = h->cur ^ 1;
        if ((rc = zero_ctl(h, slot))) return rc;
        KmcArgs a = base_args(h, slot);
        a.fin = h->frontier[h->cur];
        a.fout = h->frontier[nxt];
        if (shadow) {  // tuning aid: the identical level first runs on a copy of the table, with KMC_XFLAGS applied
            if (!h->table2) HIP_TRY(hipMalloc(&h->table2, h->table_cap * h->slot_words * 8));
            HIP_TRY(hipMemcpyAsync(h->table2, h->table, h->table_cap * h->slot_words * 8, hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(hipMemsetAsync(h->ctl + 2, 0, sizeof(KmcLevelCtl), h->stream));
            KmcArgs x = a;
            x.table = h->table2;
            x.ctl = h->ctl + 2;
            x.flags |= getenv("KMC_XFLAGS") ? (uint32_t)atoi(getenv("KMC_XFLAGS")) : 0u;
            HIP_TRY(hipEventRecord(h->ev0, h->stream));
            if ((rc = launch_expand(h, KMC_MODE_LOCAL, x, expand_grid(h, h->n_cur)))) return rc;
            HIP_TRY(hipEventRecord(h->ev1, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            float xms = 0;
            HIP_TRY(hipEventElapsedTime(&xms, h->ev0, h->ev1));
            h->dry_seconds += 1e-3 * xms;
        }
        HIP_TRY(hipEventRecord(h->ev0, h->stream));
        if ((rc = launch_expand(h, KMC_MODE_LOCAL, a, expand_grid(h, h->n_cur)))) return rc;
        HIP_TRY(hipEventRecord(h->ev1, h->stream));
        if ((rc = read_ctl(h, slot))) return rc;
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
        r.seconds_expand += 1e-3 * ms;
        r.expand_launches++;
        const KmcLevelCtl c = *h->ctl_host;
        for (int k = 0; k < 8; ++k) h->prof[k] += c.prof[k];
        if (h->f_expand_verify) {  // KMC_VERIFY: the second build regenerates this level; the counts must agree
            KmcArgs v = a;
            v.ctl = h->ctl + 2;
            if ((rc = zero_ctl(h, 2))) return rc;
            if ((rc = launch_expand(h, KMC_MODE_DRY, v, expand_grid(h, h->n_cur), nullptr, true))) return rc;
            KmcLevelCtl vc;
            HIP_TRY(hipMemcpyAsync(&vc, h->ctl + 2, KMC_CTL_LOCAL_BYTES, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            bool same = vc.deadlock_count == c.deadlock_count;
            for (int k = 0; k < KMC_MAX_KINDS; ++k) same = same && vc.generated[k] == c.generated[k];
            for (int k = 0; k < 4; ++k) same = same && vc.viol_count[k] == c.viol_count[k];
            // (orbit counting: the deficits taken when a state is expanded — the second build finds the stabilisers again)
            for (int k = 0; k < KMC_MAX_KINDS; ++k) same = same && vc.corr_gen[k] == c.corr_gen[k];
            for (int k = 0; k < 4; ++k) same = same && vc.corr_viol[k] == c.corr_viol[k];
            same = same && vc.corr_dead == c.corr_dead && vc.corr_repeats == c.corr_repeats;
            // ... and the successors themselves: how many reached the sink, and the order-independent checksum of their
            // fingerprints (taken where a successor enters the sink — behind the ring and the flush, where round 1's
            // miscompiled kernel lost some while every count above still agreed)
            const bool same_succ = vc.probed == c.probed && vc.fp_sum == c.fp_sum && vc.fp_xor == c.fp_xor &&
                                   vc.repeats == c.repeats && vc.outside == c.outside;
            if (!same || !same_succ) {
                r.verdict = KMC_V_ERROR;
                return fail(KMC_E_DEVICE, "KMC_VERIFY: the two builds of kmc_expand_%s disagree at level %llu (%s): one of them "
                                          "is miscompiled", h->kname.c_str(), (unsigned long long)h->level,
                            same ? "the successors reaching the seen-set differ: count or fingerprint checksum"
                                 : "generated / deadlock / violation counts differ");
            }
            h->verify_levels++;
        }
        const int dry = dry_mode;
        if (dry) {  // tuning aid: time the same level again without table writes / frontier traffic
            KmcArgs d = a;  // 1: no table access at all, 2: + read-only probes, 3: + invariants on every successor
            if (dry >= 2) d.flags |= KMC_FLAG_DRY_PROBE;
            if (dry == 3) d.flags |= KMC_FLAG_DRY_INV;
            if (dry == 4) d.flags |= KMC_FLAG_DRY_ATOM;
            if (dry == 5) d.flags |= KMC_FLAG_DRY_RAND;
            d.ctl = h->ctl + 2;
            HIP_TRY(hipEventRecord(h->ev0, h->stream));
            if ((rc = launch_expand(h, KMC_MODE_DRY, d, expand_grid(h, h->n_cur)))) return rc;
            HIP_TRY(hipEventRecord(h->ev1, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            float dms = 0;
            HIP_TRY(hipEventElapsedTime(&dms, h->ev0, h->ev1));
            h->dry_seconds += 1e-3 * dms;
            KmcLevelCtl dc;
            HIP_TRY(hipMemcpy(&dc, h->ctl + 2, sizeof dc, hipMemcpyDeviceToHost));
            for (int k = 0; k < 8; ++k) h->prof_dry[k] += dc.prof[k];
            HIP_TRY(hipMemsetAsync(h->ctl + 2, 0, sizeof(KmcLevelCtl), h->stream));
        }
        uint64_t new_seg[KMC_SEGS];
        const uint64_t produced = produced_segments(h, c, new_seg);
        stop = absorb(h, c, h->frontier[h->cur], h->seg_n, &rc);
        if (rc) return rc;
        if (stop) {  // invariant (produced level rolled back), deadlock, table/frontier full
            r.queue_left = queue_now(h);
            break;
        }
        if (produced == 0) {
            h->n_cur = 0;
            break;
        }
        h->cur = nxt;
        h->n_cur = produced;
        for (int sg = 0; sg < KMC_SEGS; ++sg) h->seg_n[sg] = new_seg[sg];
        h->level++;
        r.depth = h->level;
        book_level(h, produced, c);
        report();
        // stop before linear probing degenerates (sized for load <= 0.5, still fine at 0.9)
        if ((double)r.orbit_representatives > 0.92 * (double)h->table_cap && r.verdict == KMC_V_OK) {
            r.verdict = KMC_V_TABLE_FULL;
            r.queue_left = queue_now(h);
            break;
        }
    }
    r.n_levels = h->levels.size();
    r.seconds_total = now_s() - h->t_start;
    if (h->prof[7]) {
        const double tot = (double)h->prof[7];
        fprintf(stderr, "[kmc] per-wave ticks: load+extract+inv %.1f%%  guards %.1f%%  effects+push(incl flush) %.1f%%  "
                        "of which flush %.1f%%  tail %.1f%%  (total %.3g ticks)\n",
                100 * h->prof[0] / tot, 100 * h->prof[1] / tot, 100 * h->prof[2] / tot, 100 * h->prof[3] / tot,
                100 * h->prof[4] / tot, tot);
        if (h->prof[6])
            fprintf(stderr, "[kmc] effect leaves dispatched per 64-state tile: %.1f (%llu tiles)\n",
                    (double)h->prof[5] / (double)h->prof[6], (unsigned long long)h->prof[6]);
        for (int k = 0; k < 8; ++k) h->prof[k] = 0;
    }
    if (h->prof_dry[7]) {
        const double tot = (double)h->prof_dry[7];
        fprintf(stderr, "[kmc] DRY per-wave ticks: load+extract+inv %.1f%%  guards %.1f%%  effects+push(incl flush) %.1f%%  "
                        "of which flush %.1f%%  tail %.1f%%  (total %.3g ticks)\n",
                100 * h->prof_dry[0] / tot, 100 * h->prof_dry[1] / tot, 100 * h->prof_dry[2] / tot,
                100 * h->prof_dry[3] / tot, 100 * h->prof_dry[4] / tot, tot);
        for (int k = 0; k < 8; ++k) h->prof_dry[k] = 0;
    }
    if (h->dry_seconds > 0) {
        fprintf(stderr, "[kmc] dry/shadow expand: %.3f ms vs real %.3f ms\n",
                1e3 * h->dry_seconds, 1e3 * r.seconds_expand);
        h->dry_seconds = 0;
    }
    return KMC_OK;
}

int kmc_timing_get(kmc_handle* h, kmc_timing* out) {
    if (!h || !out) return fail(KMC_E_ARG, "null argument");
    *out = h->timing;
    return KMC_OK;
}

int kmc_result_get(kmc_handle* h, kmc_result* out) {
    if (!h || !out) return fail(KMC_E_ARG, "null argument");
    h->res.n_levels = h->levels.size();
    *out = h->res;
    return KMC_OK;
}

uint64_t kmc_level_sizes(kmc_handle* h, uint64_t* out, uint64_t cap) {
    if (!h) return 0;
    for (uint64_t i = 0; i < h->levels.size() && i < cap; ++i) out[i] = h->levels[i];
    return h->levels.size();
}

int kmc_frontier_states(kmc_handle* h, uint64_t* words, uint64_t cap_states, uint64_t* n_out) {
    if (!h || !n_out) return fail(KMC_E_ARG, "null argument");
    if (!h->table) return fail(KMC_E_STATE, "host-only handle");
    HIP_TRY(hipSetDevice(h->cfg.device));
    const uint64_t n = h->n_cur < cap_states ? h->n_cur : cap_states;
    *n_out = h->n_cur;
    if (n == 0) return KMC_OK;
    std::vector<uint64_t> plane(h->seg_cap);
    uint64_t at = 0;
    for (int sg = 0; sg < KMC_SEGS && at < n; ++sg) {
        const uint64_t m = h->seg_n[sg] < n - at ? h->seg_n[sg] : n - at;
        for (int k = 0; k < h->W && m; ++k) {
            HIP_TRY(hipMemcpy(plane.data(), h->frontier[h->cur] + (uint64_t)k * h->fcap + (uint64_t)sg * h->seg_cap,
                              m * 8, hipMemcpyDeviceToHost));
            for (uint64_t i = 0; i < m; ++i) words[(at + i) * h->W + k] = plane[i];
        }
        at += m;
    }
    return KMC_OK;
}

int kmc_successors(kmc_handle* h, const uint64_t* words, uint64_t* out, uint64_t cap, uint64_t* n_out) {
    if (!h || !words || !n_out) return fail(KMC_E_ARG, "null argument");
    if (!h->table) return fail(KMC_E_STATE, "host-only handle");
    HIP_TRY(hipSetDevice(h->cfg.device));
    // the auxiliary frontier is the scratch buffer viewed as SoA with stride 1... planes must be
    // fin[k*stride + 0], so stride 1 puts the W words back to back
    HIP_TRY(hipMemcpyAsync(h->scratch, words, h->W * 8, hipMemcpyHostToDevice, h->stream));
    if (h->cfg.symmetry) {   // plane W of this one-state frontier: the stabiliser's order (no count is taken from an ENUM pass)
        static const uint64_t one = 1;
        HIP_TRY(hipMemcpyAsync(h->scratch + h->W, &one, 8, hipMemcpyHostToDevice, h->stream));
    }
    int rc = zero_ctl(h, 2);
    if (rc) return rc;
    KmcArgs a = base_args(h, 2);
    a.fin = h->scratch;
    a.fin_stride = 1;
    for (int sg = 0; sg < KMC_SEGS; ++sg) a.seg_count[sg] = sg == 0 ? 1 : 0;
    a.send = h->enum_out;
    a.send_cap = h->enum_cap;
    a.inv_mask = 0;
    if ((rc = launch_expand(h, KMC_MODE_ENUM, a, 1))) return rc;
    if ((rc = read_ctl(h, 2))) return rc;
    const uint64_t n = h->ctl_host->enum_count < h->enum_cap ? h->ctl_host->enum_count : h->enum_cap;
    // The kind word of a record also says how many FURTHER satisfying bindings of the same disjunct yield this very successor
    // (Kip279.tla:47-51, Kip320.tla:82-83: two disjuncts of one binding hold at once): the list handed out repeats such a
    // record, so that it is TLC's enumeration of Next on this state — one entry per generated successor, as `generated` counts.
    const uint64_t rw = (uint64_t)h->W + 2;
    std::vector<uint64_t> recs(n * rw);
    if (n) HIP_TRY(hipMemcpy(recs.data(), h->enum_out, n * rw * 8, hipMemcpyDeviceToHost));
    uint64_t total = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t reps = 1 + (recs[i * rw + h->W + 1] >> 8);
        recs[i * rw + h->W + 1] &= 0xFFull;
        for (uint64_t k = 0; k < reps; ++k, ++total)
            if (out && total < cap) memcpy(out + total * rw, &recs[i * rw], rw * 8);
    }
    *n_out = total;
    return KMC_OK;
}

// The invariants of `mask` each of n packed states violates, from the device's own predicate (M::violated_pre, the one
// k_expand applies to every state it expands): one single-state pass of k_expand per state in its dry mode (successors are
// generated and dropped, no table or frontier is touched), the per-invariant violation counters of the control block
// read back.  A differential-testing entry point (states as data), not a search.
int kmc_check_states(kmc_handle* h, const uint64_t* words, uint64_t n, uint32_t mask, uint32_t* violated) {
    if (!h || !words || !violated) return fail(KMC_E_ARG, "null argument");
    if (!h->table) return fail(KMC_E_STATE, "host-only handle");
    HIP_TRY(hipSetDevice(h->cfg.device));
    for (uint64_t i = 0; i < n; ++i) {
        HIP_TRY(hipMemcpyAsync(h->scratch, words + i * h->W, h->W * 8, hipMemcpyHostToDevice, h->stream));
        if (h->cfg.symmetry) {
            static const uint64_t one = 1;
            HIP_TRY(hipMemcpyAsync(h->scratch + h->W, &one, 8, hipMemcpyHostToDevice, h->stream));
        }
        int rc = zero_ctl(h, 2);
        if (rc) return rc;
        KmcArgs a = base_args(h, 2);
        a.fin = h->scratch;
        a.fin_stride = 1;
        for (int sg = 0; sg < KMC_SEGS; ++sg) a.seg_count[sg] = sg == 0 ? 1 : 0;
        a.inv_mask = mask & 15u;
        if ((rc = launch_inv(h, a, 1))) return rc;
        if ((rc = read_ctl(h, 2))) return rc;
        uint32_t bits = 0;
        for (int k = 0; k < 4; ++k)
            if (h->ctl_host->viol_count[k]) bits |= 1u << k;
        violated[i] = bits;
    }
    return KMC_OK;
}

int kmc_witness(kmc_handle* h, uint64_t* words) {
    if (!h || !words) return fail(KMC_E_ARG, "null argument");
    if (!h->have_witness) return fail(KMC_E_STATE, "no witness recorded");
    for (int k = 0; k < h->W; ++k) words[k] = h->witness[k];
    return KMC_OK;
}

// Looks fp up in the device table from the host (a few 8-byte reads); returns the slot.
static int table_lookup(kmc_handle* h, uint64_t fp, uint64_t* slot) {
    const uint64_t mask = h->table_cap - 1;
    uint64_t i = fp & mask;
    for (uint64_t probes = 0; probes <= mask; ++probes) {
        uint64_t v = 0;
        HIP_TRY(hipMemcpy(&v, h->table + i * h->slot_words, 8, hipMemcpyDeviceToHost));
        // (with wide slots two distinct states may carry this fingerprint; the first one is reported — the check word
        // needs the state, which the callers of this lookup do not have)
        if (v == fp) {
            *slot = i;
            return KMC_OK;
        }
        if (v == 0) break;
        i = (i + 1) & mask;
    }
    return fail(KMC_E_STATE, "fingerprint %016llx not in table", (unsigned long long)fp);
}

// FPSet.contains analogue: is this packed state's fingerprint in the seen-set of the last run?
int kmc_contains(kmc_handle* h, const uint64_t* words, int32_t* present) {
    if (!h || !words || !present) return fail(KMC_E_ARG, "null argument");
    if (!h->table) return fail(KMC_E_STATE, "host-only handle");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    uint64_t slot = 0;
    uint64_t rep_words[KMC_MAXW];
    if (h->cfg.symmetry) {   // the table holds one state per orbit: ask for this state's representative
        kmc_canonical_state(h, words, rep_words, nullptr);
        words = rep_words;
    }
    const int rc = table_lookup(h, kmc_fingerprint_of(h, words), &slot);
    *present = rc == KMC_OK;
    g_err.clear();
    return KMC_OK;
}

int kmc_pred_of(kmc_handle* h, uint64_t fp, uint64_t* pred, int32_t* found) {
    if (!h || !pred || !found) return fail(KMC_E_ARG, "null argument");
    if (!h->pred) return fail(KMC_E_STATE, "kmc_pred_of needs keep_trace=1");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    uint64_t slot = 0;
    *found = table_lookup(h, fp, &slot) == KMC_OK;
    g_err.clear();
    *pred = 0;
    if (*found) HIP_TRY(hipMemcpy(pred, h->pred + slot, 8, hipMemcpyDeviceToHost));
    return KMC_OK;
}

int32_t kmc_owner_of(uint64_t fp, int32_t n_shards) {
    if (n_shards < 1 || n_shards > KMC_MAX_SHARDS) return -1;
    return (int32_t)kmc_owner(fp, (uint32_t)n_shards);
}

int kmc_init_state(kmc_handle* h, uint64_t* words) {
    if (!h || !words) return fail(KMC_E_ARG, "null argument");
    if (h->init_words.empty()) return fail(KMC_E_STATE, "no run has started on this handle");
    for (int k = 0; k < h->W; ++k) words[k] = h->init_words[k];
    return KMC_OK;
}

int kmc_trace(kmc_handle* h, uint8_t* canon_states, int32_t* kinds, uint64_t cap, uint64_t* n_out) {
    if (!h || !n_out) return fail(KMC_E_ARG, "null argument");
    if (!h->pred) return fail(KMC_E_STATE, "kmc_trace needs keep_trace=1");
    if (!h->have_witness) return fail(KMC_E_STATE, "no violation witness recorded");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->xstream) HIP_TRY(hipStreamSynchronize(h->xstream));   // (a pipelined level's inserts write the table on that stream)
    // 1. walk predecessor fingerprints back to the initial state (pred == 0)
    std::vector<uint64_t> chain;
    uint64_t fp = h->res.violation_fp;
    if (h->witness_outside) {  // not in the table: the chain starts at the parent it was generated from
        chain.push_back(fp);
        fp = h->witness_parent_fp;
    }
    for (uint64_t guard = 0; guard < (1u << 20); ++guard) {
        chain.push_back(fp);
        uint64_t slot = 0;
        int rc = table_lookup(h, fp, &slot);
        if (rc) return rc;
        uint64_t p = 0;
        HIP_TRY(hipMemcpy(&p, h->pred + slot, 8, hipMemcpyDeviceToHost));
        if (p == 0) break;
        fp = p;
    }
    // 2. replay forward from Init, picking at each step the successor with the next fingerprint
    const uint64_t n = chain.size();
    *n_out = n;
    const uint64_t cb = kmc_canon_bytes(h);
    std::vector<uint64_t> cur = h->init_words;
    std::vector<uint64_t> succ(h->enum_cap * (h->W + 2));
    if (kmc_fingerprint_of(h, cur.data()) != chain[n - 1]) return fail(KMC_E_STATE, "trace does not start at Init");
    for (uint64_t step = 0; step < n; ++step) {
        if (step < cap) {
            if (canon_states) kmc_unpack_state(h, cur.data(), canon_states + step * cb);
        }
        if (step + 1 == n) break;
        const uint64_t want = chain[n - 2 - step];
        uint64_t ns = 0;
        int rc = kmc_successors(h, cur.data(), succ.data(), h->enum_cap, &ns);
        if (rc) return rc;
        bool found = false;
        for (uint64_t i = 0; i < ns && i < h->enum_cap; ++i) {
            const uint64_t* rec = &succ[i * (h->W + 2)];
            if (rec[h->W] == want) {
                cur.assign(rec, rec + h->W);
                if (step + 1 < cap && kinds) kinds[step + 1] = (int32_t)rec[h->W + 1];
                found = true;
                break;
            }
        }
        if (!found) return fail(KMC_E_STATE, "trace replay lost the path at step %llu", (unsigned long long)step);
    }
    if (kinds && cap) kinds[0] = -1;
    return KMC_OK;
}

// ---- checkpoint / recover (TLC -checkpoint / -recover [TLC-recall]) --------------------------
// File: header, kmc_result, level sizes, segment sizes, then the fingerprint table (and the
// predecessor table when traces are kept) and the current frontier's planes, segment by segment.
namespace {
struct CkptHeader {
    char magic[8];          // "KMCCKPT4"
    kmc_config cfg;         // pointers inside are not meaningful in the file
    uint64_t table_cap, fcap, seg_cap, level, n_cur, n_levels, w, has_pred;
    uint64_t layout_form;   // KmcLayout::rm of the packed states in the file (0 tight, 1 / 2 replica-major): the same constants
                            // can be packed in more than one way (KMC_LAYOUT), often into the same number of words
};
bool wr(FILE* f, const void* p, size_t n) { return fwrite(p, 1, n, f) == n; }
bool rd(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n; }
// device <-> file through a bounded pinned staging buffer
int dev_to_file(FILE* f, const u64* dev, uint64_t words) {
    const uint64_t chunk = 1ull << 24;  // 128 MiB
    std::vector<uint64_t> buf(words < chunk ? words : chunk);
    for (uint64_t at = 0; at < words; at += chunk) {
        const uint64_t n = words - at < chunk ? words - at : chunk;
        HIP_TRY(hipMemcpy(buf.data(), dev + at, n * 8, hipMemcpyDeviceToHost));
        if (!wr(f, buf.data(), n * 8)) return fail(KMC_E_STATE, "checkpoint: short write");
    }
    return KMC_OK;
}
int file_to_dev(FILE* f, u64* dev, uint64_t words) {
    const uint64_t chunk = 1ull << 24;
    std::vector<uint64_t> buf(words < chunk ? words : chunk);
    for (uint64_t at = 0; at < words; at += chunk) {
        const uint64_t n = words - at < chunk ? words - at : chunk;
        if (!rd(f, buf.data(), n * 8)) return fail(KMC_E_STATE, "checkpoint: short read");
        HIP_TRY(hipMemcpy(dev + at, buf.data(), n * 8, hipMemcpyHostToDevice));
    }
    return KMC_OK;
}
}  // namespace

int kmc_checkpoint_save(kmc_handle* h, const char* path) {
    if (!h || !path) return fail(KMC_E_ARG, "null argument");
    if (!h->table) return fail(KMC_E_STATE, "checkpoints are for device handles");
    // a shard of a multi-GPU search (level-step interface) saves its own table / frontier between kmc_step_finish and
    // the next kmc_step_expand; the driver keeps the global counters (sharded.py) and sets the verdict first
    if (h->cfg.n_shards != 1 && (!h->stepping || h->step_expanded))
        return fail(KMC_E_STATE, "a shard is checkpointed between kmc_step_finish and the next kmc_step_expand");
    if (h->xstream) HIP_TRY(hipStreamSynchronize(h->xstream));
    if (h->levels.empty()) return fail(KMC_E_STATE, "nothing to checkpoint: run first");
    // Only a level boundary is a consistent state: after a stop inside a level (invariant, deadlock, table / frontier
    // full) the table already holds the fingerprints of the rolled-back or partial level while the frontier is still
    // its parent — a search resumed from that would find every successor "seen" and end with states missing.
    if (h->res.verdict != KMC_V_LEVEL_LIMIT)
        return fail(KMC_E_STATE, "a checkpoint can only be taken at a level boundary: after a run that stopped at max_levels "
                                 "(verdict level_limit); this run ended with verdict %d", h->res.verdict);
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    FILE* f = fopen(path, "wb");
    if (!f) return fail(KMC_E_ARG, "cannot open %s for writing", path);
    CkptHeader hd{};
    memcpy(hd.magic, "KMCCKPT4", 8);
    hd.cfg = h->cfg;
    hd.cfg.cache_dir = nullptr;
    hd.table_cap = h->table_cap; hd.fcap = h->fcap; hd.seg_cap = h->seg_cap; hd.level = h->level;
    hd.n_cur = h->n_cur; hd.n_levels = h->levels.size(); hd.w = h->W; hd.has_pred = h->pred != nullptr;
    hd.layout_form = (uint64_t)h->lay.rm;
    int rc = KMC_OK;
    bool ok = wr(f, &hd, sizeof hd) && wr(f, &h->res, sizeof h->res) && wr(f, h->levels.data(), h->levels.size() * 8) &&
              wr(f, h->seg_n, sizeof h->seg_n) && wr(f, h->init_words.data(), h->W * 8);
    if (!ok) rc = fail(KMC_E_STATE, "checkpoint: short write");
    if (!rc) rc = dev_to_file(f, h->table, h->table_cap * h->slot_words);
    if (!rc && h->pred) rc = dev_to_file(f, h->pred, h->table_cap);
    for (int sg = 0; sg < KMC_SEGS && !rc; ++sg)
        for (int k = 0; k < h->planes && !rc; ++k)
            if (h->seg_n[sg])
                rc = dev_to_file(f, h->frontier[h->cur] + (uint64_t)k * h->fcap + (uint64_t)sg * h->seg_cap, h->seg_n[sg]);
    fclose(f);
    return rc;
}

int kmc_checkpoint_load(kmc_handle* h, const char* path) {
    if (!h || !path) return fail(KMC_E_ARG, "null argument");
    if (!h->table) return fail(KMC_E_STATE, "checkpoints are for device handles");
    HIP_TRY(hipSetDevice(h->cfg.device));
    FILE* f = fopen(path, "rb");
    if (!f) return fail(KMC_E_ARG, "cannot open %s", path);
    CkptHeader hd{};
    int rc = KMC_OK;
    if (!rd(f, &hd, sizeof hd) || memcmp(hd.magic, "KMCCKPT4", 8) != 0)
        rc = fail(KMC_E_ARG, "%s is not a checkpoint of this version", path);
    const kmc_config& a = hd.cfg;
    const kmc_config& b = h->cfg;
    if (!rc && (a.model != b.model || a.n_replicas != b.n_replicas || a.log_size != b.log_size ||
                a.max_records != b.max_records || a.max_leader_epoch != b.max_leader_epoch ||
                a.n_log_records != b.n_log_records || a.max_id != b.max_id || a.hash_seed != b.hash_seed ||
                a.n_shards != b.n_shards || a.shard_id != b.shard_id || hd.w != (uint64_t)h->W ||
                hd.layout_form != (uint64_t)h->lay.rm || (a.wide_fingerprint != 0) != (b.wide_fingerprint != 0) ||
                (a.symmetry != 0) != (b.symmetry != 0)))
        rc = fail(KMC_E_ARG, "checkpoint was taken for a different model / constants / hash seed / shard / fingerprint width / "
                             "state layout / symmetry setting");
    if (!rc && (hd.table_cap != h->table_cap || hd.fcap != h->fcap || hd.seg_cap != h->seg_cap ||
                hd.has_pred != (uint64_t)(h->pred != nullptr)))
        rc = fail(KMC_E_ARG, "checkpoint capacities differ: open the handle with table_capacity=%llu frontier_capacity=%llu keep_trace=%d",
                  (unsigned long long)hd.table_cap, (unsigned long long)hd.fcap, (int)hd.has_pred);
    // the file is not trusted: every size is checked against the handle before it sizes a buffer or a device copy
    if (!rc && (hd.n_levels == 0 || hd.n_levels > 4096 || hd.level != hd.n_levels || hd.n_cur > hd.fcap))
        rc = fail(KMC_E_ARG, "checkpoint header is inconsistent (levels %llu, level %llu, frontier %llu of %llu)",
                  (unsigned long long)hd.n_levels, (unsigned long long)hd.level, (unsigned long long)hd.n_cur,
                  (unsigned long long)hd.fcap);
    // (as do_begin: a stepped search that stopped between kmc_step_expand and kmc_step_finish may still have a pipelined
    // level's transfer and insert in flight on the second stream — they must not land in the restored table — and its records
    // are still booked for a conservation check that belongs to the abandoned level)
    if (!rc && h->xstream && hipStreamSynchronize(h->xstream) != hipSuccess) rc = fail(KMC_E_DEVICE, "stream sync failed");
    if (!rc) h->inserted_level = 0;
    if (!rc) rc = reset_run(h);
    if (!rc && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(KMC_E_DEVICE, "stream sync failed");
    if (!rc) {
        kmc_result saved{};
        std::vector<uint64_t> lv(hd.n_levels);
        uint64_t segs[KMC_SEGS];
        std::vector<uint64_t> init(h->W);
        bool ok = rd(f, &saved, sizeof saved) && rd(f, lv.data(), hd.n_levels * 8) && rd(f, segs, sizeof segs) &&
                  rd(f, init.data(), h->W * 8);
        if (!ok) rc = fail(KMC_E_STATE, "checkpoint: short read");
        uint64_t seg_sum = 0, lv_sum = 0;
        for (int sg = 0; sg < KMC_SEGS && !rc; ++sg) {
            if (segs[sg] > h->seg_cap) rc = fail(KMC_E_ARG, "checkpoint: segment %d holds %llu states, capacity %llu", sg,
                                                 (unsigned long long)segs[sg], (unsigned long long)h->seg_cap);
            seg_sum += segs[sg];
        }
        for (uint64_t x : lv) lv_sum += x;
        // (under symmetry the level sizes and `distinct` are the weighted numbers; the stored states are orbit_representatives)
        if (!rc && (seg_sum != hd.n_cur || (!h->cfg.symmetry && lv.back() != hd.n_cur) || lv_sum != saved.distinct ||
                    saved.orbit_representatives > h->table_cap || saved.orbit_representatives > saved.distinct ||
                    saved.verdict != KMC_V_LEVEL_LIMIT || saved.state_words != (uint64_t)h->W ||
                    saved.table_capacity != h->table_cap || saved.frontier_capacity != h->fcap))
            rc = fail(KMC_E_ARG, "checkpoint body is inconsistent with its header / this handle");
        if (!rc && kmc_fingerprint_of(h, init.data()) == 0) rc = fail(KMC_E_ARG, "checkpoint: bad initial state");
        if (!rc) {
            h->res = saved;
            h->levels = lv;
            memcpy(h->seg_n, segs, sizeof segs);
            h->init_words = init;
        }
    }
    if (!rc) rc = file_to_dev(f, h->table, h->table_cap * h->slot_words);
    if (!rc && h->pred) rc = file_to_dev(f, h->pred, h->table_cap);
    h->cur = 0;
    for (int sg = 0; sg < KMC_SEGS && !rc; ++sg)
        for (int k = 0; k < h->planes && !rc; ++k)
            if (h->seg_n[sg])
                rc = file_to_dev(f, h->frontier[0] + (uint64_t)k * h->fcap + (uint64_t)sg * h->seg_cap, h->seg_n[sg]);
    fclose(f);
    if (rc) return rc;
    h->level = hd.level;
    h->n_cur = hd.n_cur;
    h->restored = true;
    return KMC_OK;
}

// ---- level-step interface ---------------------------------------------------------------
int kmc_step_begin(kmc_handle* h) {
    if (!h) return fail(KMC_E_ARG, "null handle");
    if (!h->table) return fail(KMC_E_STATE, "host-only handle (device = -1) cannot run");
    HIP_TRY(hipSetDevice(h->cfg.device));
    int rc = do_begin(h);
    h->stepping = true;
    h->step_expanded = false;
    return rc;
}

int kmc_step_expand(kmc_handle* h, uint64_t* send_counts /* [KMC_MAX_SHARDS][KMC_SEND_SUBS] */) {
    if (!h || !h->stepping) return fail(KMC_E_STATE, "kmc_step_begin first");
    // (one shard buckets nothing: every successor is its own, so it may run without a send area)
    if (!h->send && h->cfg.n_shards > 1)
        return fail(KMC_E_STATE, "no send area: open with n_shards > 1 or call kmc_step_set_send_buffer");
    HIP_TRY(hipSetDevice(h->cfg.device));
    const int slot = (int)(h->level & 1);
    int rc = zero_ctl(h, slot);
    if (rc) return rc;
    KmcArgs a = base_args(h, slot);
    a.fin = h->frontier[h->cur];
    a.fout = h->frontier[h->cur ^ 1];
    a.send = h->send;
    a.send_cap = h->send_cap;
    if ((rc = ensure_mode(h, KMC_MODE_SHARDED))) return rc;   // (a cold cache compiles here, outside the timed events)
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    if (h->n_cur) {
        if ((rc = launch_expand(h, KMC_MODE_SHARDED, a, expand_grid(h, h->n_cur)))) return rc;
    }
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    if ((rc = read_ctl(h, slot))) return rc;
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->res.seconds_expand += 1e-3 * ms;
    h->res.expand_launches++;
    for (int d = 0; d < KMC_MAX_SHARDS; ++d)
        for (int sb = 0; sb < KMC_SEGS; ++sb) {
            uint64_t c = h->ctl_host->send_count[d][sb].v;
            h->last_send_counts[d * KMC_SEGS + sb] = c < h->send_cap ? c : h->send_cap;
            if (send_counts) send_counts[d * KMC_SEGS + sb] = h->last_send_counts[d * KMC_SEGS + sb];
        }
    h->xcounts_valid = false;
    h->step_expanded = true;
    return KMC_OK;
}

int kmc_step_set_send_buffer(kmc_handle* h, void* dev_ptr, uint64_t records_per_sub_buffer) {
    const uint64_t records_per_destination = records_per_sub_buffer;
    if (!h || !dev_ptr || records_per_destination == 0) return fail(KMC_E_ARG, "bad send buffer");
    if (h->send && h->send_owned) hipFree(h->send);  // n_shards == 1 is allowed: one destination, itself
    h->send = (u64*)dev_ptr;
    h->send_cap = records_per_destination;
    h->send_owned = false;
    return KMC_OK;
}

int kmc_step_send_buffer(kmc_handle* h, int32_t dst, int32_t sub, void** dev_ptr, uint64_t* record_words) {
    if (!h || !h->send || dst < 0 || dst >= h->cfg.n_shards || sub < 0 || sub >= KMC_SEGS)
        return fail(KMC_E_ARG, "bad destination / sub-buffer");
    *dev_ptr = h->send + ((uint64_t)dst * KMC_SEGS + sub) * h->send_cap * h->rec_words;
    *record_words = h->rec_words;
    return KMC_OK;
}

int kmc_step_insert(kmc_handle* h, const void* dev_records, uint64_t n_records) {
    if (!h || !h->stepping || !h->step_expanded) return fail(KMC_E_STATE, "kmc_step_expand first");
    if (n_records == 0) return KMC_OK;
    HIP_TRY(hipSetDevice(h->cfg.device));
    const int slot = (int)(h->level & 1);
    KmcArgs a = base_args(h, slot);
    a.recv = (const u64*)dev_records;
    a.n_in = n_records;
    h->inserted_level += n_records;
    a.fout = h->frontier[h->cur ^ 1];
    uint64_t blocks = (n_records + KMC_BLOCK - 1) / KMC_BLOCK;
    const uint64_t maxb = (uint64_t)h->n_cus * 8;
    if (blocks > maxb) blocks = maxb;
    return launch(h, h->f_insert, a, (unsigned)blocks);
}

int kmc_step_finish(kmc_handle* h, kmc_level_info* info) {
    if (!h || !h->stepping || !h->step_expanded) return fail(KMC_E_STATE, "kmc_step_expand first");
    HIP_TRY(hipSetDevice(h->cfg.device));
    const int slot = (int)(h->level & 1);
    if (h->xstream) HIP_TRY(hipStreamSynchronize(h->xstream));   // a pipelined level's last transfer and insert
    int rc = read_ctl(h, slot);
    if (rc) return rc;
    const KmcLevelCtl c = *h->ctl_host;
    if (c.err & KMC_ERR_CHECK_WORD)
        return fail(KMC_E_DEVICE, "wide fingerprints: a claimed slot's check word did not appear (level %llu)", (unsigned long long)h->level);
    // the level's kernels on this shard: one k_expand (local successors probed at once, remote ones bucketed — both enter
    // the sink) and the k_insert launches over what the other shards sent
    if ((rc = check_conservation(h, c, h->inserted_level))) return rc;
    h->inserted_level = 0;
    const int nxt = h->cur ^ 1;
    uint64_t new_seg[KMC_SEGS];
    const uint64_t produced = produced_segments(h, c, new_seg);
    kmc_result& r = h->res;
    // kmc_config.symmetry: every count of this shard is weighed as book_level / absorb weigh kmc_run's — N! x the stored
    // states' count less the summed deficits of their orbits (KmcLevelCtl::corr_*).  A state is weighed where it is CLAIMED
    // (its owner: corr_won of k_expand's local path or of k_insert), an expansion where it is EXPANDED (this shard), so the
    // sums over the shards are the plain search's numbers.
    uint64_t gen_w[KMC_MAX_KINDS];
    for (int k = 0; k < KMC_MAX_KINDS; ++k) {
        gen_w[k] = weighted(h, c.generated[k], c.corr_gen[k]);
        r.generated += gen_w[k];
        r.action_generated[k] += gen_w[k];
    }
    r.generated_repeats += weighted(h, c.repeats, c.corr_repeats);
    const uint64_t dead_w = weighted(h, c.deadlock_count, c.corr_dead);
    r.deadlock_states += dead_w;
    const uint64_t produced_w = weighted(h, produced, c.corr_won);
    h->cur = nxt;
    h->n_cur = produced;
    for (int sg = 0; sg < KMC_SEGS; ++sg) { h->prev_seg_n[sg] = h->seg_n[sg]; h->seg_n[sg] = new_seg[sg]; }
    h->level++;
    if (produced) r.depth = h->level;
    r.distinct += produced_w;
    r.orbit_representatives += produced;
    h->levels.push_back(produced_w);
    h->step_expanded = false;
    r.seconds_total = now_s() - h->t_start;
    if (info) {
        memset(info, 0, sizeof *info);
        info->depth = h->level;
        info->new_states = produced_w;   // (symmetry: the states of the level as the plain search counts them; 0 iff none stored)
        info->generated_total = r.generated;
        info->distinct_total = r.distinct;
        info->seconds = r.seconds_total;
        for (int k = 0; k < KMC_MAX_KINDS; ++k) info->generated_level[k] = gen_w[k];
        for (int k = 0; k < 4; ++k) {
            info->violation_count[k] = weighted(h, c.viol_count[k], c.corr_viol[k]);
            info->violation_fp[k] = c.viol_count[k] ? ~c.viol_fp_inv[k] : 0;
        }
        for (int k = 0; k < 4; ++k) {
            info->outside_violation_count[k] = c.oviol_count[k];
            info->outside_violation_fp[k] = c.oviol_count[k] ? ~c.oviol_fp_inv[k] : 0;
        }
        info->deadlocks_level = dead_w;
        info->send_filtered = c.send_filtered;
        info->error_flags = c.err;
    }
    return rc;
}

// ---- the per-level exchange under the ABI ---------------------------------------
// After kmc_step_expand every shard holds, per (destination, sub-buffer), a dense run of records in its
// send area.  One level's exchange is
//   (1) an all-gather of one small row per rank: its KMC_SEGS send counts per destination and the caller's
//       statistics vector (the statistics of the PREVIOUS expansion ride along: one collective decides
//       termination and verdicts identically on every rank) — one stream synchronisation, because the host
//       must know the counts to post the receives;
//   (2) grouped ncclSend / ncclRecv of every non-empty (peer, sub-buffer) run, straight from the send area
//       into one contiguous receive area, on the engine's stream; and
//   (3) ONE k_insert over what arrived, queued behind the receives on the same stream — no host wait.
// The plan (who sends how many words from which offset, where each run lands) is a pure function of the
// count matrix, shared by the RCCL transport and by the in-process transport that moves the runs with
// device-to-device copies between P logical shards on one GPU (RCCL refuses two ranks on one device).
namespace {

struct KmcRccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

// librccl is bound at run time: libkmc.so must load on a box without RCCL (single-GPU use, the CPU-side
// ABI tests), and inside a PyTorch process the name resolves to the copy the wheel has already loaded
// (same SONAME), so both sides of the process talk to one RCCL.
std::string g_rccl_error;   // why librccl could not be bound (dlerror() is read ONCE, where it happens: a second call returns NULL)

void rccl_bind(KmcRccl& r) {
    const char* names[] = {getenv("KMC_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        if ((r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        const char* e = dlerror();
        g_rccl_error += std::string(g_rccl_error.empty() ? "" : "; ") + n + ": " + (e ? e : "dlopen failed");
    }
    if (!r.lib) return;
#define KMC_SYM(field, name)                                                     \
    r.field = (decltype(r.field))dlsym(r.lib, name);                             \
    if (!r.field) { const char* e = dlerror(); g_rccl_error = std::string(name) + ": " + (e ? e : "symbol not found"); r.lib = nullptr; return; }
    KMC_SYM(GetUniqueId, "ncclGetUniqueId")
    KMC_SYM(CommInitRank, "ncclCommInitRank")
    KMC_SYM(CommDestroy, "ncclCommDestroy")
    KMC_SYM(AllGather, "ncclAllGather")
    KMC_SYM(Send, "ncclSend")
    KMC_SYM(Recv, "ncclRecv")
    KMC_SYM(GroupStart, "ncclGroupStart")
    KMC_SYM(GroupEnd, "ncclGroupEnd")
    KMC_SYM(GetErrorString, "ncclGetErrorString")
#undef KMC_SYM
}

KmcRccl* rccl() {   // bound once per process, also when several host threads arrive at the same time
    static KmcRccl r;
    static std::once_flag once;
    std::call_once(once, rccl_bind, std::ref(r));
    return r.lib ? &r : nullptr;
}

#define NCCL_TRY(expr)                                                                                          \
    do {                                                                                                        \
        ncclResult_t e_ = (expr);                                                                               \
        if (e_ != ncclSuccess) return fail(KMC_E_DEVICE, "%s failed: %s", #expr, rccl()->GetErrorString(e_));  \
    } while (0)

// One message of a level's plan: `words` 64-bit words at `offset_words` of the send area (a send) or of the
// receive area (a receive), exchanged with `peer`.
struct KmcXfer {
    uint64_t peer, offset_words, words;
};
// A single message stays below 1 GiB: this RCCL build corrupted all-to-all messages above 2 GiB
// (tools/a2a_probe.py), so long runs are cut; both sides cut identically.
constexpr uint64_t KMC_XFER_MAX_WORDS = 1ull << 27;

// counts[(s * P + d) * KMC_SEGS + sub] = records shard s sends to shard d from its sub-buffer `sub`.
// Sends of `me` in (destination, sub-buffer) order; receives in (source, sub-buffer) order — RCCL matches the
// messages of a pair in posting order, and both lists enumerate a pair's runs in sub-buffer order.
void plan_level(const uint64_t* counts, int P, int me, uint64_t send_cap, uint64_t rec_words,
                std::vector<KmcXfer>* sends, std::vector<KmcXfer>* recvs, uint64_t* recv_records) {
    sends->clear();
    recvs->clear();
    auto cut = [](std::vector<KmcXfer>* out, uint64_t peer, uint64_t off, uint64_t words) {
        while (words) {
            const uint64_t n = words < KMC_XFER_MAX_WORDS ? words : KMC_XFER_MAX_WORDS;
            out->push_back(KmcXfer{peer, off, n});
            off += n;
            words -= n;
        }
    };
    for (int d = 0; d < P; ++d) {
        if (d == me) continue;
        for (int sb = 0; sb < KMC_SEGS; ++sb) {
            const uint64_t n = counts[((uint64_t)me * P + d) * KMC_SEGS + sb];
            if (n) cut(sends, (uint64_t)d, ((uint64_t)d * KMC_SEGS + sb) * send_cap * rec_words, n * rec_words);
        }
    }
    uint64_t at = 0;  // records received so far: the receive area is filled densely, source by source
    for (int s2 = 0; s2 < P; ++s2) {
        if (s2 == me) continue;
        for (int sb = 0; sb < KMC_SEGS; ++sb) {
            const uint64_t n = counts[((uint64_t)s2 * P + me) * KMC_SEGS + sb];
            if (n) cut(recvs, (uint64_t)s2, at * rec_words, n * rec_words);
            at += n;
        }
    }
    *recv_records = at;
}

int ensure_exchange_buffers(kmc_handle* h) {
    const int P = h->cfg.n_shards;
    if (!h->send || !h->send_owned)
        return fail(KMC_E_STATE, "the exchange under the ABI needs the engine-owned send area (n_shards > 1, no "
                                 "kmc_step_set_send_buffer)");
    if (!h->recv) {
        // worst case: every other shard fills all its sub-buffers for this one
        h->recv_cap = (uint64_t)(P - 1) * KMC_SEGS * h->send_cap;
        if (hipMalloc(&h->recv, h->recv_cap * h->rec_words * 8ull) != hipSuccess) {
            h->recv = nullptr;
            return fail(KMC_E_NOMEM, "cannot allocate the receive area (%llu records)", (unsigned long long)h->recv_cap);
        }
    }
    const size_t row = (size_t)P * KMC_SEGS + KMC_EXCHANGE_STATS;
    if (!h->xrow_dev) HIP_TRY(hipMalloc(&h->xrow_dev, (size_t)(P + 1) * row * 8));
    if (!h->xrow_host) HIP_TRY(hipHostMalloc(&h->xrow_host, (size_t)(P + 1) * row * 8));
    return KMC_OK;
}

int insert_received(kmc_handle* h, uint64_t n_records, hipStream_t stream = nullptr) {
    if (n_records == 0) return KMC_OK;
    const int slot = (int)(h->level & 1);
    KmcArgs a = base_args(h, slot);
    a.recv = h->recv;
    a.n_in = n_records;
    h->inserted_level += n_records;
    a.fout = h->frontier[h->cur ^ 1];
    uint64_t blocks = (n_records + KMC_BLOCK - 1) / KMC_BLOCK;
    const uint64_t maxb = (uint64_t)h->n_cus * 8;
    if (blocks > maxb) blocks = maxb;
    return launch(h, h->f_insert, a, (unsigned)blocks, stream);
}

}  // namespace

static void comm_release(kmc_handle* h) {
    if (h->comm && rccl()) rccl()->CommDestroy(h->comm);
    h->comm = nullptr;
}

int kmc_comm_unique_id(uint8_t* id) {
    if (!id) return fail(KMC_E_ARG, "null argument");
    KmcRccl* r = rccl();
    if (!r) return fail(KMC_E_DEVICE, "librccl could not be bound: %s", g_rccl_error.empty() ? "?" : g_rccl_error.c_str());
    ncclUniqueId u;
    NCCL_TRY(r->GetUniqueId(&u));
    static_assert(sizeof u == KMC_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id, &u, sizeof u);
    return KMC_OK;
}

int kmc_comm_init(kmc_handle* h, const uint8_t* id) {
    if (!h || !id) return fail(KMC_E_ARG, "null argument");
    if (!h->table || h->cfg.n_shards < 1) return fail(KMC_E_STATE, "kmc_comm_init needs a device handle");
    KmcRccl* r = rccl();
    if (!r) return fail(KMC_E_DEVICE, "librccl not found (dlopen librccl.so.1)");
    HIP_TRY(hipSetDevice(h->cfg.device));
    comm_release(h);
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    NCCL_TRY(r->CommInitRank(&h->comm, h->cfg.n_shards, u, h->cfg.shard_id));
    if (h->cfg.n_shards > 1) return ensure_exchange_buffers(h);
    return KMC_OK;
}

// Exercises every RCCL entry point the exchange uses on this handle's communicator and stream: an all-gather of
// one row per rank and a grouped send/receive ring (rank r sends a pattern to r+1 and receives from r-1; with one
// rank that is a send to itself).  Verifies what arrived.  A world_size-1 run thereby covers the binding, the
// argument conventions and the stream ordering although a one-shard search has no remote traffic.
int kmc_comm_selftest(kmc_handle* h) {
    if (!h || !h->comm) return fail(KMC_E_STATE, "kmc_comm_init first");
    KmcRccl* r = rccl();
    HIP_TRY(hipSetDevice(h->cfg.device));
    const int P = h->cfg.n_shards, me = h->cfg.shard_id;
    const size_t n = 4096;
    u64* buf = nullptr;
    HIP_TRY(hipMalloc(&buf, (size_t)(2 + P) * n * 8));
    struct Free { u64* p; ~Free() { hipFree(p); } } free_buf{buf};   // also on the error returns below
    std::vector<uint64_t> host((size_t)(2 + P) * n);
    for (size_t i = 0; i < n; ++i) host[i] = ((uint64_t)(me + 1) << 32) | i;
    HIP_TRY(hipMemcpyAsync(buf, host.data(), n * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemsetAsync(buf + n, 0, (size_t)(1 + P) * n * 8, h->stream));
    NCCL_TRY(r->GroupStart());
    NCCL_TRY(r->Send(buf, n, ncclUint64, (me + 1) % P, h->comm, h->stream));
    NCCL_TRY(r->Recv(buf + n, n, ncclUint64, (me + P - 1) % P, h->comm, h->stream));
    NCCL_TRY(r->GroupEnd());
    NCCL_TRY(r->AllGather(buf, buf + 2 * n, n, ncclUint64, h->comm, h->stream));
    HIP_TRY(hipMemcpyAsync(host.data(), buf, host.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const uint64_t from = (uint64_t)((me + P - 1) % P + 1);
    for (size_t i = 0; i < n; ++i) {
        if (host[n + i] != ((from << 32) | i)) return fail(KMC_E_DEVICE, "selftest: send/recv word %zu is wrong", i);
        for (int q = 0; q < P; ++q)
            if (host[(2 + q) * n + i] != (((uint64_t)(q + 1) << 32) | i))
                return fail(KMC_E_DEVICE, "selftest: all-gather word %zu of rank %d is wrong", i, q);
    }
    return KMC_OK;
}

int kmc_exchange_plan(const uint64_t* counts, int32_t n_shards, int32_t me, uint64_t send_cap, uint64_t rec_words,
                      uint64_t* sends, uint64_t* recvs, uint64_t cap, uint64_t* n_sends, uint64_t* n_recvs,
                      uint64_t* recv_records) {
    if (!counts || n_shards < 1 || n_shards > KMC_MAX_SHARDS || me < 0 || me >= n_shards || !n_sends || !n_recvs ||
        !recv_records)
        return fail(KMC_E_ARG, "bad argument");
    std::vector<KmcXfer> sv, rv;
    plan_level(counts, n_shards, me, send_cap, rec_words, &sv, &rv, recv_records);
    *n_sends = sv.size();
    *n_recvs = rv.size();
    for (uint64_t i = 0; i < sv.size() && i < cap && sends; ++i) {
        sends[3 * i] = sv[i].peer; sends[3 * i + 1] = sv[i].offset_words; sends[3 * i + 2] = sv[i].words;
    }
    for (uint64_t i = 0; i < rv.size() && i < cap && recvs; ++i) {
        recvs[3 * i] = rv[i].peer; recvs[3 * i + 1] = rv[i].offset_words; recvs[3 * i + 2] = rv[i].words;
    }
    return KMC_OK;
}

int kmc_step_exchange_counts(kmc_handle* h, const int64_t* stats, int32_t n_stats, int64_t* stats_sum,
                             uint64_t* recv_records) {
    if (!h || !h->stepping || !h->step_expanded) return fail(KMC_E_STATE, "kmc_step_expand first");
    if (!h->comm) return fail(KMC_E_STATE, "kmc_comm_init first");
    if (n_stats < 0 || n_stats > KMC_EXCHANGE_STATS || (n_stats && (!stats || !stats_sum)))
        return fail(KMC_E_ARG, "bad statistics vector");
    KmcRccl* r = rccl();
    HIP_TRY(hipSetDevice(h->cfg.device));
    const int P = h->cfg.n_shards, me = h->cfg.shard_id;
    const size_t row = (size_t)P * KMC_SEGS + KMC_EXCHANGE_STATS;
    int rc = P > 1 ? ensure_exchange_buffers(h) : KMC_OK;
    if (rc) return rc;
    h->xcounts.assign((size_t)P * P * KMC_SEGS, 0);
    if (P == 1) {  // nothing to gather
        for (int k = 0; k < n_stats; ++k) stats_sum[k] = stats[k];
        if (recv_records) *recv_records = 0;
        h->xcounts_valid = true;
        return KMC_OK;
    }
    int64_t* mine = h->xrow_host;
    for (int d = 0; d < P; ++d)
        for (int sb = 0; sb < KMC_SEGS; ++sb)
            mine[d * KMC_SEGS + sb] = d == me ? 0 : (int64_t)h->last_send_counts[d * KMC_SEGS + sb];
    for (int k = 0; k < KMC_EXCHANGE_STATS; ++k) mine[P * KMC_SEGS + k] = k < n_stats ? stats[k] : 0;
    HIP_TRY(hipMemcpyAsync(h->xrow_dev, mine, row * 8, hipMemcpyHostToDevice, h->stream));
    NCCL_TRY(r->AllGather(h->xrow_dev, h->xrow_dev + row, row, ncclInt64, h->comm, h->stream));
    HIP_TRY(hipMemcpyAsync(h->xrow_host + row, h->xrow_dev + row, (size_t)P * row * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int k = 0; k < n_stats; ++k) stats_sum[k] = 0;
    for (int s2 = 0; s2 < P; ++s2) {
        const int64_t* g = h->xrow_host + (size_t)(1 + s2) * row;
        for (int d = 0; d < P; ++d)
            for (int sb = 0; sb < KMC_SEGS; ++sb) {
                const int64_t c = g[d * KMC_SEGS + sb];
                if (c < 0 || (uint64_t)c > h->send_cap)
                    return fail(KMC_E_STATE, "exchange: rank %d announces %lld records for a sub-buffer of %llu", s2,
                                (long long)c, (unsigned long long)h->send_cap);
                h->xcounts[((size_t)s2 * P + d) * KMC_SEGS + sb] = (uint64_t)c;
            }
        for (int k = 0; k < n_stats; ++k) stats_sum[k] += g[P * KMC_SEGS + k];
    }
    std::vector<KmcXfer> sv, rv;
    uint64_t nrec = 0;
    plan_level(h->xcounts.data(), P, me, h->send_cap, (uint64_t)h->rec_words, &sv, &rv, &nrec);
    if (recv_records) *recv_records = nrec;
    h->xcounts_valid = true;
    return KMC_OK;
}

// kmc_step_expand + kmc_step_exchange_counts with one stream synchronisation (round 2 took two, with a host-to-device copy
// of the counts in between): k_expand fills the control block, k_packrow turns its send counters into this shard's row
// of the all-gather on the device, the collective runs behind it on the same stream, and the host reads the gathered
// rows back once.  Its own counts come out of the same rows.
int kmc_step_expand_counts(kmc_handle* h, const int64_t* stats, int32_t n_stats, int64_t* stats_sum, uint64_t* recv_records,
                           uint64_t* send_counts) {
    if (!h || !h->stepping) return fail(KMC_E_STATE, "kmc_step_begin first");
    const int P = h->cfg.n_shards, me = h->cfg.shard_id;
    if (P == 1 || !h->comm) {  // nothing to gather, or no communicator: the two-step path
        int rc = kmc_step_expand(h, send_counts);
        return rc ? rc : kmc_step_exchange_counts(h, stats, n_stats, stats_sum, recv_records);
    }
    if (!h->send) return fail(KMC_E_STATE, "no send area");
    if (n_stats < 0 || n_stats > KMC_EXCHANGE_STATS || (n_stats && (!stats || !stats_sum)))
        return fail(KMC_E_ARG, "bad statistics vector");
    static_assert(KMC_ROW_STATS == KMC_EXCHANGE_STATS, "row layout");
    KmcRccl* r = rccl();
    HIP_TRY(hipSetDevice(h->cfg.device));
    int rc = ensure_exchange_buffers(h);
    if (rc) return rc;
    const int slot = (int)(h->level & 1);
    if ((rc = zero_ctl(h, slot))) return rc;
    KmcArgs a = base_args(h, slot);
    a.fin = h->frontier[h->cur];
    a.fout = h->frontier[h->cur ^ 1];
    a.send = h->send;
    a.send_cap = h->send_cap;
    if ((rc = ensure_mode(h, KMC_MODE_SHARDED))) return rc;   // (a cold cache compiles here, outside the timed events)
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    if (h->n_cur && (rc = launch_expand(h, KMC_MODE_SHARDED, a, expand_grid(h, h->n_cur)))) return rc;
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    const size_t row = (size_t)P * KMC_SEGS + KMC_EXCHANGE_STATS;
    KmcPackArgs pa{};
    pa.ctl = h->ctl + slot;
    pa.row = (long long*)h->xrow_dev;
    pa.send_cap = h->send_cap;
    pa.nshards = (uint32_t)P;
    pa.shard = (uint32_t)me;
    for (int k = 0; k < KMC_EXCHANGE_STATS; ++k) pa.stats[k] = k < n_stats ? stats[k] : 0;
    {
        size_t size = sizeof(pa);
        void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &pa, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
        HIP_TRY(hipModuleLaunchKernel(h->f_packrow, 1, 1, 1, KMC_BLOCK, 1, 1, 0, h->stream, nullptr, config));
    }
    NCCL_TRY(r->AllGather(h->xrow_dev, h->xrow_dev + row, row, ncclInt64, h->comm, h->stream));
    HIP_TRY(hipMemcpyAsync(h->xrow_host + row, h->xrow_dev + row, (size_t)P * row * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));   // the level's only host wait before the payload is posted
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->res.seconds_expand += 1e-3 * ms;
    h->res.expand_launches++;
    h->xcounts.assign((size_t)P * P * KMC_SEGS, 0);
    for (int k = 0; k < n_stats; ++k) stats_sum[k] = 0;
    for (int s2 = 0; s2 < P; ++s2) {
        const int64_t* g = h->xrow_host + (size_t)(1 + s2) * row;
        for (int d = 0; d < P; ++d)
            for (int sb = 0; sb < KMC_SEGS; ++sb) {
                const int64_t c = g[d * KMC_SEGS + sb];
                if (c < 0 || (uint64_t)c > h->send_cap)
                    return fail(KMC_E_STATE, "exchange: rank %d announces %lld records for a sub-buffer of %llu", s2,
                                (long long)c, (unsigned long long)h->send_cap);
                h->xcounts[((size_t)s2 * P + d) * KMC_SEGS + sb] = (uint64_t)c;
                if (s2 == me) {
                    h->last_send_counts[d * KMC_SEGS + sb] = (uint64_t)c;
                    if (send_counts) send_counts[d * KMC_SEGS + sb] = (uint64_t)c;
                }
            }
        for (int k = 0; k < n_stats; ++k) stats_sum[k] += g[P * KMC_SEGS + k];
    }
    std::vector<KmcXfer> sv, rv;
    uint64_t nrec = 0;
    plan_level(h->xcounts.data(), P, me, h->send_cap, (uint64_t)h->rec_words, &sv, &rv, &nrec);
    if (recv_records) *recv_records = nrec;
    h->xcounts_valid = true;
    h->step_expanded = true;
    return KMC_OK;
}

// One BFS level of a shard as a PIPELINE of `parts` parts (2, 4 or 8 groups of the frontier's KMC_SEGS segments): part c is
// expanded into send area c mod 2 on the engine's stream while part c-1's counts are gathered, its records travel and are
// inserted on a second stream — the wire of a level hides behind its own expansion (inserts append to
// the NEXT frontier and to the seen-set with atomics, so they do not disturb the expansion of the current one).  Every part
// costs the host one wait (it must know the counts to post the receives), which is why small levels keep the one-shot path
// (kmc_step_expand_counts + kmc_step_exchange_payload).  The caller's statistics ride with part 0.  Afterwards the level
// stands where kmc_step_exchange_payload leaves it: kmc_step_finish is next.
int kmc_step_level_parts(kmc_handle* h, int32_t parts, const int64_t* stats, int32_t n_stats, int64_t* stats_sum,
                         uint64_t* recv_records) {
    if (!h || !h->stepping) return fail(KMC_E_STATE, "kmc_step_begin first");
    const int P = h->cfg.n_shards, me = h->cfg.shard_id;
    if (P < 2 || !h->comm) return fail(KMC_E_STATE, "a pipelined level needs a communicator (n_shards > 1, kmc_comm_init)");
    if (parts != 2 && parts != 4 && parts != 8) return fail(KMC_E_ARG, "parts must be 2, 4 or 8");
    if (!h->send) return fail(KMC_E_STATE, "no send area");
    if (n_stats < 0 || n_stats > KMC_EXCHANGE_STATS || (n_stats && (!stats || !stats_sum)))
        return fail(KMC_E_ARG, "bad statistics vector");
    const uint64_t half_cap = h->send_cap / 2;
    if (half_cap < 64) return fail(KMC_E_STATE, "send area too small to be split for a pipelined level");
    KmcRccl* r = rccl();
    HIP_TRY(hipSetDevice(h->cfg.device));
    int rc = ensure_exchange_buffers(h);
    if (rc) return rc;
    const size_t row = (size_t)P * KMC_SEGS + KMC_EXCHANGE_STATS;
    if (!h->xstream) {
        HIP_TRY(hipStreamCreateWithFlags(&h->xstream, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            HIP_TRY(hipEventCreateWithFlags(&h->ev_row[i], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&h->ev_xfer[i], hipEventDisableTiming));
            HIP_TRY(hipMalloc(&h->prow_dev[i], (size_t)(P + 1) * row * 8));
            HIP_TRY(hipHostMalloc(&h->prow_host[i], (size_t)(P + 1) * row * 8));
        }
    }
    const int slot = (int)(h->level & 1);
    if ((rc = zero_ctl(h, slot))) return rc;
    const size_t area_words = (size_t)P * KMC_SEGS * half_cap * (size_t)h->rec_words;
    const int per = KMC_SEGS / parts;
    uint64_t total_recv = 0;
    for (int k = 0; k < n_stats; ++k) stats_sum[k] = 0;

    // stage A: part c's expansion and its row, on the engine's stream
    auto stage_a = [&](int c) -> int {
        const int a2 = c & 1;
        if (c >= 2) HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_xfer[a2], 0));   // part c-2 has left this send area
        HIP_TRY(hipMemsetAsync(&(h->ctl + slot)->send_count, 0, sizeof(KmcLevelCtl) - KMC_CTL_LOCAL_BYTES, h->stream));
        KmcArgs a = base_args(h, slot);
        a.fin = h->frontier[h->cur];
        a.fout = h->frontier[h->cur ^ 1];
        a.send = h->send + (size_t)a2 * area_words;
        a.send_cap = half_cap;
        uint64_t n_part = 0;
        for (int sg = 0; sg < KMC_SEGS; ++sg) {
            if (sg / per != c) a.seg_count[sg] = 0;
            n_part += a.seg_count[sg];
        }
        if (!h->ev_chain[2 * c]) {
            HIP_TRY(hipEventCreate(&h->ev_chain[2 * c]));
            HIP_TRY(hipEventCreate(&h->ev_chain[2 * c + 1]));
        }
        HIP_TRY(hipEventRecord(h->ev_chain[2 * c], h->stream));
        int rc2 = KMC_OK;
        if (n_part && (rc2 = launch_expand(h, KMC_MODE_SHARDED, a, expand_grid(h, n_part)))) return rc2;
        HIP_TRY(hipEventRecord(h->ev_chain[2 * c + 1], h->stream));
        KmcPackArgs pa{};
        pa.ctl = h->ctl + slot;
        pa.row = (long long*)h->prow_dev[a2];
        pa.send_cap = half_cap;
        pa.nshards = (uint32_t)P;
        pa.shard = (uint32_t)me;
        for (int k = 0; k < KMC_EXCHANGE_STATS; ++k) pa.stats[k] = (c == 0 && k < n_stats) ? stats[k] : 0;
        size_t size = sizeof(pa);
        void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &pa, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
        HIP_TRY(hipModuleLaunchKernel(h->f_packrow, 1, 1, 1, KMC_BLOCK, 1, 1, 0, h->stream, nullptr, config));
        HIP_TRY(hipEventRecord(h->ev_row[a2], h->stream));
        return KMC_OK;
    };

    if ((rc = stage_a(0))) return rc;
    for (int c = 0; c < parts; ++c) {
        const int a2 = c & 1;
        if (c + 1 < parts && (rc = stage_a(c + 1))) return rc;   // queued BEFORE the host waits for part c's counts
        // stage B: part c's counts, gathered on the second stream
        HIP_TRY(hipStreamWaitEvent(h->xstream, h->ev_row[a2], 0));
        NCCL_TRY(r->AllGather(h->prow_dev[a2], h->prow_dev[a2] + row, row, ncclInt64, h->comm, h->xstream));
        HIP_TRY(hipMemcpyAsync(h->prow_host[a2] + row, h->prow_dev[a2] + row, (size_t)P * row * 8, hipMemcpyDeviceToHost, h->xstream));
        HIP_TRY(hipStreamSynchronize(h->xstream));   // this part's host wait (the previous part's insert is behind it too)
        // stage C: the plan, the transfer and the insert of part c, on the second stream
        h->xcounts.assign((size_t)P * P * KMC_SEGS, 0);
        for (int s2 = 0; s2 < P; ++s2) {
            const int64_t* g = h->prow_host[a2] + (size_t)(1 + s2) * row;
            for (int d = 0; d < P; ++d)
                for (int sb = 0; sb < KMC_SEGS; ++sb) {
                    const int64_t cnt = g[d * KMC_SEGS + sb];
                    if (cnt < 0 || (uint64_t)cnt > half_cap)
                        return fail(KMC_E_STATE, "exchange: rank %d announces %lld records for a sub-buffer of %llu", s2,
                                    (long long)cnt, (unsigned long long)half_cap);
                    h->xcounts[((size_t)s2 * P + d) * KMC_SEGS + sb] = (uint64_t)cnt;
                    if (s2 == me) h->last_send_counts[d * KMC_SEGS + sb] = (uint64_t)cnt;
                }
            if (c == 0)
                for (int k = 0; k < n_stats; ++k) stats_sum[k] += g[P * KMC_SEGS + k];
        }
        std::vector<KmcXfer> sv, rv;
        uint64_t nrec = 0;
        plan_level(h->xcounts.data(), P, me, half_cap, (uint64_t)h->rec_words, &sv, &rv, &nrec);
        if (nrec > h->recv_cap) return fail(KMC_E_STATE, "exchange: %llu records exceed the receive area", (unsigned long long)nrec);
        const u64* area = h->send + (size_t)a2 * area_words;
        if (!sv.empty() || !rv.empty()) {
            NCCL_TRY(r->GroupStart());
            for (const KmcXfer& x : sv) NCCL_TRY(r->Send(area + x.offset_words, x.words, ncclUint64, (int)x.peer, h->comm, h->xstream));
            for (const KmcXfer& x : rv) NCCL_TRY(r->Recv(h->recv + x.offset_words, x.words, ncclUint64, (int)x.peer, h->comm, h->xstream));
            NCCL_TRY(r->GroupEnd());
        }
        HIP_TRY(hipEventRecord(h->ev_xfer[a2], h->xstream));     // the send area may be refilled (part c+2)
        if ((rc = insert_received(h, nrec, h->xstream))) return rc;   // behind the receives; the receive area is reused by
                                                                      // part c+1's transfer, which this stream orders behind it
        total_recv += nrec;
    }
    for (int c = 0; c < parts; ++c) {   // every part's expansion has completed: its row was gathered
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, h->ev_chain[2 * c], h->ev_chain[2 * c + 1]));
        h->res.seconds_expand += 1e-3 * ms;
        h->res.expand_launches++;
    }
    if (recv_records) *recv_records = total_recv;
    h->xcounts_valid = false;
    h->step_expanded = true;
    return KMC_OK;
}

int kmc_step_exchange_payload(kmc_handle* h) {
    if (!h || !h->stepping || !h->step_expanded) return fail(KMC_E_STATE, "kmc_step_expand first");
    if (!h->xcounts_valid) return fail(KMC_E_STATE, "kmc_step_exchange_counts first");
    h->xcounts_valid = false;
    const int P = h->cfg.n_shards, me = h->cfg.shard_id;
    if (P == 1) return KMC_OK;
    KmcRccl* r = rccl();
    HIP_TRY(hipSetDevice(h->cfg.device));
    std::vector<KmcXfer> sv, rv;
    uint64_t nrec = 0;
    plan_level(h->xcounts.data(), P, me, h->send_cap, (uint64_t)h->rec_words, &sv, &rv, &nrec);
    if (nrec > h->recv_cap) return fail(KMC_E_STATE, "exchange: %llu records exceed the receive area", (unsigned long long)nrec);
    if (!sv.empty() || !rv.empty()) {
        NCCL_TRY(r->GroupStart());
        for (const KmcXfer& x : sv) NCCL_TRY(r->Send(h->send + x.offset_words, x.words, ncclUint64, (int)x.peer, h->comm, h->stream));
        for (const KmcXfer& x : rv) NCCL_TRY(r->Recv(h->recv + x.offset_words, x.words, ncclUint64, (int)x.peer, h->comm, h->stream));
        NCCL_TRY(r->GroupEnd());
    }
    return insert_received(h, nrec);  // queued behind the receives on the same stream
}

// The same level step for P logical shards living in ONE process on ONE device (tests, `tlc -gpus P` on a
// single GPU): counts and statistics are combined on the host, the runs move with device-to-device copies,
// every shard then inserts what it received.  stats: [n_shards][n_stats].
int kmc_step_exchange_local(kmc_handle** hs, int32_t n_shards, const int64_t* stats, int32_t n_stats, int64_t* stats_sum) {
    if (!hs || n_shards < 1 || n_shards > KMC_MAX_SHARDS) return fail(KMC_E_ARG, "bad shard list");
    const int P = n_shards;
    for (int s2 = 0; s2 < P; ++s2) {
        kmc_handle* h = hs[s2];
        if (!h || !h->stepping || !h->step_expanded) return fail(KMC_E_STATE, "kmc_step_expand first (shard %d)", s2);
        if (h->cfg.n_shards != P || h->cfg.shard_id != s2) return fail(KMC_E_ARG, "handle %d is not shard %d of %d", s2, s2, P);
        if (h->cfg.device != hs[0]->cfg.device || h->send_cap != hs[0]->send_cap || h->rec_words != hs[0]->rec_words)
            return fail(KMC_E_ARG, "local exchange: shards must share the device and the send geometry");
    }
    HIP_TRY(hipSetDevice(hs[0]->cfg.device));
    std::vector<uint64_t> counts((size_t)P * P * KMC_SEGS, 0);
    for (int s2 = 0; s2 < P; ++s2)
        for (int d = 0; d < P; ++d)
            for (int sb = 0; sb < KMC_SEGS; ++sb)
                counts[((size_t)s2 * P + d) * KMC_SEGS + sb] = d == s2 ? 0 : hs[s2]->last_send_counts[d * KMC_SEGS + sb];
    for (int k = 0; k < n_stats; ++k) {
        stats_sum[k] = 0;
        for (int s2 = 0; s2 < P; ++s2) stats_sum[k] += stats[(size_t)s2 * n_stats + k];
    }
    for (int s2 = 0; s2 < P; ++s2) {
        hs[s2]->xcounts = counts;
        hs[s2]->xcounts_valid = true;
        if (P > 1) {
            int rc = ensure_exchange_buffers(hs[s2]);
            if (rc) return rc;
        }
    }
    return KMC_OK;
}

int kmc_step_deliver_local(kmc_handle** hs, int32_t n_shards) {
    if (!hs || n_shards < 1 || n_shards > KMC_MAX_SHARDS) return fail(KMC_E_ARG, "bad shard list");
    const int P = n_shards;
    for (int s2 = 0; s2 < P; ++s2)
        if (!hs[s2] || !hs[s2]->xcounts_valid) return fail(KMC_E_STATE, "kmc_step_exchange_local first");
    HIP_TRY(hipSetDevice(hs[0]->cfg.device));
    // every shard's k_expand has completed (kmc_step_expand waits for its control block), so the send areas are final
    for (int me = 0; me < P; ++me) {
        kmc_handle* h = hs[me];
        h->xcounts_valid = false;
        std::vector<KmcXfer> sv, rv;
        uint64_t nrec = 0;
        plan_level(h->xcounts.data(), P, me, h->send_cap, (uint64_t)h->rec_words, &sv, &rv, &nrec);
        if (P > 1 && nrec > h->recv_cap) return fail(KMC_E_STATE, "exchange: receive area too small");
        // a receive from `peer` is matched by that peer's sends to `me`, in posting order, cut identically
        std::vector<size_t> cursor(P, 0);
        std::vector<std::vector<KmcXfer>> peer_sends(P);
        for (int q = 0; q < P; ++q) {
            if (q == me) continue;
            std::vector<KmcXfer> qs, qr;
            uint64_t dummy = 0;
            plan_level(h->xcounts.data(), P, q, h->send_cap, (uint64_t)h->rec_words, &qs, &qr, &dummy);
            for (const KmcXfer& x : qs)
                if ((int)x.peer == me) peer_sends[q].push_back(x);
        }
        for (const KmcXfer& x : rv) {
            const int q = (int)x.peer;
            if (cursor[q] >= peer_sends[q].size() || peer_sends[q][cursor[q]].words != x.words)
                return fail(KMC_E_STATE, "exchange plan mismatch between shards %d and %d", q, me);
            const KmcXfer& sx = peer_sends[q][cursor[q]++];
            HIP_TRY(hipMemcpyAsync(h->recv + x.offset_words, hs[q]->send + sx.offset_words, x.words * 8,
                                   hipMemcpyDeviceToDevice, h->stream));
        }
        for (int q = 0; q < P; ++q)
            if (q != me && cursor[q] != peer_sends[q].size())
                return fail(KMC_E_STATE, "exchange plan mismatch: unmatched sends from shard %d to %d", q, me);
        int rc = insert_received(h, nrec);
        if (rc) return rc;
    }
    return KMC_OK;
}

// The invariants of the CURRENT frontier without expanding it: what kmc_run does for the last level under
// max_levels (every state is normally checked when it is expanded; an unexpanded last level would otherwise go
// unchecked).  Fills violation_count / violation_fp only.
int kmc_step_check_frontier(kmc_handle* h, kmc_level_info* info) {
    if (!h || !h->stepping || !info) return fail(KMC_E_STATE, "kmc_step_begin first");
    if (h->step_expanded) return fail(KMC_E_STATE, "kmc_step_check_frontier between kmc_step_expand and kmc_step_finish");
    HIP_TRY(hipSetDevice(h->cfg.device));
    memset(info, 0, sizeof *info);
    info->depth = h->level;
    info->new_states = queue_now(h);
    if (h->n_cur == 0 || h->cfg.invariant_mask == 0) return KMC_OK;
    int rc = zero_ctl(h, 2);
    if (rc) return rc;
    KmcArgs d = base_args(h, 2);
    d.fin = h->frontier[h->cur];
    if ((rc = launch_inv(h, d, h->n_cur))) return rc;
    if ((rc = read_ctl(h, 2))) return rc;
    for (int k = 0; k < 4; ++k) {
        info->violation_count[k] = weighted(h, h->ctl_host->viol_count[k], h->ctl_host->corr_viol[k]);
        info->violation_fp[k] = h->ctl_host->viol_count[k] ? ~h->ctl_host->viol_fp_inv[k] : 0;
    }
    return KMC_OK;
}

// A violating successor OUTSIDE the state constraint is in no shard's table and no frontier.  After the
// kmc_step_finish of the expansion that generated it (and before the next kmc_step_expand overwrites that level),
// this looks for it among the successors of the retired level: *found = 1 gives its packed words and the
// fingerprint of the parent it was generated from (the smallest one).
int kmc_step_find_outside(kmc_handle* h, uint64_t fp, uint64_t* words, uint64_t* parent_fp, int32_t* found) {
    if (!h || !h->stepping || !words || !parent_fp || !found) return fail(KMC_E_ARG, "bad argument");
    if (h->step_expanded) return fail(KMC_E_STATE, "the retired level has been overwritten by kmc_step_expand");
    HIP_TRY(hipSetDevice(h->cfg.device));
    *found = 0;
    uint64_t n = 0;
    for (int sg = 0; sg < KMC_SEGS; ++sg) n += h->prev_seg_n[sg];
    if (n == 0) return KMC_OK;
    const bool had = h->have_witness;
    int rc = find_outside_witness(h, h->frontier[h->cur ^ 1], h->prev_seg_n, fp);
    if (rc) {  // "not found among the successors" is an answer here, not an error
        g_err.clear();
        h->witness_outside = false;
        h->have_witness = had;
        return KMC_OK;
    }
    for (int k = 0; k < h->W; ++k) words[k] = h->witness[k];
    *parent_fp = h->witness_parent_fp;
    *found = 1;
    return KMC_OK;
}

// Continue a sharded search from a shard checkpoint: after kmc_checkpoint_load the handle is back at the level
// boundary it was saved at; the next call is kmc_step_expand.
int kmc_step_resume(kmc_handle* h) {
    if (!h) return fail(KMC_E_ARG, "null handle");
    if (!h->table || !h->restored) return fail(KMC_E_STATE, "kmc_step_resume needs a handle restored by kmc_checkpoint_load");
    h->restored = false;
    h->stepping = true;
    h->step_expanded = false;
    h->xcounts_valid = false;
    h->t_start = now_s() - h->res.seconds_total;
    if (h->res.verdict == KMC_V_LEVEL_LIMIT) h->res.verdict = KMC_V_OK;
    h->res.queue_left = 0;
    return KMC_OK;
}

int kmc_step_set_verdict(kmc_handle* h, int32_t verdict) {
    if (!h) return fail(KMC_E_ARG, "null handle");
    h->res.verdict = verdict;
    return KMC_OK;
}

}  // extern "C"
