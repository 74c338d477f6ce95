/* Oracle-O: the C oracle's successor function (kmc_oracle.c, included below) under a breadth-first search over ORBITS of the
 * permutations of Replicas, every count weighted by the orbit's size — an independent check of the HIP engine's
 * kmc_config.symmetry where the plain search no longer fits anybody's memory (BASELINE config 5: 7 brokers, LogSize 8).
 * TEST INFRASTRUCTURE ONLY: nothing in the product links, loads or runs it.
 *
 * It shares the IDEA with the device (one stored state per orbit, weights N!/|Stab|) and nothing else:
 *   - states are the oracle's canonical bytes, renamed by permute() below (the device renames bit fields through LDS tables);
 *   - the representative of an orbit is the lexicographically smallest BYTE STRING among the images whose replicas stand in
 *     ascending order of (end, hw, epoch, log) — all arrangements of replicas with equal keys are tried (the device sorts by a
 *     finer key, checks tied neighbours for being interchangeable, and compares packed words);
 *   - an exact seen-set of byte strings (optionally fingerprints for the LAST level only, whose states are never expanded).
 * The premise — every renaming is an automorphism of the spec's state graph — is checked on the reference's own text by
 * tests/test_oracle_r_cpu.py; the agreement of this program with the plain oracle on small configurations by
 * tests/test_oracle_known_answers.py.
 *
 *   gcc -O2 -pthread -std=gnu11 -o oracle/orbit_oracle oracle/orbit_oracle.c
 *   oracle/orbit_oracle --model Kip320 --N 7 --L 8 --R 8 --E 3 --levels 14 --threads 8 [--last-level-fp] [--table-log2 28]
 * prints one JSON object: levels (weighted = the plain search's), stored per level, distinct, generated, action_generated.
 *
 * --compact (round 5): the same EXACT search with the arena bit-packed — every canonical byte in just the bits its range
 * needs (22 bytes per stored state instead of 48 at Kip320 3/6/6/3) and a table of 32-bit indices — so that the 1.08 G orbit
 * representatives of Kip320 3/6/6/3 (6.45 G states) fit this container's 62 GB: 23.7 GB of states + 8.6 GB of table.  Still
 * full states compared bit for bit: no fingerprint anywhere.
 */
#include "kmc_oracle.c"

typedef struct {
    P p;
    int N, sb, rs;            /* rs = sb + 2: the state and the order of its stabiliser */
    uint64_t nf;              /* N! */
    uint8_t *arena;           /* cap_states records */
    uint64_t cap_states;
    _Atomic uint64_t nstates;
    _Atomic uint64_t *table;  /* 0 empty, 1 busy, else index + 2 */
    uint64_t tcap;
    _Atomic uint64_t *fpt;    /* last level, fingerprints only */
    uint64_t fcap;
    int fp_mode;              /* the level being produced is the last one and is not stored */
    uint32_t inv_mask;        /* --inv: invariants checked on every stored state when it is expanded (a renaming of Replicas
                                 maps violating states to violating states: the orbit's weight counts them all) */
    _Atomic uint64_t cursor;
    uint64_t lo, hi;
    int overflow;
    /* --compact */
    int compact, crs;          /* crs = bytes of a packed record (state fields + the stabiliser's order) */
    uint8_t width[KMO_MAXSB + 2];  /* bits of canonical byte i (sb: stab low byte, sb + 1: unused = 0) */
    _Atomic uint32_t *table32; /* 0 empty, 1 busy, else index + 2 */
} OE;

typedef struct {
    OE *e;
    uint64_t w;                /* weight of the state being expanded */
    uint64_t generated, new_weight, new_states, deadlocks_w, nsucc;
    uint64_t action_generated[KMO_MAX_ACTIONS];
    uint64_t viol_w[4];        /* --inv: states of the expanded level (weighted) violating each checked invariant */
} OW;

static void o_permute(const P *p, const int *img, const uint8_t *s, uint8_t *t) {
    const int N = p->N;
    for (int r = 0; r < N; r++) {
        uint8_t *d = t + img[r] * p->rstride;
        memcpy(d, REP(s, r), p->rstride);
        const int l = d[3];
        d[3] = (uint8_t)((l == 0 || l > N) ? l : img[l - 1] + 1);
        unsigned m = d[4], pm = 0;
        for (int i = 0; i < N; i++)
            if (m >> i & 1u) pm |= 1u << img[i];
        d[4] = (uint8_t)pm;
    }
    const uint8_t *g = s + p->goff;
    uint8_t *h = t + p->goff;
    memcpy(h, g, 3);
    for (int f = 0; f < 1 + p->EP1; f++) {   /* quorumState, then one (leader, isr) pair per request epoch */
        const int at = f == 0 ? 3 : 5 + 2 * (f - 1);
        const int l = g[at];
        h[at] = (uint8_t)((l == 0 || l > N) ? l : img[l - 1] + 1);
        unsigned m = g[at + 1], pm = 0;
        for (int i = 0; i < N; i++)
            if (m >> i & 1u) pm |= 1u << img[i];
        h[at + 1] = (uint8_t)pm;
    }
}
static int o_keycmp(const P *p, const uint8_t *s, int a, int b) {   /* (end, hw, epoch, log): what no renaming touches */
    int c = memcmp(REP(s, a), REP(s, b), 3);
    return c ? c : memcmp(REP(s, a) + 5, REP(s, b) + 5, p->L);
}
/* c = the representative of s's orbit, returns |Stab(s)| */
static int o_canon(const P *p, const uint8_t *s, uint8_t *c) {
    const int N = p->N, sb = p->sb;
    int order[KMO_MAXN];
    for (int i = 0; i < N; i++) {   /* insertion sort of the replicas by key */
        int j = i;
        while (j > 0 && o_keycmp(p, s, order[j - 1], i) > 0) { order[j] = order[j - 1]; j--; }
        order[j] = i;
    }
    int gstart[KMO_MAXN + 1], ng = 0;
    for (int i = 0; i < N; i++)
        if (i == 0 || o_keycmp(p, s, order[i - 1], order[i]) != 0) gstart[ng++] = i;
    gstart[ng] = N;
    /* every arrangement of the replicas inside the groups of equal keys: an odometer of per-group permutations (Heap) */
    int arr[KMO_MAXN], cnt[KMO_MAXN] = {0}, img[KMO_MAXN], count = 0;
    uint8_t t[KMO_MAXSB];
    memcpy(arr, order, sizeof(int) * N);
    for (;;) {
        for (int d = 0; d < N; d++) img[arr[d]] = d;
        o_permute(p, img, s, t);
        const int cmp = count == 0 ? -1 : memcmp(t, c, sb);
        if (cmp < 0) { memcpy(c, t, sb); count = 1; }
        else if (cmp == 0) count++;
        /* next arrangement: Heap's algorithm on the first group that still has one, resetting the groups before it */
        int g = 0;
        for (; g < ng; g++) {
            const int lo = gstart[g], n = gstart[g + 1] - lo;
            int i = 1, advanced = 0;
            while (i < n) {
                if (cnt[lo + i] < i) {
                    const int a = (i % 2 == 0) ? 0 : cnt[lo + i];
                    const int x = arr[lo + a]; arr[lo + a] = arr[lo + i]; arr[lo + i] = x;
                    cnt[lo + i]++;
                    advanced = 1;
                    break;
                }
                cnt[lo + i] = 0;
                i++;
            }
            if (advanced) break;
            /* this group wrapped around (Heap's algorithm ends on a permutation, not on the start): restore its start */
            for (int k = 0; k < n; k++) arr[lo + k] = order[lo + k];
        }
        if (g == ng) break;
    }
    return count;
}

static inline uint64_t o_hash(const uint8_t *b, int n) { return hash_bytes_strong(b, n); }

/* ---- --compact: canonical bytes <-> packed bits (every field in the bits its range needs) ---- */
static int o_bits(unsigned maxv) { int b = 0; while (maxv >> b) b++; return b; }
static void o_make_widths(OE *e) {
    const P *p = &e->p;
    const int N = p->N, L = p->L, E = p->E, R = p->R;
    int at = 0, total = 0;
    for (int r = 0; r < N; r++) {
        e->width[at++] = (uint8_t)o_bits((unsigned)L);           /* endOffset 0..L */
        e->width[at++] = (uint8_t)o_bits((unsigned)L);           /* hw */
        e->width[at++] = (uint8_t)o_bits((unsigned)E + 1);       /* leaderEpoch + 1: 0..E+1 */
        e->width[at++] = (uint8_t)o_bits((unsigned)N);           /* leader + 1: 0..N */
        e->width[at++] = (uint8_t)N;                             /* isr mask */
        for (int o = 0; o < L; o++) e->width[at++] = (uint8_t)o_bits((unsigned)(R * (E + 1)));  /* 0 = Nil, 1 + id * (E+1) + epoch */
    }
    e->width[at++] = (uint8_t)o_bits((unsigned)R);               /* nextRecordId 0..R */
    e->width[at++] = (uint8_t)o_bits((unsigned)E + 1);           /* nextLeaderEpoch 0..E+1 */
    e->width[at++] = (uint8_t)o_bits((unsigned)E + 1);           /* quorum.leaderEpoch + 1 */
    e->width[at++] = (uint8_t)o_bits((unsigned)N);               /* quorum.leader + 1 */
    e->width[at++] = (uint8_t)N;                                 /* quorum.isr */
    for (int q = 0; q < p->EP1; q++) { e->width[at++] = (uint8_t)o_bits((unsigned)N); e->width[at++] = (uint8_t)N; }
    if (at != e->sb) { fprintf(stderr, "orbit_oracle: width table does not cover the state (%d of %d bytes)\n", at, e->sb); exit(2); }
    e->width[at++] = (uint8_t)o_bits((unsigned)e->nf);           /* |Stab| <= N! */
    for (int i = 0; i < at; i++) total += e->width[i];
    e->crs = (total + 7) / 8;
}
/* returns 0 when a byte does not fit its width (a TypeOk violation would: the search stops there, loudly) */
static int o_pack(const OE *e, const uint8_t *st, int stab, uint8_t *out) {
    memset(out, 0, (size_t)e->crs);
    int bit = 0;
    for (int i = 0; i <= e->sb; i++) {
        const unsigned v = i < e->sb ? st[i] : (unsigned)stab;
        const int w = e->width[i];
        if (v >> w) return 0;
        for (int b = 0; b < w; b++, bit++)
            if (v >> b & 1u) out[bit >> 3] |= (uint8_t)(1u << (bit & 7));
    }
    return 1;
}
static int o_unpack(const OE *e, const uint8_t *in, uint8_t *st) {
    int bit = 0, stab = 0;
    for (int i = 0; i <= e->sb; i++) {
        unsigned v = 0;
        const int w = e->width[i];
        for (int b = 0; b < w; b++, bit++) v |= (unsigned)(in[bit >> 3] >> (bit & 7) & 1u) << b;
        if (i < e->sb) st[i] = (uint8_t)v; else stab = (int)v;
    }
    return stab;
}
/* --compact: returns 1 when st (a representative) is new.  The stabiliser's order is a function of the state, so the packed
 * record (state bits, then |Stab|) of an equal state is equal in every bit: one memcmp decides. */
static int o_insert_compact(OE *e, const uint8_t *st, int stab) {
    uint8_t rec[KMO_MAXSB + 8];
    if (!o_pack(e, st, stab, rec)) { fprintf(stderr, "orbit_oracle: a field exceeds its range (TypeOk broken?)\n"); e->overflow = 1; return 0; }
    const uint64_t h = o_hash(st, e->sb);
    uint64_t i = h & (e->tcap - 1);
    for (uint64_t probes = 0;; ) {
        if (probes > e->tcap) { e->overflow = 1; return 0; }
        uint32_t v = atomic_load_explicit(&e->table32[i], memory_order_acquire);
        if (v == 0) {
            uint32_t exp = 0;
            if (!atomic_compare_exchange_strong(&e->table32[i], &exp, 1u)) continue;
            const uint64_t idx = atomic_fetch_add(&e->nstates, 1);
            if (idx >= e->cap_states || idx + 2 > 0xFFFFFFFFull) { e->overflow = 1; atomic_store(&e->table32[i], 0u); return 0; }
            memcpy(e->arena + idx * (uint64_t)e->crs, rec, (size_t)e->crs);
            atomic_store_explicit(&e->table32[i], (uint32_t)(idx + 2), memory_order_release);
            return 1;
        }
        if (v == 1) continue;
        if (memcmp(e->arena + (uint64_t)(v - 2) * (uint64_t)e->crs, rec, (size_t)e->crs) == 0) return 0;
        probes++;
        i = (i + 1) & (e->tcap - 1);
    }
}

/* returns 1 when st (a representative) is new; records it with its stabiliser's order */
static int o_insert(OE *e, const uint8_t *st, int stab) {
    if (e->compact) return o_insert_compact(e, st, stab);
    const int sb = e->sb;
    const uint64_t h = o_hash(st, sb);
    uint64_t i = h & (e->tcap - 1);
    for (uint64_t probes = 0;; ) {
        if (probes > e->tcap) { e->overflow = 1; return 0; }
        uint64_t v = atomic_load_explicit(&e->table[i], memory_order_acquire);
        if (v == 0) {
            if (e->fp_mode) break;   /* not among the stored states: go on to the fingerprints of the last level */
            uint64_t exp = 0;
            if (!atomic_compare_exchange_strong(&e->table[i], &exp, 1)) continue;
            const uint64_t idx = atomic_fetch_add(&e->nstates, 1);
            if (idx >= e->cap_states) { e->overflow = 1; atomic_store(&e->table[i], 0); return 0; }
            uint8_t *r = e->arena + idx * (uint64_t)e->rs;
            memcpy(r, st, sb);
            r[sb] = (uint8_t)(stab & 0xFF);
            r[sb + 1] = (uint8_t)(stab >> 8);
            atomic_store_explicit(&e->table[i], idx + 2, memory_order_release);
            return 1;
        }
        if (v == 1) continue;
        if (memcmp(e->arena + (v - 2) * (uint64_t)e->rs, st, sb) == 0) return 0;
        probes++;
        i = (i + 1) & (e->tcap - 1);
    }
    const uint64_t hv = h < 2 ? h + 2 : h;
    i = (h >> 7) & (e->fcap - 1);
    for (uint64_t probes = 0;; ) {
        if (probes > e->fcap) { e->overflow = 1; return 0; }
        uint64_t v = atomic_load_explicit(&e->fpt[i], memory_order_acquire);
        if (v == 0) {
            uint64_t exp = 0;
            if (atomic_compare_exchange_strong(&e->fpt[i], &exp, hv)) return 1;
            continue;
        }
        if (v == hv) return 0;
        probes++;
        i = (i + 1) & (e->fcap - 1);
    }
}

static void o_emit(void *ctx, int action, const uint8_t *succ) {
    OW *w = ctx;
    OE *e = w->e;
    w->generated += w->w;
    w->action_generated[action] += w->w;
    w->nsucc++;
    uint8_t c[KMO_MAXSB];
    const int stab = o_canon(&e->p, succ, c);
    if (o_insert(e, c, stab)) {
        w->new_states++;
        w->new_weight += e->nf / (uint64_t)stab;
    }
}
static void *o_worker(void *a) {
    OW *w = a;
    OE *e = w->e;
    for (;;) {
        const uint64_t lo = atomic_fetch_add(&e->cursor, 256);
        if (lo >= e->hi) break;
        const uint64_t hi = lo + 256 < e->hi ? lo + 256 : e->hi;
        for (uint64_t idx = lo; idx < hi; idx++) {
            uint8_t ub[KMO_MAXSB];
            const uint8_t *r;
            int stab;
            if (e->compact) { stab = o_unpack(e, e->arena + idx * (uint64_t)e->crs, ub); r = ub; }
            else { r = e->arena + idx * (uint64_t)e->rs; stab = r[e->sb] | (r[e->sb + 1] << 8); }
            w->w = e->nf / (uint64_t)stab;
            w->nsucc = 0;
            for (int inv = 0; inv < 4; inv++)
                if ((e->inv_mask >> inv & 1u) && !model_invariant(&e->p, inv, r)) w->viol_w[inv] += w->w;
            model_expand(&e->p, r, o_emit, w);
            if (w->nsucc == 0) w->deadlocks_w += w->w;
        }
    }
    return NULL;
}
static void *o_map(uint64_t bytes) {
    void *m = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (m == MAP_FAILED) { fprintf(stderr, "orbit_oracle: mmap of %llu bytes failed\n", (unsigned long long)bytes); exit(2); }
    return m;
}

int main(int argc, char **argv) {
    kmo_config c = {.model = M_KIP320, .N = 3, .L = 2, .R = 2, .E = 1, .K = 2, .MaxId = 10, .inv_mask = 0, .threads = 4};
    int levels = 0, last_fp = 0, tlog = 26, flog = 0, compact = 0;
    uint32_t inv_mask = 0;
    uint64_t cap_states = 0;
    static const char *names[] = {"IdSequence", "FiniteReplicatedLog", "KafkaTruncateToHighWatermark", "Kip101", "Kip279", "Kip320", "Kip320FirstTry"};
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--model")) { ++i; for (int m = 2; m <= 6; m++) if (!strcmp(argv[i], names[m])) c.model = m; }
        else if (!strcmp(argv[i], "--N")) c.N = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--L")) c.L = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--R")) c.R = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--E")) c.E = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--levels")) levels = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--threads")) c.threads = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--table-log2")) tlog = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--fp-table-log2")) flog = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--max-stored")) cap_states = strtoull(argv[++i], NULL, 0);
        else if (!strcmp(argv[i], "--last-level-fp")) last_fp = 1;
        else if (!strcmp(argv[i], "--compact")) compact = 1;
        else if (!strcmp(argv[i], "--inv")) inv_mask = (uint32_t)strtoul(argv[++i], NULL, 0);
        else { fprintf(stderr, "unknown arg %s\n", argv[i]); return 2; }
    }
    if (c.model < M_TRUNC_HW || c.model > M_KIP320_FIRST) { fprintf(stderr, "Kafka family only\n"); return 2; }
    OE *e = calloc(1, sizeof *e);
    if (!setup_params(&e->p, &c)) { fprintf(stderr, "bad constants\n"); return 2; }
    e->N = c.N; e->sb = e->p.sb; e->rs = e->sb + 2;
    e->inv_mask = inv_mask;
    e->nf = 1;
    for (int i = 2; i <= c.N; i++) e->nf *= (uint64_t)i;
    e->tcap = 1ull << tlog;
    e->cap_states = cap_states ? cap_states : e->tcap / 2;
    if (compact && last_fp) { fprintf(stderr, "--compact is the exact search: no --last-level-fp\n"); return 2; }
    e->compact = compact;
    if (compact) {
        o_make_widths(e);
        e->arena = o_map(e->cap_states * (uint64_t)e->crs);
        e->table32 = o_map(e->tcap * 4);
    } else {
        e->arena = o_map(e->cap_states * (uint64_t)e->rs);
        e->table = o_map(e->tcap * 8);
    }
    if (last_fp) { e->fcap = 1ull << (flog ? flog : tlog + 2); e->fpt = o_map(e->fcap * 8); }
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    uint8_t init[KMO_MAXSB] = {0}, ci[KMO_MAXSB];
    kafka_init(&e->p, init);
    const int st0 = o_canon(&e->p, init, ci);
    o_insert(e, ci, st0);
    uint64_t lv_w[KMO_MAX_LEVELS] = {0}, lv_n[KMO_MAX_LEVELS] = {0}, action_generated[KMO_MAX_ACTIONS] = {0};
    uint64_t generated = 1, distinct = e->nf / (uint64_t)st0, deadlocks = 0, stored = 1;
    int nl = 1, exhausted = 0;
    lv_w[0] = distinct; lv_n[0] = 1;
    uint64_t lo = 0, hi = 1;
    uint64_t viol_total[4] = {0}, viol_first_count[4] = {0};
    int viol_first_depth[4] = {0};
    OW *ws = calloc((size_t)c.threads, sizeof *ws);
    pthread_t th[256];
    while (!levels || nl < levels) {
        e->fp_mode = last_fp && levels && nl == levels - 1;
        e->lo = lo; e->hi = hi;
        atomic_store(&e->cursor, lo);
        for (int t = 0; t < c.threads; t++) { memset(&ws[t], 0, sizeof ws[t]); ws[t].e = e; pthread_create(&th[t], NULL, o_worker, &ws[t]); }
        uint64_t nw = 0, nn = 0;
        for (int t = 0; t < c.threads; t++) {
            pthread_join(th[t], NULL);
            generated += ws[t].generated; nw += ws[t].new_weight; nn += ws[t].new_states; deadlocks += ws[t].deadlocks_w;
            for (int a = 0; a < KMO_MAX_ACTIONS; a++) action_generated[a] += ws[t].action_generated[a];
        }
        for (int inv = 0; inv < 4; inv++) {   /* the level just expanded is depth nl */
            uint64_t v = 0;
            for (int t = 0; t < c.threads; t++) v += ws[t].viol_w[inv];
            if (v && !viol_first_depth[inv]) { viol_first_depth[inv] = nl; viol_first_count[inv] = v; }
            viol_total[inv] += v;
        }
        if (e->overflow) { fprintf(stderr, "orbit_oracle: table or arena full at level %d\n", nl + 1); return 3; }
        if (nn == 0) { exhausted = 1; break; }
        lv_w[nl] = nw; lv_n[nl] = nn; nl++;
        distinct += nw; stored += nn;
        lo = hi; hi = atomic_load(&e->nstates);
        fprintf(stderr, "level %d: %llu states from %llu stored\n", nl, (unsigned long long)nw, (unsigned long long)nn);
        if (e->fp_mode) break;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    printf("{\"model\": \"%s\", \"N\": %d, \"L\": %d, \"R\": %d, \"E\": %d, \"orbit_counting\": true, \"exhausted\": %s, \"last_level_fingerprints_only\": %s, \"compact_exact\": %s, \"stored_record_bytes\": %d, "
           "\"distinct\": %llu, \"generated\": %llu, \"depth\": %d, \"stored\": %llu, \"deadlock_states\": %llu, \"levels\": [",
           names[c.model], c.N, c.L, c.R, c.E, exhausted ? "true" : "false", last_fp ? "true" : "false", compact ? "true" : "false", compact ? e->crs : e->rs, (unsigned long long)distinct,
           (unsigned long long)generated, nl, (unsigned long long)stored, (unsigned long long)deadlocks);
    for (int i = 0; i < nl; i++) printf("%s%llu", i ? ", " : "", (unsigned long long)lv_w[i]);
    printf("], \"stored_per_level\": [");
    for (int i = 0; i < nl; i++) printf("%s%llu", i ? ", " : "", (unsigned long long)lv_n[i]);
    printf("], \"action_generated\": [");
    for (int a = 0; a < KMO_MAX_ACTIONS; a++) printf("%s%llu", a ? ", " : "", (unsigned long long)action_generated[a]);
    /* (on a level budget the last level is stored but not expanded, hence not checked: as the device's kmc_run leaves it
     * to the invariant pass over the last frontier) */
    printf("], \"inv_mask\": %u, \"violating_states\": [%llu, %llu, %llu, %llu], \"first_violation_depth\": [%d, %d, %d, %d], "
           "\"violating_at_first_depth\": [%llu, %llu, %llu, %llu",
           inv_mask, (unsigned long long)viol_total[0], (unsigned long long)viol_total[1], (unsigned long long)viol_total[2],
           (unsigned long long)viol_total[3], viol_first_depth[0], viol_first_depth[1], viol_first_depth[2], viol_first_depth[3],
           (unsigned long long)viol_first_count[0], (unsigned long long)viol_first_count[1], (unsigned long long)viol_first_count[2],
           (unsigned long long)viol_first_count[3]);
    printf("], \"threads\": %d, \"seconds\": %.1f}\n", c.threads, (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec));
    return 0;
}
