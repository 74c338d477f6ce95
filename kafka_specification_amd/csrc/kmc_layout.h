// kmc_layout.h — bit-packed state vector layouts for the Kafka replication specs.
//
// Shared verbatim by the host engine (runtime parameters) and by the device code (the same
// constexpr function evaluated at compile time for one specialised (model,N,L,R,E)), so the
// two can never disagree.  Self-contained: no includes (it is also fed to hiprtc).
//
// Kafka-family state vector (`vars`, KafkaReplication.tla:75).  Two arrangements of the same fields (KMC_LAYOUT_*):
//
// TIGHT — sequential packing, least-significant bit first, fields may straddle 64-bit words:
//   for r in Replicas:  log[r]     L records x BR bits      replicaLog[r].records (FiniteReplicatedLog.tla:42)
//                                  record = 0 (Nil) | ((id+1) << BEr | epoch)    LogRecords, KafkaReplication.tla:82
//   for r in Replicas:  end[r] BO  replicaLog[r].endOffset   (FiniteReplicatedLog.tla:41)
//                       hw[r]  BO  replicaState[r].hw        (KafkaReplication.tla:96)
//                       ep[r]  BE  replicaState[r].leaderEpoch + 1   (Nil = -1 -> 0)   (:97, :39)
//                       ldr[r] BL  replicaState[r].leader: 0 = None, else index+1      (:98, :38)
//                       isr[r] N   replicaState[r].isr bitmask                          (:99)
//   nextRecordId BNR (:78) | nextLeaderEpoch BE (:77) | quorumState: ep BE, ldr BL, isr N (:87-89)
//   for e in 0..E: request with leaderEpoch e: ldr BL, isr N   (zero while e >= nextLeaderEpoch)
// leaderAndIsrRequests is a set, but ControllerUpdateIsr (KafkaReplication.tla:138-145) is its
// only writer and always adds the record whose leaderEpoch equals the old nextLeaderEpoch, so
// the set is in bijection with this epoch-indexed array.  Unwritten log slots are 0
// (FiniteReplicatedLog.tla:93,108), so equal states have equal bits.
//
// REPLICA-MAJOR (rm = 1, 2) — a replica's log and its small fields (end, hw, ep, ldr, isr: one group) sit at a place that is
// the same function of the replica index for every replica and never straddle a word, so that a field of a replica chosen
// at RUN TIME is "select a word, shift by a multiple of a stride, extract at a compile-time offset": what k_expand's
// kind-major pass 2 needs (kmc_device.h).
//   rm = 1, one replica per word: word r = log from bit 0 (alone in the low 32-bit half when it fits there), then the small
//       group (from bit 32 when both halves fit).  No shift at all; at the headline's constants no field of any replica
//       straddles a 32-bit register.
//   rm = 2, grouped: the logs packed lg_q per word into the first words, the small groups sm_q per word into the next.
//       For replicas that are much smaller than a word (5 brokers, LogSize 2: three words either way) or larger than one
//       (7 brokers, LogSize 8: a 48-bit log and a 21-bit group — 11 words where the tight packing needs 9).
// The global fields (nextRecordId ... the requests) go first-fit into the bits these words leave free — never straddling a
// word — and into extra words behind them.  Automatic choice: rm = 1 when it costs no word over TIGHT; else rm = 2 when
// it costs at most a quarter more; else TIGHT.  KMC_LAYOUT=tight|rm|rmg (host) overrides, for tests and A/B runs.
//
// FiniteReplicatedLog standalone: for r: log[r] (L x BK bits, record = 0 Nil | 1..K), then
// for r: end[r] (BO).   IdSequence standalone: one 64-bit word = nextId.
//
// AsyncIsr standalone (AsyncIsr.tla:31-35), checked under the state constraint
// offsets[Leader] <= L /\ controllerState.version <= E (L = MaxOffset, E = MaxVersion; replica 0 is
// `Leader`); every field has room for the one value beyond the bound that a successor outside the
// constraint can carry:
//   controllerState: isr N | version BV (0..E+1)
//   leaderState: isr N | version BV | pendingIsr N | pendingVersion+1 BV (Nil = -1 -> 0) |
//                offsets[r] BF (0..L+1) for r in Replicas
//   requests  (a set of [isr, version], version 0..E): (E+1) x 2^N bits, bit = version * 2^N + isr mask
//   updates   (a set of [isr, version]): the controller is its only writer and always adds
//             [isr, version = controllerState.version + 1] (AsyncIsr.tla:68-86), so the set is in
//             bijection with the array, indexed by version 1..E+1, of the isr written at that
//             version: (E+1) x N bits, zero while version > controllerState.version.
#pragma once

#define KMC_MAXN 8
#define KMC_MAXE1 8
#define KMC_MAXW 12

#define KMC_MODEL_IDSEQUENCE 0
#define KMC_MODEL_FINITE_REPLICATED_LOG 1
#define KMC_MODEL_TRUNCATE_TO_HW 2
#define KMC_MODEL_KIP101 3
#define KMC_MODEL_KIP279 4
#define KMC_MODEL_KIP320 5
#define KMC_MODEL_KIP320_FIRST_TRY 6
#define KMC_MODEL_ASYNC_ISR 7

#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
#define KMC_HD __host__ __device__
#else
#define KMC_HD
#endif

// number of bits needed for values 0..nvalues-1
KMC_HD constexpr int kmc_bits_for(long long nvalues) {
    int b = 0;
    while ((1ll << b) < nvalues) ++b;
    return b;
}

#define KMC_LAYOUT_AUTO 0   // replica-major when it fits and costs no extra word, else tight
#define KMC_LAYOUT_TIGHT 1
#define KMC_LAYOUT_RM 2     // replica-major: one replica per word when a replica fits one, else grouped
#define KMC_LAYOUT_RMG 3    // replica-major, grouped

struct KmcLayout {
    int model, N, L, R, E, K;
    int rm;                                           // 0 tight, 1 replica-major one per word, 2 replica-major grouped
    // rm != 0: the log of replica r is the LB bits at bit (r % lg_q) * lg_stride + lg_base of word lg_word0 + r / lg_q,
    // its small group (end | hw | ep | ldr | isr, SB bits) at bit (r % sm_q) * sm_stride + sm_base of word sm_word0 + r / sm_q
    int LB, SB;
    int lg_q, lg_stride, lg_base, lg_word0, lg_words;
    int sm_q, sm_stride, sm_base, sm_word0, sm_words;
    int BO, BR, BEr, BId, BE, BL, BI, BNR;  // field widths
    int log_off[KMC_MAXN], end_off[KMC_MAXN], hw_off[KMC_MAXN], ep_off[KMC_MAXN], ldr_off[KMC_MAXN],
        isr_off[KMC_MAXN];
    int nextrec_off, nextep_off, qep_off, qldr_off, qisr_off;
    int reqldr_off[KMC_MAXE1], reqisr_off[KMC_MAXE1];
    // AsyncIsr
    int BV, BF;
    int a_cisr, a_cver, a_lisr, a_lver, a_pisr, a_pver, a_off[KMC_MAXN], a_req, a_upd;
    int bits, W;
    int valid;  // 0 when the parameters cannot be packed (see kmc_make_layout)
};

KMC_HD constexpr KmcLayout kmc_make_layout(int model, int N, int L, int R, int E, int K, int lm = KMC_LAYOUT_AUTO) {
    KmcLayout y{};
    y.model = model; y.N = N; y.L = L; y.R = R; y.E = E; y.K = K;
    y.valid = 0; y.rm = 0;
    if (model == KMC_MODEL_IDSEQUENCE) {
        y.bits = 64; y.W = 1; y.valid = 1;
        return y;
    }
    if (N < 1 || N > KMC_MAXN || L < 1) return y;
    int pos = 0;
    if (model == KMC_MODEL_ASYNC_ISR) {
        if (N > 6 || E < 0 || E + 1 > KMC_MAXE1 || L > 250) return y;  // 2^N request bits per version must fit one u64
        y.BV = kmc_bits_for(E + 2);
        y.BF = kmc_bits_for(L + 2);
        y.a_cisr = pos; pos += N;
        y.a_cver = pos; pos += y.BV;
        y.a_lisr = pos; pos += N;
        y.a_lver = pos; pos += y.BV;
        y.a_pisr = pos; pos += N;
        y.a_pver = pos; pos += y.BV;
        for (int r = 0; r < N; ++r) { y.a_off[r] = pos; pos += y.BF; }
        y.a_req = pos; pos += (E + 1) * (1 << N);
        y.a_upd = pos; pos += (E + 1) * N;
        y.bits = pos; y.W = (pos + 63) / 64;
        y.valid = y.W >= 1 && y.W <= KMC_MAXW;
        return y;
    }
    y.BO = kmc_bits_for(L + 1);
    if (model == KMC_MODEL_FINITE_REPLICATED_LOG) {
        if (K < 1) return y;
        y.BR = kmc_bits_for(K + 1);
        if (y.BR * L > 64) return y;
        for (int r = 0; r < N; ++r) { y.log_off[r] = pos; pos += y.BR * L; }
        for (int r = 0; r < N; ++r) { y.end_off[r] = pos; pos += y.BO; }
        y.bits = pos; y.W = (pos + 63) / 64;
        y.valid = y.W >= 1 && y.W <= KMC_MAXW;
        return y;
    }
    if (R < 1 || E < 0 || E + 1 > KMC_MAXE1) return y;
    y.BEr = kmc_bits_for(E + 1);      // record.epoch in 0..E
    y.BId = kmc_bits_for(R + 1);      // record.id+1 in 1..R, 0 reserved for Nil
    y.BR = y.BEr + y.BId;
    if (y.BR * L > 64) return y;      // one log must fit one 64-bit lane value
    y.BE = kmc_bits_for(E + 2);       // leaderEpoch+1 in 0..E+1 ; nextLeaderEpoch in 0..E+1
    y.BL = kmc_bits_for(N + 1);
    y.BI = N;
    y.BNR = kmc_bits_for(R + 1);      // nextRecordId in 0..R
    for (int r = 0; r < N; ++r) { y.log_off[r] = pos; pos += y.BR * L; }
    for (int r = 0; r < N; ++r) {
        y.end_off[r] = pos; pos += y.BO;
        y.hw_off[r] = pos; pos += y.BO;
        y.ep_off[r] = pos; pos += y.BE;
        y.ldr_off[r] = pos; pos += y.BL;
        y.isr_off[r] = pos; pos += y.BI;
    }
    y.nextrec_off = pos; pos += y.BNR;
    y.nextep_off = pos; pos += y.BE;
    y.qep_off = pos; pos += y.BE;
    y.qldr_off = pos; pos += y.BL;
    y.qisr_off = pos; pos += y.BI;
    for (int e = 0; e <= E; ++e) {
        y.reqldr_off[e] = pos; pos += y.BL;
        y.reqisr_off[e] = pos; pos += y.BI;
    }
    y.bits = pos; y.W = (pos + 63) / 64;
    y.valid = y.W >= 1 && y.W <= KMC_MAXW;
    if (lm == KMC_LAYOUT_TIGHT) return y;
    // ---- replica-major arrangements of the same fields ----
    const int logbits = y.BR * L, small = 2 * y.BO + y.BE + y.BL + y.BI;
    KmcLayout best = y;
    int have = 0;
    for (int form = 1; form <= 2 && !have; ++form) {
        if (form == 1 && (lm == KMC_LAYOUT_RMG || logbits + small > 64)) continue;   // a replica does not fit one word
        KmcLayout z = y;
        z.rm = form; z.LB = logbits; z.SB = small;
        // free regions for the global fields: the tails of the words holding replica fields (for form 1 also the gap
        // between a short log and bit 32), then extra words
        int reg_off[2 * KMC_MAXN + KMC_MAXW] = {}, reg_len[2 * KMC_MAXN + KMC_MAXW] = {};
        int nreg = 0, words = 0;
        if (form == 1) {
            const int small0 = (logbits <= 32 && small <= 32) ? 32 : logbits;
            z.lg_q = 1; z.lg_stride = 0; z.lg_base = 0; z.lg_word0 = 0; z.lg_words = N;
            z.sm_q = 1; z.sm_stride = 0; z.sm_base = small0; z.sm_word0 = 0; z.sm_words = N;
            words = N;
            for (int r = 0; r < N; ++r) { reg_off[nreg] = 64 * r + small0 + small; reg_len[nreg] = 64 - small0 - small; ++nreg; }
            if (small0 == 32)
                for (int r = 0; r < N; ++r) { reg_off[nreg] = 64 * r + logbits; reg_len[nreg] = 32 - logbits; ++nreg; }
        } else {
            z.lg_q = 64 / logbits; z.lg_stride = logbits; z.lg_base = 0; z.lg_word0 = 0;
            z.lg_words = (N + z.lg_q - 1) / z.lg_q;
            z.sm_q = 64 / small; z.sm_stride = small; z.sm_base = 0; z.sm_word0 = z.lg_words;
            z.sm_words = (N + z.sm_q - 1) / z.sm_q;
            words = z.lg_words + z.sm_words;
            if (words > KMC_MAXW) continue;
            for (int i = 0; i < z.lg_words; ++i) {
                const int n = (i + 1) * z.lg_q <= N ? z.lg_q : N - i * z.lg_q;
                reg_off[nreg] = 64 * i + n * logbits; reg_len[nreg] = 64 - n * logbits; ++nreg;
            }
            for (int i = 0; i < z.sm_words; ++i) {
                const int n = (i + 1) * z.sm_q <= N ? z.sm_q : N - i * z.sm_q;
                reg_off[nreg] = 64 * (z.sm_word0 + i) + n * small; reg_len[nreg] = 64 - n * small; ++nreg;
            }
        }
        for (int r = 0; r < N; ++r) {
            z.log_off[r] = 64 * (z.lg_word0 + r / z.lg_q) + (r % z.lg_q) * z.lg_stride + z.lg_base;
            const int so = 64 * (z.sm_word0 + r / z.sm_q) + (r % z.sm_q) * z.sm_stride + z.sm_base;
            z.end_off[r] = so; z.hw_off[r] = so + y.BO; z.ep_off[r] = so + 2 * y.BO; z.ldr_off[r] = so + 2 * y.BO + y.BE;
            z.isr_off[r] = so + 2 * y.BO + y.BE + y.BL;
        }
        const int nglob = 5 + 2 * (E + 1);
        bool fits = true;
        for (int g = 0; g < nglob && fits; ++g) {
            const int e = (g - 5) / 2;
            const int bits = g == 0 ? y.BNR : g == 1 ? y.BE : g == 2 ? y.BE : g == 3 ? y.BL : g == 4 ? y.BI
                             : ((g - 5) % 2 == 0 ? y.BL : y.BI);
            int at = -1;
            for (int k = 0; k < nreg && at < 0; ++k)
                if (reg_len[k] >= bits) { at = reg_off[k]; reg_off[k] += bits; reg_len[k] -= bits; }
            if (at < 0) {
                if (words >= KMC_MAXW) { fits = false; break; }
                reg_off[nreg] = 64 * words + bits; reg_len[nreg] = 64 - bits; ++nreg;
                at = 64 * words; ++words;
            }
            if (g == 0) z.nextrec_off = at;
            else if (g == 1) z.nextep_off = at;
            else if (g == 2) z.qep_off = at;
            else if (g == 3) z.qldr_off = at;
            else if (g == 4) z.qisr_off = at;
            else if ((g - 5) % 2 == 0) z.reqldr_off[e] = at;
            else z.reqisr_off[e] = at;
        }
        if (!fits) continue;
        z.W = words;
        z.valid = z.W >= 1 && z.W <= KMC_MAXW;
        if (!z.valid) continue;
        // the automatic choice: one replica per word when it costs no word over the tight packing, grouped when it costs
        // at most a quarter more
        if (lm == KMC_LAYOUT_AUTO && z.W > (form == 1 ? y.W : y.W + (y.W + 3) / 4)) continue;
        best = z;
        have = 1;
    }
    if (!have && lm != KMC_LAYOUT_AUTO) best.valid = 0;
    return best;
}

// Generic bit-field access on a packed state.  With compile-time `off`/`bits` (the device
// path after unrolling) every branch below folds away.
KMC_HD inline unsigned long long kmc_getbits(const unsigned long long* w, int off, int bits) {
    if (bits == 0) return 0ull;
    const int i = off >> 6, s = off & 63;
    unsigned long long v = w[i] >> s;
    if (s + bits > 64) v |= w[i + 1] << (64 - s);
    return bits >= 64 ? v : (v & ((1ull << bits) - 1ull));
}
KMC_HD inline void kmc_setbits(unsigned long long* w, int off, int bits, unsigned long long val) {
    if (bits == 0) return;
    const int i = off >> 6, s = off & 63;
    const unsigned long long m = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
    w[i] = (w[i] & ~(m << s)) | ((val & m) << s);
    if (s + bits > 64) {
        const int lo = 64 - s;  // bits already written into word i
        w[i + 1] = (w[i + 1] & ~(m >> lo)) | ((val & m) >> lo);
    }
}

// OR a (masked) value into a field that is still zero: what building a state from scratch needs (no read-modify-write mask)
KMC_HD inline void kmc_orbits(unsigned long long* w, int off, int bits, unsigned long long val) {
    if (bits == 0) return;
    const int i = off >> 6, s = off & 63;
    w[i] |= val << s;
    if (s + bits > 64) w[i + 1] |= val >> (64 - s);
}

// XOR a value into a field: with d = (field a) ^ (field b), XOR-ing d into both swaps them
KMC_HD inline void kmc_xorbits(unsigned long long* w, int off, int bits, unsigned long long val) {
    if (bits == 0) return;
    const int i = off >> 6, s = off & 63;
    w[i] ^= val << s;
    if (s + bits > 64) w[i + 1] ^= val >> (64 - s);
}

// ---- permutations of Replicas (symmetry reduction with orbit counting: kmc_config.symmetry, kmc_device.h KmcSymm) ----
// The specs never tell two replicas apart (KafkaReplication.tla quantifies over Replicas everywhere, :158-310; Init :109-120
// treats them alike), so a permutation of Replicas maps reachable states to reachable states, successors to successors and
// keeps every invariant.  A permutation acts on a packed state by moving the per-replica fields and renaming the replica
// ids held in fields (leader: 0 = None | index + 1) and the isr bit masks.
KMC_HD constexpr int kmc_factorial(int n) {
    int f = 1;
    for (int i = 2; i <= n; ++i) f *= i;
    return f;
}
// image of r under the P-th permutation of 0..N-1, permutations ranked by the lexicographic order of their image
// sequences (P = 0 is the identity)
KMC_HD constexpr int kmc_perm_image(int N, int P, int r) {
    int avail[KMC_MAXN] = {};
    for (int i = 0; i < N; ++i) avail[i] = i;
    int n = N, img = 0;
    for (int pos = 0; pos <= r; ++pos) {
        const int f = kmc_factorial(N - 1 - pos);
        const int idx = P / f;
        P %= f;
        img = avail[idx];
        for (int j = idx; j + 1 < n; ++j) avail[j] = avail[j + 1];
        --n;
    }
    return img;
}
// models whose state is a function of Replicas alone (AsyncIsr singles out `Leader`, AsyncIsr.tla:24,29)
KMC_HD constexpr bool kmc_model_symmetric(int model) {
    return model == KMC_MODEL_FINITE_REPLICATED_LOG || (model >= KMC_MODEL_TRUNCATE_TO_HW && model <= KMC_MODEL_KIP320_FIRST_TRY);
}
KMC_HD inline unsigned long long kmc_permute_mask(int N, const int* img, unsigned long long m) {
    unsigned long long out = 0;
    for (int i = 0; i < N; ++i)
        if (m >> i & 1ull) out |= 1ull << img[i];
    return out;
}
// t = the state s with replica r renamed img[r] — the run-time-layout form (host engine: witness of the initial state's
// orbit, kmc_contains / kmc_canonical_state; the device's compile-time form is KmcSymm::permute<P>)
KMC_HD inline void kmc_permute_state(const KmcLayout& y, const int* img, const unsigned long long* s, unsigned long long* t) {
    for (int k = 0; k < y.W; ++k) t[k] = 0;
    const bool kafka = y.model != KMC_MODEL_FINITE_REPLICATED_LOG;
    for (int r = 0; r < y.N; ++r) {
        const int d = img[r];
        kmc_orbits(t, y.log_off[d], y.BR * y.L, kmc_getbits(s, y.log_off[r], y.BR * y.L));
        kmc_orbits(t, y.end_off[d], y.BO, kmc_getbits(s, y.end_off[r], y.BO));
        if (!kafka) continue;
        kmc_orbits(t, y.hw_off[d], y.BO, kmc_getbits(s, y.hw_off[r], y.BO));
        kmc_orbits(t, y.ep_off[d], y.BE, kmc_getbits(s, y.ep_off[r], y.BE));
        const unsigned long long l1 = kmc_getbits(s, y.ldr_off[r], y.BL);
        kmc_orbits(t, y.ldr_off[d], y.BL, l1 == 0 || l1 > (unsigned long long)y.N ? l1 : (unsigned long long)img[l1 - 1] + 1);
        kmc_orbits(t, y.isr_off[d], y.BI, kmc_permute_mask(y.N, img, kmc_getbits(s, y.isr_off[r], y.BI)));
    }
    if (!kafka) return;
    kmc_orbits(t, y.nextrec_off, y.BNR, kmc_getbits(s, y.nextrec_off, y.BNR));
    kmc_orbits(t, y.nextep_off, y.BE, kmc_getbits(s, y.nextep_off, y.BE));
    kmc_orbits(t, y.qep_off, y.BE, kmc_getbits(s, y.qep_off, y.BE));
    const unsigned long long ql = kmc_getbits(s, y.qldr_off, y.BL);
    kmc_orbits(t, y.qldr_off, y.BL, ql == 0 || ql > (unsigned long long)y.N ? ql : (unsigned long long)img[ql - 1] + 1);
    kmc_orbits(t, y.qisr_off, y.BI, kmc_permute_mask(y.N, img, kmc_getbits(s, y.qisr_off, y.BI)));
    for (int e = 0; e <= y.E; ++e) {
        const unsigned long long rl = kmc_getbits(s, y.reqldr_off[e], y.BL);
        kmc_orbits(t, y.reqldr_off[e], y.BL, rl == 0 || rl > (unsigned long long)y.N ? rl : (unsigned long long)img[rl - 1] + 1);
        kmc_orbits(t, y.reqisr_off[e], y.BI, kmc_permute_mask(y.N, img, kmc_getbits(s, y.reqisr_off[e], y.BI)));
    }
}
// Up to this many replicas the representative of an orbit is its smallest image outright; beyond, the smallest among the
// images whose replica KEYS ascend with the position (kmc_device.h, KmcSymm::canon_sorted: 120 / 720 images per successor
// were what the orbit-counting search spent its time on)
// orbit counting up to this many replicas (7! - 1 = 5039 steps in the table of the walk through all images; 8! would be 40319)
#define KMC_SYMM_MAX_REPLICAS 7
#ifndef KMC_SYMM_UNROLLED_MAX   // (a JIT define moves the DEVICE's threshold only — timing runs; the host forms follow the default)
#define KMC_SYMM_UNROLLED_MAX 3
#endif
// The key of the replica at position r of t — everything about it that does not depend on how the replicas are named:
// *a = its log; *b = end | hw << BO | ep << 2 BO, then (from bit 2 BO + BE) 1 bit each: names itself as leader, holds itself
// in its ISR, names nobody, quorumState names it, quorumState's ISR holds it; 3 bits each: size of its ISR, how many OTHER
// replicas hold it in their ISR, how many name it as leader; then per LeaderAndIsr request e, 1 bit each: the request names
// it, the request's ISR holds it.  (The run-time-layout twin of KmcSymm::key_at.)
KMC_HD inline void kmc_replica_key_generic(const KmcLayout& y, const unsigned long long* t, int r, unsigned long long* a,
                                           unsigned long long* b) {
    *a = kmc_getbits(t, y.log_off[r], y.BR * y.L);
    *b = kmc_getbits(t, y.end_off[r], y.BO);
    if (y.model == KMC_MODEL_FINITE_REPLICATED_LOG) return;
    const int gb = 2 * y.BO + y.BE;
    *b |= kmc_getbits(t, y.hw_off[r], y.BO) << y.BO | kmc_getbits(t, y.ep_off[r], y.BE) << (2 * y.BO);
    const unsigned long long self = (unsigned long long)r + 1;
    const unsigned long long ldr = kmc_getbits(t, y.ldr_off[r], y.BL), isr = kmc_getbits(t, y.isr_off[r], y.BI);
    unsigned long long f = (ldr == self ? 1ull : 0ull) | (isr >> r & 1ull) << 1 | (ldr == 0 ? 1ull : 0ull) << 2 |
                           (kmc_getbits(t, y.qldr_off, y.BL) == self ? 1ull : 0ull) << 3 |
                           (kmc_getbits(t, y.qisr_off, y.BI) >> r & 1ull) << 4;
    unsigned long long pop = 0, held = 0, named = 0;
    for (int i = 0; i < y.N; ++i) pop += isr >> i & 1ull;
    for (int o = 0; o < y.N; ++o) {
        if (o == r) continue;
        held += kmc_getbits(t, y.isr_off[o], y.BI) >> r & 1ull;
        named += kmc_getbits(t, y.ldr_off[o], y.BL) == self ? 1ull : 0ull;
    }
    f |= pop << 5 | held << 8 | named << 11;
    for (int e = 0; e <= y.E; ++e) {
        f |= (kmc_getbits(t, y.reqldr_off[e], y.BL) == self ? 1ull : 0ull) << (14 + 2 * e);
        f |= (kmc_getbits(t, y.reqisr_off[e], y.BI) >> r & 1ull) << (15 + 2 * e);
    }
    *b |= f << gb;
}
// The representative of s's orbit: the smallest image under the N! permutations (beyond KMC_SYMM_UNROLLED_MAX replicas: among
// the images whose keys ascend with the position), states compared as the tuple (word 0, word 1, ...) of unsigned 64-bit
// values.  *stab = the permutations that fix s (the orbit has N! / *stab states).
KMC_HD inline void kmc_canonical_state_generic(const KmcLayout& y, const unsigned long long* s, unsigned long long* c, int* stab) {
    const int nf = kmc_factorial(y.N);
    unsigned long long t[KMC_MAXW];
    int img[KMC_MAXN];
    int cnt = 0;
    for (int P = 0; P < nf; ++P) {
        for (int r = 0; r < y.N; ++r) img[r] = kmc_perm_image(y.N, P, r);
        kmc_permute_state(y, img, s, t);
        if (y.N > KMC_SYMM_UNROLLED_MAX) {
            bool ascending = true;
            unsigned long long pa = 0, pb = 0;
            for (int r = 0; r < y.N && ascending; ++r) {
                unsigned long long a = 0, b = 0;
                kmc_replica_key_generic(y, t, r, &a, &b);
                if (r > 0 && (a < pa || (a == pa && b < pb))) ascending = false;
                pa = a; pb = b;
            }
            if (!ascending) continue;
        }
        int cmp = 0;   // t against c
        if (cnt == 0) cmp = -1;
        for (int k = 0; k < y.W && cmp == 0; ++k) cmp = t[k] < c[k] ? -1 : t[k] > c[k] ? 1 : 0;
        if (cmp < 0) {
            for (int k = 0; k < y.W; ++k) c[k] = t[k];
            cnt = 1;
        } else if (cmp == 0) {
            ++cnt;
        }
    }
    if (stab) *stab = cnt;
}
