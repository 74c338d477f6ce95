// kmc_symm.h — symmetry reduction with orbit counting: orbit representatives and stabilisers under the permutations of Replicas.
// Part of the device source (kmc_device.h lists the parts; the host engine hands their concatenation to hiprtc).
#pragma once
#include "kmc_common.h"
// ========================================================================================
// Symmetry reduction with orbit counting (kmc_config.symmetry; the kernels use it in KMC_SYMM builds)
// ========================================================================================
// The specs quantify over Replicas and never tell two of them apart (KafkaReplication.tla:109-120, :158-310; the five
// modules' own actions: `\E leader, replica \in Replicas`), so the N! permutations of Replicas are automorphisms of the state
// graph: they map Init to Init, successors to successors (binding by binding, so also the per-disjunct "generated" counts
// and the doubly satisfied disjuncts), and keep every invariant and the BFS depth.  The search therefore only stores and
// expands ONE state per orbit — the smallest image under the N! permutations, words compared in order — and every count it
// reports is weighted by the orbit's size N! / |stabiliser|: distinct states, states per level, generated per disjunct,
// deadlocks and violating states all come out as the numbers of the plain search (and of TLC without SYMMETRY), from
// ~1/N! of the probes.  (TLC's own SYMMETRY reports the REDUCED counts — SURVEY.md rules that out; this does not change them.)
// permute<P> is the compile-time form of kmc_permute_state (kmc_layout.h): fields move between compile-time offsets.  The
// replica ids inside a state sit in (leader, isr) PAIRS — one per replica, one in quorumState, one per LeaderAndIsr request
// — and a pair's images under ALL the permutations come from one table lookup (LDS in the kernels): entry idx = leader |
// isr << BL holds the renamed pair for PER permutations per 32-bit word (all six at three replicas), so that a permutation
// costs one bit-field extract and one insert per pair instead of two shift-indexed constant lookups each.
template <class M> struct KmcSymm {
    static constexpr KmcLayout Y = M::Y;
    static constexpr int N = Y.N, W = Y.W;
    static constexpr bool KAFKA = Y.model != KMC_MODEL_FINITE_REPLICATED_LOG;
    static constexpr int NFACT = kmc_factorial(N);
    static_assert(kmc_model_symmetric(Y.model), "this model singles out a replica: no symmetry reduction");
    static_assert(N <= KMC_SYMM_MAX_REPLICAS, "orbit counting: the walk through all images (canon_sorted's last resort) is a table of N! - 1 steps");
    static constexpr bool UNROLLED = N <= KMC_SYMM_UNROLLED_MAX;   // N! - 1 statically specialised permutations; beyond: the sorted images (canon_sorted)
    static constexpr int PB = KAFKA ? Y.BL + Y.BI : 1;       // bits of a (leader, isr) pair: 5 at N = 3, 7 at N = 4
    static constexpr int PER = 32 / PB;                      // images per table word
    static constexpr int NG = (NFACT + PER - 1) / PER;       // table words per pair value
    // (five and six replicas: the table holds, per adjacent transposition a <-> a + 1, the image of every pair value)
    static constexpr int TABLE_WORDS = !KAFKA ? 1 : UNROLLED ? (NG << PB) : ((N - 1) << PB);
    static constexpr int NPAIR = KAFKA ? N + 1 + (Y.E + 1) : 0;
    static constexpr u32 MP = (1u << PB) - 1, ML = (1u << Y.BL) - 1;

    // the pair idx with every replica in it renamed by permutation P (leader: 0 = None or index + 1; isr: a bit mask)
    static KMC_HD constexpr u32 pair_image(int P, u32 idx) {
        const u32 l = idx & ML, m = idx >> Y.BL;
        const u32 pl = (l == 0 || l > (u32)N) ? l : (u32)kmc_perm_image(N, P, (int)l - 1) + 1;
        u32 pm = 0;
        for (int i = 0; i < N; ++i)
            if (m >> i & 1u) pm |= 1u << kmc_perm_image(N, P, i);
        return pl | pm << Y.BL;
    }
    // the pair idx with the names of replicas a and a + 1 exchanged
    static KMC_HD constexpr u32 exchange_image(int a, u32 idx) {
        const u32 l = idx & ML, m = idx >> Y.BL;
        const u32 pl = l == (u32)a + 1 ? l + 1 : l == (u32)a + 2 ? l - 1 : l;
        const u32 y = ((m >> a) ^ (m >> (a + 1))) & 1u;
        return pl | ((m ^ (y << a) ^ (y << (a + 1))) << Y.BL);
    }
    // word i of the table: the images of pair (i & MP) under permutations (i >> PB) * PER ... + PER - 1, PB bits each
    static KMC_HD constexpr u32 table_word(int i) {
        const u32 idx = (u32)i & MP;
        const int g = i >> PB;
        u32 w = 0;
        for (int j = 0; j < PER; ++j)
            if (g * PER + j < NFACT) w |= pair_image(g * PER + j, idx) << (j * PB);
        return w;
    }
    struct Table { u32 w[TABLE_WORDS]; };
    static constexpr Table make_table() {
        Table t{};
        for (int i = 0; i < TABLE_WORDS; ++i) t.w[i] = !KAFKA ? 0u : UNROLLED ? table_word(i) : exchange_image(i >> PB, (u32)i & MP);
        return t;
    }
    static constexpr Table TABLE = make_table();   // (constant memory; k_expand copies it to LDS once per block)
    // pair f: replica f for f < N, quorumState for f = N, the request of epoch f - N - 1 beyond
    static constexpr int pair_ldr_off(int f) { return f < N ? Y.ldr_off[f] : f == N ? Y.qldr_off : Y.reqldr_off[f - N - 1]; }
    static constexpr int pair_isr_off(int f) { return f < N ? Y.isr_off[f] : f == N ? Y.qisr_off : Y.reqisr_off[f - N - 1]; }
    static constexpr bool pair_adjacent(int f) { return pair_isr_off(f) == pair_ldr_off(f) + Y.BL; }
    // the global fields no permutation touches (nextRecordId, nextLeaderEpoch, quorumState.leaderEpoch), as a mask of word k
    static constexpr u64 keep_mask(int k) {
        u64 m = 0;
        if (!KAFKA) return m;
        const int off[3] = {Y.nextrec_off, Y.nextep_off, Y.qep_off}, bits[3] = {Y.BNR, Y.BE, Y.BE};
        for (int f = 0; f < 3; ++f)
            for (int b = off[f]; b < off[f] + bits[f]; ++b)
                if ((b >> 6) == k) m |= 1ull << (b & 63);
        return m;
    }

    // the images of a state's pairs under every permutation: NPAIR x NG table reads, once per state
    struct Prep { u32 img[NPAIR > 0 ? NPAIR : 1][NG]; };
    static KMC_DEV void prepare(const u64* s, const u32* tab, Prep& p) {
        kmc_static_for<0, NPAIR>([&](auto FF) {
            constexpr int f = decltype(FF)::value;
            u32 idx;
            if constexpr (pair_adjacent(f)) idx = (u32)kmc_getbits(s, pair_ldr_off(f), PB);
            else idx = (u32)kmc_getbits(s, pair_ldr_off(f), Y.BL) | ((u32)kmc_getbits(s, pair_isr_off(f), Y.BI) << Y.BL);
#pragma unroll
            for (int g = 0; g < NG; ++g) p.img[f][g] = tab[(g << PB) | idx];
        });
    }
    template <int P> static KMC_DEV void permute(const u64* s, const Prep& p, u64* t) {
        kmc_static_for<0, W>([&](auto KK) {
            constexpr int k = decltype(KK)::value;
            t[k] = s[k] & keep_mask(k);
        });
        kmc_static_for<0, N>([&](auto RR) {
            constexpr int r = decltype(RR)::value, d = kmc_perm_image(N, P, r);
            kmc_orbits(t, Y.log_off[d], Y.BR * Y.L, kmc_getbits(s, Y.log_off[r], Y.BR * Y.L));
            if constexpr (!KAFKA) {
                kmc_orbits(t, Y.end_off[d], Y.BO, kmc_getbits(s, Y.end_off[r], Y.BO));
            } else {
                // end | hw | ep | ldr | isr are adjacent in every arrangement of the state vector (kmc_layout.h)
                constexpr int GB = 2 * Y.BO + Y.BE;
                static_assert(Y.hw_off[r] == Y.end_off[r] + Y.BO && Y.ep_off[r] == Y.hw_off[r] + Y.BO &&
                              Y.ldr_off[r] == Y.ep_off[r] + Y.BE && Y.isr_off[r] == Y.ldr_off[r] + Y.BL, "small group not contiguous");
                kmc_orbits(t, Y.end_off[d], GB, kmc_getbits(s, Y.end_off[r], GB));
                kmc_orbits(t, Y.ldr_off[d], PB, (p.img[r][P / PER] >> ((P % PER) * PB)) & MP);
            }
        });
        kmc_static_for<N, NPAIR>([&](auto FF) {
            constexpr int f = decltype(FF)::value;
            const u32 pi = (p.img[f][P / PER] >> ((P % PER) * PB)) & MP;
            if constexpr (pair_adjacent(f)) {
                kmc_orbits(t, pair_ldr_off(f), PB, pi);
            } else {
                kmc_orbits(t, pair_ldr_off(f), Y.BL, pi & ML);
                kmc_orbits(t, pair_isr_off(f), Y.BI, pi >> Y.BL);
            }
        });
    }
    // ---- five and six replicas: 119 / 719 statically specialised permutations are too much code, so the images are visited
    // one ADJACENT TRANSPOSITION at a time (Steinhaus-Johnson-Trotter: every permutation exactly once, consecutive ones differ
    // by exchanging two neighbouring replicas): a wave-uniform loop whose body dispatches to one of N - 1 specialised
    // "exchange replicas a and a + 1" steps working in place on the current image.
    static constexpr int NSTEPS = NFACT - 1;
    // at[k] = a: step k exchanges the replicas at positions a and a + 1; cross[k]: bit b set when the arrangement after step k
    // has moved some replica across the boundary between positions b and b + 1 (it then mixes two runs of the sorted order)
    struct Seq { unsigned char at[NSTEPS > 0 ? NSTEPS : 1], cross[NSTEPS > 0 ? NSTEPS : 1]; };
    static constexpr Seq make_sequence() {
        Seq q{};
        int perm[KMC_MAXN] = {}, dir[KMC_MAXN] = {};
        for (int i = 0; i < N; ++i) { perm[i] = i; dir[i] = -1; }
        for (int k = 0; k < NSTEPS; ++k) {
            int mp = -1, mv = -1;    // the largest element that can move in its direction past a smaller one
            for (int pos = 0; pos < N; ++pos) {
                const int v = perm[pos], np = pos + dir[v];
                if (np >= 0 && np < N && perm[np] < v && v > mv) { mv = v; mp = pos; }
            }
            const int np = mp + dir[mv];
            q.at[k] = (unsigned char)(mp < np ? mp : np);
            const int x = perm[mp]; perm[mp] = perm[np]; perm[np] = x;
            for (int v = mv + 1; v < N; ++v) dir[v] = -dir[v];
            unsigned c = 0;
            for (int b = 0; b + 1 < N; ++b)
                for (int pos = 0; pos <= b; ++pos)
                    if (perm[pos] > b) c |= 1u << b;
            q.cross[k] = (unsigned char)c;
        }
        return q;
    }
    static constexpr Seq SEQUENCE = make_sequence();
    // the image of t under the transposition of replicas A and A + 1, in place; `on` = 0 leaves this lane's t as it is
    // (everything below is an XOR of a difference: a lane that does not take part XORs zeros)
    template <int A> static KMC_DEV void exchange(u64* t, const u32* tab, u64 on = ~0ull) {
        constexpr int B = A + 1, LB = Y.BR * Y.L;
        {
            const u64 d = (kmc_getbits(t, Y.log_off[A], LB) ^ kmc_getbits(t, Y.log_off[B], LB)) & on;
            kmc_xorbits(t, Y.log_off[A], LB, d);
            kmc_xorbits(t, Y.log_off[B], LB, d);
        }
        if constexpr (!KAFKA) {
            const u64 d = (kmc_getbits(t, Y.end_off[A], Y.BO) ^ kmc_getbits(t, Y.end_off[B], Y.BO)) & on;
            kmc_xorbits(t, Y.end_off[A], Y.BO, d);
            kmc_xorbits(t, Y.end_off[B], Y.BO, d);
        } else {
            constexpr int SB = 2 * Y.BO + Y.BE + Y.BL + Y.BI;   // end | hw | ep | ldr | isr, adjacent in every arrangement
            static_assert(Y.isr_off[A] == Y.end_off[A] + SB - Y.BI && Y.isr_off[B] == Y.end_off[B] + SB - Y.BI, "small group not contiguous");
            const u64 d = (kmc_getbits(t, Y.end_off[A], SB) ^ kmc_getbits(t, Y.end_off[B], SB)) & on;
            kmc_xorbits(t, Y.end_off[A], SB, d);
            kmc_xorbits(t, Y.end_off[B], SB, d);
            // ... and the two names trade places in every (leader, isr) pair (leader values A + 1 <-> B + 1, isr bits A <-> B):
            // one table read per pair
            kmc_static_for<0, NPAIR>([&](auto FF) {
                constexpr int f = decltype(FF)::value;
                if constexpr (pair_adjacent(f)) {
                    const u32 idx = (u32)kmc_getbits(t, pair_ldr_off(f), PB);
                    kmc_xorbits(t, pair_ldr_off(f), PB, (idx ^ tab[(A << PB) | idx]) & (u32)on);
                } else {
                    const u32 idx = (u32)kmc_getbits(t, pair_ldr_off(f), Y.BL) | ((u32)kmc_getbits(t, pair_isr_off(f), Y.BI) << Y.BL);
                    const u32 x = (idx ^ tab[(A << PB) | idx]) & (u32)on;
                    kmc_xorbits(t, pair_ldr_off(f), Y.BL, x & ML);
                    kmc_xorbits(t, pair_isr_off(f), Y.BI, x >> Y.BL);
                }
            });
        }
    }
    // visits every image of s: MINIMISE keeps the smallest in c and counts how often it occurs (= the stabiliser's order);
    // otherwise c stays s and the images equal to s are counted.  `runs`: bit b set = the replicas at positions b and b + 1 of
    // s may trade places (canon_sorted: their keys are equal); an image whose arrangement crosses any other boundary is passed
    // over.  All ones: every image counts.
    template <bool MINIMISE> static KMC_DEV u32 walk(const u64* s, const u32* tab, u64* c, u32 runs = ~0u) {
        u64 t[W];
#pragma unroll
        for (int k = 0; k < W; ++k) { t[k] = s[k]; c[k] = s[k]; }
        u32 n = 1;
#pragma clang loop unroll(disable)
        for (int k = 0; k < NSTEPS; ++k) {
            int a = SEQUENCE.at[k];
            u32 crossed = SEQUENCE.cross[k];
#ifndef KMC_HOST_EMU
            a = __builtin_amdgcn_readfirstlane(a);   // the step index is wave-uniform: keep the dispatch scalar
            crossed = __builtin_amdgcn_readfirstlane(crossed);
#endif
            kmc_dispatch<0, (N > 1 ? N - 1 : 1)>(a, [&](auto AA) { exchange<decltype(AA)::value>(t, tab); });
            const bool counts = (crossed & ~runs) == 0;
            bool lt = false, eq = true;
#pragma unroll
            for (int q = 0; q < W; ++q) {
                lt = lt || (eq && t[q] < c[q]);
                eq = eq && t[q] == c[q];
            }
            lt = lt && counts;
            eq = eq && counts;
            if constexpr (MINIMISE) {
                n = lt ? 1u : n + (eq ? 1u : 0u);
#pragma unroll
                for (int q = 0; q < W; ++q) c[q] = lt ? t[q] : c[q];
            } else {
                n += eq ? 1u : 0u;
            }
        }
        return n;
    }

    // ---- five and six replicas: the representative among the SORTED images -----------------------------------------------
    // Walking through all 120 / 720 images of every successor is what the orbit-counting search spent its time on at five
    // brokers.  So the representative is chosen among far fewer: every replica gets a KEY that does not depend on how the
    // replicas are named — its log, end offset, high watermark and epoch; whether it names itself / nobody as leader, whether
    // its ISR holds itself and how many it holds; whether quorumState and each LeaderAndIsr request name it as leader / in the
    // ISR; how many OTHER replicas hold it in their ISR / name it as leader — and the representative of an orbit is its
    // smallest image (words in order, as before) AMONG THE IMAGES WHOSE KEYS ASCEND WITH THE POSITION.  Renaming permutes the
    // keys with the replicas, so every state of an orbit sees the same set of sorted images: a representative all the same.
    //   1. sort: an odd-even transposition network of N (N - 1) / 2 conditional exchanges of neighbours (exchange<A> with a
    //      lane mask; the keys trade places with the replicas);
    //   2. where neighbours' keys are equal, the sorted image is one of several.  Almost always exchanging such neighbours
    //      gives the SAME state (two followers nobody tells apart): when that holds at every tied boundary, the tied runs
    //      generate the stabiliser — a permutation that fixes the state keeps every key where it is — the sorted image is
    //      unique, and |Stab| = the product of the run lengths' factorials;
    //   3. otherwise (a tie between replicas that ARE told apart by something the key does not see: not met in 1.8 M
    //      successors of BASELINE config 4 and of the headline, tools/tie_stats.py — but nothing rests on that) the wave walks
    //      through all the images of the sorted one and keeps the smallest of those that only move replicas inside tied runs.
    struct Key { u64 a, b; };
    template <int r> static KMC_DEV Key key_at(const u64* t) {
        Key k;
        k.a = kmc_getbits(t, Y.log_off[r], Y.BR * Y.L);
        if constexpr (!KAFKA) {
            k.b = kmc_getbits(t, Y.end_off[r], Y.BO);
        } else {
            constexpr int GB = 2 * Y.BO + Y.BE;   // end | hw | ep: adjacent in every arrangement (static_assert in exchange)
            static_assert(GB + 14 + 2 * (Y.E + 1) <= 64, "replica key does not fit 64 bits");
            const u32 ldr = (u32)kmc_getbits(t, Y.ldr_off[r], Y.BL), isr = (u32)kmc_getbits(t, Y.isr_off[r], Y.BI);
            const u32 qldr = (u32)kmc_getbits(t, Y.qldr_off, Y.BL), qisr = (u32)kmc_getbits(t, Y.qisr_off, Y.BI);
            u32 f = (ldr == (u32)r + 1u ? 1u : 0u) | ((isr >> r & 1u) << 1) | ((ldr == 0u ? 1u : 0u) << 2) |
                    ((qldr == (u32)r + 1u ? 1u : 0u) << 3) | ((qisr >> r & 1u) << 4) | ((u32)__builtin_popcount(isr) << 5);
            u32 held = 0, named = 0;   // by the other replicas
            kmc_static_for<0, N>([&](auto OO) {
                constexpr int o = decltype(OO)::value;
                if constexpr (o != r) {
                    held += (u32)kmc_getbits(t, Y.isr_off[o], Y.BI) >> r & 1u;
                    named += (u32)kmc_getbits(t, Y.ldr_off[o], Y.BL) == (u32)r + 1u ? 1u : 0u;
                }
            });
            f |= held << 8 | named << 11;
            kmc_static_for<0, Y.E + 1>([&](auto EE) {
                constexpr int e = decltype(EE)::value;
                f |= ((u32)kmc_getbits(t, Y.reqldr_off[e], Y.BL) == (u32)r + 1u ? 1u : 0u) << (14 + 2 * e);
                f |= ((u32)kmc_getbits(t, Y.reqisr_off[e], Y.BI) >> r & 1u) << (15 + 2 * e);
            });
            k.b = kmc_getbits(t, Y.end_off[r], GB) | (u64)f << GB;
        }
        return k;
    }
    // t = s with replica r renamed rank[r] (a run-time permutation, per lane): every replica's log and small fields move to the
    // position of its rank — destination-major select chains over compile-time offsets, never an indexed array — and every
    // (leader, isr) pair is renamed through the ranks (the leader by a packed 3-bit lookup, the isr bit by bit).
    static KMC_DEV void permute_by_rank(const u64* s, const u32* rank, u64* t) {
        constexpr int LB = Y.BR * Y.L;
        constexpr int GB = KAFKA ? 2 * Y.BO + Y.BE : Y.BO;          // end | hw | ep (FiniteReplicatedLog: end)
        static_assert(GB + PB <= 32, "a replica's small fields do not fit one register");
        kmc_static_for<0, W>([&](auto KK) {
            constexpr int k = decltype(KK)::value;
            t[k] = s[k] & keep_mask(k);
        });
        u64 lg[N];
        u32 sm[N];   // end | hw | ep, then the replica's own (leader, isr) pair from bit GB
        kmc_static_for<0, N>([&](auto RR) {
            constexpr int r = decltype(RR)::value;
            lg[r] = kmc_getbits(s, Y.log_off[r], LB);
            sm[r] = (u32)kmc_getbits(s, Y.end_off[r], GB);
            if constexpr (KAFKA) sm[r] |= (u32)kmc_getbits(s, Y.ldr_off[r], PB) << GB;   // (adjacent: static_assert in exchange)
        });
        u32 packed = 0, bit[N];   // rank of replica i at bits 3 i; 1 << rank[i]
        kmc_static_for<0, N>([&](auto II) {
            constexpr int i = decltype(II)::value;
            packed |= rank[i] << (3 * i);
            bit[i] = 1u << rank[i];
        });
        auto rename = [&](u32 pair) -> u32 {   // (leader | isr << BL) with every replica in it renamed by its rank
            const u32 l = pair & ML, m = pair >> Y.BL;
            const u32 nl = (l == 0u || l > (u32)N) ? l : ((packed >> (3u * (l - 1u))) & 7u) + 1u;
            u32 nm = 0;
            kmc_static_for<0, N>([&](auto II) {
                constexpr int i = decltype(II)::value;
                nm += ((m >> i) & 1u) * bit[i];
            });
            return nl | nm << Y.BL;
        };
        kmc_static_for<0, N>([&](auto DD) {
            constexpr int d = decltype(DD)::value;
            // (every step opaque: the plain chain is recognised as lg[the r with rank d] — the arrays go to scratch memory and
            // every destination loads them back with a per-lane address, as KmcKafka::sel_word found in round 3)
            u32 llo = (u32)lg[0], lhi = (u32)(lg[0] >> 32), v = sm[0];
            kmc_static_for<1, N>([&](auto RR) {
                constexpr int r = decltype(RR)::value;
                const bool here = rank[r] == (u32)d;
                llo = here ? (u32)lg[r] : llo;
                if constexpr (LB > 32) lhi = here ? (u32)(lg[r] >> 32) : lhi;
                v = here ? sm[r] : v;
                KMC_OPAQUE_PURE(llo);
                if constexpr (LB > 32) KMC_OPAQUE_PURE(lhi);
                KMC_OPAQUE_PURE(v);
            });
            const u64 l = LB > 32 ? (((u64)lhi << 32) | llo) : (u64)llo;
            kmc_orbits(t, Y.log_off[d], LB, l);
            kmc_orbits(t, Y.end_off[d], GB, v & ((1u << GB) - 1u));
            if constexpr (KAFKA) kmc_orbits(t, Y.ldr_off[d], PB, rename(v >> GB));
        });
        if constexpr (KAFKA) {
            kmc_static_for<N, NPAIR>([&](auto FF) {
                constexpr int f = decltype(FF)::value;
                if constexpr (pair_adjacent(f)) {
                    kmc_orbits(t, pair_ldr_off(f), PB, rename((u32)kmc_getbits(s, pair_ldr_off(f), PB)));
                } else {
                    const u32 pi = rename((u32)kmc_getbits(s, pair_ldr_off(f), Y.BL) | ((u32)kmc_getbits(s, pair_isr_off(f), Y.BI) << Y.BL));
                    kmc_orbits(t, pair_ldr_off(f), Y.BL, pi & ML);
                    kmc_orbits(t, pair_isr_off(f), Y.BI, pi >> Y.BL);
                }
            });
        }
    }

    // Is exchanging the replicas at positions A and A + 1 of t the identity on t, GIVEN that their keys are equal (so their
    // logs, end offsets, high watermarks and epochs are)?  Then what is left to agree is the naming: every (leader, isr) pair
    // of the OTHER replicas, of quorumState and of the requests must be unchanged by the renaming A <-> A + 1, and replica A's
    // own pair must be replica A + 1's renamed (the renaming is an involution: the converse follows).  One table read per pair,
    // no image built (until round 5 the exchanged state was built and compared word by word: twice the instructions).
    template <int A> static KMC_DEV bool trade_is_identity(const u64* t, const u32* tab) {
        if constexpr (!KAFKA) {
            return true;   // FiniteReplicatedLog: a replica is its log and end offset — the key
        } else {
            auto pair_of = [&](auto FF) -> u32 {
                constexpr int f = decltype(FF)::value;
                if constexpr (pair_adjacent(f)) return (u32)kmc_getbits(t, pair_ldr_off(f), PB);
                else return (u32)kmc_getbits(t, pair_ldr_off(f), Y.BL) | ((u32)kmc_getbits(t, pair_isr_off(f), Y.BI) << Y.BL);
            };
            u32 changed = 0;
            kmc_static_for<0, NPAIR>([&](auto FF) {
                constexpr int f = decltype(FF)::value;
                if constexpr (f != A && f != A + 1) {
                    const u32 idx = pair_of(FF);
                    changed |= idx ^ tab[(A << PB) | idx];
                }
            });
            const u32 pa = pair_of(KmcIC<A>{}), pb = pair_of(KmcIC<A + 1>{});
            changed |= pa ^ tab[(A << PB) | pb];
            return changed == 0;
        }
    }

    static KMC_DEV void canon_sorted(const u64* s, const u32* tab, u64* c, u32& stab) {
        u64 t[W];
        Key key[N];
        kmc_static_for<0, N>([&](auto RR) { key[decltype(RR)::value] = key_at<decltype(RR)::value>(s); });
        // 1. a sorted image.  Round 5: the RANK of every replica among the keys (N (N - 1) / 2 comparisons; equal keys keep their
        //    order) and ONE run-time permutation (permute_by_rank), instead of an odd-even transposition network of
        //    N (N - 1) / 2 masked exchanges each renaming every pair through the table — at seven replicas ~750 vector
        //    instructions where the network took ~2,100 (profiles/r05_orbit_counting.txt).  Any sorted image serves: where
        //    tied neighbours are interchangeable all of them are the same state, and where they are not the walk below
        //    goes through every arrangement of the tied runs whichever it starts from.
        u32 rank[N], runs = 0;
        kmc_static_for<0, N>([&](auto RR) { rank[decltype(RR)::value] = 0; });
        u64 equal = 0;   // bit a * N + b: keys of a < b equal
        kmc_static_for<0, N>([&](auto AA) {
            kmc_static_for<decltype(AA)::value + 1, N>([&](auto BB) {
                constexpr int a = decltype(AA)::value, b = decltype(BB)::value;
                const bool eq = key[a].a == key[b].a && key[a].b == key[b].b;
                const bool b_first = key[b].a < key[a].a || (key[b].a == key[a].a && key[b].b < key[a].b);
                rank[a] += b_first ? 1u : 0u;
                rank[b] += b_first ? 0u : 1u;
                equal |= (u64)(eq ? 1u : 0u) << (a * N + b);
            });
        });
        permute_by_rank(s, rank, t);
        // 2. ties: equal keys stand next to each other — boundary rank[a] is tied when some b > a with the same key follows directly
        kmc_static_for<0, N>([&](auto AA) {
            kmc_static_for<decltype(AA)::value + 1, N>([&](auto BB) {
                constexpr int a = decltype(AA)::value, b = decltype(BB)::value;
                const bool eq = (equal >> (a * N + b)) & 1ull;
                runs |= (eq && rank[b] == rank[a] + 1u) ? 1u << rank[a] : 0u;
            });
        });
        // tied neighbours: is trading them the identity on t?
        u32 told_apart = 0;
        kmc_static_for<0, N - 1>([&](auto AA) {
            constexpr int a = decltype(AA)::value;
            const bool tie = (runs >> a) & 1u;
            if (kmc_any_lane(tie)) told_apart |= (tie && !trade_is_identity<a>(t, tab)) ? 1u : 0u;
        });
        if (kmc_any_lane(told_apart != 0)) {   // (every lane takes the walk's answer: where nothing is told apart it is the same)
            stab = walk<true>(t, tab, c, runs);
            return;
        }
#pragma unroll
        for (int k = 0; k < W; ++k) c[k] = t[k];
        u32 n = 1, len = 1;
#pragma unroll
        for (int a = 0; a + 1 < N; ++a) {
            len = (runs >> a & 1u) ? len + 1u : 1u;
            n *= len;
        }
        stab = n;
    }

    // c = the orbit's representative (the smallest image, word 0 first; beyond KMC_SYMM_UNROLLED_MAX replicas among the sorted
    // images), stab = the permutations that fix s
    // (the unrolled forms sit in `else` branches: N! - 1 instantiations of permute<P> must not even be attempted at 7 replicas)
    static KMC_DEV void canon(const u64* s, const u32* tab, u64* c, u32& stab) {
        if constexpr (!UNROLLED) {
            canon_sorted(s, tab, c, stab);
        } else {
            Prep p;
            prepare(s, tab, p);
#pragma unroll
            for (int k = 0; k < W; ++k) c[k] = s[k];
            u32 n = 1;
            kmc_static_for<1, NFACT>([&](auto PP) {
                u64 t[W];
                permute<decltype(PP)::value>(s, p, t);
                bool lt = false, eq = true;
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    lt = lt || (eq && t[k] < c[k]);
                    eq = eq && t[k] == c[k];
                }
                n = lt ? 1u : n + (eq ? 1u : 0u);
#pragma unroll
                for (int k = 0; k < W; ++k) c[k] = lt ? t[k] : c[k];
            });
            stab = n;
        }
    }
    static KMC_DEV u32 stabiliser(const u64* s, const u32* tab) {
        if constexpr (!UNROLLED) {
            u64 c[W];
            return walk<false>(s, tab, c);
        } else {
            Prep p;
            prepare(s, tab, p);
            u32 n = 1;
            kmc_static_for<1, NFACT>([&](auto PP) {
                u64 t[W];
                permute<decltype(PP)::value>(s, p, t);
                bool eq = true;
#pragma unroll
                for (int k = 0; k < W; ++k) eq = eq && t[k] == s[k];
                n += eq ? 1u : 0u;
            });
            return n;
        }
    }
    // The same number for k_insert's records (Init, and every record a shard receives).  Beyond KMC_SYMM_UNROLLED_MAX replicas
    // stabiliser() walks through all N! images — 5,039 steps at seven replicas, for every 64 records: the sorted form gives the
    // order of the stabiliser as the product of its tied runs' factorials for ~2 k instructions (canon_sorted; the walk only
    // where tied replicas are told apart).  stabiliser() stays what the host emulation holds canon()'s count to.
    static KMC_DEV u32 stabiliser_of_record(const u64* s, const u32* tab) {
        if constexpr (!UNROLLED) {
            u64 c[W];
            u32 st;
            canon_sorted(s, tab, c, st);
            return st;
        } else {
            return stabiliser(s, tab);
        }
    }
    // what a state of stabiliser order `stab` lacks to a full orbit: N! - N!/stab (0 for almost every state)
    static KMC_DEV u32 deficit(u32 stab) { return stab == 1 ? 0u : (u32)NFACT - (u32)NFACT / stab; }
};

