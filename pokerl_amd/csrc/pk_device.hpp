// pk_device.hpp -- gfx950 device code of the vectorised NLHE hot path: one table per lane, all per-table state in
// VGPRs (compile-time-indexed arrays, seat states as bitmasks), money in IEEE binary64 in the reference's operation
// order (translation unit is built with -ffp-contract=off).  Citations: paths relative to the reference root.
#pragma once
#ifndef PK_HOST_SIM  // tools/host_sim (dev-only CPU build of this header) supplies shims instead
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "../../include/pokerl_hip.h"

namespace pk {

// pokerl/enums.py:9-18, :104-114, :130-136
enum : int { HR_SF = 1, HR_POKER = 2, HR_FULL = 3, HR_FLUSH = 4, HR_STRAIGHT = 5, HR_TRIS = 6, HR_TWO_PAIR = 7, HR_PAIR = 8, HR_HIGH = 9, HR_NONE = 10 };
enum : int { MV_FOLD = 0, MV_CHECK = 1, MV_CALL = 2, MV_RAISE_ANY = 3, MV_ALL_IN = 6 };
enum : int { PS_FOLDED = 0, PS_ACTIVE = 1, PS_CALLED = 2, PS_ALL_IN = 3, PS_BROKEN = 4 };
constexpr uint32_t STREAM_DECK = 0x4445434Bu, STREAM_ACTION = 0x41435432u;  // 'DECK', 'ACT2' (RNG spec: DESIGN.md section 3)
constexpr uint32_t NONE_V = (uint32_t)HR_NONE << 20;  // ranking value (rank<<20 | kickers) of judger's (NONE, [])

constexpr int EVAL7_TAB_WORDS = 8192;   // entries of the table-driven evaluator's rank-mask table (eval7_tab_entry): 32 KB

// ---------------------------------------------------------------------------------------------- HBM layout
// Structure-of-arrays, seat-major: element (seat p, table t) of a per-seat array lives at [p*T + t], so the 64 lanes
// of a wave (64 consecutive tables) touch 64 consecutive 8-byte (or 4-byte) words per load: fully coalesced.
struct State {
    double *credits, *bets, *pending, *payoffs;  // [N][T]   Game.credits/.bets/.pending_bets/.payoffs (game.py:260-264)
    double *min_raise;                           // [T]      Game.minimum_raise_value (game.py:263)
    uint64_t *seat_states;                       // [T]      ACTIVE | CALLED<<16 | ALL_IN<<32 | BROKEN<<48 seat bitmasks (FOLDED = in none)
    uint32_t *cursors;                           // [T]      active | dealer<<4 | sb<<8 | bb<<12 | turn<<16 (game.py:251-258)
                                                 //          | step-in-flight bits 20..30 (Table::store), 0 for an idle table
    int32_t *hand;                               // [T]      Game.hand
    uint64_t *hand_serial, *step_serial;         // [T]      RNG-spec counters (64-bit: a table's streams never repeat)
    uint32_t *cards;                             // [W][T]   4 Card.value bytes per word, deck[0:5+2N] (game.py:385-395)
    uint32_t *show;                              // [N][T]   last showdown: HandRanking<<20 | kickers value
    uint32_t *owed;                              // [T]      Game.step()s requested by pk_rollout and not executed yet (deferred launches)
    uint32_t *mid;                               // [T]      hands rolled so far by a step that is in flight across launches
    uint64_t *env_ctx;                           // [T]      PokerGameEnv.step in flight across pk_env_step_async_d launches (0: none):
                                                 //          1 | phase<<1 | done<<4 | hand<<5 | terr<<8 | (budget+1)<<16 | (reset budget+1)<<32
    double *env_rew;                             // [T]      ... and its reward so far (game_env.py:34, :47)
    uint8_t *valid;                              // [T]      valid-action bitmask of the active player (game.py:339-383)
    uint8_t *terr;                               // [T]      PK_TERR_* of the last call
    unsigned long long *counters;                // [waves][PK_NUM_COUNTERS]: one slot per wavefront, no atomics (4 096
                                                 // contended atomicAdds per launch cost ~46 us); summed by k_sum_counters
    unsigned long long *prof;                    // [PF_SLOTS], diagnostic build only
    const uint32_t *evtab;                       // [EVAL7_TAB_WORDS] the table of eval7_tab (per device; k_rollout_tab stages it in LDS), or NULL
    double start_credits[PK_MAX_PLAYERS];
    double big_blind, small_blind;
    uint32_t key0, key1, table_id_base;
    int T;
};

// The few scalars the step machine needs inside its loop, passed BY VALUE (kernel-argument SGPRs); the array bases of
// State are taken by pointer and s_load-ed only where a table is loaded / stored.  (All of State by value kept ~35 SGPR
// pairs live across the loop and spilled 66 SGPRs into VGPR lanes; all of it by pointer made the compiler re-issue
// scalar loads inside the loop.)
// A table right after Game.reset(dealer = 0) minus the deal: it depends on the configuration only (seats, start
// credits, blinds), so the auto-reset of a finished game inside k_rollout is a copy of these constants (computed once
// per handle by k_make_fresh with the ordinary reset_state) instead of a second inlined copy of setup_hand.
struct Fresh {
    double credits[PK_MAX_PLAYERS], pending[PK_MAX_PLAYERS];
    double min_raise;
    uint32_t st_active, st_called, st_allin, st_broken;
    int active, dealer, sb, bb;
};

struct Hot {
    const Fresh *fresh;           // device memory
    double big_blind, small_blind;
    const double *start_credits;  // [N], device memory: per-seat start credits (read on Game.reset only)
    double start_uniform;         // the common case, every seat starts with the same credits: no memory read at all
    int start_is_uniform;
    uint32_t *show;               // State::show
    uint32_t key0, key1, table_id_base;
    int T;
    unsigned long long *prof;  // State::prof (diagnostic builds only)
    int tpb;  // tables per wavefront (= per workgroup): 64 when the batch fills the chip, fewer for small batches so that
              // every SIMD gets a wave (a wave-step costs the same however many lanes are live, idle SIMDs cost nothing)
};

// ---------------------------------------------------------------------------------------------- helpers
// Compile-time expansion of `for p in 0..N-1`: every state-array subscript is a constant when the IR is first built, so
// the whole table is scalarised into VGPRs by the first SROA pass.  (With `#pragma unroll` loops the arrays are still
// loop-indexed allocas at that point, and LLVM later folds the select chains of sel()/put() back into dynamically
// indexed scratch loads -- the table then lives in scratch memory.)
template <typename F, int... I>
__device__ __forceinline__ void unroll_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void unroll(F &&f) { unroll_impl(f, std::make_integer_sequence<int, N>{}); }
#define PK_FOR(p, N) pk::unroll<N>([&](auto p##_c) { constexpr int p = decltype(p##_c)::value;
#define PK_END });
template <int N>
__device__ __forceinline__ double sel(const double (&a)[N], int i) {
    double r = a[0];
    PK_FOR(p, N) if (p > 0) r = (i == p) ? a[p] : r; PK_END
    return r;
}
template <int N>
__device__ __forceinline__ void put(double (&a)[N], int i, double v) {
    PK_FOR(p, N) a[p] = (i == p) ? v : a[p]; PK_END
}
template <int N>
__device__ __forceinline__ double vmax(const double (&a)[N]) {  // np.max
    double m = a[0];
    PK_FOR(p, N) if (p > 0) m = (a[p] > m) ? a[p] : m; PK_END
    return m;
}
// np.sum over contiguous f64[N] in numpy's association order (numpy's pairwise_sum for n <= 128): N<8 left to
// right; N>=8: eight partial sums r[j] = a[j] (+ a[8+j] + ... one whole block of eight at a time), combined
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the tail left to right -- one block up to 15 seats, two at 16.
template <int N>
__device__ __forceinline__ double np_sum(const double (&a)[N]) {
    if constexpr (N < 8) {
        double r = a[0];
        PK_FOR(p, N) if (p > 0) r = r + a[p]; PK_END
        return r;
    } else {
        constexpr int BLOCKS = N / 8;
        double r8[8];
        PK_FOR(j, 8) r8[j] = a[j]; PK_END
        PK_FOR(p, N) if constexpr (p >= 8 && p < 8 * BLOCKS) r8[p % 8] = r8[p % 8] + a[p]; PK_END
        double r = ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
        PK_FOR(p, N) if constexpr (p >= 8 * BLOCKS) r = r + a[p]; PK_END
        return r;
    }
}

// byte 0 of x in all four bytes (one v_perm_b32)
__device__ __forceinline__ uint32_t rep4(uint32_t x) {
#ifdef PK_HOST_SIM
    return (x & 0xffu) * 0x01010101u;
#else
    return __builtin_amdgcn_perm(x, x, 0u);
#endif
}
// A pointer that was LOADED from memory (the array bases behind `const State *Sp`) is a flat pointer to the compiler: every
// access through it becomes flat_load / flat_store (aperture check, counts on vmcnt AND lgkmcnt).  All table state lives in
// hipMalloc'ed memory, so say so: the round trip through address space 1 makes the accesses global_load /
// global_store.  (Pointers that arrive as kernel arguments, also inside by-value structs, are global already.)
#ifdef PK_HOST_SIM
#define PK_GLOBAL
#else
#define PK_GLOBAL __attribute__((address_space(1)))
#endif
template <typename T>
__device__ __forceinline__ PK_GLOBAL T *as_global(T *p) { return (PK_GLOBAL T *)p; }   // (keep the result's TYPE: `auto g = as_global(p)`;
                                                                                       //  cast back to a flat pointer and the optimiser folds the pair away)
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// ---------------------------------------------------------------------------------------------- judger.eval_hand
__device__ __forceinline__ void cex_desc(uint32_t &a, uint32_t &b) {  // compare-exchange, larger first
    uint32_t hi = a > b ? a : b, lo = a > b ? b : a;
    a = hi; b = lo;
}
__device__ __forceinline__ void sort7_desc(uint32_t (&k)[7]) {  // 16-comparator network (0/1-principle checked on host)
    cex_desc(k[0], k[6]); cex_desc(k[2], k[3]); cex_desc(k[4], k[5]);
    cex_desc(k[0], k[2]); cex_desc(k[1], k[4]); cex_desc(k[3], k[6]);
    cex_desc(k[0], k[1]); cex_desc(k[2], k[5]); cex_desc(k[3], k[4]);
    cex_desc(k[1], k[2]); cex_desc(k[4], k[6]);
    cex_desc(k[2], k[3]); cex_desc(k[4], k[5]);
    cex_desc(k[1], k[2]); cex_desc(k[3], k[4]); cex_desc(k[5], k[6]);
}

// General evaluator: 0..7 cards, multiset semantics, every quirk of pokerl/judger.py:7-99.
// c[i] = Card.value ((suit<<4)|rank0, cards.py:28-62) for i < n.  Returns HandRanking<<20 | get_kickers_value(kickers)
// (judger.py:101-109); nk = len(kickers).
__device__ inline uint32_t eval_hand(const uint32_t (&c)[7], int n, int &nk) {
    nk = 0;
    if (n == 0) return NONE_V;                                                     // :30
    uint32_t rk[7], sk[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        uint32_t r = c[i] & 0xf; r = r ? r : 13;                                   // cards.py:14
        uint32_t s = (c[i] >> 4) & 3;                                              // cards.py:20
        bool real = i < n;
        rk[i] = real ? r : 0;                                                      // pads sort last (real keys >= 1)
        sk[i] = real ? ((s << 4) | r) : 0;
    }
    if (n == 1) { nk = 1; return ((uint32_t)HR_HIGH << 20) | rk[0]; }             // :31
    if (n == 2) {                                                                  // :32-35
        if (rk[0] == rk[1]) { nk = 1; return ((uint32_t)HR_PAIR << 20) | rk[0]; }
        uint32_t hi = rk[0] > rk[1] ? rk[0] : rk[1], lo = rk[0] > rk[1] ? rk[1] : rk[0];
        nk = 2; return ((uint32_t)HR_HIGH << 20) | (hi << 4) | lo;
    }
    sort7_desc(rk);                                                                // :38 (ties: only .rank is read)
    sort7_desc(sk);                                                                // :39 (keys of distinct cards differ; equal keys are identical)
    uint32_t flush = 4, both = 4, kind = 0, straight = 0, flush_start = 0;         // :41-45
    uint32_t two = 0, three = 0, four = 0, n2 = 0, n3 = 0, n4 = 0;                 // lists as nibbles, append order
#pragma unroll
    for (int idx = 0; idx < 7; ++idx) {                                            // :50
        if (idx < n) {
            uint32_t rr = rk[idx], sr = sk[idx] & 0xf, ss = sk[idx] >> 4;
            if (ss == (flush & 0xf)) {                                             // :52
                flush += 0x100;
                if (sr + (both >> 8) == ((both >> 4) & 0xf)) both += 0x100;        // :56
                else both = 0x100 | (sr << 4) | ss;                                // :57
            } else if ((flush >> 8) < 5) { both = flush = 0x100 | (sr << 4) | ss; flush_start = idx; }  // :58
            if (rr == (kind & 0xf)) kind += 0x10;                                  // :61
            else {
                uint32_t numakind = kind >> 4, kr = kind & 0xf;                    // :64-67
                if (numakind == 2) { two |= kr << (4 * n2); ++n2; }
                else if (numakind == 3) { three |= kr << (4 * n3); ++n3; }
                else if (numakind == 4) { four |= kr << (4 * n4); ++n4; }
                kind = 0x10 | rr;                                                  // :70
                if (rr + (straight >> 4) == (straight & 0xf)) straight += 0x10;    // :71
                else if ((straight >> 4) < 5) straight = 0x10 | rr;                // :72
            }
        }
    }
    {
        uint32_t numakind = kind >> 4, kr = kind & 0xf;                            // :77-80
        if (numakind == 2) { two |= kr << (4 * n2); ++n2; }
        else if (numakind == 3) { three |= kr << (4 * n3); ++n3; }
        else if (numakind == 4) { four |= kr << (4 * n4); ++n4; }
    }
    if ((both >> 8) == 4 && ((both >> 4) & 0xf) == 4) {                            // :83-85  (CardRank.FIVE == 4)
        bool ace = false;
#pragma unroll
        for (int i = 0; i < 7; ++i) ace = ace || (i < n && sk[i] == (((both & 0xf) << 4) | 13));
        if (ace) { nk = 1; return ((uint32_t)HR_SF << 20) | 4; }
    } else if ((straight >> 4) == 4 && (straight & 0xf) == 4) {                    // :86-88
        if (rk[0] == 13) { nk = 1; return ((uint32_t)HR_STRAIGHT << 20) | 4; }     // any ace: it sorts first
    }
    auto others = [&](uint32_t ex0, uint32_t ex1, int count, uint32_t &kick, int &cnt) {
        int k = 0;                                                                 // islice(... if c.rank != ex ..., count)
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            bool take = i < n && rk[i] != ex0 && rk[i] != ex1 && k < count;
            kick = take ? ((kick << 4) | rk[i]) : kick;
            k += take;
        }
        cnt += k;
    };
    uint32_t kick = 0;
    if ((both >> 8) >= 5) { nk = 1; return ((uint32_t)HR_SF << 20) | ((both >> 4) & 0xf); }                 // :90
    if (n4) { kick = four & 0xf; nk = 1; others(four & 0xf, 99, 1, kick, nk); return ((uint32_t)HR_POKER << 20) | kick; }  // :91
    if (n3 > 1) { nk = 2; return ((uint32_t)HR_FULL << 20) | ((three & 0xf) << 4) | ((three >> 4) & 0xf); }  // :92
    if (n3 && n2) { nk = 2; return ((uint32_t)HR_FULL << 20) | ((three & 0xf) << 4) | (two & 0xf); }          // :93
    if ((flush >> 8) >= 5) {                                                                                  // :94
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            bool take = i < n && (uint32_t)i >= flush_start && (uint32_t)i < flush_start + 5;
            kick = take ? ((kick << 4) | (sk[i] & 0xf)) : kick;
            nk += take;
        }
        return ((uint32_t)HR_FLUSH << 20) | kick;
    }
    if ((straight >> 4) >= 5) { nk = 1; return ((uint32_t)HR_STRAIGHT << 20) | (straight & 0xf); }            // :95
    if (n3) { kick = three & 0xf; nk = 1; others(three & 0xf, 99, 2, kick, nk); return ((uint32_t)HR_TRIS << 20) | kick; }  // :96
    if (n2 > 1) {                                                                                             // :97
        uint32_t t0 = two & 0xf, t1 = (two >> 4) & 0xf;
        kick = (t0 << 4) | t1; nk = 2; others(t0, t1, 1, kick, nk);
        return ((uint32_t)HR_TWO_PAIR << 20) | kick;
    }
    if (n2) { kick = two & 0xf; nk = 1; others(two & 0xf, 99, 3, kick, nk); return ((uint32_t)HR_PAIR << 20) | kick; }  // :98
#pragma unroll
    for (int i = 0; i < 5; ++i) {                                                                             // :99
        bool take = i < n;
        kick = take ? ((kick << 4) | rk[i]) : kick;
        nk += take;
    }
    return ((uint32_t)HR_HIGH << 20) | kick;
}

// judger.compare_rankings (judger.py:111-158) over values v[p] = rank<<20|kick; returns the winners bitmask.
// The reference walks the list keeping best_rank / best_kicker / winners; its line 148 (`kicker = best_kicker`) never
// raises best_kicker, so best_kicker stays the kicker k0 of the FIRST hand holding the best rank:
//   a later hand with kicker > k0 replaces the winners list, one with kicker == k0 is appended.
// Closed form: E = {best rank, kicker == k0}, G = {best rank, kicker > k0};  G empty -> E;  else the LAST seat g of G
// plus the seats of E after g.
template <int N>
__device__ __forceinline__ uint32_t compare_rankings(const uint32_t (&v)[N], int &nw) {
    uint32_t best_rank = HR_NONE;
    PK_FOR(p, N) best_rank = min(best_rank, v[p] >> 20); PK_END                    // :140 (lower rank number wins)
    uint32_t k0 = 0; bool have = false;                                            // initial best_kicker = 0 (:135)
    PK_FOR(p, N)
        bool first = !have && (v[p] >> 20) == best_rank;
        k0 = first ? (v[p] & 0xFFFFF) : k0; have = have || first;
    PK_END
    uint32_t E = 0, G = 0;
    PK_FOR(p, N)
        bool br = (v[p] >> 20) == best_rank;
        uint32_t k = v[p] & 0xFFFFF;
        E |= (br && k == k0) ? (1u << p) : 0;
        G |= (br && k > k0) ? (1u << p) : 0;
    PK_END
    uint32_t g = 31 - __clz((int)G);
    uint32_t win = G ? ((1u << g) | (E & ~((2u << g) - 1))) : E;
    nw = __popc(win);
    return win;
}

// Fast evaluator for 7 DISTINCT cards (every in-game showdown hand): per-suit rank bitmasks + bit-parallel rank counts
// instead of the reference's sort-and-scan, reproducing each quirk of judger.py:7-99:
//   * `both` (straight-flush tracker) ends as the LOWEST run inside the `flush` suit group, no >=5 guard (:56-57);
//   * the `flush` group is the suit with >=5 cards, else the lowest suit present (:52-58);
//   * wheel checks are if/elif (:83-88): a 5-4-3-2 run in that group without its ace suppresses the plain wheel.
// Equality with the reference on all C(52,7) hands is a test (tests/test_hip_parity.py, eval7 digest).
// ONE body for the in-game evaluator (SEVEN: the hand holds seven cards, every kicker tail is complete -- eval7_distinct) and for
// n = 3 .. 7 DISTINCT cards with len(kickers) (eval_distinct_n, the register fast path of pk_eval_hands(_d) -- the reference's sort-and-scan,
// eval_hand above, ~570 executed instructions, is needed only for hands that repeat a card, which its own tests feed it and a game never
// does).  Same derivation for both (the scan of judger.py:50-99 visits fewer cards, its rules are the same); what changes with fewer
// cards is the LENGTH of the kicker lists -- `islice(others, count)` yields what is there (judger.py:91-99) -- and get_kickers_value
// (judger.py:101-109) packs the list as it is, so the tail takes min(nm, ranks left) nibbles.  Equal to eval_hand on EVERY 3-, 4-, 5-,
// 6- and 7-card subset of the deck (rank, kickers value and count: tools/host_sim `evaln`); round 6 folded the two former copies of this
// body into one template with the ISA of k_rollout<6> unchanged instruction for instruction.
// `bits`: OR of (1 << Card.value) over the hand: a card byte (suit<<4)|rank0 (cards.py:28-62) is already a bit index into a 64-bit
// word of four 16-bit suit lanes.
template <bool SEVEN>
__device__ __forceinline__ uint32_t eval_distinct_bits(uint64_t bits, int &nk) {
    // ace-high inside every lane at once (cards.py:14): rank0 0 (ace) -> bit 12, rank0 k -> bit k-1
    const uint64_t hi = ((bits >> 1) & 0x0fff0fff0fff0fffull) | ((bits & 0x0001000100010001ull) << 12);
    const uint32_t h01 = (uint32_t)hi, h23 = (uint32_t)(hi >> 32);
    const uint32_t sa = h01 & 0x1fff, sb = h01 >> 16, sc = h23 & 0x1fff, sd = h23 >> 16;
    const uint32_t um = sa | sb | sc | sd;
    const uint32_t s1 = sa ^ sb, c1 = sa & sb, s2 = sc ^ sd, c2 = sc & sd;      // per-rank count = bit0 + 2*t + 4*quads
    const uint32_t bit0 = s1 ^ s2, t = c1 ^ c2 ^ (s1 & s2), quads = c1 & c2;
    const uint32_t pairs = t & ~bit0, trips = t & bit0;
    // `flush` group (:52-58) and `both` = lowest run in it (:56-57)
    const bool fa = __popc(sa) >= 5, fb = __popc(sb) >= 5, fc = __popc(sc) >= 5, fd = __popc(sd) >= 5;
    const bool has_flush = fa || fb || fc || fd;
    uint32_t gm = sa ? sa : (sb ? sb : (sc ? sc : sd));                            // no flush: lowest suit present
    gm = fa ? sa : (fb ? sb : (fc ? sc : (fd ? sd : gm)));
    const uint32_t run = gm & ~(gm + (gm & (0u - gm)));
    const int bcount = __popc(run), btop = 31 - __clz((int)run);
    const uint32_t m5 = um & (um >> 1) & (um >> 2) & (um >> 3) & (um >> 4);        // bit i: ranks i+1..i+5 all present
    // Category cascade of :90-99 from the weakest to the strongest (later assignments override), as selects.  A hand is
    // [up to 2 ranks taken from the top of L] + [up to 5 ranks from the top of (base minus what was taken)], or a
    // straight-type hand with a single kicker `direct`.
    uint32_t cat = HR_HIGH, L = 0, base = um, direct = 0;
    int nl = 0, nm = 5;                                                            // :99
    if (pairs) { cat = HR_PAIR; L = pairs; nl = 1; nm = 3; }                       // :98
    if (pairs & (pairs - 1)) { cat = HR_TWO_PAIR; nl = 2; nm = 1; }                // :97
    if (trips) { cat = HR_TRIS; L = trips; nl = 1; nm = 2; }                       // :96
    if (m5) { cat = HR_STRAIGHT; direct = (uint32_t)(31 - __clz((int)m5) + 5); nl = 0; nm = 0; }  // :95
    if (has_flush) { cat = HR_FLUSH; L = 0; nl = 0; base = gm; nm = 5; direct = 0; }               // :94
    if (trips && pairs) { cat = HR_FULL; L = trips; nl = 1; base = pairs; nm = 1; direct = 0; }    // :93
    if (trips & (trips - 1)) { cat = HR_FULL; L = trips; nl = 2; nm = 0; direct = 0; }             // :92
    if (quads) { cat = HR_POKER; L = quads; nl = 1; base = um; nm = 1; direct = 0; }               // :91
    if (bcount >= 5) { cat = HR_SF; direct = (uint32_t)(btop + 1); nl = 0; nm = 0; }               // :90
    if (bcount == 4 && btop == 3) {                                                // :83-85 (if/elif: see header note)
        if (gm & (1u << 12)) { cat = HR_SF; direct = 4; nl = 0; nm = 0; }
    } else if (m5 == 0 && (um & 0x1f) == 0xf) {                                    // :86-88
        if (um & (1u << 12)) { cat = HR_STRAIGHT; direct = 4; nl = 0; nm = 0; }
    }
    uint32_t kick = direct, taken = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        uint32_t bit = 0x80000000u >> (__clz((int)L) & 31);  // (& 31: defined for an exhausted mask, whose bit is never used)
        bool take = i < nl;                                                        // nl > 0 implies L has that many bits
        kick = take ? ((kick << 4) | (uint32_t)(32 - __clz((int)L))) : kick;
        taken |= take ? bit : 0; L = take ? (L & ~bit) : L;
    }
    uint32_t m = base & ~taken;
    if constexpr (!SEVEN) {
        const int left = __popc(m);
        nm = nm < left ? nm : left;                                                // islice yields what is there
    }
    // the top five ranks of m unconditionally (an exhausted mask yields rank nibble 0), then the top nm of them by a shift
    uint32_t k5 = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const uint32_t lz = (uint32_t)__clz((int)m);
        k5 = (k5 << 4) | (32u - lz);
        m &= ~(0x80000000u >> (lz & 31));
    }
    kick = (kick << (4 * nm)) | (nm ? (k5 >> (4 * (5 - nm))) : 0u);
    nk = (direct ? 1 : 0) + nl + nm;
    return (cat << 20) | kick;
}
__device__ __forceinline__ uint32_t eval7_distinct(const uint32_t (&c)[7]) {
    uint64_t bits = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) bits |= 1ull << (c[i] & 63);
    int nk;
    return eval_distinct_bits<true>(bits, nk);
}
__device__ __forceinline__ uint32_t eval_distinct_n(const uint32_t (&c)[7], int n, int &nk) {
    uint64_t bits = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) bits |= (i < n) ? (1ull << (c[i] & 63)) : 0ull;
    return eval_distinct_bits<false>(bits, nk);
}
// pk_eval_hands' evaluator: the fast path above for 3..7 distinct cards, the literal scan otherwise (0..2 cards: its first lines)
__device__ __forceinline__ bool distinct_valid_cards(const uint32_t (&c)[7], int n);
__device__ __forceinline__ uint32_t eval_hand_any(const uint32_t (&c)[7], int n, int &nk) {
    if (distinct_valid_cards(c, n)) return eval_distinct_n(c, n, nk);   // (a byte that is no card -- suit > 3, rank nibble 13..15 -- would alias
    return eval_hand(c, n, nk);                                          //  onto a real card in the bitmask: such hands take the scan)
}

// ---------------------------------------------------------------------------------------------- table-driven evaluator
// The same function as eval7_distinct for the STREAMING evaluator (k_eval7_stream: eight waves per SIMD hide the LDS
// latency that made a lookup table a dead end inside k_rollout's one-wave-per-SIMD loop): the five-iteration `clz`
// extraction of the top five ranks, the straight detection and the lowest-run scan become one 32-bit table entry per
// 13-bit rank mask, and the category cascade becomes a max over candidates.  8 192 entries x 4 B = 32 KB of LDS.
//
// Entry of a RAW rank mask m (bit r0 set = a card of Card rank0 r0 present; bit 0 is the ace, cards.py:14) describing
// the ace-high mask a (bit k = rank k+1, ace = bit 12) -- indexing by the raw mask folds the ace-high conversion into
// the table, and every other operation of the evaluator is position-agnostic:
//   bits  0..19  the five highest ranks of a as nibbles, highest first (rank 1..13, 0 = none left)    judger.py:94, :99
//   bits 20..23  top rank (5..13) of the highest run of five in a, 0 = none (no wheel)                 judger.py:71-72, :95
//   bits 24..27  length, bits 28..31 top bit index (0..12) of the LOWEST run of a: where the reference's `both`
//                tracker ends up inside the flush group (judger.py:56-57), and (length 4, top 3) = "exactly 5-4-3-2"
__device__ inline uint32_t eval7_tab_entry(uint32_t m) {
    const uint32_t a = ((m >> 1) & 0xfffu) | ((m & 1u) << 12);
    uint32_t k5 = 0, x = a;
    for (int i = 0; i < 5; ++i) {
        const uint32_t lz = (uint32_t)__clz((int)x);
        k5 = (k5 << 4) | (32u - lz);
        x &= ~(0x80000000u >> (lz & 31));
    }
    const uint32_t m5 = a & (a >> 1) & (a >> 2) & (a >> 3) & (a >> 4);
    const uint32_t st = m5 ? (uint32_t)(31 - __clz((int)m5) + 5) : 0u;
    const uint32_t run = a & ~(a + (a & (0u - a)));
    const uint32_t len = (uint32_t)__popc(run), top = run ? (uint32_t)(31 - __clz((int)run)) : 0u;
    return k5 | (st << 20) | (len << 24) | (top << 28);
}

// lo = card bytes 0..3, hi = card bytes 4..6 (byte 7 ignored) of 7 DISTINCT cards; T = the table above (LDS).
// Candidates are encoded stronger = larger -- c' = 10 - HandRanking in bits 24..27, then (shift of the tail lookup)/4
// in bits 20..22, then the leading kickers -- so that the category cascade of judger.py:90-99 is a max.
// Two halves, so that a caller with several hands per lane can issue the lookups of all of them before any is used.
struct Eval7Front {
    uint32_t um, gm, pairs, trips, quads;
    uint32_t e_um, e_gm, e_p, e_t, e_q;
    bool has_flush;
};
// Every mask below is kept SHIFTED LEFT BY TWO (rank0 r0 = bit r0 + 2 of a 16-bit suit lane), i.e. it is already the byte
// offset of its table entry: six address shifts per evaluation less.
__device__ __forceinline__ uint32_t eval7_tab_at(const uint32_t *T, uint32_t mask4) {
    return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(T) + mask4);
}
// `bits` = OR of (4 << Card.value) over the hand's cards: four 16-bit suit lanes of raw rank masks, shifted left by two
__device__ __forceinline__ Eval7Front eval7_tab_front_bits(uint64_t bits, const uint32_t *T) {
    Eval7Front f;
    const uint32_t w01 = (uint32_t)bits, w23 = (uint32_t)(bits >> 32);
    const uint32_t sa = w01 & 0xffffu, sb = w01 >> 16, sc = w23 & 0xffffu, sd = w23 >> 16;   // raw rank masks per suit (x 4)
    f.um = sa | sb | sc | sd;
    const uint32_t s1 = sa ^ sb, c1 = sa & sb, s2 = sc ^ sd, c2 = sc & sd;        // per-rank count = bit0 + 2*t + 4*quads
    const uint32_t bit0 = s1 ^ s2, t = c1 ^ c2 ^ (s1 & s2);
    f.quads = c1 & c2; f.pairs = t & ~bit0; f.trips = t & bit0;
    // `flush` group (judger.py:52-58): the suit with >= 5 cards, else the lowest suit present
    const uint32_t ka = ((uint32_t)__popc(sa) << 15) | sa, kb = ((uint32_t)__popc(sb) << 15) | sb;
    const uint32_t kc = ((uint32_t)__popc(sc) << 15) | sc, kd = ((uint32_t)__popc(sd) << 15) | sd;
    const uint32_t gk = max(max(ka, kb), max(kc, kd));
    f.has_flush = gk >= (5u << 15);
    const uint32_t lp = sa ? sa : (sb ? sb : (sc ? sc : sd));
    f.gm = f.has_flush ? (gk & 0x7fffu) : lp;
    f.e_um = eval7_tab_at(T, f.um); f.e_gm = eval7_tab_at(T, f.gm); f.e_p = eval7_tab_at(T, f.pairs);
    f.e_t = eval7_tab_at(T, f.trips); f.e_q = eval7_tab_at(T, f.quads);
    return f;
}
__device__ __forceinline__ Eval7Front eval7_tab_front(uint32_t lo, uint32_t hi, const uint32_t *T) {
    const uint64_t bits = (4ull << (lo & 63)) | (4ull << ((lo >> 8) & 63)) | (4ull << ((lo >> 16) & 63)) | (4ull << ((lo >> 24) & 63)) |
                          (4ull << (hi & 63)) | (4ull << ((hi >> 8) & 63)) | (4ull << ((hi >> 16) & 63));
    return eval7_tab_front_bits(bits, T);
}
// ONE back end (round 6: the two former copies folded, ISA of k_eval7_tab_stream unchanged instruction for instruction).  SEVEN: the caller
// knows the hand holds 7 cards -- every tail is complete (the streaming evaluator); else n = 3 .. 7 DISTINCT cards with len(kickers): the table
// path of pk_eval_hands(_d) (the partial-hand rank feature of examples/q_learning.py:29-33).  What changes with fewer cards is what
// eval_distinct_bits states: `islice(others, count)` yields what is there (judger.py:91-99), so the tail takes min(nm, ranks left) nibbles and
// the whole kicker word moves down by the nibbles that are missing; len(kickers) = the leading kickers of the category + that tail.  Equal to
// eval_hand on EVERY 3-, 4-, 5-, 6- and 7-card subset of the deck (value and count: tools/host_sim `evalntab`).
template <bool SEVEN, bool NK>   // NK: len(kickers) is wanted
__device__ __forceinline__ uint32_t eval_tab_back(const Eval7Front &f, const uint32_t *T, int &nk) {
    const uint32_t p1 = (f.e_p >> 16) & 15u, p12 = (f.e_p >> 12) & 0xffu, p2 = p12 & 15u, p3 = (f.e_p >> 8) & 15u;
    const uint32_t t1 = (f.e_t >> 16) & 15u, t2 = (f.e_t >> 12) & 15u, q1 = (f.e_q >> 16) & 15u;
    const uint32_t st = (f.e_um >> 20) & 15u;
    // an empty mask's entry is 0, so "the family exists" is "its first rank is not 0"
    uint32_t W = 1u << 24;                                                                     // :99 HIGH, tail = top five of um
    const uint32_t w_p = p2 ? ((3u << 24) | (4u << 20) | (p12 << 4))                           // :97 TWO_PAIR [p1, p2, x]
                            : ((2u << 24) | (2u << 20) | (p1 << 12));                          // :98 PAIR [p1, x, x, x]
    W = max(W, p1 ? w_p : 0u);
    const uint32_t x2 = t2 ? t2 : p1;                                                          // :92 second trips, else :93 highest pair
    const uint32_t w_t = (x2 ? ((7u << 24) | (5u << 20) | x2) : ((4u << 24) | (3u << 20)))    // FULL [t1, x2] / :96 TRIS [t1, x, x]
                         | (t1 << (x2 ? 4u : 8u));
    W = max(W, t1 ? w_t : 0u);
    W = max(W, st ? ((5u << 24) | (5u << 20) | st) : 0u);                                      // :95 STRAIGHT
    W = max(W, f.has_flush ? ((6u << 24) | (5u << 20) | (f.e_gm & 0xfffffu)) : 0u);            // :94 FLUSH, five highest of the suit
    W = max(W, q1 ? ((8u << 24) | (4u << 20) | (q1 << 4)) : 0u);                               // :91 POKER [q, x]
    W = max(W, ((f.e_gm >> 24) & 15u) >= 5u ? ((9u << 24) | (5u << 20) | ((f.e_gm >> 28) + 1u)) : 0u);   // :90 STRAIGHT_FLUSH
    // ranks the tail must skip: the quads / the trips / the pair or the two highest pairs (a third pair p3 is the lowest
    // and never the ace, so its raw bit is 1 << p3 -- 4 << p3 in the shifted masks; p3 == 0 must clear nothing)
    const uint32_t taken2 = f.pairs & ~((4u << p3) & ~4u);
    const uint32_t taken = q1 ? f.quads : (t1 ? f.trips : taken2);
    const uint32_t rest = f.um & ~taken;
    uint32_t v;
    if constexpr (SEVEN && !NK) {     // (the streaming evaluator's own spelling of the same value: its ISA is what round 3 tuned, kept to the instruction)
        const uint32_t tail = (eval7_tab_at(T, rest) & 0xfffffu) >> (((W >> 20) & 7u) << 2);
        v = ((10u - (W >> 24)) << 20) | (W & 0xfffffu) | tail;
    } else {
        const uint32_t shift = (W >> 20) & 7u, cat = W >> 24;
        const uint32_t nm0 = 5u - shift, left = SEVEN ? 5u : (uint32_t)__popc(rest);
        const uint32_t nm = SEVEN ? nm0 : min(nm0, left);                                      // islice yields what is there
        const uint32_t tail = (eval7_tab_at(T, rest) & 0xfffffu) >> (shift << 2);
        const uint32_t kick = ((W & 0xfffffu) | tail) >> ((nm0 - nm) << 2);
        // len(kickers): the leading kickers per category (HIGH 0, PAIR 1, TWO_PAIR 2, TRIS 1, STRAIGHT 1, FLUSH 5, FULL 2, POKER 1, SF 1) + the tail
        if constexpr (NK) nk = (int)(((0x1125112100ull >> (cat << 2)) & 15u) + nm);
        v = ((10u - cat) << 20) | kick;
    }
    // wheel checks of judger.py:83-88, if / elif: a 5-4-3-2 run as the flush group's lowest run decides alone
    const bool low4 = (f.e_gm >> 24) == 0x34u;
    const bool wheel_sf = low4 && (f.gm & 4u);
    const bool wheel_st = !low4 && st == 0 && (f.e_um >> 24) == 0x34u && (f.um & 4u);
    v = wheel_st ? (((uint32_t)HR_STRAIGHT << 20) | 4u) : v;
    v = wheel_sf ? (((uint32_t)HR_SF << 20) | 4u) : v;
    if constexpr (NK) nk = (wheel_st || wheel_sf) ? 1 : nk;
    return v;
}
__device__ __forceinline__ uint32_t eval7_tab_back(const Eval7Front &f, const uint32_t *T) {
    int nk = 0;
    return eval_tab_back<true, false>(f, T, nk);
}
__device__ __forceinline__ uint32_t eval7_tab(uint32_t lo, uint32_t hi, const uint32_t *T) {
    const Eval7Front f = eval7_tab_front(lo, hi, T);
    return eval7_tab_back(f, T);
}
// The table path of pk_eval_hands(_d) on the suit-lane bit set (see tab_bits_of); SEVEN: ncards == NULL, every hand holds seven cards.
template <bool SEVEN = false>
__device__ __forceinline__ uint32_t eval_tab_bits(uint64_t bits, const uint32_t *T, int &nk) {
    const Eval7Front f = eval7_tab_front_bits(bits, T);
    return eval_tab_back<SEVEN, true>(f, T, nk);
}
// Front end on the PACKED hand (card i = byte i of w, n = 3 .. 7 cards; bytes from n on are ignored): the suit-lane bit set of
// eval_tab_bits and, with it, whether the hand may take the table at all -- every used byte a real card (suit < 4: byte < 0x40; rank0 < 13)
// and no card twice.  Unused bytes are replaced by a copy of card 0, which sets no new bit.  A rank nibble of 13 lands on bit 15 of its suit
// lane, 14 / 15 on bits 0 / 1 of the next lane (masks are shifted by two, so those bits are otherwise never set) or beyond bit 63, where it
// is lost and the popcount test fails.
template <bool SEVEN = false>
__device__ __forceinline__ bool tab_bits_of(uint64_t w, int n, uint64_t &bits) {
    uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32) & 0x00ffffffu;
    if (!SEVEN) {   // bytes n .. 6 := card 0 (a bit-field select per half: mask = the used bytes)
        const uint32_t r4 = rep4(lo), ulo = n >= 4 ? 0xffffffffu : ((1u << ((8 * n) & 31)) - 1u);
        const uint32_t uhi = n <= 4 ? 0u : ((1u << ((8 * (n - 4)) & 31)) - 1u);
        lo = (lo & ulo) | (r4 & ~ulo); hi = ((hi & uhi) | (r4 & ~uhi)) & 0x00ffffffu;
    }
    bits = (4ull << (lo & 63)) | (4ull << ((lo >> 8) & 63)) | (4ull << ((lo >> 16) & 63)) | (4ull << ((lo >> 24) & 63)) |
           (4ull << (hi & 63)) | (4ull << ((hi >> 8) & 63)) | (4ull << ((hi >> 16) & 63));
    const bool cards_ok = ((lo & 0xc0c0c0c0u) | (hi & 0x00c0c0c0u)) == 0 && (((uint32_t)bits | (uint32_t)(bits >> 32)) & 0x80038003u) == 0;
    return (SEVEN || n >= 3) && cards_ok && __popcll(bits) == (SEVEN ? 7 : n);
}
// judger.eval_hand's first lines for hands of 0, 1, 2 cards (judger.py:30-35), on the packed hand: what the scan returns for them
__device__ __forceinline__ uint32_t eval_small(uint64_t w, int n, int &nk) {
    uint32_t r0 = (uint32_t)w & 0xf, r1 = (uint32_t)(w >> 8) & 0xf;
    r0 = r0 ? r0 : 13u; r1 = r1 ? r1 : 13u;                                       // cards.py:14
    const uint32_t hi = r0 > r1 ? r0 : r1, lo = r0 > r1 ? r1 : r0;
    const bool pair = n == 2 && r0 == r1;
    nk = n == 0 ? 0 : ((n == 1 || pair) ? 1 : 2);
    const uint32_t v2 = pair ? (((uint32_t)HR_PAIR << 20) | r0) : (((uint32_t)HR_HIGH << 20) | (hi << 4) | lo);
    return n == 0 ? NONE_V : (n == 1 ? (((uint32_t)HR_HIGH << 20) | r0) : v2);
}
__device__ __forceinline__ uint32_t eval_tab_n(const uint32_t (&c)[7], int n, const uint32_t *T, int &nk) {   // (array form: tests, k_eval7_prefix)
    uint64_t bits = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) bits |= (i < n) ? (4ull << (c[i] & 63)) : 0ull;
    return eval_tab_bits(bits, T, nk);
}
// Which hands may take a bitmask / table evaluator: 3..7 cards, every used byte a real card (suit < 4: byte < 0x40; rank0 < 13), no card
// twice.  Anything else -- the reference's own tests feed eval_hand repeated cards -- takes the literal scan (its reading of a byte that is
// no card: suit = bits 4..5, rank nibble as it is -- what pk_eval_hands returned for such bytes before the bitmask paths existed).
__device__ __forceinline__ bool distinct_valid_cards(const uint32_t (&c)[7], int n) {
    uint64_t bits = 0;
    bool ok = n >= 3;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const bool used = i < n;
        ok = ok && (!used || (c[i] < 0x40u && (c[i] & 15u) < 13u));
        bits |= used ? (1ull << (c[i] & 63)) : 0ull;
    }
    return ok && __popcll(bits) == n;
}

// ---------------------------------------------------------------------------------------------- the table
// One table per lane.  Game.step's nested calls (next_player -> next_turn -> end_hand -> setup_hand, game.py:578-619)
// are flattened into a per-lane state machine so that a wavefront executes each expensive block ONCE per step for all
// the lanes that need it, instead of once per call site and per lane-divergent path:
//   LS_SCAN  next_player's walk to the next ACTIVE seat incl. next_turn's commit / turn roll-over (game.py:554-611)
//   LS_END   end_hand + setup_hand (game.py:453-539, 414-451): lanes park here until no lane is in SCAN/TURN, then
//            the whole wave runs the block together; showdown hands of all parked lanes are compacted through LDS and
//            evaluated one hand per lane (eval7_distinct).
//   LS_POT   a showdown whose side-pot loop (game.py:498-525) needs ANOTHER full iteration (18 % of showdowns under
//            random agents: three or more players all-in for different amounts).  end_block runs one full iteration
//            of that loop per call (plus the closing `num_potential_winners == 1` pass, :500-505, which 67 % of
//            showdowns end with) for the showdowns that arrive and the ones still in the loop together, instead of
//            iterating until the slowest arrival is done: that took 3.8 wave-iterations of the ~300-instruction body
//            with 14 of 64 lanes active, a third of all instructions of the kernel.  Never survives a kernel: the
//            complete kernels run until nothing is parked, k_rollout takes it back to LS_END when it ends early.
enum : int { LS_DONE = 0, LS_SCAN = 1, LS_POT = 2, LS_END = 3 };
#ifndef PK_WAVE
#define PK_WAVE 64  // lanes per wavefront on gfx950 (tools/host_sim builds this header with 1)
#endif
// Decks a lone dealing table computes at once, on as many lanes, for the hands its step may roll on (end_block<false>'s deal stock)
#define PK_STOCK (PK_WAVE >= 4 ? 4 : 1)
#if PK_WAVE >= 16
__device__ __forceinline__ uint32_t lane_read(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }   // lane: wave-uniform
__device__ __forceinline__ uint64_t lane_read64(uint64_t v, int lane) { return (uint64_t)lane_read((uint32_t)v, lane) | ((uint64_t)lane_read((uint32_t)(v >> 32), lane) << 32); }
__device__ __forceinline__ bool wave_uniform(bool b) { return __builtin_amdgcn_readfirstlane((int)b) != 0; }   // b holds the same value in every lane
#endif

// Diagnostic build only (-DPK_PROFILE, libpokerl_hip_prof.so; never the shipped library): per-wave cycle stamps
// (s_memtime) around the blocks of the step machine, summed into State::prof.  Shares, not run times, are read from it.
enum : int { PF_ACTION = 0, PF_CURSOR = 1, PF_END_PRE = 2, PF_EVAL = 3, PF_SIDEPOT = 4, PF_SETUP = 5, PF_DEAL = 6, PF_OTHER = 7,
             PF_N_CURSOR = 8, PF_N_END = 9, PF_N_EVALPASS = 10, PF_N_SIDEPOT = 11,
             PF_N_SIDEPOT_LANES = 12, PF_N_END_LANES = 13, PF_SLOTS = 16 };
#ifdef PK_PROFILE
struct Prof {
    unsigned long long acc[PF_SLOTS] = {0}, t0 = 0;
    __device__ __forceinline__ void start() { t0 = __builtin_readcyclecounter(); }
    __device__ __forceinline__ void lap(int slot) { unsigned long long t = __builtin_readcyclecounter(); acc[slot] += t - t0; t0 = t; }
    __device__ __forceinline__ void count(int slot, unsigned n = 1) { acc[slot] += n; }
    // wave-level event inside divergent control flow: counted once (by the first active lane), plus the active lanes
    // (global atomics: the build that uses this is for COUNTS only, its timings are meaningless)
    __device__ __forceinline__ void count_wave(unsigned long long *dst, int slot_events, int slot_lanes) {
        const unsigned long long act = __ballot(1);
        if ((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1) {
            atomicAdd(&dst[slot_events], 1ull); atomicAdd(&dst[slot_lanes], (unsigned long long)__popcll(act));
        }
    }
    __device__ __forceinline__ void flush(unsigned long long *dst) {
        if ((threadIdx.x & 63) == 0) for (int i = 0; i < PF_SLOTS; ++i) atomicAdd(&dst[i], acc[i]);
    }
};
#define PK_PROF(x) x
#else
#define PK_PROF(x)
#endif

template <int N>
struct Lds {  // per workgroup (= one wavefront); ~10 KB at N = 10
    static constexpr bool TAB = false;   // (LdsTab: the showdown hands are ranked by the table-driven evaluator)
    uint32_t item[64 * N + 64][2];  // [0] = community cards 0..3 (bytes); [1] = card4 | hole0<<6 | hole1<<12 | dest<<18
                               // (the last 64 entries: one scratch slot per lane for the writes of seats not in the showdown)
    uint32_t res[64 * N];      // dest = lane*N + seat -> HandRanking<<20 | kickers
    uint32_t act[8][64];       // k_rollout's action draws: two Philox blocks per lane, [slot * 4 + word][lane]
    uint32_t show[N][64];      // rankings of each lane's last showdown; written back by Table::store_show at kernel end
                               // (keeps global stores, and the vmcnt waits they drag along, out of the step loop)
    Fresh fresh;               // workgroup copy of *Hot::fresh (Table::stage_fresh), read with broadcast ds_reads
    alignas(16) uint8_t nth[128][8];   // nth[mask][k] = k-th (0-based) set bit of a 7-bit valid-action mask (stage_nth)
    // (last, so that the layout above is what the kernels without single-table tails were tuned with)
    uint32_t lone[(5 + 2 * N + 3) / 4];              // the deck words of a wave's ONE arriving table (end_block<false>), read by byte
    uint32_t stock[PK_STOCK][(5 + 2 * N + 3) / 4];   // decks dealt ahead for ONE table of the wave (end_block<false>: a lone dealing table's step may roll on)
};

// The same for k_rollout_tab (round 6): the workgroup also holds the 32 KB rank-mask table of eval7_tab, and the showdown hands are ranked by
// that evaluator (~110 instructions + 6 LDS lookups against eval7_distinct's ~230).  A CU's 160 KB of LDS hold FOUR such workgroups -- one wave
// per SIMD, what 65 536 tables need -- only if each stays within 40 960 bytes (measured: 40 864 bytes run at full rate, 42 912 at half,
// profiles/r06_lds_pad.txt), i.e. 8 192 bytes beside the table: up to six seats, with the rankings returned THROUGH the queue slots (no `res`),
// ONE dummy slot for the writes of seats not in the showdown, and without the single-table arrays (the kernel runs end_block<true>).
// RING: the in-kernel agent draws (random agents: the action ring and the k-th-valid-action table); without them -- the all-in agents, whose
// every hand is a showdown: BASELINE configs[4] -- the budget holds up to ten seats.
template <int N, bool RING>
struct LdsTab;
template <int N>
struct LdsTab<N, true> {
    static constexpr bool TAB = true, PAIRS = false;   // (PAIRS: two hands per lane per evaluation round -- end_block; the random agents' queues rarely hold > 64 hands: -0.6 % there)
    uint32_t item[64 * N + 1][2];      // [0] = community cards 0..3 (bytes); [1] = card4 | hole0<<6 | hole1<<12; [0] is overwritten by the hand's ranking
    uint32_t act[8][64];
    uint32_t show[N][64];
    Fresh fresh;
    alignas(16) uint8_t nth[128][8];
    alignas(16) uint32_t evtab[EVAL7_TAB_WORDS];
};
template <int N>
struct LdsTab<N, false> {
    static constexpr bool TAB = true, PAIRS = true;    // (all-in agents: ~290 hands per call at nine seats: +5 %)
    uint32_t item[64 * N + 1][2];
    uint32_t show[N][64];
    Fresh fresh;
    alignas(16) uint32_t evtab[EVAL7_TAB_WORDS];
};
#define PK_TAB_LDS_BUDGET 40960        // a quarter of a CU's 160 KB
static_assert(sizeof(LdsTab<6, true>) <= PK_TAB_LDS_BUDGET && sizeof(LdsTab<10, false>) <= PK_TAB_LDS_BUDGET,
              "four one-wave workgroups of k_rollout_tab<6> / k_rollout_allin_tab<10> must fit a CU's 160 KB of LDS");

struct ActionRng {  // one Philox block serves EIGHT consecutive steps of a table: 16-bit draws (RNG spec)
    uint64_t idx = ~0ull;
    uint32_t w[4];
    __device__ __forceinline__ uint32_t draw16(const Hot &S, uint32_t table_id, uint64_t step_serial) {
        const uint64_t q = step_serial >> 3;
        if (q != idx) {
            idx = q;
            philox4x32_10(table_id, (uint32_t)q, STREAM_ACTION, (uint32_t)(q >> 32), S.key0, S.key1, w);
        }
        const uint32_t j = (uint32_t)step_serial & 7;
        uint32_t lo = (j & 2) ? w[1] : w[0], hi = (j & 2) ? w[3] : w[2];
        uint32_t x = (j & 4) ? hi : lo;
        return (j & 1) ? (x >> 16) : (x & 0xffffu);
    }
};
// The same draws for k_rollout, where the lanes of a wave sit at different step serials: on demand, the Philox block
// of a lane that crosses a block boundary would be computed by the whole wave at 1/8 occupancy nearly every iteration.
// Instead every lane keeps TWO blocks in LDS (its current one and the next), and all lanes refill their free slot
// together, in wave-uniform control flow, only when some lane has run out: one Philox pass per ~8 iterations at
// ~3/4 occupancy.  Lanes only ever read what they wrote themselves (no cross-lane traffic, no barrier).
struct ActionRing {
    uint32_t filled = 0;  // (low 32 bits of) the first block index NOT yet in LDS; blocks filled-2, filled-1 are
    bool primed = false;
    // `need`: this lane picks an action now.  Must be called from wave-uniform control flow.
    template <typename LDS>
    __device__ __forceinline__ uint32_t draw16(LDS &lds, const Hot &S, uint32_t table_id, uint64_t step_serial, bool need) {
        const int lane = threadIdx.x & (PK_WAVE - 1);
        const uint64_t qfull = step_serial >> 3;
        const uint32_t q = (uint32_t)qfull;
        if (!primed || (int32_t)(filled - q) < 0) filled = q;       // nothing useful held (start, or the table idled)
        primed = true;
        if (__any(need && filled == q)) {                           // some lane's current block is missing
#pragma unroll 1
            for (int r = 0; r < 2; ++r) {                           // every lane fills its free slot(s)
                const bool fill = (int32_t)(filled - q) < 2;
                if (!__any(fill)) break;
                if (fill) {
                    const uint64_t b = qfull + (uint64_t)(filled - q);
                    uint32_t w[4];
                    philox4x32_10(table_id, (uint32_t)b, STREAM_ACTION, (uint32_t)(b >> 32), S.key0, S.key1, w);
                    const int slot = (filled & 1) * 4;
                    lds.act[slot + 0][lane] = w[0]; lds.act[slot + 1][lane] = w[1];
                    lds.act[slot + 2][lane] = w[2]; lds.act[slot + 3][lane] = w[3];
                    filled += 1;
                }
            }
        }
        const uint32_t j = (uint32_t)step_serial & 7;
        return lds.act[(q & 1) * 4 + (j >> 1)][lane];   // the word holding draw j; half_of() picks the 16 bits at the use site,
    }                                                    // so that the LDS latency hides behind the valid-mask arithmetic
    // The same split in two for a caller that draws up to `ahead` (<= 8) times per lane between two wave-uniform points:
    // ensure() once, peek() per draw.  `may`: this lane may pick actions before the next ensure().
    template <typename LDS>
    __device__ __forceinline__ void ensure(LDS &lds, const Hot &S, uint32_t table_id, uint64_t step_serial, bool may, int ahead) {
        const int lane = threadIdx.x & (PK_WAVE - 1);
        const uint64_t qfull = step_serial >> 3;
        const uint32_t q = (uint32_t)qfull;
        if (!primed || (int32_t)(filled - q) < 0) filled = q;
        primed = true;
        const int have = (int)(filled - q) * 8 - (int)((uint32_t)step_serial & 7);   // draws of this lane held in LDS
        if (__any(may && have < ahead)) {
#pragma unroll 1
            for (int r = 0; r < 2; ++r) {
                const bool fill = (int32_t)(filled - q) < 2;
                if (!__any(fill)) break;
                if (fill) {
                    const uint64_t b = qfull + (uint64_t)(filled - q);
                    uint32_t w[4];
                    philox4x32_10(table_id, (uint32_t)b, STREAM_ACTION, (uint32_t)(b >> 32), S.key0, S.key1, w);
                    const int slot = (filled & 1) * 4;
                    lds.act[slot + 0][lane] = w[0]; lds.act[slot + 1][lane] = w[1];
                    lds.act[slot + 2][lane] = w[2]; lds.act[slot + 3][lane] = w[3];
                    filled += 1;
                }
            }
        }
    }
    template <typename LDS>
    __device__ __forceinline__ static uint32_t peek(const LDS &lds, uint64_t step_serial) {
        const uint32_t s = (uint32_t)step_serial;
        return lds.act[((s >> 3) & 1) * 4 + ((s & 7) >> 1)][threadIdx.x & (PK_WAVE - 1)];
    }
    __device__ __forceinline__ static uint32_t half_of(uint32_t word, uint64_t step_serial) {
        return ((uint32_t)step_serial & 1) ? (word >> 16) : (word & 0xffffu);
    }
};
// The same through the workgroup's LDS table (k_rollout): one byte read instead of a six-step select chain.
struct alignas(16) NthTable {   // (16-byte aligned: stage_nth copies it with uint4 loads) nth[mask][k] = k-th (0-based) set bit of a 7-bit valid-action mask, for every mask: a compile-time constant
    uint8_t e[128][8];
    constexpr NthTable() : e{} {
        for (int m = 0; m < 128; ++m) {
            int rest = m;
            for (int k = 0; k < 8; ++k) {
                int low = 0;
                while (rest && !((rest >> low) & 1)) ++low;
                e[m][k] = rest ? (uint8_t)low : (uint8_t)0;
                rest &= rest - 1;
            }
        }
    }
};
__device__ __constant__ const NthTable g_nth{};
template <typename LDS>
__device__ __forceinline__ void stage_nth(LDS &lds) {  // call once from wave-uniform control flow: 16 bytes per lane
    static_assert(alignof(NthTable) >= 16 && sizeof(NthTable) == 64 * sizeof(uint4), "stage_nth copies the table as 64 uint4");
    for (int i = threadIdx.x & (PK_WAVE - 1); i < 64; i += PK_WAVE)
        reinterpret_cast<uint4 *>(&lds.nth[0][0])[i] = reinterpret_cast<const uint4 *>(&g_nth.e[0][0])[i];
    __syncthreads();
}
template <typename LDS>
__device__ __forceinline__ int action_from_draw_lds(const LDS &lds, uint32_t r16, uint32_t mask) {
    return lds.nth[mask & 127][__umul24(r16, (uint32_t)__popc(mask)) >> 16];
}
// k-th (0-based) valid action of the mask for a 16-bit draw r: k = (r * popcount(mask)) >> 16 (RNG spec)
__device__ __forceinline__ int action_from_draw(uint32_t r16, uint32_t mask) {
    uint32_t k = __umul24(r16, (uint32_t)__popc(mask)) >> 16;
    uint32_t m = mask;
#pragma unroll
    for (uint32_t i = 0; i < 6; ++i) m = (i < k) ? (m & (m - 1)) : m;  // drop the k lowest set bits (k <= 6)
    return __ffs(m) - 1;
}

// The call agent of the RNG / agent spec: CALL if valid, else CHECK if valid, else ALL_IN.
__device__ __forceinline__ int call_action(uint32_t mask) {
    return ((mask >> MV_CALL) & 1) ? (int)MV_CALL : (((mask >> MV_CHECK) & 1) ? (int)MV_CHECK : (int)MV_ALL_IN);
}
// Synthetic agents (RandomAgent semantics of pokerl/agents/random.py:12-16 under the RNG spec).
__device__ __forceinline__ int pick_action(const Hot &S, ActionRng &rng, uint32_t table_id, uint64_t step_serial, uint32_t mask, int policy) {
    if (policy == PK_POLICY_ALLIN) return MV_ALL_IN;
    if (policy == PK_POLICY_CALL) return call_action(mask);
    return action_from_draw(rng.draw16(S, table_id, step_serial), mask);
}
// policy nibble of seat p in a per-seat policy word (PokerGameEnv's agents list, envs/game_env.py:13-18)
__device__ __forceinline__ int seat_policy(uint64_t seatpol, int p) { return (int)((seatpol >> (4 * p)) & 15); }

// The showdown queue's synchronisation points.  Every table kernel's workgroup is ONE wavefront (PK_TABLE_BLOCK = 64), and a wave's LDS instructions
// execute in issue order: a write followed by a read of the same location -- also by another lane of that wave -- needs program order, not a barrier.
// Rounds 1-5 wrote __syncthreads() here: the compiler drops its s_barrier for a one-wave workgroup but keeps its fences -- a full s_waitcnt (every
// LDS and global access drained) and no scheduling across it.  Round 6: a compiler-only barrier (no memory operation moves across it, nothing is
// emitted; the reads' own s_waitcnt is placed at their first use): k_rollout_tab<6> 30.84 -> 31.13 G on one box (profiles/r06_tab_variants.txt).
// -DPK_HEAVY_SYNC brings the old form back.  NOT valid for workgroups of several waves (there are none among the table kernels: static_assert below).
#if defined(PK_HOST_SIM) || defined(PK_HEAVY_SYNC)
#define PK_QSYNC() __syncthreads()
#else
#define PK_QSYNC() __builtin_amdgcn_wave_barrier()
#endif
template <int N>
struct Table {
    static constexpr int K = 5 + 2 * N;      // cards ever read (game.py:278,388-395)
    static constexpr int W = (K + 3) / 4;    // packed words
    static_assert(N <= 16, "seat bitmasks are 16 bits wide (State::seat_states), policy words hold 16 nibbles");
    static constexpr uint32_t FULL = (1u << N) - 1;
    double credits[N], bets[N], pending[N], payoffs[N];
    double min_raise;
#ifdef PK_CARRY_HB
    double hb;                   // np.max(pending_bets) carried across the step instead of recomputed (experiment, round 6: see valid_mask)
#endif
    uint32_t st_active, st_called, st_allin, st_broken;  // seat bitmasks; FOLDED = in none of them
    int active, dealer, sb, bb, turn, hand;
    uint64_t hand_serial, step_serial;
    uint32_t cards[W];
    // showdown in progress (valid from end_hand's showdown branch until its side-pot loop is over: lstate LS_POT between calls)
    double pot_wb[N];            // `bets` working copy of game.py:485
    uint32_t pot_hv[N];          // hand_rankings with processed seats set to NONE (:522)
    uint32_t pot_todo;           // showdown seats not yet processed (:495-496)
    int pot_npw;                 // num_potential_winners (:472, :525)
    // per-step machine state
    int lstate, current, hands_this_step;
    uint32_t flags, terr, stepped;  // stepped: 1 while a Game.step is in flight on this lane
    bool foldout;
    // counters since load
    uint32_t evals, games, hands, seen;
    bool showed;  // lds.show holds a showdown of this launch
    bool pay_dirty;   // an end_hand of this launch has rewritten payoffs (load<false> / store<false>)
    uint32_t stock_lane; uint64_t stock_serial;   // wave-uniform: lds.stock holds the decks of hand serials stock_serial .. +PK_STOCK-1 of lane stock_lane's table
    PK_PROF(Prof prof;)

    // PAYOFFS == false (the single-step kernels, whose launch time is the bytes they move): Game.payoffs is not read -- the step machine
    // only ever WRITES it (end_hand zeroes it first, game.py:468) -- and store<false> writes it back only where a hand ended (pay_dirty).
    template <bool PAYOFFS = true, typename ST>   // ST: State, or a State in address space 1 (`*as_global(Sp)`)
    __device__ __forceinline__ void load(const ST &S, int t) {
        const auto g_credits = as_global(S.credits), g_bets = as_global(S.bets), g_pending = as_global(S.pending), g_payoffs = as_global(S.payoffs);
        const size_t T = (size_t)S.T;
        PK_FOR(p, N)
            credits[p] = g_credits[(size_t)p * T + t]; bets[p] = g_bets[(size_t)p * T + t];
            pending[p] = g_pending[(size_t)p * T + t]; payoffs[p] = PAYOFFS ? g_payoffs[(size_t)p * T + t] : 0.0;
         PK_END
        pay_dirty = false;
#ifdef PK_CARRY_HB
        hb = vmax<N>(pending);
#endif
        min_raise = as_global(S.min_raise)[t];
        uint64_t ss = as_global(S.seat_states)[t];
        st_active = (uint32_t)ss & 0xffff; st_called = (uint32_t)(ss >> 16) & 0xffff;
        st_allin = (uint32_t)(ss >> 32) & 0xffff; st_broken = (uint32_t)(ss >> 48) & 0xffff;
        uint32_t cur = as_global(S.cursors)[t];
        active = cur & 0xf; dealer = (cur >> 4) & 0xf; sb = (cur >> 8) & 0xf; bb = (cur >> 12) & 0xf; turn = (cur >> 16) & 0xf;
        hand = as_global(S.hand)[t];
        hand_serial = as_global(S.hand_serial)[t]; step_serial = as_global(S.step_serial)[t];
        const auto g_cards = as_global(S.cards);
        PK_FOR(w, W) cards[w] = g_cards[(size_t)w * T + t]; PK_END
        // A step left in flight by a deferred rollout launch (all-zero bits = idle table; only k_rollout ever finds
        // anything else: the host flushes deferred work before every other kernel).
        current = (cur >> 20) & 0xf; lstate = (cur >> 24) & 3; foldout = (cur >> 26) & 1; stepped = (cur >> 27) & 1;
        flags = (cur >> 28) & 7; hands_this_step = 0; terr = 0;
        PK_FOR(p, N) pot_wb[p] = 0.0; pot_hv[p] = NONE_V; PK_END
        pot_todo = 0; pot_npw = 0;
        evals = 0; games = 0; hands = 0; seen = 0; showed = false;
        stock_lane = ~0u; stock_serial = 0;
    }
    // A lane with no table (t >= T) still walks the wave-uniform control flow: give it inert, well-defined state.
    __device__ __forceinline__ void blank() {
        PK_FOR(p, N) credits[p] = bets[p] = pending[p] = payoffs[p] = 0.0;  PK_END
        min_raise = 0.0; st_active = st_called = st_allin = 0; st_broken = FULL;
#ifdef PK_CARRY_HB
        hb = 0.0;
#endif
        active = dealer = sb = bb = turn = hand = 0; hand_serial = step_serial = 0;
        PK_FOR(w, W) cards[w] = 0; PK_END
        PK_FOR(p, N) pot_wb[p] = 0.0; pot_hv[p] = NONE_V; PK_END
        pot_todo = 0; pot_npw = 0;
        idle();
        evals = 0; games = 0; hands = 0; seen = 0; showed = false; pay_dirty = false;
        stock_lane = ~0u; stock_serial = 0;
    }
    __device__ __forceinline__ void idle() { lstate = LS_DONE; current = 0; hands_this_step = 0; flags = 0; terr = 0; stepped = 0; foldout = false; }
    template <bool PAYOFFS = true, typename ST>
    __device__ __forceinline__ void store(const ST &S, int t) const {
        const auto g_credits = as_global(S.credits), g_bets = as_global(S.bets), g_pending = as_global(S.pending), g_payoffs = as_global(S.payoffs);
        const size_t T = (size_t)S.T;
        PK_FOR(p, N)
            g_credits[(size_t)p * T + t] = credits[p]; g_bets[(size_t)p * T + t] = bets[p];
            g_pending[(size_t)p * T + t] = pending[p];
            if (PAYOFFS) g_payoffs[(size_t)p * T + t] = payoffs[p];
         PK_END
        if (!PAYOFFS && pay_dirty) { PK_FOR(p, N) g_payoffs[(size_t)p * T + t] = payoffs[p]; PK_END }
        as_global(S.min_raise)[t] = min_raise;
        as_global(S.seat_states)[t] = (uint64_t)st_active | ((uint64_t)st_called << 16) | ((uint64_t)st_allin << 32) | ((uint64_t)st_broken << 48);
        const uint32_t inflight = ((uint32_t)current << 20) | ((uint32_t)lstate << 24) | ((uint32_t)foldout << 26) | (stepped << 27) | (flags << 28);
        as_global(S.cursors)[t] = (uint32_t)active | ((uint32_t)dealer << 4) | ((uint32_t)sb << 8) | ((uint32_t)bb << 12) | ((uint32_t)turn << 16) |
                                  (lstate == LS_DONE ? 0u : inflight);
        as_global(S.hand)[t] = hand;
        as_global(S.hand_serial)[t] = hand_serial; as_global(S.step_serial)[t] = step_serial;
        const auto g_cards = as_global(S.cards);
        PK_FOR(w, W) g_cards[(size_t)w * T + t] = cards[w]; PK_END
    }

    __device__ __forceinline__ void set_state(int p, int st) {  // player_states[p] = st
        uint32_t b = 1u << p;
        st_active &= ~b; st_called &= ~b; st_allin &= ~b; st_broken &= ~b;
        st_active |= (st == PS_ACTIVE) ? b : 0; st_called |= (st == PS_CALLED) ? b : 0;
        st_allin |= (st == PS_ALL_IN) ? b : 0; st_broken |= (st == PS_BROKEN) ? b : 0;
    }
    __device__ __forceinline__ static uint32_t rotr(uint32_t m, int a) {  // bit k of result = bit (a+k)%N of m
        return ((m >> a) | (m << (N - a))) & FULL;
    }
    // Game.get_first_playing, game.py:334-337 (idx in [0, N]; all-BROKEN -> idx % N)
    __device__ __forceinline__ int first_playing(int idx) const {
        int i = idx >= N ? idx - N : idx;
        uint32_t nb = ~st_broken & FULL;
        int r = i + (__ffs(rotr(nb, i)) - 1);
        r = r >= N ? r - N : r;
        return nb ? r : i;
    }
    __device__ __forceinline__ bool game_over() const { return __popc(~st_broken & FULL) == 1; }  // game.py:317-320

    // Game.get_valid_actions(active player) as a bitmask, game.py:339-383.  high_bet is returned for the step.
    __device__ __forceinline__ uint32_t valid_mask(double &high_bet) const {
#ifdef PK_CARRY_HB
        high_bet = hb;
#ifdef PK_HOST_SIM
        if (hb != vmax<N>(pending)) { printf("CARRIED HIGH BET %.17g != np.max(pending_bets) %.17g\n", hb, vmax<N>(pending)); abort(); }
#endif
#else
        high_bet = vmax<N>(pending);                                               // :365
#endif
        double credit = sel<N>(credits, active);                                   // :366
        uint32_t mask = (1u << MV_FOLD) | (1u << MV_ALL_IN);                       // :367
        double d = credit - high_bet;
        double rv0 = 0.1 * d, rv1 = 0.25 * d, rv2 = 0.5 * d;                       // :370
        mask |= (rv0 > min_raise && (high_bet + rv0) < credit) ? (1u << 3) : 0;    // :371
        mask |= (rv1 > min_raise && (high_bet + rv1) < credit) ? (1u << 4) : 0;
        mask |= (rv2 > min_raise && (high_bet + rv2) < credit) ? (1u << 5) : 0;
        mask |= (high_bet == 0.0) ? (1u << MV_CHECK) : 0;                          // :375
        mask |= (high_bet < credit) ? (1u << MV_CALL) : 0;                         // :376
        return mask;
    }

    // Deck of this hand (RNG spec: DESIGN.md, "RNG specification") -> cards[]; replaces random.shuffle(self.deck), game.py:424.
    __device__ __forceinline__ void deal(const Hot &S, uint32_t table_id) {
        deal_cards(S, table_id, hand_serial, cards);
        hand_serial += 1;
    }
    // The deck of (table_id, hand_serial) as packed Card.value words: a pure function of its arguments (any lane may compute any table's)
    __device__ __forceinline__ static void deal_cards(const Hot &S, uint32_t table_id, uint64_t hand_serial, uint32_t (&cards)[W]) {
        uint32_t c[K];
        constexpr int NB = (K + 17) / 18;
        PK_FOR(b, NB)
            uint32_t w[4];
            philox4x32_10(table_id, (uint32_t)hand_serial, STREAM_DECK + (uint32_t)b, (uint32_t)(hand_serial >> 32), S.key0, S.key1, w);
            PK_FOR(h, 2)
                uint32_t xlo = w[2 * h], xhi = w[2 * h + 1];
                PK_FOR(j, 9)
                    constexpr int i = b * 18 + h * 9 + j;
                    if constexpr (i < K) {  // chained multiply-high: c_i = (x * (52-i)) >> 64, x = low 64 bits
                        uint64_t t = (uint64_t)xlo * (uint32_t)(52 - i);
                        uint64_t u = (uint64_t)xhi * (uint32_t)(52 - i) + (t >> 32);
                        c[i] = (uint32_t)(u >> 32); xlo = (uint32_t)t; xhi = (uint32_t)u;
                    }
                PK_END
            PK_END
        PK_END
        // Lehmer decode ("c_i-th card not yet dealt") without arrays: packed bytes (bit 7 kept set), processed from the
        // last draw to the first; for each earlier-processed (later-drawn) byte b: b += (b >= c_i).  A word operation costs
        // four instructions whatever the number of live bytes in it, so a LAST word that holds a single card (K = 4k + 1:
        // six seats, K = 17) is kept as a plain register instead: compare + add-with-carry, two instructions per step.
        constexpr bool LONE = (K % 4) == 1 && K > 1;
        constexpr int WP = LONE ? W - 1 : W;           // packed words
        uint32_t a[W];
        PK_FOR(w, W)
            uint32_t v = 0x80808080u;
            PK_FOR(j, 4) if constexpr (4 * w + j < K) v |= c[4 * w + j] << (8 * j); PK_END
            a[w] = v;
        PK_END
        uint32_t lone = LONE ? c[K - 1] : 0u;
        PK_FOR(ii, K)
            constexpr int i = K - 1 - ii;
            if constexpr (!(LONE && i == K - 1)) {
                const uint32_t bc = rep4(c[i]);                                    // the draw in all four bytes
                if constexpr (LONE) lone += (lone >= c[i]) ? 1u : 0u;
                PK_FOR(w, WP)
                    if constexpr (w > i / 4) a[w] += ((a[w] - bc) & 0x80808080u) >> 7;
                    else if constexpr (w == i / 4 && (i % 4) != 3)
                        a[w] += ((a[w] - bc) & 0x80808080u & (0x80808080u << (8 * ((i % 4) + 1)))) >> 7;
                PK_END
            }
        PK_END
        if constexpr (LONE) a[W - 1] = 0x80808080u | lone;
        PK_FOR(w, W)  // canonical index k -> Card.value ((k%4)<<4)|(k//4), cards.py:77
            uint32_t idx = a[w] & 0x3f3f3f3fu;
            cards[w] = ((idx & 0x03030303u) << 4) | ((idx >> 2) & 0x0f0f0f0fu);
        PK_END
    }
    __device__ __forceinline__ uint32_t card(int i) const { return (cards[i >> 2] >> (8 * (i & 3))) & 0xff; }  // compile-time i

    // Game.setup_hand minus the shuffle, game.py:414-451
    __device__ __forceinline__ void setup_state(const Hot &S) {
        hand += 1; turn = 0;                                                       // :417-418
        st_active = ~st_broken & FULL; st_called = 0; st_allin = 0;                // :421
        dealer = first_playing(dealer + 1);                                        // :432
        sb = first_playing(dealer + 1);                                            // :433
        bb = first_playing(sb + 1);                                                // :434
        active = first_playing(bb + 1);                                            // :435
        PK_FOR(p, N)                                              // :438-440 (fancy index: sb written last)
            bets[p] = 0.0;
            double v = (p == bb) ? S.big_blind : 0.0;
            pending[p] = (p == sb) ? S.small_blind : v;
         PK_END
        set_state(bb, PS_CALLED);                                                  // :441
        uint32_t over = 0;
        PK_FOR(p, N)
            over |= (pending[p] > credits[p]) ? (1u << p) : 0;                     // :444 (also revives BROKEN seats with credits < 0)
            pending[p] = (credits[p] < pending[p]) ? credits[p] : pending[p];      // :445 np.minimum
         PK_END
        st_active &= ~over; st_called &= ~over; st_broken &= ~over; st_allin |= over;
        min_raise = vmax<N>(pending);                                              // :446
#ifdef PK_CARRY_HB
        hb = min_raise;
#endif
        hands_this_step += 1;
    }
    // Game.reset minus the shuffle, game.py:397-412
    __device__ __forceinline__ void reset_state(const Hot &S, int dealer_cfg) {
        dealer = dealer_cfg; hand = 0; active = 0;                                 // :403-407
        if (S.start_is_uniform) {                                                  // :408 (scalar branch: S is wave-uniform)
            PK_FOR(p, N) credits[p] = S.start_uniform; PK_END
        } else {
            PK_FOR(p, N) credits[p] = S.start_credits[p]; PK_END
        }
        st_active = FULL; st_called = st_allin = st_broken = 0;                    // :409
        setup_state(S);                                                            // :412
    }

    // Rankings of the last showdown (game.py:488-489) -> State::show; every kernel that runs end_block calls this last.
    template <typename LDS>
    __device__ __forceinline__ void store_show(uint32_t *show, int T, int t, const LDS &lds) const {
        if (showed) {
            const int lane = threadIdx.x & (PK_WAVE - 1);
            const auto g_show = as_global(show);
            PK_FOR(p, N) g_show[(size_t)p * T + t] = lds.show[p][lane]; PK_END
        }
    }
    // Workgroup copy of the fresh-table constants; call once, from wave-uniform control flow, before the first end_block.
    template <typename LDS>
    __device__ __forceinline__ static void stage_fresh(LDS &lds, const Fresh *src) {
        const int lane = threadIdx.x & (PK_WAVE - 1);
        constexpr int words = (int)(sizeof(Fresh) / 4);
        const auto s32 = as_global(reinterpret_cast<const uint32_t *>(src));
        uint32_t *d32 = reinterpret_cast<uint32_t *>(&lds.fresh);
        for (int i = lane; i < words; i += PK_WAVE) d32[i] = s32[i];
        __syncthreads();
    }
    // Game.reset(dealer = 0) minus the shuffle as a copy of the per-handle constants (see Fresh); payoffs are kept,
    // as reset() keeps them.
    __device__ __forceinline__ void load_fresh(const Fresh &f) {
        PK_FOR(p, N) credits[p] = f.credits[p]; bets[p] = 0.0; pending[p] = f.pending[p]; PK_END
        min_raise = f.min_raise;
#ifdef PK_CARRY_HB
        hb = f.min_raise;
#endif
        st_active = f.st_active; st_called = f.st_called; st_allin = f.st_allin; st_broken = f.st_broken;
        active = f.active; dealer = f.dealer; sb = f.sb; bb = f.bb;
        hand = 1; turn = 0;
        hands_this_step += 1;
    }

    // Game.step up to the call of next_player (game.py:656-699) for an action already checked against the mask.
    // Branch-free: every lane of the wave takes a different action, so the three arms are merged into selects.
    __device__ __forceinline__ void begin_step(const Hot &S, int action, double high_bet) {
        const int a = active;
        const uint32_t b = 1u << a;
        hands_this_step = 0; terr = 0; flags = 0; stepped = 1;
        const bool money = action >= MV_CALL;                                      // :662 (CALL / RAISE* / ALL_IN)
        const double credit = sel<N>(credits, a);                                  // :666
        const double base = (S.big_blind > high_bet) ? S.big_blind : high_bet;     // :665 max(high_bet, big_blind)
        const double f = action == 3 ? 0.1 : (action == 4 ? 0.25 : 0.5);           // :676
        const double raised = base + (credit - base) * f;                          // :675-678
        const bool is_raise = action >= MV_RAISE_ANY && action < MV_ALL_IN;
        const double bet_value = (action == MV_ALL_IN) ? credit : (is_raise ? raised : base);  // :669-678
        const bool raises = money && bet_value > high_bet;                         // :680
        st_active |= raises ? st_called : 0;                                       // :683 CALLED -> ACTIVE
        st_called = raises ? 0 : st_called;
        min_raise = raises ? bet_value - high_bet : min_raise;                     // :687
#ifdef PK_CARRY_HB
        hb = raises ? bet_value : hb;        // (a bet that is no raise never exceeds the high bet; the acting seat's own earlier bet is never the sole maximum it undercuts)
#endif
        st_active &= ~b; st_called &= ~b; st_allin &= ~b; st_broken &= ~b;         // player_states[a] = ...
        st_allin |= (action == MV_ALL_IN) ? b : 0;                                 // :671
        st_called |= (action != MV_FOLD && action != MV_ALL_IN) ? b : 0;           // :660, :667 (FOLD: in no mask, :657)
        PK_FOR(p, N) pending[p] = (money && p == a) ? bet_value : pending[p]; PK_END  // :696
        // next_player's entry (game.py:598-605,616-619)
        const bool many = __popc((st_active | st_called | st_allin) & FULL) > 1;
        current = a;
        active = many ? ((a + 1 == N) ? 0 : a + 1) : a;
        lstate = many ? LS_SCAN : LS_END;
        foldout = !many;
    }

    // The `while states[active] != ACTIVE` walk of next_player (game.py:607-611) for one lane, O(1) with seat bitmasks.
    __device__ __forceinline__ bool scan() {
        int d = current - active; d = d < 0 ? d + N : d;                           // seats active..current (cyclic)
        uint32_t window = rotr(st_active, active) & ((2u << d) - 1);
        int a = active + (__ffs(window) - 1); a = a >= N ? a - N : a;
        active = window ? a : current;                                             // else: walked up to current_player
        return window != 0;
    }
    // Takes a lane from LS_SCAN to LS_DONE or LS_END in straight-line code (no wave-wide loop):
    //   scan -> [next_turn (game.py:554-576) -> scan] -> DONE | END.
    // If the scan after a next_turn() fails as well, every further next_turn() before turn 4 is a no-op for the
    // outcome: pending bets are already zero (x + 0.0 == x), CALLED seats cannot be re-activated a second time (there
    // are none left, or there was at most one), and the walk starts from the same seat over the same states, so it
    // fails again -- the loop can only end in end_hand() at turn 4.  (This is how the reference skips betting rounds,
    // SURVEY A.3; the flags of those intermediate turns are overwritten by end_hand's.)
    __device__ __forceinline__ void cursor() {
        if (lstate != LS_SCAN) return;
        if (scan()) { lstate = LS_DONE; return; }
        next_turns();
    }
    // The same in two halves for k_rollout's betting pass: scan_first() is branch-free (nearly every step ends there);
    // cursor_tail() is the divergent rest for the lanes whose walk failed.  cursor_tail() ONLY after scan_first().
    __device__ __forceinline__ void scan_first() {
        int d = current - active; d = d < 0 ? d + N : d;
        uint32_t window = rotr(st_active, active) & ((2u << d) - 1);
        int a = active + (__ffs(window) - 1); a = a >= N ? a - N : a;
        const bool hit = lstate == LS_SCAN && window != 0;
        active = hit ? a : active;
        lstate = hit ? (int)LS_DONE : lstate;
    }
    __device__ __forceinline__ void cursor_tail() {
        if (lstate != LS_SCAN) return;
        active = current;                                                          // the failed walk ends at current_player
        next_turns();
    }
    __device__ __forceinline__ void next_turns() {
        PK_FOR(p, N) bets[p] = bets[p] + pending[p]; credits[p] = credits[p] - pending[p]; pending[p] = 0.0; PK_END  // :554-557
        min_raise = 0.0;                                                           // :558
#ifdef PK_CARRY_HB
        hb = 0.0;
#endif
        turn += 1;                                                                 // :561
        const bool more = turn < 4;                                                // :566-576, as selects (one basic block)
        const bool merge = more && __popc(st_called) > 1;
        st_active |= merge ? st_called : 0u; st_called = merge ? 0u : st_called;
        const int first = first_playing(dealer + 1);
        int d = current - first; d = d < 0 ? d + N : d;                            // scan() from `first`
        const uint32_t window = rotr(st_active, first) & ((2u << d) - 1);
        int a = first + (__ffs(window) - 1); a = a >= N ? a - N : a;
        const bool found = more && window != 0;
        active = more ? (window ? a : current) : active;
        flags = more ? (uint32_t)PK_FLAG_TURN_OVER : flags;
        lstate = found ? LS_DONE : LS_END;                                         // :563-565 (turn 4, now or after no-op turns)
        turn = found ? turn : 4;
        foldout = false;
    }

    __device__ __forceinline__ bool parked() const { return lstate == LS_END || lstate == LS_POT; }

    // end_hand + setup_hand for every lane parked at LS_END (game.py:453-539), executed by the WHOLE wave.
    // auto_reset: a finished game (or a table that hit PK_HAND_CAP, which the reference would never leave) is
    // Game.reset() on the spot, as the rollout/bench loop does on the host side of the reference.
    // ONE_PASS == false: the side-pot loop runs to its end inside the call (no LS_POT).  Slower for the multi-step rollout at
    // every batch size (21.9 vs 23.4 G at 65 536 x 6, 34.5 vs 37.0 G at 1 M x 6), faster where a launch ends with a few lanes
    // per wave anyway, so that every further end_block call serves one or two tables: the single-step kernels (k_step,
    // k_rollout_single: +3.9 %), the synchronous PokerGameEnv.step (+3.1 %) and k_env_reset.
    // LONE (only with !ONE_PASS): the two single-table paths below -- a lone arriving table's showdown without the queue, a lone dealing
    // table's decks four at a time.  For the kernels whose launches end in a tail of single tables: k_step / k_rollout_single (+7 %: a
    // step that rolls hand after hand), k_env_step / k_env_reset (+1 %); NOT the bounded asynchronous env launches (-3 %: their
    // end_blocks serve one or two tables that rarely deal twice within a launch) -- profiles/r05_ab_lone.txt.
    template <bool ONE_PASS = true, bool LONE = !ONE_PASS, typename LDS = Lds<N>>
    __device__ __forceinline__ void end_block(const Hot &S, int t, uint32_t table_id, LDS &lds, bool auto_reset) {
        constexpr bool TAB = LDS::TAB;      // (LdsTab: rank the showdown hands with the table-driven evaluator; the rankings come back through the queue slots)
        static_assert(!TAB || (ONE_PASS && !LONE), "the table evaluator's LDS layout has no single-table arrays");
        static_assert(!(ONE_PASS && LONE), "the lone-table paths belong to the kernels that run the whole side-pot loop per call");
        const bool e = lstate == LS_END;
        const bool resumed = ONE_PASS && lstate == LS_POT;
        PK_PROF(prof.lap(PF_OTHER); prof.count(PF_N_END);)
#ifdef PK_PROFILE_COUNTS
        if (e) { const unsigned long long act = __ballot(1); if ((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1) atomicAdd(&S.prof[PF_N_END_LANES], (unsigned long long)__popcll(act)); }
#endif
        bool sd = false, nowin = false, need_deal = false;
        uint32_t showdown = 0;
        if (e) {
            pay_dirty = true;
            PK_FOR(p, N)                                          // :457-461, :468
                bets[p] = bets[p] + pending[p]; credits[p] = credits[p] - pending[p];
                pending[p] = 0.0; payoffs[p] = 0.0;
             PK_END
            min_raise = 0.0;
#ifdef PK_CARRY_HB
            hb = 0.0;
#endif
            uint32_t pw = (st_active | st_called | st_allin) & FULL;               // :471 (not BROKEN, not FOLDED)
            const int npw = __popc(pw);                                            // :472
            nowin = npw <= 0;                                                      // :473
            {                                                                      // :475-480 (selects: one basic block)
                const int winner = (npw == 1) ? __ffs(pw) - 1 : -1;
                const double pot = np_sum<N>(bets);
                PK_FOR(p, N)
                    payoffs[p] = (p == winner) ? pot : payoffs[p];
                    credits[p] = (p == winner) ? credits[p] + pot : credits[p];
                 PK_END
            }
            sd = npw > 1;
            showdown = sd ? (st_called | st_allin) & FULL : 0u;                    // :488, :496
            pot_npw = sd ? npw : pot_npw;
        }
        PK_PROF(prof.lap(PF_END_PRE);)
        const int lane = threadIdx.x & 63;
        // ---- ONE arriving lane (the kernels whose launches end in a tail of single tables: a step that rolls hand after hand, a busted
        //      seat 0 waiting for the end of its game): no queue, no barrier -- the lane's cards travel by v_readlane, lane p evaluates
        //      seat p's hand, the rankings travel back the same way.  Same function on the same cards as the queue below.
        bool lone = false;
#if PK_WAVE >= 16 && !defined(PK_NO_LONE)   // (PK_NO_LONE: A/B builds without the lone-table paths)
        if constexpr (LONE) {
            const unsigned long long eb = __ballot(e);
            lone = __popcll(eb) == 1;                                              // wave-uniform
            if (lone) {
                const int sl = __ffsll((long long)eb) - 1;
                const uint32_t sm = lane_read(showdown, sl);
                if (sm) {
                    // the lane's deck words go through ONE row of LDS: lane p picks seat p's hole cards out of it by byte address (as
                    // a select chain over readlane'd words the compiler spilled the words to scratch memory and indexed them there)
                    if (e) { PK_FOR(w, W) lds.lone[w] = cards[w]; PK_END }
                    __syncthreads();
                    const int q = lane < N ? lane : 0;
                    const uint8_t *lb = reinterpret_cast<const uint8_t *>(lds.lone);
                    const uint32_t c0 = lds.lone[0];
                    const uint32_t h[7] = {c0 & 0xff, (c0 >> 8) & 0xff, (c0 >> 16) & 0xff, c0 >> 24, lb[4],
                                           lb[5 + 2 * q], lb[6 + 2 * q]};               // hand = deck[:5] + hole cards (:394-395)
                    const uint32_t v = eval7_distinct(h);
                    uint32_t res[N];
                    PK_FOR(p, N) res[p] = lane_read(v, p); PK_END
                    if (sd) { PK_FOR(p, N) pot_hv[p] = ((showdown >> p) & 1) ? res[p] : NONE_V; PK_END }
                    evals += __popc(showdown);
                } else if (sd) {
                    PK_FOR(p, N) pot_hv[p] = NONE_V; PK_END
                }
            }
        }
#endif
        // ---- showdown hands of all arriving lanes -> LDS queue -> one hand per lane (game.py:488-489)
        uint32_t total = 0, my_base[N];
        if (!lone) {
        PK_FOR(p, N)
            unsigned long long bal = __ballot((showdown >> p) & 1);
            my_base[p] = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            total += (uint32_t)__popcll(bal);
         PK_END
        }
        if (lone) {
        } else if (total) {  // wave-uniform
            PK_FOR(p, N)                                                           // hand = deck[:5] + hole cards (:394-395)
                const uint32_t slot = ((showdown >> p) & 1) ? my_base[p] : (uint32_t)(64 * N + (TAB ? 0 : lane));   // no branch per seat (TAB: ONE dummy slot)
                lds.item[slot][0] = cards[0];
                lds.item[slot][1] = card(4) | (card(5 + 2 * p) << 6) | (card(6 + 2 * p) << 12) | (TAB ? 0u : ((uint32_t)(lane * N + p) << 18));
             PK_END
            PK_QSYNC();
            uint32_t base0 = 0;
            if constexpr (TAB) {
                // More than 64 hands in the queue (the all-in agents: ~290 per call at nine seats): TWO hands per lane per round, branch-free, so that
                // the ten lookups of a pair are in flight together and one hand's arithmetic fills the other's LDS waits -- at one wave per SIMD each
                // of the three dependent LDS round trips of an evaluation is otherwise exposed.  A lane without a (second) hand evaluates the dummy
                // slot.  The remainder (and the usual case of the random agents, <= 64 hands) takes the single-hand loop below.
                if constexpr (LDS::PAIRS) {
                    auto bits_of = [](uint32_t w0, uint32_t w1) {
                        return (4ull << (w0 & 63)) | (4ull << ((w0 >> 8) & 63)) | (4ull << ((w0 >> 16) & 63)) | (4ull << ((w0 >> 24) & 63)) |
                               (4ull << (w1 & 63)) | (4ull << ((w1 >> 6) & 63)) | (4ull << ((w1 >> 12) & 63));
                    };
                    for (; base0 + PK_WAVE < total; base0 += 2 * PK_WAVE) {
                        PK_PROF(prof.count(PF_N_EVALPASS, 2);)
                        const uint32_t ia = base0 + lane, ib0 = ia + PK_WAVE;
                        const uint32_t ib = ib0 < total ? ib0 : (uint32_t)(64 * N);          // (ia < total: the loop condition)
                        const uint32_t a0 = lds.item[ia][0], a1 = lds.item[ia][1], b0 = lds.item[ib][0], b1 = lds.item[ib][1];
                        const Eval7Front fa = eval7_tab_front_bits(bits_of(a0, a1), lds.evtab), fb = eval7_tab_front_bits(bits_of(b0, b1), lds.evtab);
                        const uint32_t va = eval7_tab_back(fa, lds.evtab), vb = eval7_tab_back(fb, lds.evtab);
                        lds.item[ia][0] = va; lds.item[ib][0] = vb;
                    }
                }
            }
            for (uint32_t base = base0; base < total; base += PK_WAVE) {
                PK_PROF(prof.count(PF_N_EVALPASS);)
                uint32_t i = base + lane;
                if (i < total) {
                    uint32_t w0 = lds.item[i][0], w1 = lds.item[i][1];
                    if constexpr (TAB) {
                        // the hand's suit-lane bit set straight from the packed words (a 64-bit shift takes its amount modulo 64: no masks needed
                        // beyond the ones C wants), then eval7_tab's lookups in the workgroup's copy of the table
                        const uint64_t bits = (4ull << (w0 & 63)) | (4ull << ((w0 >> 8) & 63)) | (4ull << ((w0 >> 16) & 63)) | (4ull << ((w0 >> 24) & 63)) |
                                              (4ull << (w1 & 63)) | (4ull << ((w1 >> 6) & 63)) | (4ull << ((w1 >> 12) & 63));
                        lds.item[i][0] = eval7_tab_back(eval7_tab_front_bits(bits, lds.evtab), lds.evtab);
                    } else {
                        uint32_t h[7] = {w0 & 0xff, (w0 >> 8) & 0xff, (w0 >> 16) & 0xff, w0 >> 24, w1 & 0x3f, (w1 >> 6) & 0x3f, (w1 >> 12) & 0x3f};
                        lds.res[w1 >> 18] = eval7_distinct(h);
                    }
                }
            }
            PK_QSYNC();
            if constexpr (TAB) { if (sd) { PK_FOR(p, N) pot_hv[p] = ((showdown >> p) & 1) ? lds.item[my_base[p]][0] : NONE_V; PK_END } }
            else if (sd) { PK_FOR(p, N) pot_hv[p] = ((showdown >> p) & 1) ? lds.res[lane * N + p] : NONE_V; PK_END }
            PK_QSYNC();  // the queue is reused by the next end_block of this wave
            evals += __popc(showdown);
        } else if (sd) {
            PK_FOR(p, N) pot_hv[p] = NONE_V; PK_END                                // potential winners, none of them CALLED / ALL_IN
        }
        PK_PROF(prof.lap(PF_EVAL);)
        PK_FOR(p, N) pot_wb[p] = sd ? bets[p] : pot_wb[p]; PK_END                  // :485 (selects: one basic block)
        pot_todo = sd ? showdown : pot_todo;                                       // :495-496 argsort(bets) filtered, stable
        showed = showed || sd;
        if (sd) { PK_FOR(p, N) lds.show[p][lane] = pot_hv[p]; PK_END }             // (an unconditional write to a scratch row: slower)
        // ---- the side-pot loop (:498-525) for the arriving showdowns and the ones still in it: per call ONE pass of
        //      the general body plus, if that leaves a single potential winner, the closing pass of :500-505.
        bool pot_over = false;
        if (sd || resumed) {
          for (;;) {
#ifdef PK_PROFILE_COUNTS
            prof.count_wave(S.prof, PF_N_SIDEPOT, PF_N_SIDEPOT_LANES);
#endif
            bool left = false;                                                     // any bet still > 0 (:499)
            PK_FOR(p, N) left = left || !(pot_wb[p] <= 0.0); PK_END
            if (pot_todo != 0 && left && pot_npw != 1) {                           // a general pass (:507-525)
                int player = 0; double best = 0.0, max_bet = 0.0; bool have = false;
                PK_FOR(p, N)                                                       // next seat in ascending ORIGINAL-bet order
                    bool cand = (pot_todo >> p) & 1;
                    bool better = cand && (!have || bets[p] < best);
                    player = better ? p : player; best = better ? bets[p] : best; max_bet = better ? pot_wb[p] : max_bet;  // :508
                    have = have || cand;
                PK_END
                pot_todo &= ~(1u << player);
                double mb[N];                                                      // :509 np.clip(bets, 0, max_bet)
                PK_FOR(p, N) double x = pot_wb[p]; x = (x < 0.0) ? 0.0 : x; x = (x > max_bet) ? max_bet : x; mb[p] = x; PK_END
                int nw;
                uint32_t win = compare_rankings<N>(pot_hv, nw);                    // :512
                double s = np_sum<N>(mb);
                // :515 single winner: += s.  :516 split: += s*onehot/k, i.e. s/k for winners ((s*1.0)/k == s/k) and
                // (s*0.0)/k == +0.0 for the rest (s >= 0), which leaves a non-negative payoff unchanged bit for bit.
                double share = (nw == 2) ? s * 0.5 : s;                           // s / 2 == s * 0.5 exactly
                if (nw > 2) share = s / (double)nw;                                // real division only for 3+-way ties
                left = false;
                PK_FOR(p, N)
                    payoffs[p] = ((win >> p) & 1) ? payoffs[p] + share : payoffs[p];
                    pot_hv[p] = (p == player) ? NONE_V : pot_hv[p];                // :522
                    pot_wb[p] = pot_wb[p] - mb[p];                                 // :523
                    left = left || !(pot_wb[p] <= 0.0);
                PK_END
                pot_npw -= 1;                                                      // :525
            }
            // what the next pass of the loop would do: stop (:498, :499), hand everything left to the last potential
            // winner (:500-505; with one potential winner left at most one seat is left in `todo`), or go on
            {                                                                      // (selects: one basic block)
                const bool stop = pot_todo == 0 || !left;
                const int player = (!stop && pot_npw == 1) ? __ffs(pot_todo) - 1 : -1;
                const double s = np_sum<N>(pot_wb);
                PK_FOR(p, N) payoffs[p] = (p == player) ? payoffs[p] + s : payoffs[p]; PK_END
                pot_over = stop || pot_npw == 1;
            }
            if (ONE_PASS || pot_over) break;
          }
            if (pot_over) { PK_FOR(p, N) credits[p] = credits[p] + payoffs[p]; PK_END }  // :528
            else lstate = LS_POT;                                                  // another general pass, in the next call
        }
        PK_PROF(prof.lap(PF_SIDEPOT);)
        if ((e && !sd) || pot_over) {
            if (nowin) {                                                           // :473 assert (state left as the reference leaves it)
                terr |= PK_TERR_NO_WINNER; lstate = LS_DONE;
            } else {
                uint32_t broke = 0;
                hands += 1;                                                        // an end_hand() ran to its end
                PK_FOR(p, N)
                    payoffs[p] = payoffs[p] - bets[p];                             // :531
                    broke |= (credits[p] <= 0.0) ? (1u << p) : 0;                  // :536
                 PK_END
                st_active &= ~broke; st_called &= ~broke; st_allin &= ~broke; st_broken |= broke;
                setup_state(S);                                                    // :539 (shuffle: deal() below)
                const bool go = game_over();
                // fold-out: the step returns (:619).  Else TURN_OVER too (:565) and the step returns only if the game
                // is over (:610) -- or if it never could: (a) dead table -- every seat's credits are exactly 0 and no
                // seat is ACTIVE, so each further hand is a zero-chip showdown that re-creates this very state;
                // (b) backstop: PK_HAND_CAP hands inside one step.  (Selects: one basic block up to the deal.)
                bool chips = false;
                PK_FOR(p, N) chips = chips || credits[p] != 0.0; PK_END
                const bool dead = !chips && st_active == 0;
                const bool capped = !foldout && !go && (dead || hands_this_step > PK_HAND_CAP);
                flags = (go ? PK_FLAG_GAME_OVER : 0) | PK_FLAG_HAND_OVER | (foldout ? 0 : PK_FLAG_TURN_OVER);
                terr |= capped ? PK_TERR_HAND_CAP : 0;
                lstate = (foldout || go || capped) ? (int)LS_DONE : (int)LS_SCAN;
                if (auto_reset && lstate == LS_DONE && (go || (terr & PK_TERR_HAND_CAP))) {
                    seen |= terr; terr = 0;
                    flags |= PK_FLAG_GAME_OVER;   // counted as a finished game by the caller (k_rollout's retire)
                    hand_serial += 1;      // the deck setup_hand() shuffled for the dead game is never looked at
                    load_fresh(lds.fresh);
                }
                PK_PROF(prof.lap(PF_SETUP);)
                if constexpr (ONE_PASS) deal(S, table_id);                         // :424
                else need_deal = true;                                             // (dealt below, outside the divergent branch)
            }
        }
        if constexpr (!ONE_PASS) {
            // :424 for the kernels with single-table tails.  ONE table deals: PK_STOCK decks at once -- hand serials hs .. hs+3 on four
            // lanes, the price of one -- kept in LDS for the hands this table's step may roll on (each further hand of the tail then
            // takes its deck from the stock: a deal is a third of a lone end_block).  A deck is a pure function of (table id, hand
            // serial), so a stocked deck IS the deck the table would have dealt.
#if PK_WAVE >= 16 && !defined(PK_NO_LONE)
            const unsigned long long db = LONE ? __ballot(need_deal) : 0ull;
            if (LONE && __popcll(db) == 1) {
                const int dl = __ffsll((long long)db) - 1;
                const uint32_t tid = lane_read(table_id, dl);
                const uint64_t hs = lane_read64(hand_serial, dl);
                uint32_t idx = (uint32_t)(hs - stock_serial);
                if (!wave_uniform(stock_lane == (uint32_t)dl && hs - stock_serial < (uint64_t)PK_STOCK)) {
                    uint32_t c[W];
                    deal_cards(S, tid, hs + (uint64_t)(lane & (PK_STOCK - 1)), c);
                    __syncthreads();                                               // (earlier reads of the stock are done)
                    if (lane < PK_STOCK) { PK_FOR(w, W) lds.stock[lane][w] = c[w]; PK_END }
                    __syncthreads();
                    stock_lane = (uint32_t)dl; stock_serial = hs; idx = 0;
                }
                if (need_deal) { PK_FOR(w, W) cards[w] = lds.stock[idx][w]; PK_END hand_serial += 1; }
            } else
#endif
            if (need_deal) deal(S, table_id);
        }
        PK_PROF(prof.lap(PF_DEAL);)
    }

    // Runs the machine until every lane of the wave is DONE.  Must be called from wave-uniform control flow.
    template <typename LDS>
    __device__ __forceinline__ void run(const Hot &S, int t, uint32_t table_id, LDS &lds, bool auto_reset) {
        PK_PROF(prof.lap(PF_ACTION);)
        for (;;) {
            cursor();
            PK_PROF(prof.lap(PF_CURSOR); prof.count(PF_N_CURSOR);)
            if (!__any(parked())) break;
            end_block<false>(S, t, table_id, lds, auto_reset);   // (k_step's form: the whole side-pot loop per call, the lone-table paths)
        }
        PK_PROF(prof.lap(PF_OTHER);)
        finish_step();
    }
    __device__ __forceinline__ void finish_step() {
        step_serial += (terr & PK_TERR_NO_WINNER) ? 0 : stepped;  // RNG spec: one serial per COMPLETED Game.step
        stepped = 0;
    }
};

}  // namespace pk
