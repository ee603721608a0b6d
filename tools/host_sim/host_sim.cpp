// Dev-only host simulation of the device step machine (one table per "wave") diffed against the CPU oracle.
//   g++ -std=c++20 -O1 -ffp-contract=off -I. tools/host_sim/host_sim.cpp oracle/pokerl_oracle.c -o /tmp/host_sim
#include "hip_shim.h"
#define hip_runtime_h_shimmed
#include <cstdio>
#include <cstdlib>
#include <vector>
// keep pk_device.hpp from including the real HIP runtime
#define HIP_INCLUDE_HIP_HIP_RUNTIME_H
#include "../../pokerl_amd/csrc/pk_device.hpp"
extern "C" {
#include "../../oracle/pokerl_oracle.h"
}
using namespace pk;

struct Cfg { double start[16]; double bb, sb; uint32_t base; int dealer; };
static Cfg default_cfg() { Cfg c; for (int p = 0; p < 16; ++p) c.start[p] = 100.0; c.bb = 2; c.sb = 1; c.base = 0; c.dealer = 0; return c; }

template <int N>
static int run(int T, int K, int policy, uint64_t seed, Cfg cfg = default_cfg(), bool quiet = false) {
    const int KK = 5 + 2 * N, W = (KK + 3) / 4;
    std::vector<double> cr(N * T), be(N * T), pe(N * T), pa(N * T), mr(T);
    std::vector<uint64_t> ss(T, (1u << N) - 1);
    std::vector<uint32_t> cur(T), cards(W * T), show(N * T, NONE_V);
    std::vector<uint64_t> hs(T), st(T);
    std::vector<int32_t> hand(T);
    std::vector<uint8_t> valid(T), terr(T);
    std::vector<uint32_t> owed(T), mid(T);
    unsigned long long counters[4] = {0}, prof[16] = {0};
    State S{};
    S.credits = cr.data(); S.bets = be.data(); S.pending = pe.data(); S.payoffs = pa.data(); S.min_raise = mr.data();
    S.seat_states = ss.data(); S.cursors = cur.data(); S.hand = hand.data(); S.hand_serial = hs.data(); S.step_serial = st.data();
    S.owed = owed.data(); S.mid = mid.data(); S.cards = cards.data(); S.show = show.data(); S.valid = valid.data(); S.terr = terr.data(); S.counters = counters; S.prof = prof;
    for (int p = 0; p < N; ++p) S.start_credits[p] = cfg.start[p];
    S.big_blind = cfg.bb; S.small_blind = cfg.sb; S.key0 = (uint32_t)seed; S.key1 = (uint32_t)(seed >> 32); S.table_id_base = cfg.base; S.T = T;
    double *sc = cfg.start;
    Hot H{}; H.big_blind = cfg.bb; H.small_blind = cfg.sb; H.start_credits = sc; H.show = show.data();
    H.key0 = S.key0; H.key1 = S.key1; H.table_id_base = cfg.base; H.T = T;
    H.start_uniform = sc[0]; H.start_is_uniform = 1;
    for (int p = 1; p < N; ++p) if (sc[p] != sc[0]) H.start_is_uniform = 0;
    Fresh fr{};
    {
        Table<N> f0; f0.blank(); f0.reset_state(H, 0);
        for (int p = 0; p < N; ++p) { fr.credits[p] = f0.credits[p]; fr.pending[p] = f0.pending[p]; }
        fr.min_raise = f0.min_raise; fr.st_active = f0.st_active; fr.st_called = f0.st_called; fr.st_allin = f0.st_allin; fr.st_broken = f0.st_broken;
        fr.active = f0.active; fr.dealer = f0.dealer; fr.sb = f0.sb; fr.bb = f0.bb;
    }
    H.fresh = &fr;
    orc_game *o = orc_create(T, N, sc, cfg.bb, cfg.sb, seed, cfg.base);
    orc_reset(o, nullptr, cfg.dealer);
    static Lds<N> lds;
    Table<N>::stage_fresh(lds, H.fresh);
    std::vector<Table<N>> tb(T);
    for (int t = 0; t < T; ++t) { tb[t].load(S, t); tb[t].reset_state(H, cfg.dealer % N); tb[t].deal(H, cfg.base + (uint32_t)t); tb[t].store(S, t); }
    std::vector<double> oc(N * T), ob(N * T), op(N * T), oy(N * T);
    std::vector<uint8_t> ost(N * T);
    std::vector<int32_t> ocur(6 * T);
    for (int k = 0; k < K; ++k) {
        uint64_t c4[4] = {0};
        orc_rollout(o, 1, policy, 1, c4);
        for (int t = 0; t < T; ++t) {
            Table<N> &x = tb[t];
            if (k & 1) x.template load<false>(S, t); else x.load(S, t);          // (k_step: payoffs are written back only where a hand ended)
            double hb; uint32_t mask = x.valid_mask(hb);
            ActionRing ring;   // the LDS ring of k_rollout (one lane)
            uint32_t draw = policy == 0 ? ActionRing::half_of(ring.draw16(lds, H, cfg.base + (uint32_t)t, x.step_serial, true), x.step_serial) : 0;
            x.begin_step(H, policy == 1 ? (int)MV_ALL_IN : action_from_draw(draw, mask), hb);
            if (k & 1) x.run(H, t, cfg.base + (uint32_t)t, lds, true);             // k_step's driver ...
            else {                                                                 // ... and k_rollout's betting pass
                for (;;) {
                    x.scan_first(); x.cursor_tail();
                    if (!x.parked()) break;
                    x.end_block(H, t, cfg.base + (uint32_t)t, lds, true);
                }
                x.finish_step();
            }
            if (k & 1) x.template store<false>(S, t); else x.store(S, t);
        }
        orc_get_f64(o, 0, oc.data()); orc_get_f64(o, 1, ob.data()); orc_get_f64(o, 2, op.data()); orc_get_f64(o, 3, oy.data());
        orc_get_states(o, ost.data()); orc_get_cursors(o, ocur.data());
        for (int t = 0; t < T; ++t)
            for (int p = 0; p < N; ++p) {
                bool bad = memcmp(&oc[t * N + p], &cr[(size_t)p * T + t], 8) || memcmp(&ob[t * N + p], &be[(size_t)p * T + t], 8) ||
                           memcmp(&op[t * N + p], &pe[(size_t)p * T + t], 8) || memcmp(&oy[t * N + p], &pa[(size_t)p * T + t], 8);
                uint64_t s = ss[t];
                int stt = ((s >> p) & 1) ? 1 : ((s >> (16 + p)) & 1) ? 2 : ((s >> (32 + p)) & 1) ? 3 : ((s >> (48 + p)) & 1) ? 4 : 0;
                bad = bad || stt != ost[t * N + p] || (int)(cur[t] & 0xf) != ocur[6 * t];
                if (bad) {
                    printf("MISMATCH N=%d step %d table %d seat %d: credits %.17g/%.17g bets %.17g/%.17g pend %.17g/%.17g pay %.17g/%.17g state %d/%d active %d/%d turn %d/%d\n",
                           N, k, t, p, oc[t * N + p], cr[(size_t)p * T + t], ob[t * N + p], be[(size_t)p * T + t], op[t * N + p], pe[(size_t)p * T + t],
                           oy[t * N + p], pa[(size_t)p * T + t], ost[t * N + p], stt, ocur[6 * t], cur[t] & 0xf, ocur[6 * t + 1], (cur[t] >> 16) & 0xf);
                    return 1;
                }
            }
    }
    if (!quiet) printf("N=%d T=%d K=%d policy=%d: host-sim == oracle\n", N, T, K, policy);
    orc_destroy(o);
    return 0;
}

// exhaustive: eval7_distinct (host build of the device function) vs the oracle over all C(52,7) hands
static int check_eval7(const uint32_t *tab = nullptr) {
    std::vector<uint32_t> out(2200000);
    size_t total = 0, bad = 0;
    for (int a = 0; a < 52; ++a)
        for (int b = a + 1; b < 52; ++b) {
            size_t n = orc_eval7_prefix(a, b, out.data());
            size_t i = 0;
            auto canon = [](int c) { return (uint32_t)(((c % 4) << 4) | (c / 4)); };
            for (int c = b + 1; c < 52; ++c) for (int d = c + 1; d < 52; ++d) for (int e = d + 1; e < 52; ++e)
            for (int f = e + 1; f < 52; ++f) for (int g = f + 1; g < 52; ++g) {
                uint32_t h[7] = {canon(a), canon(b), canon(c), canon(d), canon(e), canon(f), canon(g)};
                uint32_t v = eval7_distinct(h);
                if (tab) {   // the table-driven variant of the streaming evaluator, cards in any order (rotate by the hand index)
                    uint32_t r[7]; for (int j = 0; j < 7; ++j) r[j] = h[(j + i) % 7];
                    v = eval7_tab(r[0] | r[1] << 8 | r[2] << 16 | r[3] << 24, r[4] | r[5] << 8 | r[6] << 16 | 0xAB000000u, tab);
                }
                if (v != out[i]) { if (bad++ < 5) printf("eval7 MISMATCH %d %d %d %d %d %d %d: %x vs %x\n", a, b, c, d, e, f, g, v, out[i]); }
                ++i; ++total;
            }
            if (i != n) { printf("count mismatch\n"); return 1; }
        }
    printf("%s vs oracle: %zu hands, %zu mismatches\n", tab ? "eval7_tab" : "eval7_distinct", total, bad);
    return bad != 0;
}

template <int N>
static int fuzz_one(uint64_t r, int T, int K) {
    auto next = [&]() { r ^= r << 13; r ^= r >> 7; r ^= r << 17; return r; };
    static const double stacks[] = {0.5, 1, 2, 3, 5, 10, 37.5, 100, 1000, 1e6}, blinds[] = {0, 0.25, 0.5, 1, 2, 3, 7.5, 40, 250};
    Cfg c;
    bool same = next() % 2;
    double s0 = stacks[next() % 10];
    for (int p = 0; p < 16; ++p) c.start[p] = same ? s0 : stacks[next() % 10];
    c.bb = blinds[next() % 9]; c.sb = blinds[next() % 9];
    c.base = (uint32_t)next(); c.dealer = (int)(next() % N);
    int policy = next() % 4 == 0 ? 1 : 0;
    uint64_t seed = next();
    int rc = run<N>(T, K, policy, seed, c, true);
    if (rc) printf("  ^ fuzz config: N=%d policy=%d bb=%g sb=%g start0=%g same=%d dealer=%d seed=%llu base=%u\n", N, policy, c.bb, c.sb, c.start[0], (int)same, c.dealer, (unsigned long long)seed, c.base);
    return rc;
}
static int fuzz(int rounds, int T, int K) {
    int bad = 0;
    for (int i = 0; i < rounds; ++i) {
        uint64_t r = 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1);
        switch (i % 15) {
            case 0: bad += fuzz_one<2>(r, T, K); break; case 1: bad += fuzz_one<3>(r, T, K); break;
            case 2: bad += fuzz_one<4>(r, T, K); break; case 3: bad += fuzz_one<5>(r, T, K); break;
            case 4: bad += fuzz_one<6>(r, T, K); break; case 5: bad += fuzz_one<7>(r, T, K); break;
            case 6: bad += fuzz_one<8>(r, T, K); break; case 7: bad += fuzz_one<9>(r, T, K); break;
            case 8: bad += fuzz_one<10>(r, T, K); break; case 9: bad += fuzz_one<11>(r, T, K); break;
            case 10: bad += fuzz_one<12>(r, T, K); break; case 11: bad += fuzz_one<13>(r, T, K); break;
            case 12: bad += fuzz_one<14>(r, T, K); break; case 13: bad += fuzz_one<15>(r, T, K); break;
            default: bad += fuzz_one<16>(r, T, K); break;
        }
    }
    printf("fuzz: %d configs, %d mismatching\n", rounds, bad);
    return bad != 0;
}

// eval_distinct_n (the fast path of pk_eval_hands) against eval_hand (the literal scan) on EVERY n-card subset of the deck,
// n = 3 .. 7 (7: 133 784 560 hands; pass a smaller max n for a quick run), cards in a rotated order so that positions vary
static int check_evaln(int nmax, const uint32_t *tab = nullptr) {   // tab: check eval_tab_n (the table path of pk_eval_hands) instead
    auto canon = [](int k) { return (uint32_t)(((k % 4) << 4) | (k / 4)); };
    long long bad = 0, total = 0;
    for (int n = 3; n <= nmax; ++n) {
        int idx[7];
        for (int i = 0; i < n; ++i) idx[i] = i;
        long long cnt = 0;
        for (;;) {
            uint32_t c[7] = {0, 0, 0, 0, 0, 0, 0};
            for (int i = 0; i < n; ++i) c[(i + (int)(cnt % 7)) % n] = canon(idx[i]);
            int nk0 = -1, nk1 = -1;
            uint32_t v1;
            const uint32_t v0 = eval_hand(c, n, nk0);
            if (tab) {                                          // the packed front end of k_eval_hands_tab (garbage in the unused bytes)
                uint64_t w = 0xABull << 56, bits = 0;
                for (int i = 0; i < 7; ++i) w |= (uint64_t)(i < n ? c[i] : (uint32_t)(0x5Au + 37u * (uint32_t)cnt + (uint32_t)i) & 0xffu) << (8 * i);
                if (!tab_bits_of(w, n, bits)) { printf("tab_bits_of refuses a valid hand\n"); ++bad; }
                v1 = eval_tab_bits(bits, tab, nk1);
                if (n == 7) {                                   // the ncards == NULL instantiation of the kernel
                    uint64_t b7; int nk7 = -1;
                    if (!tab_bits_of<true>(w, 7, b7) || eval_tab_bits<true>(b7, tab, nk7) != v1 || nk7 != nk1) { printf("SEVEN variant differs\n"); ++bad; }
                }
                if (!distinct_valid_cards(c, n)) { printf("distinct_valid_cards refuses a valid hand\n"); ++bad; }
            } else v1 = eval_hand_any(c, n, nk1);
            if (v0 != v1 || nk0 != nk1) {
                if (bad < 5) printf("MISMATCH n=%d hand %lld: scan %08x nk %d, fast %08x nk %d\n", n, cnt, v0, nk0, v1, nk1);
                ++bad;
            }
            ++cnt;
            int i = n - 1;
            while (i >= 0 && idx[i] == 52 - n + i) --i;
            if (i < 0) break;
            ++idx[i];
            for (int j = i + 1; j < n; ++j) idx[j] = idx[j - 1] + 1;
        }
        printf("n=%d: %lld hands\n", n, cnt);
        total += cnt;
    }
    // hands that repeat a card must take the literal scan (and 0..2 cards its first lines): spot-check the dispatch
    uint32_t d[7] = {0x20, 0x20, 0x0c, 0x1c, 0x2c, 0x3c, 0x01};
    int a, b;
    if (eval_hand(d, 7, a) != eval_hand_any(d, 7, b) || a != b) { printf("MISMATCH on a hand with a repeated card\n"); ++bad; }
    for (int n = 0; n <= 2; ++n) if (eval_hand(d, n, a) != eval_hand_any(d, n, b) || a != b) { printf("MISMATCH n=%d\n", n); ++bad; }
    // ... and so must a byte that is no card (suit > 3 or rank nibble 13..15: it would alias onto a real card in the bitmask)
    const uint32_t odd[4][7] = {{0x4c, 0x0c, 0x1c, 0x2c, 0x3c, 0x01, 0x02}, {0x0d, 0x00, 0x1c, 0x2c, 0x3c, 0x01, 0x02},
                                {0x0f, 0x1f, 0x2f, 0x3f, 0x01, 0x02, 0x03}, {0x80, 0x00, 0x10, 0x20, 0x30, 0x05, 0x06}};
    for (int k = 0; k < 4; ++k)
        for (int n = 3; n <= 7; ++n) {
            if (distinct_valid_cards(odd[k], n)) { printf("distinct_valid_cards accepts a malformed byte (case %d, n %d)\n", k, n); ++bad; }
            uint64_t w = 0, bits;
            for (int i = 0; i < 7; ++i) w |= (uint64_t)odd[k][i] << (8 * i);
            if (tab_bits_of(w, n, bits)) { printf("tab_bits_of accepts a malformed byte (case %d, n %d)\n", k, n); ++bad; }
            if (eval_hand(odd[k], n, a) != eval_hand_any(odd[k], n, b) || a != b) { printf("MISMATCH on a malformed byte (case %d, n %d)\n", k, n); ++bad; }
        }
    // eval_small = the scan on every 0-, 1-, 2-card hand over all byte values
    for (int n = 0; n <= 2; ++n)
        for (uint32_t x = 0; x < 256; ++x)
            for (uint32_t y = 0; y < 256; ++y) {
                const uint32_t c2[7] = {x, y, 0x77, 0, 0, 0, 0};
                int ka = -1, kb = -1;
                if (eval_hand(c2, n, ka) != eval_small((uint64_t)x | ((uint64_t)y << 8) | 0x770000ull, n, kb) || ka != kb) {
                    if (bad < 5) printf("eval_small differs: n %d bytes %02x %02x\n", n, x, y);
                    ++bad;
                }
            }
    // tab_bits_of: EVERY byte value in EVERY position of a few valid hands, n = 3..7: accepted iff it is a real card no other used position holds
    {
        const uint32_t base[3][7] = {{0x00, 0x11, 0x22, 0x33, 0x04, 0x15, 0x26}, {0x3c, 0x2c, 0x1c, 0x0c, 0x0b, 0x1b, 0x2b}, {0x30, 0x31, 0x32, 0x33, 0x34, 0x35, 0x3c}};
        for (int k = 0; k < 3; ++k)
            for (int n = 3; n <= 7; ++n)
                for (int pos = 0; pos < 7; ++pos)
                    for (uint32_t byte = 0; byte < 256; ++byte) {
                        uint32_t c[7];
                        for (int i = 0; i < 7; ++i) c[i] = base[k][i];
                        c[pos] = byte;
                        uint64_t w = 0, bits;
                        for (int i = 0; i < 7; ++i) w |= (uint64_t)c[i] << (8 * i);
                        bool want = true;                       // (a byte outside the hand does not matter)
                        for (int i = 0; i < n; ++i) {
                            want = want && c[i] < 0x40 && (c[i] & 15) < 13;
                            for (int j = 0; j < i; ++j) want = want && c[i] != c[j];
                        }
                        if (tab_bits_of(w, n, bits) != want || distinct_valid_cards(c, n) != want || (n == 7 && tab_bits_of<true>(w, 7, bits) != want)) {
                            if (bad < 5) printf("tab_bits_of / distinct_valid_cards wrong: hand %d n %d pos %d byte %02x\n", k, n, pos, byte);
                            ++bad;
                        }
                    }
    }
    printf("evaln: %lld hands, %lld mismatching\n", total, bad);
    return bad != 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "evaln")) return check_evaln(argc > 2 ? atoi(argv[2]) : 7);
    if (argc > 1 && !strcmp(argv[1], "evalntab")) {
        std::vector<uint32_t> tab(EVAL7_TAB_WORDS);
        for (int m = 0; m < EVAL7_TAB_WORDS; ++m) tab[m] = eval7_tab_entry((uint32_t)m);
        return check_evaln(argc > 2 ? atoi(argv[2]) : 7, tab.data());
    }
    if (argc > 1 && !strcmp(argv[1], "eval7")) return check_eval7();
    if (argc > 1 && !strcmp(argv[1], "eval7tab")) {
        std::vector<uint32_t> tab(EVAL7_TAB_WORDS);
        for (int m = 0; m < EVAL7_TAB_WORDS; ++m) tab[m] = eval7_tab_entry((uint32_t)m);
        return check_eval7(tab.data());
    }
    if (argc > 1 && !strcmp(argv[1], "fuzz")) return fuzz(argc > 2 ? atoi(argv[2]) : 90, argc > 3 ? atoi(argv[3]) : 64, argc > 4 ? atoi(argv[4]) : 300);
    int T = argc > 1 ? atoi(argv[1]) : 256, K = argc > 2 ? atoi(argv[2]) : 400;
    int rc = 0;
    rc |= run<2>(T, K, 0, 0x706F6B65726Cull);
    rc |= run<6>(T, K, 0, 0x706F6B65726Cull);
    rc |= run<9>(T, K, 1, 0x706F6B65726Cull);
    rc |= run<3>(T, K, 0, 7);
    rc |= run<10>(T, K, 0, 99);
    rc |= run<13>(T, K, 0, 1313);
    rc |= run<15>(T, K, 1, 1515);
    rc |= run<16>(T, K, 0, 1616);
    rc |= run<16>(T, K, 1, 1616);
    return rc;
}
