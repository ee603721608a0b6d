/* pokerl_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, not product code).  See pokerl_oracle.h.
 *
 * Literal restatement: control flow, operation order and every quirk of the
 * reference are kept.  All money is IEEE binary64, no
 * FMA contraction (build with -ffp-contract=off), numpy's np.sum association
 * order is reproduced by orc_np_sum.  File:line citations are relative to
 * /root/reference/.
 */
#include "pokerl_oracle.h"

#include <stdlib.h>
#include <string.h>

/* enums.py:9-18, :104-114, :130-136 */
enum { HR_SF = 1, HR_POKER = 2, HR_FULL = 3, HR_FLUSH = 4, HR_STRAIGHT = 5, HR_TRIS = 6, HR_TWO_PAIR = 7, HR_PAIR = 8, HR_HIGH = 9, HR_NONE = 10 };
enum { MV_FOLD = 0, MV_CHECK = 1, MV_CALL = 2, MV_RAISE_ANY = 3, MV_ALL_IN = 6, MV_NUM = 7 };
enum { PS_FOLDED = 0, PS_ACTIVE = 1, PS_CALLED = 2, PS_ALL_IN = 3, PS_BROKEN = 4 };
enum { RANK_FIVE = 4, RANK_ACE = 13, NUM_SUITS = 4 };

#define STREAM_DECK 0x4445434Bu
#define STREAM_ACTION 0x41435432u

typedef struct {
    uint8_t deck[52]; /* Card.value, game.py:253 */
    int turn, hand, big_blind_idx, small_blind_idx, active_player, dealer_idx; /* game.py:254-258,251 */
    uint8_t states[ORC_MAX_PLAYERS];                                        /* game.py:259 */
    double credits[ORC_MAX_PLAYERS], bets[ORC_MAX_PLAYERS], pending[ORC_MAX_PLAYERS], payoffs[ORC_MAX_PLAYERS];
    double minimum_raise_value; /* game.py:263 */
    uint32_t table_id;
    uint64_t hand_serial, step_serial; /* rng_spec: 64-bit serials */
    uint8_t srank[ORC_MAX_PLAYERS];
    uint32_t skick[ORC_MAX_PLAYERS];
    uint8_t err;
    int hands_this_step;
    uint64_t evals, games, hands; /* hands = end_hand() calls that ran to their end */
} table_t;

struct orc_game {
    int T, N;
    double start_credits[ORC_MAX_PLAYERS];
    double big_blind, small_blind;
    uint64_t seed;
    table_t *t;
};

/* ------------------------------------------------------------------ RNG spec (oracle/rng_spec.py) */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void orc_deck(uint64_t seed, uint32_t table_id, uint64_t hand_serial, uint8_t out[52]) {
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint8_t remaining[52];
    int nrem = 52;
    uint64_t words[2] = {0, 0}, x = 0;
    for (int i = 0; i < 52; ++i) remaining[i] = (uint8_t)(((i % 4) << 4) | (i / 4)); /* cards.py:77 */
    for (int i = 0; i < 52; ++i) {
        if (i % 18 == 0) {
            uint32_t ctr[4] = {table_id, (uint32_t)hand_serial, STREAM_DECK + (uint32_t)(i / 18), (uint32_t)(hand_serial >> 32)}, w[4];
            orc_philox4x32_10(ctr, key, w);
            words[0] = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
            words[1] = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
        }
        if (i % 9 == 0) x = words[(i / 9) % 2];
        unsigned __int128 p = (unsigned __int128)x * (unsigned)(52 - i);
        int c = (int)(p >> 64);
        x = (uint64_t)p;
        out[i] = remaining[c];
        for (int k = c; k + 1 < nrem; ++k) remaining[k] = remaining[k + 1];
        --nrem;
    }
}

static int pick_action(uint64_t seed, const table_t *t, int policy, unsigned mask) {
    if (policy == 1) return MV_ALL_IN;
    if (policy == 2) return ((mask >> MV_CALL) & 1) ? MV_CALL : (((mask >> MV_CHECK) & 1) ? MV_CHECK : MV_ALL_IN); /* call agent, rng_spec.py */
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, o[4];
    uint64_t q = t->step_serial >> 3;
    int j = (int)(t->step_serial & 7);
    uint32_t ctr[4] = {t->table_id, (uint32_t)q, STREAM_ACTION, (uint32_t)(q >> 32)};
    orc_philox4x32_10(ctr, key, o);
    uint32_t r = (o[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
    int n = __builtin_popcount(mask);
    int k = (int)((r * (uint32_t)n) >> 16);
    for (int a = 0; a < MV_NUM; ++a)
        if ((mask >> a) & 1) { if (k == 0) return a; --k; }
    return -1;
}

/* ------------------------------------------------------------------ numpy reductions */
/* np.sum over a contiguous f64[n]: n<8 left-to-right; n>=8 eight strided partial sums combined
 * ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the tail left-to-right (numpy pairwise_sum, n<=128). */
double orc_np_sum(const double *a, int n) {
    if (n <= 0) return 0.0;
    if (n < 8) {
        double res = a[0];
        for (int i = 1; i < n; ++i) res = res + a[i];
        return res;
    }
    double r[8];
    int i;
    for (i = 0; i < 8; ++i) r[i] = a[i];
    for (i = 8; i + 8 <= n; i += 8)
        for (int j = 0; j < 8; ++j) r[j] = r[j] + a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res = res + a[i];
    return res;
}

static double np_max(const double *a, int n) {
    double m = a[0];
    for (int i = 1; i < n; ++i) if (a[i] > m) m = a[i];
    return m;
}

/* ------------------------------------------------------------------ cards.py */
static inline int card_rank(uint8_t v) { int r = v & 0xf; return r ? r : 13; } /* cards.py:14 */
static inline int card_suit(uint8_t v) { return v >> 4; }                        /* cards.py:20 */

/* ------------------------------------------------------------------ judger.py */
typedef struct { int rank; int kick[7]; int nk; } ranking_t;

static uint32_t kickers_value(const ranking_t *r) { /* judger.py:101-109 */
    uint32_t v = 0;
    for (int i = 0; i < r->nk; ++i) v |= (uint32_t)r->kick[r->nk - 1 - i] << (i << 2);
    return v;
}

/* sorted(hand, key=..., reverse=True): stable, descending (judger.py:38-39) */
static void sort_desc(const uint8_t *hand, int n, int by_suit, uint8_t *out) {
    for (int i = 0; i < n; ++i) {
        uint8_t c = hand[i];
        int key = by_suit ? ((card_suit(c) << 4) | card_rank(c)) : card_rank(c);
        int j = i;
        while (j > 0) {
            uint8_t p = out[j - 1];
            int pk = by_suit ? ((card_suit(p) << 4) | card_rank(p)) : card_rank(p);
            if (pk >= key) break;
            out[j] = p; --j;
        }
        out[j] = c;
    }
}

static int others(const uint8_t *rank_sorted, int n, int ex0, int ex1, int count, int *dst) {
    /* islice((c.rank for c in rank_sorted if c.rank != ex0 [and != ex1]), count) */
    int k = 0;
    for (int i = 0; i < n && k < count; ++i) {
        int r = card_rank(rank_sorted[i]);
        if (r != ex0 && r != ex1) dst[k++] = r;
    }
    return k;
}

static void eval_hand(const uint8_t *hand, int n, ranking_t *out) { /* judger.py:7-99 */
    out->nk = 0;
    if (n == 0) { out->rank = HR_NONE; return; }                                  /* :30 */
    if (n == 1) { out->rank = HR_HIGH; out->kick[0] = card_rank(hand[0]); out->nk = 1; return; } /* :31 */
    if (n == 2) {                                                                  /* :32-35 */
        int a = card_rank(hand[0]), b = card_rank(hand[1]);
        if (a == b) { out->rank = HR_PAIR; out->kick[0] = a; out->nk = 1; }
        else { out->rank = HR_HIGH; out->kick[0] = a > b ? a : b; out->kick[1] = a > b ? b : a; out->nk = 2; }
        return;
    }
    uint8_t rank_sorted[8], suit_sorted[8];
    sort_desc(hand, n, 0, rank_sorted);
    sort_desc(hand, n, 1, suit_sorted);
    int flush = NUM_SUITS, both = NUM_SUITS, kind = 0, straight = 0, flush_start = 0; /* :41-45 */
    int four[4], three[4], two[4], n4 = 0, n3 = 0, n2 = 0;
    for (int idx = 0; idx < n; ++idx) {                                            /* :50 */
        int rr = card_rank(rank_sorted[idx]);
        int sr = card_rank(suit_sorted[idx]), ss = card_suit(suit_sorted[idx]);
        if (ss == (flush & 0xf)) {                                                 /* :52 */
            flush += 0x100;
            if (sr + (both >> 8) == ((both >> 4) & 0xf)) both += 0x100;            /* :56 */
            else both = 0x100 | (sr << 4) | ss;                                    /* :57 */
        } else if ((flush >> 8) < 5) { both = flush = 0x100 | (sr << 4) | ss; flush_start = idx; } /* :58 */
        if (rr == (kind & 0xf)) kind += 0x10;                                      /* :61 */
        else {
            int numakind = kind >> 4;                                              /* :64-67 */
            if (numakind == 2) two[n2++] = kind & 0xf;
            else if (numakind == 3) three[n3++] = kind & 0xf;
            else if (numakind == 4) four[n4++] = kind & 0xf;
            kind = 0x10 | rr;                                                      /* :70 */
            if (rr + (straight >> 4) == (straight & 0xf)) straight += 0x10;        /* :71 */
            else if ((straight >> 4) < 5) straight = 0x10 | rr;                    /* :72 */
        }
    }
    {
        int numakind = kind >> 4;                                                  /* :77-80 */
        if (numakind == 2) two[n2++] = kind & 0xf;
        else if (numakind == 3) three[n3++] = kind & 0xf;
        else if (numakind == 4) four[n4++] = kind & 0xf;
    }
    if ((both >> 8) == 4 && ((both >> 4) & 0xf) == RANK_FIVE) {                    /* :83-85 */
        int ace = 0;
        for (int i = 0; i < n; ++i) if (card_rank(hand[i]) == RANK_ACE && card_suit(hand[i]) == (both & 0xf)) ace = 1;
        if (ace) { out->rank = HR_SF; out->kick[0] = RANK_FIVE; out->nk = 1; return; }
    } else if ((straight >> 4) == 4 && (straight & 0xf) == RANK_FIVE) {            /* :86-88 */
        int ace = 0;
        for (int i = 0; i < n; ++i) if (card_rank(hand[i]) == RANK_ACE) ace = 1;
        if (ace) { out->rank = HR_STRAIGHT; out->kick[0] = RANK_FIVE; out->nk = 1; return; }
    }
    if ((both >> 8) >= 5) { out->rank = HR_SF; out->kick[0] = (both >> 4) & 0xf; out->nk = 1; }          /* :90 */
    else if (n4) { out->rank = HR_POKER; out->kick[0] = four[0]; out->nk = 1 + others(rank_sorted, n, four[0], -1, 1, out->kick + 1); } /* :91 */
    else if (n3 > 1) { out->rank = HR_FULL; out->kick[0] = three[0]; out->kick[1] = three[1]; out->nk = 2; } /* :92 */
    else if (n3 && n2) { out->rank = HR_FULL; out->kick[0] = three[0]; out->kick[1] = two[0]; out->nk = 2; } /* :93 */
    else if ((flush >> 8) >= 5) {                                                  /* :94 */
        out->rank = HR_FLUSH;
        for (int i = flush_start; i < flush_start + 5 && i < n; ++i) out->kick[out->nk++] = card_rank(suit_sorted[i]);
    } else if ((straight >> 4) >= 5) { out->rank = HR_STRAIGHT; out->kick[0] = straight & 0xf; out->nk = 1; } /* :95 */
    else if (n3) { out->rank = HR_TRIS; out->kick[0] = three[0]; out->nk = 1 + others(rank_sorted, n, three[0], -1, 2, out->kick + 1); } /* :96 */
    else if (n2 > 1) { out->rank = HR_TWO_PAIR; out->kick[0] = two[0]; out->kick[1] = two[1]; out->nk = 2 + others(rank_sorted, n, two[0], two[1], 1, out->kick + 2); } /* :97 */
    else if (n2) { out->rank = HR_PAIR; out->kick[0] = two[0]; out->nk = 1 + others(rank_sorted, n, two[0], -1, 3, out->kick + 1); } /* :98 */
    else { out->rank = HR_HIGH; for (int i = 0; i < 5 && i < n; ++i) out->kick[out->nk++] = card_rank(rank_sorted[i]); } /* :99 */
}

void orc_eval_hands(const uint8_t *cards, const uint8_t *ncards, size_t m, uint8_t *rank, uint32_t *kick, uint8_t *nkick) {
    for (size_t i = 0; i < m; ++i) {
        ranking_t r;
        eval_hand(cards + 7 * i, ncards ? ncards[i] : 7, &r);
        rank[i] = (uint8_t)r.rank;
        kick[i] = kickers_value(&r);
        if (nkick) nkick[i] = (uint8_t)r.nk;
    }
}

int orc_compare_rankings(const uint8_t *rank, const uint32_t *kick, int n, uint8_t *onehot) { /* judger.py:111-158 */
    int winners[64], nw = 0;
    int best_rank = HR_NONE;
    uint32_t best_kicker = 0;
    for (int idx = 0; idx < n; ++idx) {
        uint32_t kicker = kick[idx];
        if (rank[idx] < best_rank) { best_rank = rank[idx]; best_kicker = kicker; nw = 0; winners[nw++] = idx; } /* :140-144 */
        else if (rank[idx] == best_rank) {
            if (kicker > best_kicker) { /* :146-149 -- `kicker = best_kicker`: best_kicker is NOT raised (A.2) */
                nw = 0; winners[nw++] = idx;
            } else if (kicker == best_kicker) winners[nw++] = idx;                  /* :150-152 */
        }
    }
    for (int i = 0; i < n; ++i) onehot[i] = 0;
    for (int i = 0; i < nw; ++i) onehot[winners[i]] = 1;
    return nw;
}

/* ------------------------------------------------------------------ game.py */
static int get_first_playing(const orc_game *g, const table_t *t, int idx) { /* game.py:334-337 */
    int n = g->N;
    for (int k = 0; k < n; ++k)
        if (t->states[(idx + k) % n] != PS_BROKEN) return (idx + k) % n;
    return idx % n; /* argmax of all-False is 0 */
}

static unsigned valid_actions(const orc_game *g, const table_t *t, int player) { /* game.py:339-383 */
    double high_bet = np_max(t->pending, g->N);                                   /* :365 */
    double credit = t->credits[player];                                           /* :366 */
    static const double f[3] = {0.1, 0.25, 0.5};
    unsigned mask = (1u << MV_FOLD) | (1u << MV_ALL_IN);                           /* :367 */
    for (int k = 0; k < 3; ++k) {
        double rv = f[k] * (credit - high_bet);                                   /* :370 */
        if (rv > t->minimum_raise_value && (high_bet + rv) < credit) mask |= 1u << (MV_RAISE_ANY + k); /* :371 */
    }
    if (high_bet == 0.0) mask |= 1u << MV_CHECK;                                   /* :375 */
    if (high_bet < credit) mask |= 1u << MV_CALL;                                  /* :376 */
    return mask;
}

static void setup_hand(orc_game *g, table_t *t) { /* game.py:414-451 */
    int n = g->N;
    t->hand += 1; t->turn = 0;                                                     /* :417-418 */
    for (int p = 0; p < n; ++p) if (t->states[p] != PS_BROKEN) t->states[p] = PS_ACTIVE; /* :421 */
    orc_deck(g->seed, t->table_id, t->hand_serial, t->deck);                       /* :424 (rng_spec) */
    t->hand_serial += 1;
    t->dealer_idx = get_first_playing(g, t, t->dealer_idx + 1);                    /* :432 */
    t->small_blind_idx = get_first_playing(g, t, t->dealer_idx + 1);               /* :433 */
    t->big_blind_idx = get_first_playing(g, t, t->small_blind_idx + 1);            /* :434 */
    t->active_player = get_first_playing(g, t, t->big_blind_idx + 1);              /* :435 */
    for (int p = 0; p < n; ++p) { t->bets[p] = 0.0; t->pending[p] = 0.0; }         /* :438-439 */
    t->pending[t->big_blind_idx] = g->big_blind;                                   /* :440 fancy-index, */
    t->pending[t->small_blind_idx] = g->small_blind;                               /*      last write wins */
    t->states[t->big_blind_idx] = PS_CALLED;                                       /* :441 */
    for (int p = 0; p < n; ++p) if (t->pending[p] > t->credits[p]) t->states[p] = PS_ALL_IN; /* :444 */
    for (int p = 0; p < n; ++p) if (t->credits[p] < t->pending[p]) t->pending[p] = t->credits[p]; /* :445 np.minimum */
    t->minimum_raise_value = np_max(t->pending, n);                                /* :446 */
    t->hands_this_step += 1;
}

static void end_hand(orc_game *g, table_t *t) { /* game.py:453-539 */
    int n = g->N;
    for (int p = 0; p < n; ++p) { t->bets[p] = t->bets[p] + t->pending[p]; t->credits[p] = t->credits[p] - t->pending[p]; } /* :457-458 */
    for (int p = 0; p < n; ++p) t->pending[p] = 0.0;                               /* :460 */
    t->minimum_raise_value = 0.0;                                                  /* :461 */
    for (int p = 0; p < n; ++p) t->payoffs[p] = 0.0;                               /* :468 */
    int pw[ORC_MAX_PLAYERS], npw = 0;
    for (int p = 0; p < n; ++p) { pw[p] = t->states[p] != PS_BROKEN && t->states[p] != PS_FOLDED; npw += pw[p]; } /* :471-472 */
    if (npw <= 0) { t->err |= ORC_ERR_NO_WINNER; return; }                          /* :473 assert */
    if (npw == 1) {                                                                /* :475-480 */
        int winner = 0;
        for (int p = 0; p < n; ++p) if (pw[p]) { winner = p; break; }
        double pot = orc_np_sum(t->bets, n);
        t->payoffs[winner] = pot;
        t->credits[winner] = t->credits[winner] + pot;
    } else {
        double bets[ORC_MAX_PLAYERS];
        memcpy(bets, t->bets, sizeof(double) * n);                                 /* :485 */
        uint8_t hr[ORC_MAX_PLAYERS]; uint32_t hk[ORC_MAX_PLAYERS];
        for (int p = 0; p < n; ++p) {                                              /* :488-489 */
            ranking_t r;
            if (t->states[p] == PS_CALLED || t->states[p] == PS_ALL_IN) {
                uint8_t hand[7];
                memcpy(hand, t->deck, 5);                                          /* :394-395 */
                hand[5] = t->deck[5 + 2 * p]; hand[6] = t->deck[6 + 2 * p];
                eval_hand(hand, 7, &r);
                t->evals += 1;
            } else eval_hand(NULL, 0, &r);
            hr[p] = (uint8_t)r.rank; hk[p] = kickers_value(&r);
            t->srank[p] = hr[p]; t->skick[p] = hk[p];
        }
        int order[ORC_MAX_PLAYERS];                                                /* :495 argsort (stable; A.6) */
        for (int i = 0; i < n; ++i) {
            int j = i;
            while (j > 0 && bets[order[j - 1]] > bets[i]) { order[j] = order[j - 1]; --j; }
            order[j] = i;
        }
        for (int oi = 0; oi < n; ++oi) {                                           /* :498 */
            int player = order[oi];
            if (!(t->states[player] == PS_CALLED || t->states[player] == PS_ALL_IN)) continue; /* :496 */
            int all_le0 = 1;
            for (int p = 0; p < n; ++p) if (!(bets[p] <= 0.0)) all_le0 = 0;
            if (all_le0) break;                                                    /* :499 */
            if (npw == 1) { t->payoffs[player] = t->payoffs[player] + orc_np_sum(bets, n); break; } /* :500-505 */
            double max_bet = bets[player], max_bets[ORC_MAX_PLAYERS];              /* :508-509 np.clip */
            for (int p = 0; p < n; ++p) { double x = bets[p]; if (x < 0.0) x = 0.0; if (x > max_bet) x = max_bet; max_bets[p] = x; }
            uint8_t onehot[ORC_MAX_PLAYERS];
            int nw = orc_compare_rankings(hr, hk, n, onehot);                      /* :512 */
            double s = orc_np_sum(max_bets, n);
            if (nw == 1) { for (int p = 0; p < n; ++p) if (onehot[p]) t->payoffs[p] = t->payoffs[p] + s; } /* :515 */
            else for (int p = 0; p < n; ++p) t->payoffs[p] = t->payoffs[p] + (s * (double)onehot[p]) / (double)nw; /* :516 */
            hr[player] = HR_NONE; hk[player] = 0;                                  /* :522 */
            for (int p = 0; p < n; ++p) bets[p] = bets[p] - max_bets[p];           /* :523 */
            npw -= 1;                                                              /* :525 */
        }
        for (int p = 0; p < n; ++p) t->credits[p] = t->credits[p] + t->payoffs[p]; /* :528 */
    }
    t->hands += 1;
    for (int p = 0; p < n; ++p) t->payoffs[p] = t->payoffs[p] - t->bets[p];        /* :531 */
    for (int p = 0; p < n; ++p) if (t->credits[p] <= 0.0) t->states[p] = PS_BROKEN; /* :536 */
    setup_hand(g, t);                                                              /* :539 */
}

static int game_over(const orc_game *g, const table_t *t) { /* game.py:317-320 */
    int c = 0;
    for (int p = 0; p < g->N; ++p) c += t->states[p] != PS_BROKEN;
    return c == 1;
}

typedef struct { int game, hand, turn; } done_t;

static done_t next_turn(orc_game *g, table_t *t) { /* game.py:541-576 */
    int n = g->N;
    for (int p = 0; p < n; ++p) { t->bets[p] = t->bets[p] + t->pending[p]; t->credits[p] = t->credits[p] - t->pending[p]; } /* :554-555 */
    for (int p = 0; p < n; ++p) t->pending[p] = 0.0;
    t->minimum_raise_value = 0.0;
    t->turn += 1;                                                                  /* :561 */
    done_t d;
    if (t->turn == 4) {                                                            /* :563-565 */
        end_hand(g, t);
        d.game = game_over(g, t); d.hand = 1; d.turn = 1;
        return d;
    }
    int c = 0;
    for (int p = 0; p < n; ++p) c += t->states[p] == PS_CALLED;                    /* :567-568 */
    if (c > 1) for (int p = 0; p < n; ++p) if (t->states[p] == PS_CALLED) t->states[p] = PS_ACTIVE; /* :570-572 */
    t->active_player = get_first_playing(g, t, t->dealer_idx + 1);                 /* :575 */
    d.game = 0; d.hand = 0; d.turn = 1;
    return d;
}

static done_t next_player(orc_game *g, table_t *t) { /* game.py:578-619 */
    int n = g->N, playing = 0;
    for (int p = 0; p < n; ++p) playing += t->states[p] != PS_BROKEN && t->states[p] != PS_FOLDED; /* :598-599 */
    done_t done = {0, 0, 0};
    if (playing > 1) {
        int current_player = t->active_player;                                    /* :604 */
        t->active_player = (t->active_player + 1) % n;                             /* :605 */
        while (t->states[t->active_player] != PS_ACTIVE) {                         /* :607 */
            if (current_player == t->active_player) {
                done = next_turn(g, t);                                            /* :609 */
                if (t->err) return done;
                if (done.game) return done;                                        /* :610 */
                /* The step can never return: (a) dead table -- every seat's credits are exactly 0 and no seat is
                 * ACTIVE, so each further hand is a zero-chip showdown that re-creates this very state (all cap events
                 * seen in 4e8 default-config steps are of this kind); (b) backstop: ORC_HAND_CAP hands in one step. */
                int dead = done.hand; /* only right after end_hand + setup_hand: mid-hand, all-in seats hold 0 credits too */
                for (int p = 0; p < n; ++p) if (t->credits[p] != 0.0 || t->states[p] == PS_ACTIVE) dead = 0;
                if (dead || t->hands_this_step > ORC_HAND_CAP) { t->err |= ORC_ERR_HAND_CAP; return done; }
            } else t->active_player = (t->active_player + 1) % n;                  /* :611 */
        }
        return done;                                                               /* :615 */
    }
    end_hand(g, t);                                                                /* :618 */
    done.game = game_over(g, t); done.hand = 1; done.turn = 0;                     /* :619 */
    return done;
}

static int step_table(orc_game *g, table_t *t, int action, uint8_t *flags) { /* game.py:621-700 */
    int n = g->N, a = t->active_player;
    t->err = 0; t->hands_this_step = 0;
    unsigned mask = valid_actions(g, t, a);                                        /* :648 */
    if (action < 0 || action >= MV_NUM || !((mask >> action) & 1)) {               /* :649-651 */
        t->err = ORC_ERR_INVALID_ACTION;
        if (flags) *flags = 0;
        return t->err;
    }
    if (action == MV_FOLD) t->states[a] = PS_FOLDED;                               /* :656-657 */
    else if (action == MV_CHECK) t->states[a] = PS_CALLED;                         /* :659-660 */
    else {
        double high_bet = np_max(t->pending, n);                                   /* :664 */
        double bet_value = g->big_blind > high_bet ? g->big_blind : high_bet;      /* :665 max(high_bet, big_blind) */
        double credit = t->credits[a];                                             /* :666 */
        t->states[a] = PS_CALLED;                                                  /* :667 */
        if (action == MV_ALL_IN) { bet_value = credit; t->states[a] = PS_ALL_IN; } /* :669-671 */
        else if (action >= MV_RAISE_ANY) {                                         /* :673-678 */
            static const double f[3] = {0.1, 0.25, 0.5};
            double future_credit = credit - bet_value;
            double raise_value = future_credit * f[action - MV_RAISE_ANY];
            bet_value = bet_value + raise_value;
        }
        if (bet_value > high_bet) {                                                /* :680-687 */
            uint8_t current_state = t->states[a];
            for (int p = 0; p < n; ++p) if (t->states[p] == PS_CALLED) t->states[p] = PS_ACTIVE;
            t->states[a] = current_state;
            t->minimum_raise_value = bet_value - high_bet;
        }
        t->pending[a] = bet_value;                                                 /* :696 */
    }
    done_t d = next_player(g, t);                                                  /* :699 */
    if (!(t->err & ORC_ERR_NO_WINNER)) t->step_serial += 1; /* Game.step returned (rng_spec: one serial per completed step) */
    if (flags) *flags = (uint8_t)((d.game ? 1 : 0) | (d.hand ? 2 : 0) | (d.turn ? 4 : 0));
    return t->err;
}

static void reset_table(orc_game *g, table_t *t, int dealer) { /* game.py:397-412 */
    t->dealer_idx = dealer;                                                        /* :403 */
    t->hand = 0; t->active_player = 0;                                             /* :406-407 */
    for (int p = 0; p < g->N; ++p) { t->credits[p] = g->start_credits[p]; t->states[p] = PS_ACTIVE; } /* :408-409 */
    setup_hand(g, t);                                                              /* :412 */
}

/* ------------------------------------------------------------------ public API */
orc_game *orc_create(int T, int N, const double *start_credits, double big_blind, double small_blind, uint64_t seed,
                     uint32_t table_id_base) {
    if (N < 1 || N > ORC_MAX_PLAYERS || T < 1) return NULL;
    orc_game *g = (orc_game *)calloc(1, sizeof(*g));
    g->T = T; g->N = N; g->big_blind = big_blind; g->small_blind = small_blind; g->seed = seed;
    for (int p = 0; p < N; ++p) g->start_credits[p] = start_credits[p];
    g->t = (table_t *)calloc((size_t)T, sizeof(table_t));
    for (int i = 0; i < T; ++i) {                                                  /* game.py:242-264 */
        table_t *t = &g->t[i];
        t->table_id = table_id_base + (uint32_t)i;
        for (int c = 0; c < 52; ++c) t->deck[c] = (uint8_t)(((c % 4) << 4) | (c / 4));
        for (int p = 0; p < N; ++p) { t->states[p] = PS_ACTIVE; t->srank[p] = HR_NONE; }
    }
    return g;
}

void orc_destroy(orc_game *g) { if (g) { free(g->t); free(g); } }

void orc_reset(orc_game *g, const uint8_t *mask, int dealer) {
    for (int i = 0; i < g->T; ++i) if (!mask || mask[i]) reset_table(g, &g->t[i], dealer);
}

int orc_step(orc_game *g, const int32_t *actions, uint8_t *flags, uint8_t *err) {
    int any = 0;
    for (int i = 0; i < g->T; ++i) {
        int e = step_table(g, &g->t[i], actions[i], flags ? &flags[i] : NULL);
        if (err) err[i] = (uint8_t)e;
        any |= e;
    }
    return any;
}

void orc_valid_actions(const orc_game *g, uint8_t *mask) {
    for (int i = 0; i < g->T; ++i) mask[i] = (uint8_t)valid_actions(g, &g->t[i], g->t[i].active_player);
}

void orc_valid_actions_for(const orc_game *g, int player, uint8_t *mask) { /* get_valid_actions(player), game.py:339-383 */
    for (int i = 0; i < g->T; ++i) mask[i] = (uint8_t)valid_actions(g, &g->t[i], player < 0 ? g->t[i].active_player : player);
}

void orc_pick_actions(const orc_game *g, int policy, int32_t *actions) {
    for (int i = 0; i < g->T; ++i) {
        const table_t *t = &g->t[i];
        actions[i] = pick_action(g->seed, t, policy, valid_actions(g, t, t->active_player));
    }
}

int orc_rollout(orc_game *g, int K, int policy, int auto_reset, uint64_t *counters) {
    int any = 0;
    for (int i = 0; i < g->T; ++i) {
        table_t *t = &g->t[i];
        uint64_t h0 = t->hands; uint64_t e0 = t->evals, steps = 0, games = 0;
        for (int k = 0; k < K; ++k) {
            uint8_t fl;
            int a = pick_action(g->seed, t, policy, valid_actions(g, t, t->active_player));
            int e = step_table(g, t, a, &fl);
            any |= e;
            if (e) { /* hand cap (reference would never return): with auto_reset it counts as a finished game */
                if (!(auto_reset && e == ORC_ERR_HAND_CAP)) break;
                t->err = 0; fl = 1;
            }
            ++steps;
            if (fl & 1) { ++games; if (auto_reset) reset_table(g, t, 0); }
        }
        if (counters) { counters[0] += steps; counters[1] += t->hands - h0; counters[2] += t->evals - e0; counters[3] += games; }
    }
    return any;
}

/* envs/game_env.py:20-29 */
/* self.agents[active_player] (:25, :43, :51): policy nibble of the seat to act */
static inline int seat_policy(uint64_t seatpol, const table_t *t) { return (int)((seatpol >> (4 * t->active_player)) & 15); }

static void env_reset_table(orc_game *g, table_t *t, uint64_t seatpol) {
    reset_table(g, t, 0);                                                          /* :23 */
    int budget = ORC_ENV_STEP_CAP;
    while (t->active_player != 0) {                                                /* :24 */
        uint8_t fl;
        int a = pick_action(g->seed, t, seat_policy(seatpol, t), valid_actions(g, t, t->active_player)); /* :25 */
        int e = step_table(g, t, a, &fl);                                          /* :26 */
        if (--budget < 0) t->err |= ORC_ERR_ENV_CAP;
        if (e || t->err) return;
        if (fl & 1) reset_table(g, t, 0);                                          /* :27 */
    }
}

static uint64_t uniform_seats(int policy) { return 0x1111111111111111ull * (uint64_t)(policy & 15); }

void orc_env_reset_seats(orc_game *g, const uint8_t *mask, uint64_t seatpol) {
    for (int i = 0; i < g->T; ++i) if (!mask || mask[i]) env_reset_table(g, &g->t[i], seatpol);
}
void orc_env_reset(orc_game *g, const uint8_t *mask, int opp_policy) { orc_env_reset_seats(g, mask, uniform_seats(opp_policy)); }

int orc_env_step(orc_game *g, const int32_t *actions, int opp_policy, double *reward, uint8_t *done_out, uint8_t *hand_out,
                 uint8_t *err) {
    return orc_env_step_seats(g, actions, uniform_seats(opp_policy), reward, done_out, hand_out, err);
}

int orc_env_step_seats(orc_game *g, const int32_t *actions, uint64_t seatpol, double *reward, uint8_t *done_out, uint8_t *hand_out,
                       uint8_t *err) { /* envs/game_env.py:31-53 */
    int any = 0;
    for (int i = 0; i < g->T; ++i) {
        table_t *t = &g->t[i];
        uint8_t fl;
        double rew = 0.0;                                                          /* :34 */
        int e = step_table(g, t, actions[i], &fl);                                 /* :35 */
        int done = fl & 1, hand = (fl >> 1) & 1;
        if (e) { if (err) err[i] = (uint8_t)e; any |= e; reward[i] = 0.0; done_out[i] = 0; hand_out[i] = 0; continue; }
        if (done || t->states[0] == PS_BROKEN) {                                   /* :37-39 */
            reward[i] = t->payoffs[0]; done_out[i] = 1; hand_out[i] = 1;
            if (err) err[i] = 0;
            continue;
        }
        int budget = ORC_ENV_STEP_CAP;
        while (!hand && t->active_player != 0) {                                   /* :41-44 */
            int a = pick_action(g->seed, t, seat_policy(seatpol, t), valid_actions(g, t, t->active_player));
            e = step_table(g, t, a, &fl);
            if (--budget < 0) { t->err |= ORC_ERR_ENV_CAP; e |= ORC_ERR_ENV_CAP; }
            if (e) break;
            done = fl & 1; hand = (fl >> 1) & 1;
        }
        if (!e && hand) rew = t->payoffs[0];                                       /* :47 */
        while (!e && !done && t->active_player != 0) {                             /* :49-52 */
            int a = pick_action(g->seed, t, seat_policy(seatpol, t), valid_actions(g, t, t->active_player));
            e = step_table(g, t, a, &fl);
            if (--budget < 0) { t->err |= ORC_ERR_ENV_CAP; e |= ORC_ERR_ENV_CAP; }
            if (e) break;
            done = fl & 1;
        }
        reward[i] = rew; done_out[i] = (uint8_t)done; hand_out[i] = (uint8_t)hand; /* :53 */
        if (err) err[i] = (uint8_t)e;
        any |= e;
    }
    return any;
}

void orc_get_f64(const orc_game *g, int field, double *out) {
    for (int i = 0; i < g->T; ++i) {
        const table_t *t = &g->t[i];
        const double *src = field == ORC_F_CREDITS ? t->credits : field == ORC_F_BETS ? t->bets : field == ORC_F_PENDING ? t->pending : t->payoffs;
        memcpy(out + (size_t)i * g->N, src, sizeof(double) * g->N);
    }
}
void orc_get_min_raise(const orc_game *g, double *out) { for (int i = 0; i < g->T; ++i) out[i] = g->t[i].minimum_raise_value; }
void orc_get_states(const orc_game *g, uint8_t *out) { for (int i = 0; i < g->T; ++i) memcpy(out + (size_t)i * g->N, g->t[i].states, g->N); }
void orc_get_cursors(const orc_game *g, int32_t *out) {
    for (int i = 0; i < g->T; ++i) {
        const table_t *t = &g->t[i];
        int32_t *o = out + 6 * (size_t)i;
        o[0] = t->active_player; o[1] = t->turn; o[2] = t->dealer_idx; o[3] = t->small_blind_idx; o[4] = t->big_blind_idx; o[5] = t->hand;
    }
}
void orc_get_serials(const orc_game *g, uint64_t *hand_serial, uint64_t *step_serial) {
    for (int i = 0; i < g->T; ++i) { if (hand_serial) hand_serial[i] = g->t[i].hand_serial; if (step_serial) step_serial[i] = g->t[i].step_serial; }
}
/* error bits the last Game.step of each table left (ORC_ERR_*): how a harness sees an error raised INSIDE orc_env_reset's loop
 * (game_env.py:24-27), which returns nothing */
void orc_get_errs(const orc_game *g, uint8_t *out) {
    for (int i = 0; i < g->T; ++i) out[i] = (uint8_t)g->t[i].err;
}
void orc_set_serials(orc_game *g, const uint64_t *hand_serial, const uint64_t *step_serial) {
    for (int i = 0; i < g->T; ++i) { if (hand_serial) g->t[i].hand_serial = hand_serial[i]; if (step_serial) g->t[i].step_serial = step_serial[i]; }
}
void orc_get_cards(const orc_game *g, uint8_t *out) {
    int nc = 5 + 2 * g->N; if (nc > 52) nc = 52;
    for (int i = 0; i < g->T; ++i) memcpy(out + (size_t)i * nc, g->t[i].deck, nc);
}
void orc_get_showdown(const orc_game *g, uint8_t *rank, uint32_t *kick) {
    for (int i = 0; i < g->T; ++i) { memcpy(rank + (size_t)i * g->N, g->t[i].srank, g->N); memcpy(kick + (size_t)i * g->N, g->t[i].skick, sizeof(uint32_t) * g->N); }
}

/* ------------------------------------------------------------------ exhaustive evaluator digest */
static inline uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    return z;
}
static inline uint8_t canon(int c) { return (uint8_t)(((c % 4) << 4) | (c / 4)); }

void orc_eval7_digest(int first_lo, int first_hi, uint64_t *per_first, uint64_t *counts) {
    /* binomials for the lexicographic rank of the first hand with first card a */
    static uint64_t C[53][8];
    for (int n = 0; n <= 52; ++n) for (int k = 0; k <= 7; ++k) C[n][k] = k == 0 ? 1 : (n == 0 ? 0 : C[n - 1][k - 1] + C[n - 1][k]);
    uint64_t idx = 0;
    for (int x = 0; x < first_lo; ++x) idx += C[51 - x][6];
    uint8_t h[7];
    for (int a = first_lo; a < first_hi && a <= 45; ++a) {
        uint64_t acc = 0;
        h[0] = canon(a);
        for (int b = a + 1; b < 52; ++b) { h[1] = canon(b);
        for (int c = b + 1; c < 52; ++c) { h[2] = canon(c);
        for (int d = c + 1; d < 52; ++d) { h[3] = canon(d);
        for (int e = d + 1; e < 52; ++e) { h[4] = canon(e);
        for (int f = e + 1; f < 52; ++f) { h[5] = canon(f);
        for (int g = f + 1; g < 52; ++g) { h[6] = canon(g);
            ranking_t r;
            eval_hand(h, 7, &r);
            uint64_t v = ((uint64_t)r.rank << 20) | kickers_value(&r);
            acc += mix64(v ^ (idx * 0x9E3779B97F4A7C15ull));
            counts[r.rank] += 1;
            ++idx;
        }}}}}}
        per_first[a] += acc;
    }
}

size_t orc_eval7_prefix(int a, int b, uint32_t *out) {
    size_t k = 0;
    uint8_t h[7];
    h[0] = canon(a); h[1] = canon(b);
    for (int c = b + 1; c < 52; ++c) { h[2] = canon(c);
    for (int d = c + 1; d < 52; ++d) { h[3] = canon(d);
    for (int e = d + 1; e < 52; ++e) { h[4] = canon(e);
    for (int f = e + 1; f < 52; ++f) { h[5] = canon(f);
    for (int g = f + 1; g < 52; ++g) { h[6] = canon(g);
        ranking_t r;
        eval_hand(h, 7, &r);
        out[k++] = ((uint32_t)r.rank << 20) | kickers_value(&r);
    }}}}}
    return k;
}
