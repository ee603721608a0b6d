/* pokerl_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, not product code).
 *
 * Scalar, single-threaded C restatement of the reference hot path
 * (pokerl/game.py, pokerl/judger.py, pokerl/cards.py, pokerl/envs/game_env.py)
 * with the deck/action streams of oracle/rng_spec.py.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (pokerl_amd/, libpokerl_hip.so) never does.
 *
 * Parity pin: checked bit-for-bit against golden vectors captured from the
 * imported reference (the .npz/.json fixtures under tests/golden, made by tests/golden/make_golden.py)
 * and against the reference's own judger known-answer tests
 * (tests/pokerl/test_judger.py:11-117 -> tests/golden/judger_kat.json).
 */
#ifndef POKERL_ORACLE_H
#define POKERL_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_PLAYERS 16

/* error bits per table (also returned OR-ed) */
#define ORC_ERR_INVALID_ACTION 1 /* game.py:649-651 ValueError, no mutation */
#define ORC_ERR_NO_WINNER 2      /* game.py:473 AssertionError (state partially mutated, as in the reference) */
#define ORC_ERR_HAND_CAP 4       /* >ORC_HAND_CAP hands inside one step (reference would keep looping) */
#define ORC_HAND_CAP 4096
#define ORC_ERR_ENV_CAP 8        /* env loops auto-played more than ORC_ENV_STEP_CAP opponent steps (reference would spin on) */
#define ORC_ENV_STEP_CAP 8192

typedef struct orc_game orc_game;

orc_game *orc_create(int num_tables, int num_players, const double *start_credits /* N */, double big_blind,
                     double small_blind, uint64_t seed, uint32_t table_id_base);
void orc_destroy(orc_game *g);

/* Game.reset(dealer=..) for tables with mask[t] != 0 (mask NULL = all). game.py:397-412 */
void orc_reset(orc_game *g, const uint8_t *mask, int dealer);
/* Game.step. flags[t] = game_over | hand_over<<1 | turn_over<<2 ; err[t] = ORC_ERR_* ; returns OR of err. game.py:621-700 */
int orc_step(orc_game *g, const int32_t *actions, uint8_t *flags, uint8_t *err);
/* get_valid_actions() of the active player as a bitmask (bit a = action a). game.py:339-383 */
void orc_valid_actions(const orc_game *g, uint8_t *mask);
/* the same for seat `player` of every table (player < 0: the active player), get_valid_actions(player) game.py:339-383 */
void orc_valid_actions_for(const orc_game *g, int player, uint8_t *mask);

/* Synthetic agents of rng_spec.py: policy 0 random, 1 all-in, 2 call. Writes the action each table would take now. */
void orc_pick_actions(const orc_game *g, int policy, int32_t *actions);

/* K lockstep steps with in-library agents; auto_reset != 0 resets finished games (dealer 0).
* A table that hits ORC_ERR_HAND_CAP is treated as a finished game when auto_reset != 0 (return value still has the bit).
 * counters[0] += steps, [1] += hands played (end_hand calls that ran to their end), [2] += showdown 7-card evals, [3] += games finished */
int orc_rollout(orc_game *g, int K, int policy, int auto_reset, uint64_t *counters);

/* PokerGameEnv.reset()/step() with seat 0 controlled and opponents playing `opp_policy`. envs/game_env.py:20-53 */
void orc_env_reset(orc_game *g, const uint8_t *mask, int opp_policy);
int orc_env_step(orc_game *g, const int32_t *actions, int opp_policy, double *reward, uint8_t *done, uint8_t *hand,
                 uint8_t *err);
/* The same with one agent per seat, as PokerGameEnv(agents=[...]) has (envs/game_env.py:13-18, :25, :43, :51): seat p plays
 * policy nibble (seat_policies >> 4p) & 15 (rng_spec.py; seat 0's nibble is not used: its actions are the arguments). */
void orc_env_reset_seats(orc_game *g, const uint8_t *mask, uint64_t seat_policies);
int orc_env_step_seats(orc_game *g, const int32_t *actions, uint64_t seat_policies, double *reward, uint8_t *done, uint8_t *hand,
                       uint8_t *err);

/* State reads, table-major [T][N] (or [T]). */
enum { ORC_F_CREDITS = 0, ORC_F_BETS = 1, ORC_F_PENDING = 2, ORC_F_PAYOFFS = 3 };
void orc_get_f64(const orc_game *g, int field, double *out);
void orc_get_min_raise(const orc_game *g, double *out);
void orc_get_states(const orc_game *g, uint8_t *out);
/* [T][6] int32: active, turn, dealer, sb, bb, hand */
void orc_get_cursors(const orc_game *g, int32_t *out);
void orc_get_serials(const orc_game *g, uint64_t *hand_serial, uint64_t *step_serial);
void orc_get_errs(const orc_game *g, uint8_t *out);   /* error bits of each table's last Game.step (incl. one made by orc_env_reset) */
/* resume a table's RNG streams at given serials (rng_spec; either pointer may be NULL) */
void orc_set_serials(orc_game *g, const uint64_t *hand_serial, const uint64_t *step_serial);
void orc_get_cards(const orc_game *g, uint8_t *out /* [T][5+2N] Card.value */);
void orc_get_showdown(const orc_game *g, uint8_t *rank, uint32_t *kick /* [T][N], last showdown */);

/* judger.eval_hand on M hands of ncards[i] (0..7) cards (Card.value bytes, row stride 7). judger.py:7-99
 * kick = get_kickers_value(kickers) (judger.py:101-109), nkick = len(kickers). */
void orc_eval_hands(const uint8_t *cards, const uint8_t *ncards, size_t m, uint8_t *rank, uint32_t *kick, uint8_t *nkick);
/* judger.compare_rankings on one list. judger.py:111-158. Returns number of winners. */
int orc_compare_rankings(const uint8_t *rank, const uint32_t *kick, int n, uint8_t *onehot);

/* Exhaustive 7-card digest (definition: tests/golden/make_eval_digest.py): all C(52,7) hands whose FIRST (lowest)
 * canonical index is in [first_lo, first_hi).  per_first[52] partial digests (mod 2^64), counts[11] per category. */
void orc_eval7_digest(int first_lo, int first_hi, uint64_t *per_first, uint64_t *counts);
/* v = rank<<20|kick for the hands of one (a,b) prefix in lexicographic order; returns how many were written. */
size_t orc_eval7_prefix(int a, int b, uint32_t *out);

/* Spec helpers exposed for tests */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void orc_deck(uint64_t seed, uint32_t table_id, uint64_t hand_serial, uint8_t out[52]);
double orc_np_sum(const double *a, int n);

#ifdef __cplusplus
}
#endif
#endif
