/* pokerl_hip.h -- C ABI of libpokerl_hip.so: the MI355X (gfx950) vectorised No-Limit Hold'em hot path.
 *
 * The reference (sneppy/pokerl) is pure Python and has no plugin/FFI layer; its boundary for this
 * path is the Python object API.  Each entry point below names the reference interface it replaces
 * (paths relative to the reference root).  T = num_tables, N = num_players.
 *
 * Conventions
 *   - every pointer is a HOST pointer owned by the caller unless the name ends in `_d` (device pointer,
 *     caller-owned, on the handle's device);  the library owns all table state behind the opaque handle;
 *   - arrays are table-major: [T][N] for per-seat data, [T] for per-table data;
 *   - return value: PK_OK (0) or a negative PK_E_* code; pk_last_error() gives text; the library never aborts;
 *   - per-table error bits (PK_TERR_*) are reported separately from the call's return code;
 *   - a handle is bound to one device and one HIP stream and is not thread-safe; different handles may be
 *     driven from different threads/processes (one process per GPU is the intended deployment);
 *   - RNG is counter-based per handle: Philox4x32-10 keyed by `seed`, indexed by the GLOBAL table id
 *     (table_id_base + t) and 64-bit per-table serials, so results do not depend on how tables are sharded over
 *     GPUs and a table's streams never repeat;
 *   - `_d` entry points are asynchronous on the handle's stream and take buffers that must be COMPLETE in stream order:
 *     either make the handle run on your stream (pk_set_stream) or order the two streams with events
 *     (pk_wait_event before the call, pk_record_event after it).  pk_sync() waits for everything requested so far.
 *
 * All money arithmetic is IEEE binary64 in the reference's operation order (no FMA contraction), so
 * valid_actions / payoffs / credits / flags / hand ranks are bit-identical to the CPU reference.  Money must be FINITE: the
 * library is built with -fno-honor-nans (np.max / np.minimum as v_max_f64 / v_min_f64), and pk_create refuses inf / NaN stacks
 * and blinds (PK_E_INVALID_ARG) -- an infinite stack would turn into NaN at the first all-in (inf - inf).
 */
#ifndef POKERL_HIP_H
#define POKERL_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PK_ABI_VERSION 6
#define PK_MIN_PLAYERS 2
#define PK_MAX_PLAYERS 16 /* the reference takes any num_players (game.py:246; the deck allows 23).  Up to 16 the numpy routines it
                             calls are restated exactly: np.sum's eight-lane pairwise blocks (one up to 15 seats, two at 16) and
                             np.argsort's stable insertion sort (the pinned numpy sorts up to 17 elements that way); from 18 seats on
                             argsort's order among EQUAL bets -- which decides side pots -- is no longer a rule the reference pins,
                             and a 17th seat does not fit the 16 nibbles of a policy word: DESIGN.md section 8, docs/history.md section 9.  13 ... 16 seats
                             run kernels that keep part of the table in AGPRs at one wave per SIMD (no scratch memory). */
#define PK_MAX_ENV_BATCHES 8 /* pk_set_env_batches */
#define PK_MAX_DEVICES 64 /* the handle-less judger calls keep one scratch arena per device index below this */
#define PK_NUM_MOVES 7 /* pokerl/enums.py:104-114 PokerMoves */

/* return codes */
#define PK_OK 0
#define PK_E_INVALID_ARG (-1)
#define PK_E_NO_DEVICE (-2)
#define PK_E_HIP (-3)
#define PK_E_OOM (-4)
#define PK_E_TABLE (-5) /* at least one table reported a PK_TERR_* bit; call completed for all other tables */
#define PK_E_BUSY (-6)  /* PokerGameEnv steps are in flight (pk_env_step_async_d); nothing was done: drain them first */

/* per-table error bits */
#define PK_TERR_INVALID_ACTION 1 /* Game.step ValueError, pokerl/game.py:649-651: that table is left untouched */
#define PK_TERR_NO_WINNER 2      /* Game.end_hand AssertionError, pokerl/game.py:473: state partially mutated as in the reference */
#define PK_TERR_HAND_CAP 4       /* the reference would (as good as) never return from this Game.step: either every seat's
                                    credits are exactly 0 with no seat ACTIVE after a hand rolled over (each further hand
                                    re-creates that state: detected at once), or more than PK_HAND_CAP hands were rolled
                                    inside ONE step (sane configs roll <= 6; blinds 40x the stacks: 1 315 observed) */
#define PK_HAND_CAP 4096
#define PK_TERR_ENV_CAP 8        /* PokerGameEnv.reset/step auto-played more than PK_ENV_STEP_CAP opponent steps without
                                    reaching seat 0 or the end of the game (the reference's loops, envs/game_env.py:24,
                                    :41, :49, would still be spinning -- e.g. seat 0 broke with the game not over) */
#define PK_ENV_STEP_CAP 8192

/* step flags (bit set) = the tuple Game.step returns, pokerl/game.py:634-641 */
#define PK_FLAG_GAME_OVER 1
#define PK_FLAG_HAND_OVER 2
#define PK_FLAG_TURN_OVER 4

/* in-kernel agents (synthetic workload; RandomAgent semantics of pokerl/agents/random.py:12-16) */
#define PK_POLICY_RANDOM 0 /* uniform over the valid mask */
#define PK_POLICY_ALLIN 1  /* always PokerMoves.ALL_IN */
#define PK_POLICY_CALL 2   /* the calling station: CALL if valid, else CHECK if valid, else ALL_IN (no random draw) */
#define PK_NUM_POLICIES 3
/* One agent per seat, as PokerGameEnv(agents=[...]) has (pokerl/envs/game_env.py:13-18, :25, :43, :51): a 64-bit word with
 * the policy of seat p in nibble p (bits 4p..4p+3).  PK_POLICY_EXTERNAL: the CALLER plays that seat (pk_env_step_multi_d). */
#define PK_POLICY_EXTERNAL 15
#define PK_SEAT_POLICY(word, p) ((int)(((word) >> (4 * (p))) & 15))
#define PK_ACTION_SKIP (-2) /* pk_env_step_multi_d: "no action for this (idle) table" */

/* f64 [T][N] fields of pk_get_f64 = Game attributes, pokerl/game.py:260-264 */
#define PK_F_CREDITS 0
#define PK_F_BETS 1
#define PK_F_PENDING_BETS 2
#define PK_F_PAYOFFS 3

/* int32 [T] fields of pk_get_i32 = Game attributes, pokerl/game.py:251-258 */
#define PK_I_ACTIVE_PLAYER 0
#define PK_I_TURN 1
#define PK_I_DEALER_IDX 2
#define PK_I_SMALL_BLIND_IDX 3
#define PK_I_BIG_BLIND_IDX 4
#define PK_I_HAND 5

/* f64 [T] per-table values of pk_get_table_f64 = Game properties */
#define PK_TF_POT 0       /* Game.pot: np.sum(bets) in numpy's association order, pokerl/game.py:281-284 */
#define PK_TF_HIGH_BET 1  /* Game.high_bet: np.max(pending_bets), pokerl/game.py:287-290 */
#define PK_TF_MIN_RAISE 2 /* Game.minimum_raise_value */

/* rollout counters */
#define PK_C_STEPS 0
#define PK_C_HANDS 1 /* hands played: end_hand() calls that ran to their end, pokerl/game.py:453-539 */
#define PK_C_EVALS 2 /* 7-card eval_hand calls made by showdowns, pokerl/game.py:489 */
#define PK_C_GAMES 3
#define PK_NUM_COUNTERS 4

typedef struct pk_handle pk_handle;

int pk_abi_version(void);
/* "abi=<PK_ABI_VERSION> src=<16 hex digits>": the hash pokerl_amd/build.py takes over the kernel sources and compiler flags this library was built
 * from (comments and white space excluded).  Every profiles/ *_summary.json carries the hash of the library it was measured on; bench.py marks a
 * roofline figure that rests on a summary of OTHER sources `profile_stale`. */
const char *pk_build_info(void);
/* Number of visible HIP devices (0 if none / no driver). */
int pk_device_count(void);
const char *pk_last_error(const pk_handle *h); /* h may be NULL: last create/standalone error of this thread */

/* Game(**config), pokerl/game.py:242-264.  start_credits: N doubles or NULL (then start_credit_scalar is
 * broadcast, the `isinstance(self.start_credits, int)` branch of game.py:408).  `dealer` is kept for API parity;
 * Game.reset() overwrites it (game.py:403).  Tables start un-reset (credits 0), as in the reference. */
int pk_create(pk_handle **out, int device, int num_tables, int num_players, const double *start_credits,
              double start_credit_scalar, double big_blind, double small_blind, int dealer, uint64_t seed,
              uint32_t table_id_base);
int pk_destroy(pk_handle *h);
int pk_num_tables(const pk_handle *h);
int pk_num_players(const pk_handle *h);

/* Game.reset(dealer=...), pokerl/game.py:397-412, on tables with mask[t] != 0 (mask NULL = all tables). */
int pk_reset(pk_handle *h, const uint8_t *mask, int dealer);
/* Same with a DEVICE mask, asynchronous on the handle's stream: tables with (mask_d[t] & mask_bits) != 0 are reset (mask_d NULL = all
 * tables).  mask_bits picks the bits of a mask byte that count, so that the flags pk_step_d wrote can serve as the mask directly:
 * pk_reset_d(h, flags_d, PK_FLAG_GAME_OVER, 0) is the `if game_over: game.reset()` of a device-resident rollout loop
 * (examples/random_game.py:8-12) without a host round trip; 0xFF = any non-zero byte. */
int pk_reset_d(pk_handle *h, const uint8_t *mask_d, int mask_bits, int dealer);

/* Game.step(action), pokerl/game.py:621-700, one action per table.
 * flags[T] (PK_FLAG_*), terr[T] (PK_TERR_*, may be NULL).  Returns PK_E_TABLE if any terr != 0. */
int pk_step(pk_handle *h, const int32_t *actions, uint8_t *flags, uint8_t *terr);
/* Same, device-resident I/O (inputs already in HBM; asynchronous on the handle's stream). */
int pk_step_d(pk_handle *h, const int32_t *actions_d, uint8_t *flags_d, uint8_t *terr_d);
/* pk_step_d with the `if game_over: game.reset()` of a rollout loop (examples/random_game.py:8-12) in the SAME launch: a table whose
 * step ends its game (or runs into PK_TERR_HAND_CAP, which the reference would never leave) is Game.reset(dealer = 0) on the spot, exactly
 * as pk_rollout's auto_reset does; flags_d[t] still reports PK_FLAG_GAME_OVER and terr_d[t] the error bits of the step that ended it.
 * Saves the pk_reset_d launch of the loop (a launch costs ~4 us of a ~25 us step at 65 536 tables). */
int pk_step_auto_d(pk_handle *h, const int32_t *actions_d, uint8_t *flags_d, uint8_t *terr_d);
/* Game.step as a BOUNDED launch, for callers that act on whichever tables are ready (the pk_env_step_async_d idea applied to Game.step).
 * The reference's next_player loop plays whole hands in which nobody can act (pokerl/game.py:607-611), so ONE step can roll through
 * several hands: about one table in 10 000 per step does -- but among 65 536 tables the slowest needs ~4 hands, each ~3 us of serial work,
 * and a synchronous step launch (pk_step_d: ~20 us) lasts as long as that one table.  Here a launch serves at most max_hands end_hand
 * rounds (1: every table's step up to its first hand end): a table whose step has returned gets ready_d[t] = 1 and its flags_d / terr_d
 * written; a table whose step rolls on gets ready_d[t] = 0, its outputs are left untouched and the step stays IN FLIGHT on the device --
 * the next call carries on with it and IGNORES actions_d[t].  An invalid action returns at once (ready, PK_TERR_INVALID_ACTION, table
 * untouched).  auto_reset != 0: as pk_step_auto_d.  Per table the sequence of steps, flags and RNG draws is exactly the synchronous one;
 * only the call that delivers them differs.  max_hands <= 0: run every step to its end (every table ready) -- which also ends the state in
 * which every other entry point that reads or changes tables returns PK_E_BUSY (pk_sync only waits) -- EXCEPT the device-resident readers a
 * caller needs to act on the ready tables: pk_pick_actions_d, pk_get_obs_d, pk_get_obs_packed_d, pk_get_valid_actions_d, pk_get_f64_d (their
 * rows for a table in flight show that table in the middle of its step: ignore them).  auto_reset must not change while
 * steps are in flight (PK_E_INVALID_ARG).  A step that has already rolled 16 hands is carried to its end whatever the budget.
 * A DRAIN (max_hands <= 0) IS A FULL STEP CALL: besides finishing what is in flight it steps every idle (ready) table with actions_d[t], like any other
 * call.  To drain WITHOUT stepping, fill actions_d with an invalid action (-1): those tables come back untouched with PK_TERR_INVALID_ACTION in terr_d[t]
 * (the error byte of a table that was merely passed over -- not an error of the call) -- or pass actions_d == NULL, which only a drain accepts and which
 * means the same "no action for anybody".  Draining with the buffer the previous launch consumed steps the ready tables a second time with stale actions. */
int pk_step_async_d(pk_handle *h, const int32_t *actions_d, uint8_t *flags_d, uint8_t *terr_d, uint8_t *ready_d, int max_hands, int auto_reset);

/* Game.get_valid_actions(player), pokerl/game.py:339-383: out[T][7] one-hot bytes.  player < 0: each table's active
 * player (the reference's `player=None`); 0 <= player < N: that seat on every table. */
int pk_get_valid_actions(pk_handle *h, int player, uint8_t *out);

/* State reads (Game attributes). */
int pk_get_f64(pk_handle *h, int field, double *out /* [T][N] */);
/* The same into a DEVICE buffer, asynchronous on the handle's stream (e.g. the send buffer of an RCCL all-gather of the
 * payoffs: pokerl_amd/sharding.py gather_f64). */
int pk_get_f64_d(pk_handle *h, int field, double *out_d /* [T][N] */);
int pk_get_min_raise(pk_handle *h, double *out /* [T] minimum_raise_value */);
int pk_get_table_f64(pk_handle *h, int field /* PK_TF_* */, double *out /* [T] */);
int pk_get_game_over(pk_handle *h, uint8_t *out /* [T] Game.game_over, pokerl/game.py:317-320 */);
int pk_get_player_states(pk_handle *h, uint8_t *out /* [T][N] PlayerState, pokerl/enums.py:130-136 */);
int pk_get_i32(pk_handle *h, int field, int32_t *out /* [T] */);
/* RNG-spec serials of each table: setup_hand() calls / completed Game.step() calls so far (64-bit; either pointer may be
 * NULL).  pk_set_serials resumes a table's streams at given serials (e.g. restoring a checkpoint); call before pk_reset. */
int pk_get_serials(pk_handle *h, uint64_t *hand_serial, uint64_t *step_serial);
int pk_set_serials(pk_handle *h, const uint64_t *hand_serial, const uint64_t *step_serial);
/* deck[0 : 5+2N] as Card.value bytes ((suit<<4)|rank0, pokerl/cards.py:28-62): community = [0:5], hole(p) = [5+2p : 7+2p]
 * (pokerl/game.py:385-395). out[T][5+2N]. */
int pk_get_cards(pk_handle *h, uint8_t *out);
/* Rankings of the last showdown of each table (pokerl/game.py:488-489): rank[T][N] HandRanking (10 = NONE for
 * seats not shown down), kick[T][N] = judger.get_kickers_value(kickers) (pokerl/judger.py:101-109). */
int pk_get_hand_ranks(pk_handle *h, uint8_t *rank, uint32_t *kick);

/* pokerl.judger.eval_hand (pokerl/judger.py:7-99) on M hands.  cards[M][7] Card.value bytes (unused slots ignored),
 * ncards[M] in 0..7 (NULL = all 7).  rank[M], kick[M] (packed kickers, judger.py:101-109), nkick[M] (may be NULL).
 * The FIRST call per device that needs the evaluator's 32 KB table builds it (unless a handle of up to ten seats exists on the device: pk_create has
 * built it): one hipMalloc, one small kernel and one hipStreamSynchronize -- on the stream the call was given (pk_eval_hands_d) or on the legacy default
 * stream (the host-buffer entry points, which run there anyway).  A one-off host block, not legal inside a stream capture: make one warm-up call
 * (any m >= 1) before capturing or timing.
 * Multiset semantics: duplicate cards are legal, as in the reference's own tests (those hands, hands of fewer than three cards and
 * hands with a byte that is no card -- suit > 3 or rank nibble > 12 -- take the reference's sort-and-scan; 3..7 DISTINCT cards a
 * table-driven evaluator that equals it on every subset of the deck: tools/host_sim `evalntab`, GPU digest fast = 4). */
int pk_eval_hands(int device, const uint8_t *cards, const uint8_t *ncards, size_t m, uint8_t *rank, uint32_t *kick,
                  uint8_t *nkick);
/* Same op on device-resident buffers, asynchronous on `stream` (a hipStream_t; NULL = the default stream): the
 * partial-hand rank feature of examples/q_learning.py:29-33 (2/5/6/7-card hands every step) without a host round trip. */
int pk_eval_hands_d(int device, const uint8_t *cards_d, const uint8_t *ncards_d, size_t m, uint8_t *rank_d, uint32_t *kick_d,
                    uint8_t *nkick_d, void *stream);
/* pokerl.judger.compare_rankings (pokerl/judger.py:111-158) on M lists of n rankings: rank[M][n], kick[M][n] ->
 * onehot[M][n].  Includes the reference's line-148 behaviour. */
int pk_compare_rankings(int device, const uint8_t *rank, const uint32_t *kick, int n, size_t m, uint8_t *onehot);

/* Streaming evaluator on device-resident data: hands_d[m] = one 7-card hand per 64-bit word (card i = byte i, byte 7
 * unused), out_d[m] = HandRanking<<20 | kickers value.  12 algorithmic bytes per evaluation (8 in + 4 out): HBM-bound.
 * distinct != 0: the caller guarantees 7 DISTINCT cards per hand (every in-game hand) and the bitmask evaluator is used;
 * distinct == 0: the general (multiset) evaluator of pk_eval_hands.  Runs on the default stream, synchronous.
 * hands_d must be 8-byte and out_d 4-byte aligned; 16-byte / 8-byte alignment (any allocation base) enables the
 * two-hands-per-lane vector path. */
int pk_eval7_d(int device, const uint64_t *hands_d, size_t m, uint32_t *out_d, int distinct);
/* Synthetic workload for it: hand i = the first 7 cards of the RNG-spec deck of (seed, table_id = i, hand_serial = 0). */
int pk_make_hands_d(int device, uint64_t seed, size_t m, uint64_t *hands_d);
/* `reps` back-to-back passes of pk_eval7_d timed with HIP events: average milliseconds per pass. */
int pk_time_eval7_d(int device, const uint64_t *hands_d, size_t m, uint32_t *out_d, int distinct, int reps, double *ms_per_pass);

/* Exhaustive-check hook for the evaluators: v = HandRanking<<20 | kickers value of every 7-card hand whose two lowest
 * canonical deck indices (pokerl/cards.py:77 order) are (a, b), in lexicographic order; out holds C(51-b, 5) words.
 * fast == 1: the 7-distinct-card evaluator the showdown kernels use; fast == 0: the general one behind pk_eval_hands;
 * fast == 2: the table-driven 7-distinct-card evaluator of pk_eval7_d (cards of hand i rotated by i inside the packed word);
 * fast == 3: the register evaluator that backed pk_eval_hands(_d) up to ABI 4 and still does under PK_EVAL_HANDS_TAB=0 (a bitmask fast path
 * for 3..7 distinct cards, the reference's scan for hands that repeat a card or hold fewer than three);
 * fast == 4: the table path pk_eval_hands(_d) takes (eval_tab_n, cards rotated as for fast == 2). */
int pk_eval7_prefix(int device, int a, int b, int fast, uint32_t *out, size_t *count_out);

/* Actions the in-kernel agent `policy` would take now (one per table) -- lets a host loop reproduce rollouts. */
int pk_pick_actions(pk_handle *h, int policy, int32_t *actions);
int pk_pick_actions_d(pk_handle *h, int policy, int32_t *actions_d);

/* Throughput path: K lockstep Game.step()s per table with in-kernel agents; finished games are reset
 * (Game.reset()) when auto_reset != 0.  fused != 0: ONE launch, table state held in registers for all K steps;
 * fused == 0: K launches, state round-trips HBM every step.  counters[PK_NUM_COUNTERS] are ADDED to (may be NULL).
 * Asynchronous unless counters != NULL.  With counters == NULL a fused call may also DEFER part of its steps: a launch
 * ends when the first lanes of a wave run out of work instead of idling until the slowest table has finished, and the
 * tables remember what they still owe; the next pk_rollout picks that up, and every other entry point (getters,
 * pk_step, pk_reset, pk_sync, pk_record_event, ...) first completes it (pk_flush), so no caller can observe a table
 * that has made fewer than the requested steps.  Results do not depend on how the steps were split over launches. */
/* Kernel choice (internal, bit-identical results either way): for random agents up to six seats and all-in agents up to ten, batches of at most one
 * wave per SIMD (<= 65 536 tables) and launches of >= 16 steps run the variant that ranks the showdown hands with the table-driven evaluator out of
 * each wave's own 32 KB LDS copy of the table (+2 ... 14 %).  That variant takes ALL of a CU's LDS (four 40 KB workgroups): two handles that roll out
 * CONCURRENTLY on one GPU then run one after the other instead of side by side -- env PK_ROLLOUT_TAB=0 switches the variant off for such a process
 * (PK_ROLLOUT_TAB=<n>: minimum steps per launch, default 16). */
int pk_rollout(pk_handle *h, int k_steps, int policy, int auto_reset, int fused, uint64_t *counters);
/* Asynchronous fused pk_rollout calls (counters == NULL, auto_reset != 0) are also COALESCED on the host: while the two
 * most recent launches are still running, a call only adds its steps to a host-side count, which is launched as ONE kernel
 * as soon as a launch slot frees up, when it reaches `max_steps`, or by the flush every observer issues -- so a stream of
 * short calls (20 steps each) runs as launches of up to max_steps steps and pays the fixed cost of a launch once per
 * launch, not per call.  Invisible like deferral: no entry point can observe fewer than the requested steps.
 * max_steps: 0 = every call launches at once; default 1024 (env PK_COALESCE).
 * What it buys depends on the CALLER: a loop that never looks at the tables between calls (bench.py's command: 20-step calls)
 * runs as ~1 000-step launches (30 G env-steps/s at 65 536 x 6); a caller that reads observations or rewards every 20 steps
 * flushes each time and gets the one-launch-per-call rate (21 G: bench.py's extra leg "one launch per call").  Held steps start
 * with the next pk_rollout, flush, getter or pk_wait_event -- not by themselves. */
int pk_set_coalesce(pk_handle *h, int max_steps);
/* Launches of the rollout kernel since the last reset: out[4] = {launches (including the 0-step flushes of deferred
 * work), steps summed over them, min and max steps per launch over the launches with steps}.  reset != 0 clears them. */
int pk_get_launch_stats(pk_handle *h, uint64_t *out, int reset);
/* Diagnostic: the steps each table still owes ON THE DEVICE (out[T]); waits for the launches queued so far but does NOT
 * complete the deferred work (and does not see steps a coalescing host still holds back). */
int pk_get_owed(pk_handle *h, uint32_t *out);
/* Completes deferred rollout steps now (asynchronous on the handle's stream); a no-op when there are none. */
int pk_flush(pk_handle *h);
/* Tuning knobs of the fused rollout (values outside 1..64 leave the knob unchanged): `park` = lanes of a wave waiting at
 * end_hand before the wave runs it (default: per kernel, 28 for random agents, 32 otherwise); `endk` = a deferred launch
 * ends once fewer than this many lanes have work (1 = never defer). */
int pk_set_tuning(pk_handle *h, int park, int endk);

/* PokerGameEnv.reset() / .step(action) (pokerl/envs/game_env.py:20-29, :31-53): seat 0 is the controlled seat,
 * the other seats play `opp_policy` in-kernel.  reward[T] f64, done[T], hand[T] bytes. */
int pk_env_reset(pk_handle *h, const uint8_t *mask, int opp_policy);
int pk_env_step(pk_handle *h, const int32_t *actions, int opp_policy, double *reward, uint8_t *done, uint8_t *hand,
                uint8_t *terr);

/* Dense observation of Game.StateView(active player) (pokerl/game.py:117-131), one row of PK_OBS_DIM(N) doubles
 * per table: [player, turn, minimum_raise_value, valid_actions[7], player_cards[2], community_cards[5] (-1 where
 * not yet visible: game.py:278), credits[N], bets[N], pending_bets[N]]. */
#define PK_OBS_DIM(n) (3 + 7 + 2 + 5 + 3 * (n))
/* player < 0: each table's active player (Game.active_state, game.py:323-332); 0 <= player < N: that SEAT's view on every
 * table.  NOTE the one difference from the reference's constructor: StateView(game, 0) there is the ACTIVE player's view
 * (`player or game.active_player`, game.py:122, treats seat 0 like None); here player == 0 is seat 0 and "the active
 * player" is spelled -1.  The Python mirror (VecGame.observations_of / state_views) maps 0 and None to -1 like the
 * reference. */
int pk_get_obs(pk_handle *h, int player, double *out /* [T][PK_OBS_DIM(N)] */);

/* The same row, compact, for callers that move observations over PCIe or keep millions of them: PK_OBS_PACKED_BYTES(N) bytes
 * per table, 8-byte aligned --
 *   byte 0 player (seat), 1 turn, 2 valid_actions as bits (bit a = PokerMoves a), 3-4 player_cards, 5-9 community_cards
 *   (Card.value bytes; 0xFF where the f64 row has -1), 10-15 zero;
 *   then (3N+1) f64: minimum_raise_value, credits[N], bets[N], pending_bets[N] -- the money stays binary64, bit for bit.
 * 168 bytes against 280 at six seats.  pokerl_amd.state_view.PACKED_DTYPE(N) is the numpy structured dtype of a row. */
#define PK_OBS_PACKED_BYTES(n) (16 + 8 * (3 * (n) + 1))
int pk_get_obs_packed(pk_handle *h, int player, uint8_t *out /* [T][PK_OBS_PACKED_BYTES(N)] */);

/* Pinned host memory for the host-pointer entry points: a getter / pk_step / pk_env_step* whose buffers were allocated here
 * copies at PCIe line rate and (pk_env_step_begin) without blocking; any other host memory works too, through the HIP
 * runtime's staged copies (about half the rate).  Wraps hipHostMalloc / hipHostFree. */
int pk_host_alloc(void **out, size_t bytes);
int pk_host_free(void *p);

/* Game.step's precondition for a whole batch, nothing mutated (pokerl/game.py:648-651: `if action not in valid_actions:
 * raise ValueError`): *first_bad = the lowest table index whose action is not valid for its active player, or -1. */
int pk_check_actions(pk_handle *h, const int32_t *actions, int32_t *first_bad);

/* Device-resident variants for a learner that lives on the same GPU (no host round trip; asynchronous on the handle's
 * stream -- see "Stream control" below for ordering against your own stream): out_d / actions_d / ... are DEVICE pointers. */
int pk_get_obs_d(pk_handle *h, int player, double *out_d /* [T][PK_OBS_DIM(N)] */);
int pk_get_valid_actions_d(pk_handle *h, int player, uint8_t *out_d /* [T][7] one-hot */);
int pk_env_step_d(pk_handle *h, const int32_t *actions_d, int opp_policy, double *reward_d, uint8_t *done_d,
                  uint8_t *hand_d, uint8_t *terr_d);
int pk_env_reset_d(pk_handle *h, const uint8_t *mask_d /* NULL = all */, int opp_policy);
/* PokerGameEnv.step with the rest of a learner's loop fused into the same launch (each option removes a launch per env
 * step): actions_d == NULL lets the in-kernel agent `seat0_policy` play seat 0; auto_reset != 0 applies
 * PokerGameEnv.reset() (pokerl/envs/game_env.py:20-29) on the spot to every table whose step returned done (or ran into
 * PK_TERR_HAND_CAP / PK_TERR_ENV_CAP) -- reward / done / hand / terr still describe the step that ended the episode, and
 * terr also carries the error bits of that reset, should its loop (game_env.py:24-27) run into one;
 * obs_d != NULL receives the dense StateView row of the player to act (PK_OBS_DIM(N) doubles per table). */
int pk_env_step_fused_d(pk_handle *h, const int32_t *actions_d, int seat0_policy, int opp_policy, int auto_reset,
                        double *reward_d, uint8_t *done_d, uint8_t *hand_d, uint8_t *terr_d, double *obs_d);

/* The compact row (PK_OBS_PACKED_BYTES) from the PokerGameEnv kernels: once a device buffer is set here, every
 * pk_env_step_fused_d / _async_d / _multi_d call writes the packed row of each table it delivers into it (beside the f64 row
 * if obs_d is given too), from registers.  NULL switches it off.  8-byte aligned, [T][PK_OBS_PACKED_BYTES(N)]. */
/* Exactly those three write it (pk_env_step_d, pk_env_step / _begin and pk_env_reset_d do not: after a reset the rows are those of the last
 * step -- refresh them with pk_get_obs_packed_d).  The handle keeps the RAW pointer: call pk_set_env_obs_packed(h, NULL) before freeing the
 * buffer.  PK_E_BUSY while env steps are in flight. */
int pk_set_env_obs_packed(pk_handle *h, uint8_t *obs_packed_d);
int pk_get_obs_packed_d(pk_handle *h, int player, uint8_t *out_d);

/* The observation from the Game.step kernels themselves (pokerl/game.py:323-332: `game.active_state`, the StateView of the player to act,
 * is what a learner that drives Game.step reads right after every step; pokerl/game.py:117-131).  Once a buffer is set here, every
 * pk_step_d / pk_step_auto_d / pk_step_async_d launch (pk_step too: it runs pk_step_d) writes, for each table whose step RETURNED in that
 * launch, the row of the player to act -- from the registers the step left, in the same launch -- into
 *   obs_d         [T][PK_OBS_DIM(N)] f64, the row of pk_get_obs_d(h, -1, ...), and / or
 *   obs_packed_d  [T][PK_OBS_PACKED_BYTES(N)], the row of pk_get_obs_packed_d(h, -1, ...);
 * either may be NULL (both NULL: off, the default).  A table whose action was refused (PK_TERR_INVALID_ACTION) gets the row of its untouched
 * table; a table pk_step_auto_d reset on the spot gets the first row of its new game; a table whose step is still in flight after a bounded
 * launch (ready_d[t] == 0) keeps the row it had.  Saves the k_obs launch per step and the second pass over the 285 B/table the step kernel has
 * just stored.  8-byte aligned buffers; the handle keeps the RAW pointers: call pk_set_step_obs(h, NULL, NULL) before freeing them.  Rows are NOT
 * written by pk_reset / pk_reset_d / pk_rollout (refresh with pk_get_obs(_packed)_d after those).  PK_E_BUSY while steps are in flight. */
int pk_set_step_obs(pk_handle *h, double *obs_d, uint8_t *obs_packed_d);

/* pk_env_step through host buffers in two halves, for callers that want the copies off their critical path:
 * pk_env_step_begin uploads actions[T], launches PokerGameEnv.step (auto_reset != 0: finished episodes are reset on the spot,
 * as in pk_env_step_fused_d) and queues the device-to-host copies of reward / done / hand / terr and, where not NULL, of the
 * f64 observation rows (obs) and / or the packed ones (obs_packed); pk_env_step_end waits for them.  With buffers from
 * pk_host_alloc nothing in `begin` blocks and the copies run at PCIe line rate, so the caller's own work -- or the step of
 * ANOTHER handle -- overlaps with them.  Per-table errors are in terr[] after `end` (which does not scan them).  The handle
 * must not be used between the two calls -- every entry point that reads or changes tables, a second pk_env_step_begin included, returns
 * PK_E_BUSY until pk_env_step_end (pk_sync only waits); pk_env_step_end without a begin is PK_E_INVALID_ARG -- and EVERY buffer --
 * actions[] included, which a pinned upload reads when the copy engine gets to it, not when `begin` returns -- must stay valid and
 * untouched until `end` has returned.  Invalid actions are NOT an error of either call: such a table is left untouched and terr[t] says
 * PK_TERR_INVALID_ACTION -- read terr[] after `end`. */
int pk_env_step_begin(pk_handle *h, const int32_t *actions, int opp_policy, int auto_reset, double *reward, uint8_t *done,
                      uint8_t *hand, uint8_t *terr, double *obs, uint8_t *obs_packed);
int pk_env_step_end(pk_handle *h);

/* The same as a BOUNDED launch for learners that act on whichever tables are ready (asynchronous vector environment).
 * One PokerGameEnv.step of a whole batch lasts as long as its slowest table: a seat 0 that goes broke during an
 * opponent's step waits for the end of the game (game_env.py:49-52), ~150 Game.steps against a mean of 6.  Here a launch
 * runs at most max_passes betting passes (each busy table executes about one Game.step per pass); a table whose
 * env.step (and, with auto_reset, the reset after it) has returned by then gets ready_d[t] = 1 and its reward / done /
 * hand / terr / obs row written; the others get ready_d[t] = 0, their outputs are left untouched and their step stays IN
 * FLIGHT on the device: the next call carries on with it and IGNORES actions_d[t].  A table with an invalid action
 * returns at once (ready, PK_TERR_INVALID_ACTION, untouched).  Per table the sequence of steps, outputs and RNG draws is
 * exactly that of pk_env_step_fused_d; only the call that delivers them differs.  max_passes <= 0: run every step to its
 * end (every table ready).  While steps may be in flight (after any call with max_passes > 0) all other entry points
 * that read or change tables return PK_E_BUSY (pk_sync only waits); a call with max_passes <= 0 ends that state.
 * Use auto_reset != 0 with bounded launches: pk_env_reset_d is one of the entry points that are busy meanwhile, so a
 * finished episode could only be reset after a drain.  Steps in flight keep their agents: while that state lasts every
 * call must pass the same seat-0 source (actions_d NULL or not, seat0_policy), opp_policy and auto_reset as the call that
 * started it, else PK_E_INVALID_ARG (nothing done).
 * How long a table can stay in flight: a Game.step that rolls hand after hand is carried to its end inside one launch once it has rolled
 * 16 hands; an env.step whose loops never reach seat 0 (seat 0 broke with the game not over: the reference would spin forever, here
 * PK_ENV_STEP_CAP = 8 192 opponent steps end it with PK_TERR_ENV_CAP) makes about max_passes Game.steps per launch, i.e. is delivered
 * after ~8 192 / max_passes launches -- slow, never stuck; the other tables are not held up by it. */
int pk_env_step_async_d(pk_handle *h, const int32_t *actions_d, int seat0_policy, int opp_policy, int auto_reset,
                        int max_passes, double *reward_d, uint8_t *done_d, uint8_t *hand_d, uint8_t *terr_d,
                        double *obs_d, uint8_t *ready_d);

/* Sub-batches of pk_env_step_async_d inside ONE handle.  A bounded launch ends with a tail (its last waves run alone) and
 * the launches of one handle are serialised on its stream, so one handle of 524 288 tables delivers 2.99 G env.step/s where
 * the same handle in three sub-batches delivers 3.56 G (profiles/r03_env_inner_sweep.txt).  pk_set_env_batches(h, B) splits
 * the handle's tables into B contiguous ranges (whole waves each; fewer than B for a small batch: right after the call pk_env_last_range reports range 0, [0, tables per range)),
 * each with an internal stream.  From then on a call with max_passes > 0
 *   - LAUNCHES one range (round robin), reading actions_d only inside it, and
 *   - DELIVERS the range launched longest ago: the call WAITS ON THE HOST for that launch (made B - 1 calls ago, so done or
 *     nearly so; ABI 3 queued a device-side wait on the handle's stream instead, whose barrier packet could stall another range's
 *     hardware queue -- docs/history.md section 5), and pk_env_last_range reports [begin, end): ready_d / reward_d / ... / obs_d are
 *     COMPLETE inside that range when the call returns (and untouched outside).  The launch itself is ordered after the work the
 *     caller queued on the handle's stream (an event pair) unless that stream is idle, in which case nothing needs ordering.
 *     fresh != 0: that range has not been launched yet since pk_set_env_batches / the last drain -- nothing was written, every
 *     table of it awaits its first action (its observation is the one pk_env_reset_d / pk_get_obs_d left).
 * The range delivered by one call is the range the next call launches, so a learner acts on [begin, end) between two
 * calls.  max_passes <= 0 drains every range and delivers [0, T).  Per table nothing changes: the sequence of steps, outputs
 * and RNG draws is that of the synchronous call.  Call it while no env step is in flight (PK_E_BUSY otherwise).
 * Use B <= 3 (524 288 tables: 3.40 / 3.65 / 1.5 G env.step/s at B = 2 / 3 / 4): the internal streams are created with the highest
 * stream priority -- their own set of hardware queues, apart from the caller's stream, the legacy default stream and whatever else
 * the process created -- and a fourth one shares a queue with another range.
 * Ranges of >= 131 072 tables are the ones that pay (smaller batches: several handles, each on its own stream). */
int pk_set_env_batches(pk_handle *h, int batches);
int pk_env_last_range(pk_handle *h, int *begin, int *end, int *fresh);

/* PokerGameEnv with ONE AGENT PER SEAT, some of them played by the CALLER (self-play, league opponents, a learner's
 * earlier checkpoints): pokerl/envs/game_env.py:13-18 takes a list of agent callables and calls
 * self.agents[active_player](state) wherever an opponent is to act (:25, :43, :51).  seat_policies holds the agent of seat p
 * in nibble p: an in-kernel policy, or PK_POLICY_EXTERNAL.  The call is pk_env_step_async_d with two more outcomes per table:
 *   ready_d[t] = 1  PokerGameEnv.step (or .reset) has returned: reward / done / hand / terr / obs row written; the next
 *                   call reads actions_d[t] as seat 0's action (ignored if seat 0's nibble is an in-kernel policy);
 *   ready_d[t] = 2  an EXTERNAL opponent seat is to act inside the env call, which stays in flight: who_d[t] = that
 *                   seat, the obs row is ITS StateView, terr_d[t] = 0; the next call reads actions_d[t] as ITS action.  An
 *                   invalid one is refused: ready 2 again, terr_d[t] = PK_TERR_INVALID_ACTION, table untouched;
 *   ready_d[t] = 0  the pass budget ran out first (only with max_passes > 0); supply nothing;
 *   ready_d[t] = 3  the table was idle and actions_d[t] was PK_ACTION_SKIP: left alone, no output written (how a caller
 *                   that serves the yielded tables keeps the tables that have already returned out of further launches).
 *                   PK_ACTION_SKIP is honoured WHATEVER seat 0's agent is: also with an in-kernel policy in nibble 0 (where
 *                   actions_d[t] is otherwise ignored for idle tables) a -2 parks the table -- write 0 there, not stale data.
 * who_d[t] is always the seat to act next.  reset_d (may be NULL): reset_d[t] != 0 starts PokerGameEnv.reset()
 * (game_env.py:20-29) on table t instead of a step -- delivered like a step, with reward 0, done 0, hand 0 -- dropping
 * whatever that table had in flight.  max_passes <= 0: run until every table has returned or yielded.
 * Per table the Game.steps, their order and every RNG draw are those of the reference's loop with the same agents; external
 * seats whose caller plays an in-kernel policy's rule reproduce that policy's trajectory bit for bit (tested).
 * While tables may be in flight all other entry points that read or change tables return PK_E_BUSY (as after
 * pk_env_step_async_d); with an external seat that state lasts until pk_env_end_multi_d, which drains and ABANDONS the env
 * calls still waiting for an external action (their tables stay where they are, between two Game.steps); call it after a
 * drain (max_passes <= 0) so that no delivered step is lost.  seat_policies and auto_reset must not change meanwhile. */
int pk_env_step_multi_d(pk_handle *h, const int32_t *actions_d, const uint8_t *reset_d, uint64_t seat_policies, int auto_reset,
                        int max_passes, double *reward_d, uint8_t *done_d, uint8_t *hand_d, uint8_t *terr_d, double *obs_d,
                        uint8_t *who_d, uint8_t *ready_d);
int pk_env_end_multi_d(pk_handle *h);

/* Stream control (no torch types: `stream` is a hipStream_t, `event` a hipEvent_t, passed as void*).  A handle creates
 * its own NON-BLOCKING stream (no implicit ordering with the legacy default stream).  pk_set_stream makes it run on the
 * caller's stream instead, which orders every `_d` call after the work already queued there.  stream == NULL IS a
 * stream: the legacy default stream -- what `torch.cuda.current_stream().cuda_stream` (0) denotes unless the caller
 * entered a torch.cuda.Stream -- so the documented pattern g.set_stream(torch.cuda.current_stream().cuda_stream) orders
 * the handle with torch in every case (ABI 2 silently went back to the handle's own stream for 0: a data race).  A
 * stream of another device is refused (PK_E_INVALID_ARG).  pk_use_own_stream goes back to the handle's own stream.
 * Or keep two streams and order them with events:
 * pk_wait_event = the handle's stream waits for `event` (record it on your stream after producing actions_d),
 * pk_record_event = records `event` on the handle's stream (wait for it on your stream before reading obs_d); deferred
 * rollout steps are completed first, and the launches pk_env_step_async_d made on the handle's internal sub-batch streams
 * (pk_set_env_batches) are waited for, so the event covers everything requested so far. */
int pk_get_stream(pk_handle *h, void **stream_out);
int pk_set_stream(pk_handle *h, void *stream);
int pk_use_own_stream(pk_handle *h);
int pk_wait_event(pk_handle *h, void *event);
int pk_record_event(pk_handle *h, void *event);
/* Completes deferred rollout steps and waits until everything requested so far has finished. */
int pk_sync(pk_handle *h);
/* Streams are recycled through a per-device pool when handles are destroyed (a process that opens and closes handles keeps its hardware
 * queues); the sub-batch streams of pk_set_env_batches are created at the HIGHEST stream priority (env PK_ENV_STREAM_PRIO=0: normal), so a
 * learner's normal-priority kernels on the same GPU yield to the env ranges while those run.  pk_stream_pool_drain destroys the pooled (idle)
 * streams of `device` (-1: all devices) and returns how many -- call it before hipDeviceReset, which would leave stale handles in the pool
 * -- any call on such a handle, a query included, crashes inside the HIP runtime -- or to give the queues back.  Seat belt for an application that
 * forgets: the pool keeps a small canary allocation per device and looks its ADDRESS up (hipMemGetAddressRange) before it hands a pooled stream out; a
 * canary that has vanished means the device was reset, and the pool forgets its handles instead of using them (a stream that cannot be synchronised when
 * its handle is destroyed is not pooled either). */
int pk_stream_pool_drain(int device);
/* Runs `reps` back-to-back fused rollouts of k_steps each (never coalesced) plus the flush of what they deferred and
 * returns the device time of all of it divided by `reps`, in milliseconds (events on the handle's stream): the time one
 * launch's k_steps of work take, the flush shared among the launches.  Diagnostic (tools/); bench.py's roofline leg
 * brackets its own timed region with events instead (pk_record_event + pk_get_launch_stats). */
int pk_time_rollout(pk_handle *h, int k_steps, int policy, int auto_reset, int fused, int reps, double *ms_per_launch,
                    uint64_t *counters);

#ifdef __cplusplus
}
#endif
#endif
