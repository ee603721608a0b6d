// pk_api.hip -- kernels + C ABI (include/pokerl_hip.h) of libpokerl_hip.so.  gfx950 only; plain HIP runtime,
// no torch types.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC (see pokerl_amd/build.py).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "pk_device.hpp"

using namespace pk;

// ================================================================================================ kernels
// One table per lane, one wavefront per workgroup.  At the headline size (65 536 tables) that is exactly one wave per
// SIMD (256 CUs x 4), so occupancy cannot hide anything and the register budget is the whole 512-entry file:
// __launch_bounds__(64) lets the compiler keep a table's full state in VGPRs instead of spilling to scratch.
#define PK_TABLE_BLOCK 64

__device__ __forceinline__ void wave_add_counters(const State &S, uint32_t steps, uint32_t hands, uint32_t evals, uint32_t games) {
    // 64-wide butterfly reduction in registers, then lane 0 adds to the slot this wavefront owns (plain RMW, no atomics).
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        steps += __shfl_down(steps, off, 64); hands += __shfl_down(hands, off, 64);
        evals += __shfl_down(evals, off, 64); games += __shfl_down(games, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        unsigned long long *slot = S.counters + (size_t)blockIdx.x * PK_NUM_COUNTERS;
        slot[PK_C_STEPS] += steps; slot[PK_C_HANDS] += hands; slot[PK_C_EVALS] += evals; slot[PK_C_GAMES] += games;
    }
}

// Sums and clears the per-wave counter slots: one workgroup, grid-stride over the slots.
__global__ void __launch_bounds__(256) k_sum_counters(unsigned long long *slots, int nslots, unsigned long long *out) {
    __shared__ unsigned long long part[256][PK_NUM_COUNTERS];
    unsigned long long acc[PK_NUM_COUNTERS] = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < nslots; i += 256)
        for (int c = 0; c < PK_NUM_COUNTERS; ++c) { acc[c] += slots[(size_t)i * PK_NUM_COUNTERS + c]; slots[(size_t)i * PK_NUM_COUNTERS + c] = 0; }
    for (int c = 0; c < PK_NUM_COUNTERS; ++c) part[threadIdx.x][c] = acc[c];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) for (int c = 0; c < PK_NUM_COUNTERS; ++c) part[threadIdx.x][c] += part[threadIdx.x + s][c];
        __syncthreads();
    }
    if (threadIdx.x < PK_NUM_COUNTERS) out[threadIdx.x] = part[0][threadIdx.x];
}

template <int N>
__global__ void __launch_bounds__(PK_TABLE_BLOCK) k_reset(State S, Hot H, const uint8_t *mask, int dealer) {  // Game.reset, game.py:397-412
    int t = blockIdx.x * H.tpb + threadIdx.x;
    if ((int)threadIdx.x >= H.tpb || t >= S.T) return;
    if (mask && !mask[t]) return;
    Table<N> tb;
    tb.load(S, t);
    tb.reset_state(H, dealer);
    tb.deal(H, H.table_id_base + (uint32_t)t);
    tb.store(S, t);
    double hb;
    S.valid[t] = (uint8_t)tb.valid_mask(hb);
    S.terr[t] = 0;
}

// The table Game.reset(dealer = 0) produces before its deal, for this handle's configuration (see pk::Fresh).
template <int N>
__global__ void k_make_fresh(Hot H, Fresh *out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Table<N> tb;
    tb.blank();
    tb.reset_state(H, 0);
    Fresh f{};
    PK_FOR(p, N) f.credits[p] = tb.credits[p]; f.pending[p] = tb.pending[p]; PK_END
    f.min_raise = tb.min_raise;
    f.st_active = tb.st_active; f.st_called = tb.st_called; f.st_allin = tb.st_allin; f.st_broken = tb.st_broken;
    f.active = tb.active; f.dealer = tb.dealer; f.sb = tb.sb; f.bb = tb.bb;
    *out = f;
}

template <int N>
__global__ void __launch_bounds__(PK_TABLE_BLOCK) k_step(const State *__restrict__ Sp, Hot H, const int32_t *actions, uint8_t *flags, uint8_t *terr) {  // Game.step, game.py:621-700
    const State &S = *Sp;
    __shared__ Lds<N> lds;
    const int t = blockIdx.x * H.tpb + threadIdx.x;
    const bool live = (int)threadIdx.x < H.tpb && t < S.T;
    const uint32_t table_id = H.table_id_base + (uint32_t)t;
    Table<N> tb;
    if (live) tb.load(S, t); else tb.blank();
    double high_bet;
    uint32_t mask = tb.valid_mask(high_bet);                                       // :648
    const int action = live ? actions[t] : -1;
    const bool ok = live && action >= 0 && action < PK_NUM_MOVES && ((mask >> action) & 1);
    bool todo = ok;
    for (;;) {  // same flat shape as k_rollout / k_env_step: one instantiation of cursor and end_block
        if (todo) { tb.begin_step(H, action, high_bet); todo = false; }
        tb.cursor();
        if (!__any(tb.parked())) break;
        tb.end_block(H, t, table_id, lds, false);
    }
    tb.finish_step();
    if (!live) return;
    tb.store_show(S.show, S.T, t, lds);
    if (!ok) {                                                                     // :649-651: no mutation
        flags[t] = 0;
        S.terr[t] = PK_TERR_INVALID_ACTION;
        if (terr) terr[t] = PK_TERR_INVALID_ACTION;
        return;
    }
    tb.store(S, t);
    flags[t] = (uint8_t)tb.flags;
    S.valid[t] = (uint8_t)tb.valid_mask(high_bet);
    S.terr[t] = (uint8_t)tb.terr;
    if (terr) terr[t] = (uint8_t)tb.terr;
}

template <int N>
__global__ void __launch_bounds__(PK_TABLE_BLOCK) k_pick(State S, Hot H, int policy, int32_t *actions) {
    int t = blockIdx.x * H.tpb + threadIdx.x;
    if ((int)threadIdx.x >= H.tpb || t >= S.T) return;
    ActionRng rng;
    actions[t] = pick_action(H, rng, H.table_id_base + (uint32_t)t, S.step_serial[t], S.valid[t], policy);
}

// K more steps per table, in-kernel agents, table state in registers for the whole launch (K == 1, endk == 1: the
// unfused form).  Tables are independent, so lanes need not stay in lockstep INSIDE the launch: a lane whose step
// reaches end_hand parks at LS_END while the other lanes of the wave run ahead on their own step counters; the expensive
// end_block (showdown + side pots + setup_hand + deal) runs only once `park` lanes are waiting (or nobody else can
// run), which raises its lane utilisation from ~25 % to ~60 %.
// Nor need they stay in lockstep ACROSS launches: every table carries the number of steps it still owes (State::owed;
// a launch adds K), and a launch may end while lanes still owe steps or are parked in the middle of one -- as soon as
// fewer than `endk` of the wave's lanes have work left, i.e. before the stragglers would run alone (the tail that
// costs a 20-step launch 45 % of its throughput).  What is left is picked up by the next launch or by the flush
// (endk == 1: run to completion) the host issues before anything can observe the tables.  Every table still makes
// exactly the requested steps with the actions the RNG spec assigns to (table, step_serial), so the observable state
// is bit-identical to the lockstep order.
#ifndef PK_ROLLOUT_ATTR
#define PK_ROLLOUT_ATTR
#endif
// POLICY (the in-kernel agents) is a template parameter: with a run-time policy the random agent's LDS lookup sat in a
// basic block of its own and its latency could not be overlapped with the action-independent part of the step
// (+2.4 % at 65 536 x 6).
template <int N, bool ONE_PASS, int POLICY>
__device__ __forceinline__ void rollout_body(const State *__restrict__ Sp, const Hot &H, int K, int auto_reset, int park, int slack, int clear_terr) {
    // Array bases by pointer (loaded only where the table is loaded / stored), loop scalars by value: see pk::Hot.
    constexpr int policy = POLICY;
    const State &S = *Sp;
    __shared__ Lds<N> lds;
    const int t = blockIdx.x * H.tpb + threadIdx.x;
    const bool live = (int)threadIdx.x < H.tpb && t < S.T;
    const uint32_t table_id = H.table_id_base + (uint32_t)t;
    Table<N> tb;
    uint32_t owed = 0;
    Table<N>::stage_fresh(lds, H.fresh);
    stage_nth(lds);
    if (live) { tb.load(S, t); tb.hands_this_step = (int)S.mid[t]; owed = S.owed[t] + (uint32_t)K; } else tb.blank();
    uint32_t steps = 0;
    bool alive = live;
    ActionRing ring;
    double high_bet;
    // lanes that can work at all in this launch; the launch ends once more than `slack` of them have run out of work
    // (slack >= 64: never, i.e. run to completion)
    const int cap = __popcll(__ballot(live && (owed > 0 || tb.lstate == LS_END)));
    const int quit = max(1, cap - slack);
    PK_PROF(tb.prof.start();)
    auto retire = [&]() {  // a lane whose Game.step() has returned (selects, not branches: nearly every lane, every pass)
        const bool r = tb.stepped && tb.lstate == LS_DONE;
        const bool bad = r && tb.terr != 0;            // table keeps its (reference-identical) state; reported through terr
        const bool good = r && tb.terr == 0;
        tb.step_serial += (r && !(tb.terr & PK_TERR_NO_WINNER)) ? 1u : 0u;   // finish_step()
        tb.stepped = r ? 0u : tb.stepped;
        owed = bad ? 0u : owed - (r ? 1u : 0u);
        alive = alive && !bad;
        steps += good ? 1u : 0u;
        tb.games += good ? (tb.flags & PK_FLAG_GAME_OVER) : 0u;
    };
// Which of the betting passes also run cursor_tail() (next_turn for the lanes whose seat walk failed: ~75 instructions for
// ~1 lane in 5).  Every other pass: the block runs half as often over twice the lanes, a lane waits at most one pass
// (29.7 vs 28.6 G at 65 536 x 6; last pass only 28.7, passes 0+3 29.1, six passes with three tails 29.2).  The LAST
// pass must be in the mask: no lane may be left in LS_SCAN when the wave looks at its parked lanes or leaves the loop.
#ifndef PK_TAIL_MASK
#define PK_TAIL_MASK 0xA
#endif
#ifndef PK_BET_PASSES
#define PK_BET_PASSES 4   // betting passes between two looks at the parked lanes: end_block then serves what four passes
#endif                    // have parked (1: 23.8 G, 2: 25.1 G, 3: 24.4 G, 4: 25.6 G, 6: 24.7 G, 8: 23.2 G at 65 536 x 6)
    static_assert(((PK_TAIL_MASK) >> (PK_BET_PASSES - 1)) & 1, "the last betting pass must run cursor_tail()");
    for (;;) {
        // Nothing is in flight at the top of an iteration.  Without this the compiler cannot rule out that a table
        // register still waits for the global loads before the loop or for end_block's LDS reads (both sit in
        // conditionally executed blocks), and parks a full s_waitcnt right behind the first LDS read of every betting
        // pass: the action ring's latency was exposed three passes out of four.
        if (policy == PK_POLICY_RANDOM) __builtin_amdgcn_s_waitcnt(0);   // (the all-in kernel has no LDS read in its passes)
        if (policy == PK_POLICY_RANDOM) ring.ensure(lds, H, table_id, tb.step_serial, alive && owed > 0, PK_BET_PASSES);   // wave-uniform
#pragma unroll
        for (int pass = 0; pass < PK_BET_PASSES; ++pass) {
            const bool go = alive && tb.lstate == LS_DONE && owed > 0;
            uint32_t word = 0;
            if (policy == PK_POLICY_RANDOM) word = ActionRing::peek(lds, tb.step_serial);
            if (go) {
                uint32_t mask = tb.valid_mask(high_bet);
                tb.begin_step(H, policy == PK_POLICY_ALLIN ? (int)MV_ALL_IN
                                                           : action_from_draw_lds(lds, ActionRing::half_of(word, tb.step_serial), mask), high_bet);
            }
            PK_PROF(tb.prof.lap(PF_ACTION);)
            tb.scan_first();      // every lane in LS_SCAN: the steps just begun and the ones end_block carried into a new hand
            if ((PK_TAIL_MASK >> pass) & 1) tb.cursor_tail();
            PK_PROF(tb.prof.count(PF_N_CURSOR);)
            retire();
            PK_PROF(tb.prof.lap(PF_CURSOR);)
        }
        // showdowns waiting in their side-pot loop (LS_POT) count towards `park` like arrivals (weights 0 and 1/2 measured no better)
        const int parked = __popcll(__ballot(tb.parked()));
        const int runnable = __popcll(__ballot(alive && tb.lstate == LS_DONE && owed > 0));
        if (parked + runnable < quit) break;
        if (parked >= park || runnable == 0) {
            tb.template end_block<ONE_PASS>(H, t, table_id, lds, auto_reset != 0);
            retire();
        }
    }
    // LS_POT never survives a kernel.  A launch that ends early (deferred work) simply takes such a showdown back to
    // LS_END: end_hand up to there is idempotent (the pending bets are committed and zero, payoffs are re-zeroed, the
    // side pots restart from the unchanged committed bets), so the next launch redoes it together with its own arrivals.
    if (tb.lstate == LS_POT) { tb.lstate = LS_END; tb.evals -= (uint32_t)__popc((tb.st_called | tb.st_allin) & Table<N>::FULL); }
    if (live) {
        tb.store(S, t);
        tb.store_show(S.show, S.T, t, lds);
        S.owed[t] = owed; S.mid[t] = (uint32_t)tb.hands_this_step;
        S.valid[t] = (uint8_t)tb.valid_mask(high_bet);
        S.terr[t] = (uint8_t)((clear_terr ? 0 : S.terr[t]) | tb.terr | tb.seen);
    }
    wave_add_counters(S, steps, tb.hands, tb.evals, tb.games);  // every lane takes part in the shuffles
    PK_PROF(tb.prof.flush(S.prof);)
}

// Batches of up to two waves per SIMD: registers capped at 256 (no instantiation needs more; N = 10 uses 245), which
// also steers the max-ILP scheduler to a slightly better schedule than an unlimited budget (24.1 vs 23.4 G at 65 536 x 6 when it was introduced) ...
template <int N>
__global__ void __launch_bounds__(PK_TABLE_BLOCK, 2) PK_ROLLOUT_ATTR k_rollout(const State *__restrict__ Sp, Hot H, int K, int auto_reset, int park, int slack, int clear_terr) {
    rollout_body<N, true, PK_POLICY_RANDOM>(Sp, H, K, auto_reset, park, slack, clear_terr);
}
template <int N>
__global__ void __launch_bounds__(PK_TABLE_BLOCK, 2) PK_ROLLOUT_ATTR k_rollout_allin(const State *__restrict__ Sp, Hot H, int K, int auto_reset, int park, int slack, int clear_terr) {
    rollout_body<N, true, PK_POLICY_ALLIN>(Sp, H, K, auto_reset, park, slack, clear_terr);
}
// ... larger batches: capped at 168 for three waves per SIMD (no spill up to N = 7).  The third wave is worth +19 % at
// 1 M x 6 (49 G env-steps/s) and +14 % at 524 288 x 9 in spite of the scratch traffic at N >= 8; a cap of 128 (four
// waves) spills too much (18.4 G at 65 536 x 6).
#ifndef PK_OCC_CAP
#define PK_OCC_CAP 3
#endif
template <int N>
__global__ void __launch_bounds__(PK_TABLE_BLOCK, PK_OCC_CAP) k_rollout_occ3(const State *__restrict__ Sp, Hot H, int K, int auto_reset, int park, int slack, int clear_terr) {
    rollout_body<N, true, PK_POLICY_RANDOM>(Sp, H, K, auto_reset, park, slack, clear_terr);
}
template <int N>
__global__ void __launch_bounds__(PK_TABLE_BLOCK, PK_OCC_CAP) k_rollout_occ3_allin(const State *__restrict__ Sp, Hot H, int K, int auto_reset, int park, int slack, int clear_terr) {
    rollout_body<N, true, PK_POLICY_ALLIN>(Sp, H, K, auto_reset, park, slack, clear_terr);
}

// PokerGameEnv.reset / .step (envs/game_env.py:20-29, :31-53) share k_rollout's shape: ONE flat loop in which every
// lane owns a small phase machine, begins its next Game.step() as soon as the previous one has returned, and the wave
// runs end_block (end_hand + setup_hand + deal) once for all lanes parked at it.  The step machine is instantiated once
// per kernel (three inlined copies of run() cost k_env_step 256 VGPRs + AGPR spills), table bases come by pointer.

// PokerGameEnv.reset: Game.reset() (:23), then opponents play until seat 0 is to act (:24-26); a game that ends before
// seat 0 ever acts is reset again (:27).
template <int N>
__global__ void __launch_bounds__(PK_TABLE_BLOCK) k_env_reset(const State *__restrict__ Sp, Hot H, const uint8_t *mask, int opp_policy, int park) {
    const State &S = *Sp;
    __shared__ Lds<N> lds;
    const int t = blockIdx.x * H.tpb + threadIdx.x;
    const bool live = (int)threadIdx.x < H.tpb && t < S.T && (!mask || mask[t < S.T ? t : 0]);
    const uint32_t table_id = H.table_id_base + (uint32_t)t;
    Table<N> tb;
    if (live) tb.load(S, t); else tb.blank();
    ActionRng rng;
    double high_bet = 0.0;
    bool more = live, due_reset = live;
    int budget = PK_ENV_STEP_CAP;  // every wave-uniform loop in this file has an exit all lanes reach
    auto retire = [&]() {          // an opponent's Game.step() has returned
        if (tb.stepped && tb.lstate == LS_DONE) {
            tb.finish_step();
            if (--budget < 0) tb.terr |= PK_TERR_ENV_CAP;
            if (tb.terr) more = false;
            else due_reset = (tb.flags & PK_FLAG_GAME_OVER) != 0;                  // :27
        }
    };
    for (;;) {
        if (more && tb.lstate == LS_DONE) {                                        // no step in flight on this lane
            if (due_reset) { tb.reset_state(H, 0); tb.deal(H, table_id); due_reset = false; }   // :23 / :27
            more = tb.active != 0;                                                 // :24
            if (more) {
                uint32_t vm = tb.valid_mask(high_bet);
                tb.begin_step(H, pick_action(H, rng, table_id, tb.step_serial, vm, opp_policy), high_bet);  // :25-26
            }
        }
        tb.cursor();
        retire();
        const int parked = __popcll(__ballot(tb.parked()));
        const int runnable = __popcll(__ballot(more && tb.lstate == LS_DONE));
        if (parked == 0 && runnable == 0) break;
        if (parked >= park || runnable == 0) {
            tb.end_block(H, t, table_id, lds, false);
            retire();
        }
    }
    if (live) {
        tb.store(S, t);
        tb.store_show(S.show, S.T, t, lds);
        S.valid[t] = (uint8_t)tb.valid_mask(high_bet);
        S.terr[t] = (uint8_t)tb.terr;
    }
}

// PokerGameEnv.step: seat 0's own step (:35), the opponents until the hand ends or seat 0 is to act (:41-44), the
// opponents until seat 0 is to act or the game is over (:49-52) -- three phases of one lane-level machine.
// Fused extras for a learner's loop (all optional, each removes a launch per env step): seat 0 can be played by an
// in-kernel agent (seat0_policy >= 0; actions == NULL), a finished episode can be reset on the spot
// (auto_reset: PokerGameEnv.reset(), game_env.py:20-29 -- what the caller would do next for `done` tables; reward /
// done / hand still describe the step that ended it), and the dense StateView row of the player to act can be
// written straight from registers (obs != NULL, layout PK_OBS_DIM).
//
// ASYNC (pk_env_step_async_d): a launch lasts at most `max_passes` betting passes.  A PokerGameEnv.step that has not
// returned by then stays IN FLIGHT: its machine state goes to State::env_ctx / env_rew (plus the step-in-flight bits of
// the table itself), the next launch carries on with it, and only tables whose step returned in this launch write
// their outputs and ready[t] = 1.  One env.step of a whole batch lasts as long as its slowest table (a seat 0 that
// busts during an opponent's step waits for the end of the game, game_env.py:49-52); a learner that acts on the ready
// tables only never waits for those.  Per table the sequence of steps, outputs and RNG draws is the synchronous one.
template <int N, bool ASYNC>
__device__ __forceinline__ void env_step_body(const State *__restrict__ Sp, const Hot &H, const int32_t *actions, int seat0_policy, int opp_policy, int auto_reset, double *reward, uint8_t *done_out, uint8_t *hand_out, uint8_t *terr, double *obs, int park, uint8_t *ready, int max_passes) {
    const State &S = *Sp;
    __shared__ Lds<N> lds;
    const int t = blockIdx.x * H.tpb + threadIdx.x;
    const bool live = (int)threadIdx.x < H.tpb && t < S.T;
    const uint32_t table_id = H.table_id_base + (uint32_t)t;
    Table<N> tb;
    if (live) tb.load(S, t); else tb.blank();
    uint64_t ctx = 0;                                          // != 0: a PokerGameEnv.step of this table is in flight
    if (ASYNC && live) ctx = S.env_ctx[t];
    const bool carried = ctx != 0;
    ActionRing ring;
    stage_nth(lds);
    double high_bet;
    const uint32_t vm0 = tb.valid_mask(high_bet);
    // seat 0's action: supplied (checked here, game.py:648-651) or drawn in the loop like the opponents' (always valid)
    const int action = (!live || carried || !actions) ? -1 : actions[t];
    const bool ok = carried || (live && (!actions || (action >= 0 && action < PK_NUM_MOVES && ((vm0 >> action) & 1))));
    enum { PH_SEAT0 = 0, PH_HAND = 1, PH_TURN = 2, PH_RESET = 3, PH_RESET_PLAY = 4, PH_END = 5 };
    int phase = ok ? PH_SEAT0 : PH_END;
    double rew = 0.0;                                                              // :34
    bool done = false, hand = false;
    uint32_t terr_step = 0;
    int budget = PK_ENV_STEP_CAP, budget_reset = PK_ENV_STEP_CAP;
    if (ASYNC && carried) {
        phase = (int)(ctx >> 1) & 7; done = (ctx >> 4) & 1; hand = (ctx >> 5) & 1; terr_step = (uint32_t)(ctx >> 8) & 0xff;
        budget = (int)((ctx >> 16) & 0xffff) - 1; budget_reset = (int)((ctx >> 32) & 0xffff) - 1;
        rew = S.env_rew[t];
        tb.hands_this_step = (int)S.mid[t];
    }
    int passes = 0;
    const uint32_t caps = PK_TERR_HAND_CAP | PK_TERR_ENV_CAP;
    auto step_finished = [&]() {   // PokerGameEnv.step has returned: its outputs are final; maybe reset the episode
        terr_step = tb.terr;
        phase = (auto_reset && (done || (tb.terr & caps))) ? PH_RESET : PH_END;
    };
    auto retire = [&]() {  // a Game.step() of this lane has returned: the reference's control flow between two steps
        if (tb.stepped && tb.lstate == LS_DONE) {
            tb.finish_step();
            if (phase == PH_RESET_PLAY) {                                          // game_env.py:24-27
                if (--budget_reset < 0) tb.terr |= PK_TERR_ENV_CAP;
                if (tb.terr) phase = PH_END;
                else if (tb.flags & PK_FLAG_GAME_OVER) phase = PH_RESET;           // :27
                else if (tb.active == 0) phase = PH_END;                           // :24
                return;
            }
            if (phase != PH_SEAT0 && --budget < 0) tb.terr |= PK_TERR_ENV_CAP;
            if (tb.terr) { step_finished(); return; }
            const bool over = (tb.flags & PK_FLAG_GAME_OVER) != 0, hand_now = (tb.flags & PK_FLAG_HAND_OVER) != 0;
            const bool seat0 = tb.active == 0;
            bool leave_hand_stretch = false;                                       // :41's loop is over (or never entered)
            if (phase == PH_SEAT0) {
                done = over; hand = hand_now;
                if (done || (tb.st_broken & 1)) { rew = tb.payoffs[0]; done = true; hand = true; step_finished(); }  // :37-39
                else if (!hand && !seat0) phase = PH_HAND;                         // :41
                else leave_hand_stretch = true;
            } else if (phase == PH_HAND) {
                done = over; hand = hand_now;                                      // :44
                leave_hand_stretch = hand || seat0;
            } else {                                                               // PH_TURN: only `done` is re-read (:52)
                done = over;
                if (done || seat0) step_finished();
            }
            if (leave_hand_stretch) {
                if (hand) rew = tb.payoffs[0];                                     // :47
                if (!done && !seat0) phase = PH_TURN; else step_finished();        // :49
            }
        }
    };
#ifndef PK_ENV_PASSES
#define PK_ENV_PASSES 4   // betting passes between two looks at the parked lanes, as in k_rollout
#endif
    const bool draws = seat0_policy == PK_POLICY_RANDOM || opp_policy == PK_POLICY_RANDOM;   // wave-uniform
    for (;;) {
        // ASYNC, pass budget used up: no lane begins another Game.step; the hands that are ending are still brought to
        // their end (a lane parked at end_hand would otherwise wait for 'park' neighbours launch after launch)
        const bool draining = ASYNC && max_passes > 0 && passes >= max_passes;
        if (!draining && phase == PH_RESET && tb.lstate == LS_DONE) {              // game_env.py:23 / :27
            tb.reset_state(H, 0); tb.deal(H, table_id);
            phase = tb.active != 0 ? PH_RESET_PLAY : PH_END;                       // :24
        }
        if (draws) __builtin_amdgcn_s_waitcnt(0);   // as in k_rollout: no stray full wait behind the passes' LDS reads
        if (draws) ring.ensure(lds, H, table_id, tb.step_serial, live && phase != PH_END, PK_ENV_PASSES);
#pragma unroll
        for (int pass = 0; pass < PK_ENV_PASSES; ++pass) {
            const bool open = !(ASYNC && max_passes > 0 && passes + pass >= max_passes);
            const uint32_t word = draws ? ActionRing::peek(lds, tb.step_serial) : 0u;
            if (open && phase != PH_END && phase != PH_RESET && tb.lstate == LS_DONE) {   // begin this lane's next Game.step()
                const uint32_t vm = tb.valid_mask(high_bet);
                const int pol = phase == PH_SEAT0 ? seat0_policy : opp_policy;
                const int a = (phase == PH_SEAT0 && actions) ? action
                            : pol == PK_POLICY_ALLIN ? (int)MV_ALL_IN
                                                     : action_from_draw_lds(lds, ActionRing::half_of(word, tb.step_serial), vm);
                tb.begin_step(H, a, high_bet);                                     // :35 / :43-44 / :51-52 / :25-26
            }
            tb.cursor();
            retire();
        }
        const int parked = __popcll(__ballot(tb.parked()));
        const int runnable = draining ? 0 : __popcll(__ballot(phase != PH_END && tb.lstate == LS_DONE));
        if (parked == 0 && runnable == 0) break;                                   // draining: the rest stays in flight
        passes += PK_ENV_PASSES;
        if (parked >= park || runnable == 0) {
            tb.end_block(H, t, table_id, lds, false);
            retire();
        }
    }
    if (!live) return;
    if (ok) {
        tb.store(S, t);
        tb.store_show(S.show, S.T, t, lds);
    }
    const uint32_t vmask = tb.valid_mask(high_bet);
    if (ASYNC) {
        const bool returned = phase == PH_END;
        S.env_ctx[t] = returned ? 0ull
                                : (1ull | ((uint64_t)phase << 1) | ((uint64_t)done << 4) | ((uint64_t)hand << 5) | ((uint64_t)(terr_step & 0xff) << 8) |
                                   ((uint64_t)(budget + 1) << 16) | ((uint64_t)(budget_reset + 1) << 32));
        S.mid[t] = returned ? 0u : (uint32_t)tb.hands_this_step;
        S.valid[t] = (uint8_t)vmask;
        ready[t] = returned;
        if (!returned) { S.env_rew[t] = rew; return; }
    }
    const uint32_t te = ok ? (terr_step | tb.terr) : (uint32_t)PK_TERR_INVALID_ACTION;
    reward[t] = ok ? rew : 0.0; done_out[t] = ok && done; hand_out[t] = ok && hand;  // :53
    if (ok) S.valid[t] = (uint8_t)vmask;
    S.terr[t] = (uint8_t)te; terr[t] = (uint8_t)te;
    if (obs) {  // Game.StateView(active player), game.py:117-131, from registers (same row k_obs builds from HBM)
        double *o = obs + (size_t)t * PK_OBS_DIM(N);
        const int who = tb.active;
        o[0] = who; o[1] = tb.turn; o[2] = tb.min_raise;
        for (int a = 0; a < PK_NUM_MOVES; ++a) o[3 + a] = (vmask >> a) & 1;
        double h0 = 0.0, h1 = 0.0;
        PK_FOR(p, N) h0 = (who == p) ? (double)tb.card(5 + 2 * p) : h0; h1 = (who == p) ? (double)tb.card(6 + 2 * p) : h1; PK_END
        o[10] = h0; o[11] = h1;                                                    // game.py:385-389
        PK_FOR(c, 5) o[12 + c] = (tb.turn != 0 && c < tb.turn + 2) ? (double)tb.card(c) : -1.0; PK_END   // game.py:278
        PK_FOR(p, N) o[17 + p] = tb.credits[p]; o[17 + N + p] = tb.bets[p]; o[17 + 2 * N + p] = tb.pending[p]; PK_END
    }
}
template <int N>
__global__ void __launch_bounds__(PK_TABLE_BLOCK) k_env_step(const State *__restrict__ Sp, Hot H, const int32_t *actions, int seat0_policy, int opp_policy, int auto_reset, double *reward, uint8_t *done_out, uint8_t *hand_out, uint8_t *terr, double *obs, int park) {
    env_step_body<N, false>(Sp, H, actions, seat0_policy, opp_policy, auto_reset, reward, done_out, hand_out, terr, obs, park, nullptr, 0);
}
template <int N>
__global__ void __launch_bounds__(PK_TABLE_BLOCK, (N <= 6 ? 3 : 2)) k_env_step_async(const State *__restrict__ Sp, Hot H, const int32_t *actions, int seat0_policy, int opp_policy, int auto_reset, double *reward, uint8_t *done_out, uint8_t *hand_out, uint8_t *terr, double *obs, int park, uint8_t *ready, int max_passes) {
    env_step_body<N, true>(Sp, H, actions, seat0_policy, opp_policy, auto_reset, reward, done_out, hand_out, terr, obs, park, ready, max_passes);
}

// ---- exports: device-side conversion from the SoA/bitmask layout to the reference's table-major arrays
__global__ void k_export_f64(const double *src, int T, int N, double *out) {  // [N][T] -> [T][N]
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * N) return;
    int t = i / N, p = i - t * N;
    out[i] = src[(size_t)p * T + t];
}
__global__ void k_export_states(const uint64_t *ss, int T, int N, uint8_t *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * N) return;
    int t = i / N, p = i - t * N;
    uint64_t s = ss[t];
    uint8_t st = PS_FOLDED;
    if ((s >> p) & 1) st = PS_ACTIVE;
    if ((s >> (16 + p)) & 1) st = PS_CALLED;
    if ((s >> (32 + p)) & 1) st = PS_ALL_IN;
    if ((s >> (48 + p)) & 1) st = PS_BROKEN;
    out[i] = st;
}
__global__ void k_export_i32(State S, int field, int32_t *out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S.T) return;
    uint32_t cur = S.cursors[t];
    int32_t v = 0;
    switch (field) {
        case PK_I_ACTIVE_PLAYER: v = cur & 0xf; break;
        case PK_I_TURN: v = (cur >> 16) & 0xf; break;
        case PK_I_DEALER_IDX: v = (cur >> 4) & 0xf; break;
        case PK_I_SMALL_BLIND_IDX: v = (cur >> 8) & 0xf; break;
        case PK_I_BIG_BLIND_IDX: v = (cur >> 12) & 0xf; break;
        case PK_I_HAND: v = S.hand[t]; break;
    }
    out[t] = v;
}
__global__ void k_export_cards(const uint32_t *cards, int T, int K, uint8_t *out) {  // [W][T] words -> [T][K] bytes
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * K) return;
    int t = i / K, c = i - t * K;
    out[i] = (uint8_t)(cards[(size_t)(c >> 2) * T + t] >> (8 * (c & 3)));
}
__global__ void k_export_show(const uint32_t *show, int T, int N, uint8_t *rank, uint32_t *kick) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * N) return;
    int t = i / N, p = i - t * N;
    uint32_t v = show[(size_t)p * T + t];
    rank[i] = (uint8_t)(v >> 20);
    kick[i] = v & 0xFFFFF;
}
// Game.get_valid_actions(player), game.py:339-383, of ANY seat as a bitmask (runtime N: export kernels are not
// templated).  Same expressions, in the same order, as Table::valid_mask.
__device__ inline uint32_t valid_bits_of(const State &S, int t, int N, int player) {
    const size_t T = (size_t)S.T;
    double high_bet = S.pending[t];                                               // :365 np.max
    for (int p = 1; p < N; ++p) { double x = S.pending[(size_t)p * T + t]; high_bet = (x > high_bet) ? x : high_bet; }
    const double credit = S.credits[(size_t)player * T + t], min_raise = S.min_raise[t];   // :366
    uint32_t mask = (1u << MV_FOLD) | (1u << MV_ALL_IN);                          // :367
    const double d = credit - high_bet;
    const double rv0 = 0.1 * d, rv1 = 0.25 * d, rv2 = 0.5 * d;                    // :370
    mask |= (rv0 > min_raise && (high_bet + rv0) < credit) ? (1u << 3) : 0;       // :371
    mask |= (rv1 > min_raise && (high_bet + rv1) < credit) ? (1u << 4) : 0;
    mask |= (rv2 > min_raise && (high_bet + rv2) < credit) ? (1u << 5) : 0;
    mask |= (high_bet == 0.0) ? (1u << MV_CHECK) : 0;                             // :375
    mask |= (high_bet < credit) ? (1u << MV_CALL) : 0;                            // :376
    return mask;
}
// player < 0: each table's active player (the cached mask); else that seat on every table.  out: one-hot [T][7]
__global__ void k_export_valid(State S, int N, int player, uint8_t *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S.T * PK_NUM_MOVES) return;
    int t = i / PK_NUM_MOVES, a = i - t * PK_NUM_MOVES;
    uint32_t m = player < 0 ? S.valid[t] : valid_bits_of(S, t, N, player);
    out[i] = (m >> a) & 1;
}
// Game.StateView(game, player), game.py:117-131, as one dense f64 row per table (layout: pokerl_hip.h PK_OBS_DIM).
// player < 0: the active player of each table (what `game.active_state` is, game.py:323-332).
__global__ void k_obs(State S, int N, int player, double *out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S.T) return;
    const int T = S.T, D = PK_OBS_DIM(N);
    double *o = out + (size_t)t * D;
    uint32_t cur = S.cursors[t];
    int active = cur & 0xf, turn = (cur >> 16) & 0xf;
    const int who = player < 0 ? active : player;
    const uint32_t vm = player < 0 ? S.valid[t] : valid_bits_of(S, t, N, who);
    auto card = [&](int c) { return (double)((S.cards[(size_t)(c >> 2) * T + t] >> (8 * (c & 3))) & 0xff); };
    o[0] = who; o[1] = turn; o[2] = S.min_raise[t];
    for (int a = 0; a < PK_NUM_MOVES; ++a) o[3 + a] = (vm >> a) & 1;
    o[10] = card(5 + 2 * who); o[11] = card(6 + 2 * who);                          // game.py:385-389
    for (int c = 0; c < 5; ++c) o[12 + c] = (turn != 0 && c < turn + 2) ? card(c) : -1.0;  // game.py:278
    for (int p = 0; p < N; ++p) {
        o[17 + p] = S.credits[(size_t)p * T + t];
        o[17 + N + p] = S.bets[(size_t)p * T + t];
        o[17 + 2 * N + p] = S.pending[(size_t)p * T + t];
    }
}
// Game.pot (np.sum(bets) in numpy's association order, game.py:281-284 + SURVEY A.5) / Game.high_bet
// (np.max(pending_bets), game.py:287-290) per table; Game.game_over (game.py:317-320).
__global__ void k_table_f64(State S, int N, int field, double *out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S.T) return;
    const size_t T = (size_t)S.T;
    double r;
    if (field == PK_TF_POT) {
        const double *a = S.bets;
        if (N < 8) {
            r = a[t];
            for (int p = 1; p < N; ++p) r = r + a[(size_t)p * T + t];
        } else {
            r = ((a[t] + a[T + t]) + (a[2 * T + t] + a[3 * T + t])) + ((a[4 * T + t] + a[5 * T + t]) + (a[6 * T + t] + a[7 * T + t]));
            for (int p = 8; p < N; ++p) r = r + a[(size_t)p * T + t];
        }
    } else if (field == PK_TF_HIGH_BET) {
        r = S.pending[t];
        for (int p = 1; p < N; ++p) { double x = S.pending[(size_t)p * T + t]; r = (x > r) ? x : r; }
    } else r = S.min_raise[t];
    out[t] = r;
}
__global__ void k_game_over(const uint64_t *ss, int T, int N, uint8_t *out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    uint32_t broken = (uint32_t)(ss[t] >> 48) & 0xffff;
    out[t] = __popc(~broken & ((1u << N) - 1)) == 1;
}

// pokerl.judger.eval_hand batched: one hand per lane, cards[M][7] bytes
__global__ void k_eval_hands(const uint8_t *cards, const uint8_t *ncards, size_t m, uint8_t *rank, uint32_t *kick, uint8_t *nkick) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    uint32_t c[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) c[j] = cards[i * 7 + j];
    int n = ncards ? ncards[i] : 7;
    n = n < 0 ? 0 : (n > 7 ? 7 : n);
    int nk;
    uint32_t v = eval_hand(c, n, nk);
    rank[i] = (uint8_t)(v >> 20);
    kick[i] = v & 0xFFFFF;
    if (nkick) nkick[i] = (uint8_t)nk;
}
// pokerl.judger.compare_rankings batched: one list of n rankings per lane (judger.py:111-158)
__global__ void k_compare(const uint8_t *rank, const uint32_t *kick, int n, size_t m, uint8_t *onehot) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    uint32_t best_rank = HR_NONE, best_kicker = 0, win = 0;
    for (int p = 0; p < n; ++p) {
        uint32_t r = rank[i * n + p], k = kick[i * n + p];
        if (r < best_rank) { best_rank = r; best_kicker = k; win = 1u << p; }
        else if (r == best_rank) {
            if (k > best_kicker) win = 1u << p;  // line 148: best_kicker is not raised
            else if (k == best_kicker) win |= 1u << p;
        }
    }
    for (int p = 0; p < n; ++p) onehot[i * n + p] = (win >> p) & 1;
}
// Streaming evaluator: two hands per lane per iteration (one 16-byte load, one 8-byte store), grid-stride.
// VEC: hands 16-byte and out 8-byte aligned (any hipMalloc'ed base); otherwise one hand per lane per iteration.
template <bool DISTINCT, bool VEC>
__global__ void __launch_bounds__(256) k_eval7_stream(const uint64_t *__restrict__ hands, size_t m, uint32_t *__restrict__ out) {
    const size_t pairs = VEC ? m / 2 : 0, stride = (size_t)gridDim.x * blockDim.x;
    auto eval1 = [](uint64_t w) {
        uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32);
        uint32_t c[7] = {lo & 0xff, (lo >> 8) & 0xff, (lo >> 16) & 0xff, lo >> 24, hi & 0xff, (hi >> 8) & 0xff, (hi >> 16) & 0xff};
        int nk;
        return DISTINCT ? eval7_distinct(c) : eval_hand(c, 7, nk);
    };
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += stride) {
        const ulonglong2 w = reinterpret_cast<const ulonglong2 *>(hands)[i];
        uint2 r;
        r.x = eval1(w.x); r.y = eval1(w.y);
        reinterpret_cast<uint2 *>(out)[i] = r;
    }
    if constexpr (VEC) {
        if ((m & 1) && blockIdx.x == 0 && threadIdx.x == 0) out[m - 1] = eval1(hands[m - 1]);
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) out[i] = eval1(hands[i]);
    }
}
// hand i = first 7 cards of the RNG-spec deck of (table_id = i, hand_serial = 0): the deal of a 1-seat table
__global__ void __launch_bounds__(256) k_make_hands(Hot H, size_t m, uint64_t *out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
        Table<1> tb;
        tb.hand_serial = 0;
        tb.deal(H, (uint32_t)i);
        out[i] = (uint64_t)tb.cards[0] | ((uint64_t)(tb.cards[1] & 0x00ffffffu) << 32);
    }
}

// Exhaustive 7-card sweep used by tests (digest definition: tests/golden/make_eval_digest.py): all hands with prefix
// (a, b); hand index within the prefix -> combination of 5 from the cards above b is decoded per lane.
__global__ void k_eval7_prefix(int a, int b, int fast, uint32_t count, uint32_t *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    // unrank i among 5-subsets of {b+1..51} in lexicographic order
    int n = 51 - b;  // pool size
    int sel5[5];
    uint32_t r = i;
    int start = 0;
    for (int k = 5; k >= 1; --k) {
        for (int x = start;; ++x) {
            // C(n - x - 1, k - 1) hands start with element x
            uint32_t cnt = 1;
            int top = n - x - 1;
            if (top < k - 1) cnt = 0;
            else for (int j = 0; j < k - 1; ++j) cnt = cnt * (uint32_t)(top - j) / (uint32_t)(j + 1);
            if (r < cnt) { sel5[5 - k] = x; start = x + 1; break; }
            r -= cnt;
        }
    }
    auto canon = [](int c) { return (uint32_t)(((c % 4) << 4) | (c / 4)); };
    uint32_t h[7] = {canon(a), canon(b), canon(b + 1 + sel5[0]), canon(b + 1 + sel5[1]), canon(b + 1 + sel5[2]),
                     canon(b + 1 + sel5[3]), canon(b + 1 + sel5[4])};
    int nk;
    out[i] = fast ? eval7_distinct(h) : eval_hand(h, 7, nk);  // in-game evaluator / general (multiset) evaluator
}

// ================================================================================================ host side
static thread_local std::string g_err;

struct pk_handle {
    int device = 0, T = 0, N = 0, block = 64, dealer = 0;
    int tpb = 64;   // tables per wavefront (Hot::tpb)
    bool occ3 = true;  // k_rollout_occ3 (registers capped for 3 waves per SIMD) vs k_rollout; knob PK_OCC3
    // lanes parked at end_hand before a wave runs end_block (the kernels look every 4 betting passes); knob PK_PARK /
    // pk_set_tuning.  0 = the measured optimum of each kernel: 28 for k_rollout with random agents (20.9 vs 20.6 G at
    // 20-step launches, the same at long ones), 32 for the all-in agents (44.3 vs 43.3 G) and for the env kernels.
    int park = 0;
    int endk = 48;  // a deferred rollout launch ends once fewer than this many of a wave's lanes have work; knob PK_ENDK
                    // (measured optimum 44..52 at 20 and at 512 steps per launch: tools/tune_sweep.py)
    // Deferred rollout work: steps requested by pk_rollout that no launch has executed yet may exist on the device
    // (State::owed, steps in flight).  Every entry point that reads or changes table state flushes first.
    bool pending = false;
    int pend_policy = 0, pend_auto = 0;
    // PokerGameEnv.steps left in flight by pk_env_step_async_d (State::env_ctx): every other entry point that touches
    // table state refuses to run until a draining call (max_passes <= 0) has completed them.
    bool env_pending = false;
    hipStream_t stream = nullptr, own_stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    State S{};
    State *d_S = nullptr;  // device copy of S (kernels that take the state by pointer)
    Hot hot{};             // loop scalars, passed by value
    unsigned long long *d_totals = nullptr;  // [PK_NUM_COUNTERS] output of k_sum_counters
    void *arena = nullptr;
    // staging (device) + pinned staging (host) of the host-buffer entry points
    int32_t *d_actions = nullptr;
    uint8_t *d_flags = nullptr, *d_terr = nullptr, *d_mask = nullptr, *d_done = nullptr, *d_handf = nullptr;
    double *d_reward = nullptr;
    void *d_export = nullptr;
    size_t export_bytes = 0;
    uint8_t *h_pinned = nullptr;  // [T] pinned: per-table error bytes of pk_step / pk_env_step
    std::string err;
    int fail(int code, const char *what, hipError_t e = hipSuccess) {
        err = what;
        if (e != hipSuccess) { err += ": "; err += hipGetErrorString(e); }
        g_err = err;
        return code;
    }
};

// Entry points run on the handle's device but leave the caller's current device as they found it (a host
// application may be driving other GPUs, e.g. through torch, from the same thread).
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define ON_DEVICE(h)                                             \
    DeviceGuard guard_((h)->device);                             \
    if (!guard_.ok) return (h)->fail(PK_E_HIP, "hipSetDevice")

#define HIPCHK(h, call)                                            \
    do {                                                           \
        hipError_t e_ = (call);                                    \
        if (e_ != hipSuccess) return (h)->fail(PK_E_HIP, #call, e_); \
    } while (0)

#define DISPATCH_N(h, KERNEL, grid, ...)                                                                         \
    do {                                                                                                         \
        dim3 g_((grid)), b_((h)->block);                                                                         \
        switch ((h)->N) {                                                                                        \
            case 2: hipLaunchKernelGGL(KERNEL<2>, g_, b_, 0, (h)->stream, __VA_ARGS__); break;                   \
            case 3: hipLaunchKernelGGL(KERNEL<3>, g_, b_, 0, (h)->stream, __VA_ARGS__); break;                   \
            case 4: hipLaunchKernelGGL(KERNEL<4>, g_, b_, 0, (h)->stream, __VA_ARGS__); break;                   \
            case 5: hipLaunchKernelGGL(KERNEL<5>, g_, b_, 0, (h)->stream, __VA_ARGS__); break;                   \
            case 6: hipLaunchKernelGGL(KERNEL<6>, g_, b_, 0, (h)->stream, __VA_ARGS__); break;                   \
            case 7: hipLaunchKernelGGL(KERNEL<7>, g_, b_, 0, (h)->stream, __VA_ARGS__); break;                   \
            case 8: hipLaunchKernelGGL(KERNEL<8>, g_, b_, 0, (h)->stream, __VA_ARGS__); break;                   \
            case 9: hipLaunchKernelGGL(KERNEL<9>, g_, b_, 0, (h)->stream, __VA_ARGS__); break;                   \
            case 10: hipLaunchKernelGGL(KERNEL<10>, g_, b_, 0, (h)->stream, __VA_ARGS__); break;                 \
        }                                                                                                        \
    } while (0)

static inline int table_grid(const pk_handle *h) { return (h->T + h->tpb - 1) / h->tpb; }
// parking threshold for waves that hold h->tpb tables instead of 64
static inline int scaled_park(const pk_handle *h, int dflt = 32) {
    int p = ((h->park > 0 ? h->park : dflt) * h->tpb + 63) / 64;
    return p < 1 ? 1 : p;
}
static inline int flat_grid(size_t n) { return (int)((n + 255) / 256); }

// One fused rollout launch: every table owes k_steps more steps; the launch ends once fewer than `endk` lanes of a
// wave have work left (endk == 1: runs to completion).
static int launch_rollout(pk_handle *h, int k_steps, int policy, int auto_reset, int endk) {
    const int slack = endk <= 1 ? PK_WAVE : ((PK_WAVE - endk) * h->tpb) / PK_WAVE;   // lanes allowed to idle before a launch ends
#define ROLLOUT_ARGS (const State *)h->d_S, h->hot, k_steps, auto_reset, scaled_park(h, policy == PK_POLICY_RANDOM ? 28 : 32), slack, h->pending ? 0 : 1
    if (!h->occ3) {
        if (policy == PK_POLICY_RANDOM) DISPATCH_N(h, k_rollout, table_grid(h), ROLLOUT_ARGS);
        else DISPATCH_N(h, k_rollout_allin, table_grid(h), ROLLOUT_ARGS);
    } else {
        if (policy == PK_POLICY_RANDOM) DISPATCH_N(h, k_rollout_occ3, table_grid(h), ROLLOUT_ARGS);
        else DISPATCH_N(h, k_rollout_occ3_allin, table_grid(h), ROLLOUT_ARGS);
    }
#undef ROLLOUT_ARGS
    HIPCHK(h, hipGetLastError());
    h->pending = slack < PK_WAVE;
    h->pend_policy = policy; h->pend_auto = auto_reset;
    return PK_OK;
}
// Completes whatever deferred rollout launches left undone.  Called by every entry point that reads or mutates tables.
static int flush_rollout(pk_handle *h) {
    if (!h->pending) return PK_OK;
    return launch_rollout(h, 0, h->pend_policy, h->pend_auto, 1);
}
static int flush(pk_handle *h) {
    if (h->env_pending)
        return h->fail(PK_E_BUSY, "PokerGameEnv steps are in flight (pk_env_step_async_d): drain them with max_passes = 0 first");
    return flush_rollout(h);
}
#define FLUSH(h)                   \
    do {                           \
        int rc_ = flush(h);        \
        if (rc_) return rc_;       \
    } while (0)

template <typename F>
static int export_to_host(pk_handle *h, void *out, size_t bytes, F launch) {
    ON_DEVICE(h);
    FLUSH(h);
    if (bytes > h->export_bytes) return h->fail(PK_E_INVALID_ARG, "export buffer too small");
    launch();
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(out, h->d_export, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PK_OK;
}

extern "C" {

int pk_abi_version(void) { return PK_ABI_VERSION; }

int pk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *pk_last_error(const pk_handle *h) { return h ? h->err.c_str() : g_err.c_str(); }

int pk_create(pk_handle **out, int device, int num_tables, int num_players, const double *start_credits,
              double start_credit_scalar, double big_blind, double small_blind, int dealer, uint64_t seed,
              uint32_t table_id_base) {
    if (!out) { g_err = "pk_create: out is NULL"; return PK_E_INVALID_ARG; }
    *out = nullptr;
    if (num_tables < 1 || num_players < PK_MIN_PLAYERS || num_players > PK_MAX_PLAYERS) {
        g_err = "pk_create: need num_tables >= 1 and 2 <= num_players <= 10";
        return PK_E_INVALID_ARG;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_err = std::string("pk_create: no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "0 devices") +
                "); this library has no CPU fallback";
        return PK_E_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) { g_err = "pk_create: device index out of range"; return PK_E_INVALID_ARG; }
    pk_handle *h = new pk_handle();
    h->device = device; h->T = num_tables; h->N = num_players; h->dealer = dealer;
    h->block = PK_TABLE_BLOCK;
    {   // small batches: spread the tables over all 1 024 SIMDs of the chip (power of two, 1..64 tables per wave)
        int tpb = 64;
        while (tpb > 1 && (long)num_tables <= 1024L * (tpb / 2)) tpb /= 2;
        if (const char *pk = getenv("PK_TPB")) { int v = atoi(pk); if (v >= 1 && v <= 64) tpb = v; }
        h->tpb = tpb;
        h->occ3 = num_tables > 2 * 65536;   // see k_rollout_occ3
        if (const char *pk = getenv("PK_OCC3")) h->occ3 = atoi(pk) != 0;
    }
    if (const char *pk = getenv("PK_PARK")) { int v = atoi(pk); if (v >= 1 && v <= 64) h->park = v; }
    if (const char *pk = getenv("PK_ENDK")) { int v = atoi(pk); if (v >= 1 && v <= 64) h->endk = v; }
    auto bail = [&](int code) { g_err = h->err; pk_destroy(h); return code; };
    DeviceGuard guard(device);
    if (!guard.ok) return bail(h->fail(PK_E_HIP, "hipSetDevice"));
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) return bail(h->fail(PK_E_HIP, "hipStreamCreate"));
    h->stream = h->own_stream;
    if (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) return bail(h->fail(PK_E_HIP, "hipEventCreate"));
    if (hipHostMalloc((void **)&h->h_pinned, (size_t)num_tables, hipHostMallocDefault) != hipSuccess) return bail(h->fail(PK_E_OOM, "hipHostMalloc"));

    const size_t T = (size_t)num_tables, N = (size_t)num_players;
    const size_t K = 5 + 2 * N, W = (K + 3) / 4;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t obs = (size_t)PK_OBS_DIM(N) * 8;
    h->export_bytes = al(T * (obs > N * 8 ? obs : N * 8));
    const size_t nwaves = (T + (size_t)h->tpb - 1) / (size_t)h->tpb;
    size_t total = 4 * al(T * N * 8) + 4 * al(T * 8) + 4 * al(T * 4) + al(W * T * 4) + al(N * T * 4) + 2 * al(T) +
                   al(nwaves * PK_NUM_COUNTERS * 8) + al(PK_NUM_COUNTERS * 8) + al(PF_SLOTS * 8) + al(sizeof(State)) +
                   al(PK_MAX_PLAYERS * 8) + al(sizeof(Fresh)) + al(T * 4) + 5 * al(T) + al(T * 8) + 2 * al(T * 8) + h->export_bytes;
    e = hipMalloc(&h->arena, total);
    if (e != hipSuccess) return bail(h->fail(PK_E_OOM, "hipMalloc(table state)", e));
    if (hipMemsetAsync(h->arena, 0, total, h->stream) != hipSuccess) return bail(h->fail(PK_E_HIP, "hipMemset"));
    char *p = (char *)h->arena;
    auto take = [&](size_t bytes) { void *r = p; p += al(bytes); return r; };
    State &S = h->S;
    S.credits = (double *)take(T * N * 8); S.bets = (double *)take(T * N * 8);
    S.pending = (double *)take(T * N * 8); S.payoffs = (double *)take(T * N * 8);
    S.min_raise = (double *)take(T * 8);
    S.seat_states = (uint64_t *)take(T * 8);
    S.hand_serial = (uint64_t *)take(T * 8); S.step_serial = (uint64_t *)take(T * 8);
    S.cursors = (uint32_t *)take(T * 4); S.hand = (int32_t *)take(T * 4);
    S.owed = (uint32_t *)take(T * 4); S.mid = (uint32_t *)take(T * 4);
    S.env_ctx = (uint64_t *)take(T * 8); S.env_rew = (double *)take(T * 8);
    S.cards = (uint32_t *)take(W * T * 4);
    S.show = (uint32_t *)take(N * T * 4);
    S.valid = (uint8_t *)take(T); S.terr = (uint8_t *)take(T);
    S.counters = (unsigned long long *)take(nwaves * PK_NUM_COUNTERS * 8);
    h->d_totals = (unsigned long long *)take(PK_NUM_COUNTERS * 8);
    S.prof = (unsigned long long *)take(PF_SLOTS * 8);
    h->d_S = (State *)take(sizeof(State));
    double *d_start = (double *)take(PK_MAX_PLAYERS * 8);
    Fresh *d_fresh = (Fresh *)take(sizeof(Fresh));
    h->d_actions = (int32_t *)take(T * 4);
    h->d_flags = (uint8_t *)take(T); h->d_terr = (uint8_t *)take(T); h->d_mask = (uint8_t *)take(T);
    h->d_done = (uint8_t *)take(T); h->d_handf = (uint8_t *)take(T);
    h->d_reward = (double *)take(T * 8);
    h->d_export = take(h->export_bytes);
    for (int i = 0; i < PK_MAX_PLAYERS; ++i)
        S.start_credits[i] = i < num_players ? (start_credits ? start_credits[i] : start_credit_scalar) : 0.0;
    S.big_blind = big_blind; S.small_blind = small_blind;
    S.key0 = (uint32_t)seed; S.key1 = (uint32_t)(seed >> 32);
    S.table_id_base = table_id_base; S.T = num_tables;

    // Game.__init__ (game.py:242-264): every seat ACTIVE, dealer cursor = config dealer, credits 0, ranks NONE.
    {
        std::vector<uint64_t> ss(T, (uint64_t)((1u << num_players) - 1));
        int d = ((dealer % num_players) + num_players) % num_players;
        std::vector<uint32_t> cur(T, (uint32_t)d << 4);
        std::vector<uint32_t> show(N * T, NONE_V);
        // get_valid_actions on the un-reset state (all credits and pending bets 0): raises invalid (0 > 0 is false),
        // CHECK valid (high_bet == 0), CALL invalid (0 < 0 is false) -> FOLD | CHECK | ALL_IN
        std::vector<uint8_t> valid(T, (uint8_t)((1u << MV_FOLD) | (1u << MV_CHECK) | (1u << MV_ALL_IN)));
        h->hot.fresh = d_fresh;
        h->hot.big_blind = big_blind; h->hot.small_blind = small_blind; h->hot.start_credits = d_start; h->hot.show = S.show;
        h->hot.key0 = S.key0; h->hot.key1 = S.key1; h->hot.table_id_base = table_id_base; h->hot.T = num_tables; h->hot.tpb = h->tpb; h->hot.prof = S.prof;
        h->hot.start_uniform = S.start_credits[0]; h->hot.start_is_uniform = 1;
        for (int i = 1; i < num_players; ++i) if (S.start_credits[i] != S.start_credits[0]) h->hot.start_is_uniform = 0;
        if (hipMemcpyAsync(d_start, S.start_credits, PK_MAX_PLAYERS * 8, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
            hipMemcpyAsync(h->d_S, &h->S, sizeof(State), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
            hipMemcpyAsync(S.seat_states, ss.data(), T * 8, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
            hipMemcpyAsync(S.cursors, cur.data(), T * 4, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
            hipMemcpyAsync(S.show, show.data(), N * T * 4, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
            hipMemcpyAsync(S.valid, valid.data(), T, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess)
            return bail(h->fail(PK_E_HIP, "initial state upload"));
        DISPATCH_N(h, k_make_fresh, 1, h->hot, d_fresh);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)
            return bail(h->fail(PK_E_HIP, "k_make_fresh"));
    }
    *out = h;
    return PK_OK;
}

int pk_destroy(pk_handle *h) {
    if (!h) return PK_OK;
    DeviceGuard guard(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->arena) (void)hipFree(h->arena);
    if (h->h_pinned) (void)hipHostFree(h->h_pinned);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return PK_OK;
}

int pk_num_tables(const pk_handle *h) { return h ? h->T : PK_E_INVALID_ARG; }
int pk_num_players(const pk_handle *h) { return h ? h->N : PK_E_INVALID_ARG; }

// ---- stream control: how a caller with its own stream (a learner on the same GPU) orders its work against ours
int pk_get_stream(pk_handle *h, void **stream_out) {
    if (!h || !stream_out) return PK_E_INVALID_ARG;
    *stream_out = (void *)h->stream;
    return PK_OK;
}
int pk_set_stream(pk_handle *h, void *stream) {
    if (!h) return PK_E_INVALID_ARG;
    ON_DEVICE(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));  // nothing of ours may still be running on the stream we leave
    h->stream = stream ? (hipStream_t)stream : h->own_stream;
    return PK_OK;
}
int pk_wait_event(pk_handle *h, void *event) {
    if (!h || !event) return PK_E_INVALID_ARG;
    ON_DEVICE(h);
    HIPCHK(h, hipStreamWaitEvent(h->stream, (hipEvent_t)event, 0));
    return PK_OK;
}
int pk_record_event(pk_handle *h, void *event) {
    if (!h || !event) return PK_E_INVALID_ARG;
    ON_DEVICE(h);
    if (!h->env_pending) FLUSH(h);  // "everything requested so far" includes deferred rollout steps (env steps in flight stay so)
    HIPCHK(h, hipEventRecord((hipEvent_t)event, h->stream));
    return PK_OK;
}

int pk_set_tuning(pk_handle *h, int park, int endk) {
    if (!h) return PK_E_INVALID_ARG;
    if (park >= 1 && park <= 64) h->park = park;
    if (endk >= 1 && endk <= 64) h->endk = endk;
    return PK_OK;
}

static int upload_mask(pk_handle *h, const uint8_t *mask, const uint8_t **dmask) {
    *dmask = nullptr;
    if (mask) {
        HIPCHK(h, hipMemcpyAsync(h->d_mask, mask, (size_t)h->T, hipMemcpyHostToDevice, h->stream));
        *dmask = h->d_mask;
    }
    return PK_OK;
}

int pk_reset(pk_handle *h, const uint8_t *mask, int dealer) {
    if (!h) return PK_E_INVALID_ARG;
    ON_DEVICE(h);
    FLUSH(h);
    const uint8_t *dmask;
    int rc = upload_mask(h, mask, &dmask);
    if (rc) return rc;
    int d = ((dealer % h->N) + h->N) % h->N;
    DISPATCH_N(h, k_reset, table_grid(h), h->S, h->hot, dmask, d);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PK_OK;
}

int pk_step_d(pk_handle *h, const int32_t *actions_d, uint8_t *flags_d, uint8_t *terr_d) {
    if (!h || !actions_d || !flags_d) return h ? h->fail(PK_E_INVALID_ARG, "pk_step_d: NULL buffer") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    FLUSH(h);
    DISPATCH_N(h, k_step, table_grid(h), (const State *)h->d_S, h->hot, actions_d, flags_d, terr_d);
    HIPCHK(h, hipGetLastError());
    return PK_OK;
}

static int any_terr(const uint8_t *terr, int T) {
    for (int i = 0; i < T; ++i) if (terr[i]) return 1;
    return 0;
}

int pk_step(pk_handle *h, const int32_t *actions, uint8_t *flags, uint8_t *terr) {
    if (!h || !actions || !flags) return h ? h->fail(PK_E_INVALID_ARG, "pk_step: NULL buffer") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    const size_t T = (size_t)h->T;
    HIPCHK(h, hipMemcpyAsync(h->d_actions, actions, T * 4, hipMemcpyHostToDevice, h->stream));
    int rc = pk_step_d(h, h->d_actions, h->d_flags, h->d_terr);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(flags, h->d_flags, T, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->h_pinned, h->d_terr, T, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (terr) memcpy(terr, h->h_pinned, T);
    if (any_terr(h->h_pinned, h->T)) return h->fail(PK_E_TABLE, "pk_step: per-table error(s), see terr");
    return PK_OK;
}

static int bad_player(const pk_handle *h, int player) { return player >= h->N; }

int pk_get_valid_actions(pk_handle *h, int player, uint8_t *out) {
    if (!h || !out || bad_player(h, player)) return h ? h->fail(PK_E_INVALID_ARG, "pk_get_valid_actions: bad argument") : PK_E_INVALID_ARG;
    size_t n = (size_t)h->T * PK_NUM_MOVES;
    return export_to_host(h, out, n, [&] {
        hipLaunchKernelGGL(k_export_valid, dim3(flat_grid(n)), dim3(256), 0, h->stream, h->S, h->N, player, (uint8_t *)h->d_export);
    });
}

int pk_get_f64(pk_handle *h, int field, double *out) {
    if (!h || !out || field < 0 || field > 3) return h ? h->fail(PK_E_INVALID_ARG, "pk_get_f64: bad field") : PK_E_INVALID_ARG;
    const double *src = field == PK_F_CREDITS ? h->S.credits : field == PK_F_BETS ? h->S.bets : field == PK_F_PENDING_BETS ? h->S.pending : h->S.payoffs;
    size_t n = (size_t)h->T * h->N;
    return export_to_host(h, out, n * 8, [&] {
        hipLaunchKernelGGL(k_export_f64, dim3(flat_grid(n)), dim3(256), 0, h->stream, src, h->T, h->N, (double *)h->d_export);
    });
}

int pk_get_table_f64(pk_handle *h, int field, double *out) {
    if (!h || !out || field < 0 || field > PK_TF_MIN_RAISE) return h ? h->fail(PK_E_INVALID_ARG, "pk_get_table_f64: bad field") : PK_E_INVALID_ARG;
    return export_to_host(h, out, (size_t)h->T * 8, [&] {
        hipLaunchKernelGGL(k_table_f64, dim3(flat_grid(h->T)), dim3(256), 0, h->stream, h->S, h->N, field, (double *)h->d_export);
    });
}

int pk_get_min_raise(pk_handle *h, double *out) { return pk_get_table_f64(h, PK_TF_MIN_RAISE, out); }

int pk_get_game_over(pk_handle *h, uint8_t *out) {
    if (!h || !out) return PK_E_INVALID_ARG;
    return export_to_host(h, out, (size_t)h->T, [&] {
        hipLaunchKernelGGL(k_game_over, dim3(flat_grid(h->T)), dim3(256), 0, h->stream, h->S.seat_states, h->T, h->N, (uint8_t *)h->d_export);
    });
}

int pk_get_player_states(pk_handle *h, uint8_t *out) {
    if (!h || !out) return PK_E_INVALID_ARG;
    size_t n = (size_t)h->T * h->N;
    return export_to_host(h, out, n, [&] {
        hipLaunchKernelGGL(k_export_states, dim3(flat_grid(n)), dim3(256), 0, h->stream, h->S.seat_states, h->T, h->N, (uint8_t *)h->d_export);
    });
}

int pk_get_i32(pk_handle *h, int field, int32_t *out) {
    if (!h || !out || field < 0 || field > PK_I_HAND) return h ? h->fail(PK_E_INVALID_ARG, "pk_get_i32: bad field") : PK_E_INVALID_ARG;
    return export_to_host(h, out, (size_t)h->T * 4, [&] {
        hipLaunchKernelGGL(k_export_i32, dim3(flat_grid(h->T)), dim3(256), 0, h->stream, h->S, field, (int32_t *)h->d_export);
    });
}

int pk_get_serials(pk_handle *h, uint64_t *hand_serial, uint64_t *step_serial) {
    if (!h) return PK_E_INVALID_ARG;
    ON_DEVICE(h);
    if (!h->env_pending) FLUSH(h);   // with env steps in flight: the counts of the hands dealt / Game.steps completed so far
    if (hand_serial) HIPCHK(h, hipMemcpyAsync(hand_serial, h->S.hand_serial, (size_t)h->T * 8, hipMemcpyDeviceToHost, h->stream));
    if (step_serial) HIPCHK(h, hipMemcpyAsync(step_serial, h->S.step_serial, (size_t)h->T * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PK_OK;
}

int pk_set_serials(pk_handle *h, const uint64_t *hand_serial, const uint64_t *step_serial) {
    if (!h) return PK_E_INVALID_ARG;
    ON_DEVICE(h);
    FLUSH(h);
    if (hand_serial) HIPCHK(h, hipMemcpyAsync(h->S.hand_serial, hand_serial, (size_t)h->T * 8, hipMemcpyHostToDevice, h->stream));
    if (step_serial) HIPCHK(h, hipMemcpyAsync(h->S.step_serial, step_serial, (size_t)h->T * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PK_OK;
}

int pk_get_cards(pk_handle *h, uint8_t *out) {
    if (!h || !out) return PK_E_INVALID_ARG;
    int K = 5 + 2 * h->N;
    size_t n = (size_t)h->T * K;
    return export_to_host(h, out, n, [&] {
        hipLaunchKernelGGL(k_export_cards, dim3(flat_grid(n)), dim3(256), 0, h->stream, h->S.cards, h->T, K, (uint8_t *)h->d_export);
    });
}

int pk_get_hand_ranks(pk_handle *h, uint8_t *rank, uint32_t *kick) {
    if (!h || !rank || !kick) return PK_E_INVALID_ARG;
    ON_DEVICE(h);
    FLUSH(h);
    size_t n = (size_t)h->T * h->N;
    uint32_t *dk = (uint32_t *)h->d_export;
    uint8_t *dr = (uint8_t *)h->d_export + n * 4;
    if (n * 5 > h->export_bytes) return h->fail(PK_E_INVALID_ARG, "export buffer too small");
    hipLaunchKernelGGL(k_export_show, dim3(flat_grid(n)), dim3(256), 0, h->stream, h->S.show, h->T, h->N, dr, dk);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(kick, dk, n * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(rank, dr, n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PK_OK;
}

int pk_get_obs(pk_handle *h, int player, double *out) {
    if (!h || !out || bad_player(h, player)) return h ? h->fail(PK_E_INVALID_ARG, "pk_get_obs: bad argument") : PK_E_INVALID_ARG;
    size_t bytes = (size_t)h->T * PK_OBS_DIM(h->N) * 8;
    return export_to_host(h, out, bytes, [&] {
        hipLaunchKernelGGL(k_obs, dim3(flat_grid(h->T)), dim3(256), 0, h->stream, h->S, h->N, player, (double *)h->d_export);
    });
}

int pk_get_obs_d(pk_handle *h, int player, double *out_d) {
    if (!h || !out_d || bad_player(h, player)) return h ? h->fail(PK_E_INVALID_ARG, "pk_get_obs_d: bad argument") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    FLUSH(h);
    hipLaunchKernelGGL(k_obs, dim3(flat_grid(h->T)), dim3(256), 0, h->stream, h->S, h->N, player, out_d);
    HIPCHK(h, hipGetLastError());
    return PK_OK;
}

int pk_get_valid_actions_d(pk_handle *h, int player, uint8_t *out_d) {
    if (!h || !out_d || bad_player(h, player)) return h ? h->fail(PK_E_INVALID_ARG, "pk_get_valid_actions_d: bad argument") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    FLUSH(h);
    size_t n = (size_t)h->T * PK_NUM_MOVES;
    hipLaunchKernelGGL(k_export_valid, dim3(flat_grid(n)), dim3(256), 0, h->stream, h->S, h->N, player, out_d);
    HIPCHK(h, hipGetLastError());
    return PK_OK;
}

int pk_env_reset_d(pk_handle *h, const uint8_t *mask_d, int opp_policy) {
    if (!h || opp_policy < 0 || opp_policy > 1) return h ? h->fail(PK_E_INVALID_ARG, "pk_env_reset_d: bad argument") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    FLUSH(h);
    DISPATCH_N(h, k_env_reset, table_grid(h), (const State *)h->d_S, h->hot, mask_d, opp_policy, scaled_park(h));
    HIPCHK(h, hipGetLastError());
    return PK_OK;
}

int pk_env_step_d(pk_handle *h, const int32_t *actions_d, int opp_policy, double *reward_d, uint8_t *done_d,
                  uint8_t *hand_d, uint8_t *terr_d) {
    if (!h || !actions_d || !reward_d || !done_d || !hand_d || !terr_d || opp_policy < 0 || opp_policy > 1)
        return h ? h->fail(PK_E_INVALID_ARG, "pk_env_step_d: bad argument") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    FLUSH(h);
    DISPATCH_N(h, k_env_step, table_grid(h), (const State *)h->d_S, h->hot, actions_d, -1, opp_policy, 0, reward_d, done_d, hand_d, terr_d, (double *)nullptr, scaled_park(h));
    HIPCHK(h, hipGetLastError());
    return PK_OK;
}

int pk_env_step_fused_d(pk_handle *h, const int32_t *actions_d, int seat0_policy, int opp_policy, int auto_reset,
                        double *reward_d, uint8_t *done_d, uint8_t *hand_d, uint8_t *terr_d, double *obs_d) {
    if (!h || !reward_d || !done_d || !hand_d || !terr_d || opp_policy < 0 || opp_policy > 1 ||
        (!actions_d && (seat0_policy < 0 || seat0_policy > 1)))
        return h ? h->fail(PK_E_INVALID_ARG, "pk_env_step_fused_d: bad argument") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    FLUSH(h);
    DISPATCH_N(h, k_env_step, table_grid(h), (const State *)h->d_S, h->hot, actions_d, actions_d ? -1 : seat0_policy, opp_policy,
               auto_reset ? 1 : 0, reward_d, done_d, hand_d, terr_d, obs_d, scaled_park(h));
    HIPCHK(h, hipGetLastError());
    return PK_OK;
}

int pk_env_step_async_d(pk_handle *h, const int32_t *actions_d, int seat0_policy, int opp_policy, int auto_reset, int max_passes,
                        double *reward_d, uint8_t *done_d, uint8_t *hand_d, uint8_t *terr_d, double *obs_d, uint8_t *ready_d) {
    if (!h || !reward_d || !done_d || !hand_d || !terr_d || !ready_d || opp_policy < 0 || opp_policy > 1 ||
        (!actions_d && (seat0_policy < 0 || seat0_policy > 1)))
        return h ? h->fail(PK_E_INVALID_ARG, "pk_env_step_async_d: bad argument") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    int rc = flush_rollout(h);
    if (rc) return rc;
    DISPATCH_N(h, k_env_step_async, table_grid(h), (const State *)h->d_S, h->hot, actions_d, actions_d ? -1 : seat0_policy, opp_policy,
               auto_reset ? 1 : 0, reward_d, done_d, hand_d, terr_d, obs_d, scaled_park(h), ready_d, max_passes > 0 ? max_passes : 0);
    HIPCHK(h, hipGetLastError());
    h->env_pending = max_passes > 0;
    return PK_OK;
}

int pk_pick_actions_d(pk_handle *h, int policy, int32_t *actions_d) {
    if (!h || !actions_d || policy < 0 || policy > 1) return h ? h->fail(PK_E_INVALID_ARG, "pk_pick_actions_d: bad argument") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    FLUSH(h);
    DISPATCH_N(h, k_pick, table_grid(h), h->S, h->hot, policy, actions_d);
    HIPCHK(h, hipGetLastError());
    return PK_OK;
}

int pk_pick_actions(pk_handle *h, int policy, int32_t *actions) {
    if (!h || !actions || policy < 0 || policy > 1) return h ? h->fail(PK_E_INVALID_ARG, "pk_pick_actions: bad argument") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    int rc = pk_pick_actions_d(h, policy, h->d_actions);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(actions, h->d_actions, (size_t)h->T * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PK_OK;
}

static int fetch_counters(pk_handle *h, uint64_t *counters) {
    unsigned long long c[PK_NUM_COUNTERS];
    hipLaunchKernelGGL(k_sum_counters, dim3(1), dim3(256), 0, h->stream, h->S.counters, table_grid(h), h->d_totals);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(c, h->d_totals, sizeof(c), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < PK_NUM_COUNTERS; ++i) counters[i] += c[i];
    return PK_OK;
}

// fused: one launch that may leave work for later (counters == NULL) or must complete it (counters != NULL);
// unfused: k_steps complete single-step launches (state round-trips HBM every step).
static int enqueue_rollout(pk_handle *h, int k_steps, int policy, int auto_reset, int fused, bool complete) {
    if (h->env_pending) return flush(h);   // PK_E_BUSY
    if (h->pending && (policy != h->pend_policy || auto_reset != h->pend_auto)) FLUSH(h);  // owed steps keep THEIR agents
    if (!fused) {
        FLUSH(h);
        for (int k = 0; k < k_steps; ++k) {
            int rc = launch_rollout(h, 1, policy, auto_reset, 1);
            if (rc) return rc;
        }
        return PK_OK;
    }
    if (k_steps == 0) return complete ? flush(h) : PK_OK;
    // deferral only in the throughput mode: without auto_reset a table that reports an error stops for the rest of THIS
    // call (and is retried by the next), so calls must not be merged
    return launch_rollout(h, k_steps, policy, auto_reset, (complete || !auto_reset) ? 1 : h->endk);
}

int pk_rollout(pk_handle *h, int k_steps, int policy, int auto_reset, int fused, uint64_t *counters) {
    if (!h || k_steps < 0 || policy < 0 || policy > 1) return h ? h->fail(PK_E_INVALID_ARG, "pk_rollout: bad argument") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    int rc = enqueue_rollout(h, k_steps, policy, auto_reset ? 1 : 0, fused, counters != nullptr);
    if (rc) return rc;
    if (counters) return fetch_counters(h, counters);
    return PK_OK;
}

int pk_get_owed(pk_handle *h, uint32_t *out) {
    if (!h || !out) return PK_E_INVALID_ARG;
    ON_DEVICE(h);
    HIPCHK(h, hipMemcpyAsync(out, h->S.owed, (size_t)h->T * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PK_OK;
}

int pk_flush(pk_handle *h) {
    if (!h) return PK_E_INVALID_ARG;
    ON_DEVICE(h);
    return flush(h);
}

int pk_time_rollout(pk_handle *h, int k_steps, int policy, int auto_reset, int fused, int reps, double *ms_per_launch,
                    uint64_t *counters) {
    if (!h || !ms_per_launch || reps < 1 || k_steps < 0 || policy < 0 || policy > 1)
        return h ? h->fail(PK_E_INVALID_ARG, "pk_time_rollout: bad argument") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    FLUSH(h);
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    for (int r = 0; r < reps; ++r) {
        // k_steps == 0: empty launches (load the tables, store them) -- the fixed cost of a launch, for diagnostics
        int rc = k_steps ? enqueue_rollout(h, k_steps, policy, auto_reset ? 1 : 0, fused, false)
                         : launch_rollout(h, 0, policy, auto_reset ? 1 : 0, 1);
        if (rc) return rc;
    }
    FLUSH(h);  // what the deferred launches left is part of the work that is being timed
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    int launches = reps * (fused ? 1 : k_steps);
    *ms_per_launch = launches ? (double)ms / launches : 0.0;
    if (counters) return fetch_counters(h, counters);
    return PK_OK;
}

int pk_env_reset(pk_handle *h, const uint8_t *mask, int opp_policy) {
    if (!h || opp_policy < 0 || opp_policy > 1) return h ? h->fail(PK_E_INVALID_ARG, "pk_env_reset: bad argument") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    const uint8_t *dmask;
    int rc = upload_mask(h, mask, &dmask);
    if (rc) return rc;
    rc = pk_env_reset_d(h, dmask, opp_policy);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PK_OK;
}

int pk_env_step(pk_handle *h, const int32_t *actions, int opp_policy, double *reward, uint8_t *done, uint8_t *hand,
                uint8_t *terr) {
    if (!h || !actions || !reward || !done || !hand || opp_policy < 0 || opp_policy > 1)
        return h ? h->fail(PK_E_INVALID_ARG, "pk_env_step: bad argument") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    const size_t T = (size_t)h->T;
    HIPCHK(h, hipMemcpyAsync(h->d_actions, actions, T * 4, hipMemcpyHostToDevice, h->stream));
    int rc = pk_env_step_d(h, h->d_actions, opp_policy, h->d_reward, h->d_done, h->d_handf, h->d_terr);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(reward, h->d_reward, T * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(done, h->d_done, T, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(hand, h->d_handf, T, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->h_pinned, h->d_terr, T, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (terr) memcpy(terr, h->h_pinned, T);
    if (any_terr(h->h_pinned, h->T)) return h->fail(PK_E_TABLE, "pk_env_step: per-table error(s), see terr");
    return PK_OK;
}

#ifdef PK_PROFILE
// Diagnostic library only: read and clear the per-block cycle sums (slots: pk_device.hpp PF_*).
int pk_prof_read(pk_handle *h, unsigned long long *out) {
    if (!h || !out) return PK_E_INVALID_ARG;
    ON_DEVICE(h);
    FLUSH(h);
    HIPCHK(h, hipMemcpyAsync(out, h->S.prof, PF_SLOTS * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemsetAsync(h->S.prof, 0, PF_SLOTS * 8, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PK_OK;
}
#endif

int pk_sync(pk_handle *h) {
    if (!h) return PK_E_INVALID_ARG;
    ON_DEVICE(h);
    if (!h->env_pending) FLUSH(h);   // env steps in flight stay in flight: only wait for the launches made so far
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PK_OK;
}

// ---- standalone judger ops (no handle)
struct tmp_handle { std::string err; int fail(int code, const char *what, hipError_t e) { err = what; err += ": "; err += hipGetErrorString(e); g_err = err; return code; } };

static int check_device(int device) {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) { g_err = "no HIP device available; this library has no CPU fallback"; return PK_E_NO_DEVICE; }
    if (device < 0 || device >= ndev) { g_err = "device index out of range"; return PK_E_INVALID_ARG; }
    return PK_OK;
}

// Grow-only device scratch of the host-buffer judger calls, one per device: no hipMalloc / hipFree per call.
static std::mutex g_scratch_mu;
static struct { void *p; size_t cap; } g_scratch[PK_MAX_DEVICES];
static void *scratch(int device, size_t bytes) {  // caller holds g_scratch_mu and has the device current
    auto &s = g_scratch[device];
    if (s.cap < bytes) {
        if (s.p) (void)hipFree(s.p);
        s.p = nullptr; s.cap = 0;
        size_t want = bytes < (1u << 20) ? (1u << 20) : bytes + bytes / 2;
        if (hipMalloc(&s.p, want) != hipSuccess) { s.p = nullptr; return nullptr; }
        s.cap = want;
    }
    return s.p;
}

int pk_eval_hands_d(int device, const uint8_t *cards_d, const uint8_t *ncards_d, size_t m, uint8_t *rank_d, uint32_t *kick_d,
                    uint8_t *nkick_d, void *stream) {
    if (!cards_d || !rank_d || !kick_d) { g_err = "pk_eval_hands_d: NULL buffer"; return PK_E_INVALID_ARG; }
    int rc = check_device(device);
    if (rc) return rc;
    DeviceGuard guard(device);
    if (!guard.ok) { g_err = "hipSetDevice failed"; return PK_E_HIP; }
    if (m == 0) return PK_OK;
    hipLaunchKernelGGL(k_eval_hands, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cards_d, ncards_d, m, rank_d, kick_d, nkick_d);
    if (hipGetLastError() != hipSuccess) { g_err = "pk_eval_hands_d: launch failed"; return PK_E_HIP; }
    return PK_OK;
}

int pk_eval_hands(int device, const uint8_t *cards, const uint8_t *ncards, size_t m, uint8_t *rank, uint32_t *kick,
                  uint8_t *nkick) {
    if (!cards || !rank || !kick) { g_err = "pk_eval_hands: NULL buffer"; return PK_E_INVALID_ARG; }
    int rc = check_device(device);
    if (rc) return rc;
    if (device >= PK_MAX_DEVICES) { g_err = "pk_eval_hands: device index beyond PK_MAX_DEVICES"; return PK_E_INVALID_ARG; }
    DeviceGuard guard(device);
    if (!guard.ok) { g_err = "hipSetDevice failed"; return PK_E_HIP; }
    if (m == 0) return PK_OK;
    tmp_handle th;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    size_t off_n = m * 7, off_r = off_n + m, off_nk = off_r + m, off_k = (off_nk + m + 3) & ~(size_t)3, total = off_k + m * 4;
    uint8_t *d = (uint8_t *)scratch(device, total);
    if (!d) { g_err = "pk_eval_hands: out of device memory"; return PK_E_OOM; }
    hipError_t e = hipMemcpyAsync(d, cards, m * 7, hipMemcpyHostToDevice, 0);
    if (e == hipSuccess && ncards) e = hipMemcpyAsync(d + off_n, ncards, m, hipMemcpyHostToDevice, 0);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_eval_hands, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, 0, d, ncards ? d + off_n : nullptr, m,
                           d + off_r, (uint32_t *)(d + off_k), d + off_nk);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(rank, d + off_r, m, hipMemcpyDeviceToHost, 0);
    if (e == hipSuccess) e = hipMemcpyAsync(kick, d + off_k, m * 4, hipMemcpyDeviceToHost, 0);
    if (e == hipSuccess && nkick) e = hipMemcpyAsync(nkick, d + off_nk, m, hipMemcpyDeviceToHost, 0);
    if (e == hipSuccess) e = hipStreamSynchronize(0);
    if (e != hipSuccess) return th.fail(PK_E_HIP, "pk_eval_hands", e);
    return PK_OK;
}

int pk_compare_rankings(int device, const uint8_t *rank, const uint32_t *kick, int n, size_t m, uint8_t *onehot) {
    if (!rank || !kick || !onehot || n < 1 || n > 32) { g_err = "pk_compare_rankings: bad argument (1 <= n <= 32)"; return PK_E_INVALID_ARG; }
    int rc = check_device(device);
    if (rc) return rc;
    if (device >= PK_MAX_DEVICES) { g_err = "pk_compare_rankings: device index beyond PK_MAX_DEVICES"; return PK_E_INVALID_ARG; }
    DeviceGuard guard(device);
    if (!guard.ok) { g_err = "hipSetDevice failed"; return PK_E_HIP; }
    if (m == 0) return PK_OK;
    tmp_handle th;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    size_t cnt = m * (size_t)n;
    size_t off_r = cnt * 4, off_o = off_r + cnt, total = off_o + cnt;
    uint8_t *d = (uint8_t *)scratch(device, total);
    if (!d) { g_err = "pk_compare_rankings: out of device memory"; return PK_E_OOM; }
    hipError_t e = hipMemcpyAsync(d, kick, cnt * 4, hipMemcpyHostToDevice, 0);
    if (e == hipSuccess) e = hipMemcpyAsync(d + off_r, rank, cnt, hipMemcpyHostToDevice, 0);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_compare, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, 0, d + off_r, (const uint32_t *)d, n, m, d + off_o);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(onehot, d + off_o, cnt, hipMemcpyDeviceToHost, 0);
    if (e == hipSuccess) e = hipStreamSynchronize(0);
    if (e != hipSuccess) return th.fail(PK_E_HIP, "pk_compare_rankings", e);
    return PK_OK;
}

static int launch_eval7(const uint64_t *hands_d, size_t m, uint32_t *out_d, int distinct) {
    const unsigned grid = 256 * 16;  // 16 workgroups per CU, grid-stride over the rest
    const bool vec = (((uintptr_t)hands_d & 15) | ((uintptr_t)out_d & 7)) == 0;  // 16-byte loads / 8-byte stores need it
    if (distinct) {
        if (vec) hipLaunchKernelGGL((k_eval7_stream<true, true>), dim3(grid), dim3(256), 0, 0, hands_d, m, out_d);
        else hipLaunchKernelGGL((k_eval7_stream<true, false>), dim3(grid), dim3(256), 0, 0, hands_d, m, out_d);
    } else {
        if (vec) hipLaunchKernelGGL((k_eval7_stream<false, true>), dim3(grid), dim3(256), 0, 0, hands_d, m, out_d);
        else hipLaunchKernelGGL((k_eval7_stream<false, false>), dim3(grid), dim3(256), 0, 0, hands_d, m, out_d);
    }
    return hipGetLastError() == hipSuccess ? PK_OK : PK_E_HIP;
}

int pk_eval7_d(int device, const uint64_t *hands_d, size_t m, uint32_t *out_d, int distinct) {
    if (!hands_d || !out_d || ((uintptr_t)hands_d & 7) || ((uintptr_t)out_d & 3)) { g_err = "pk_eval7_d: NULL or misaligned buffer"; return PK_E_INVALID_ARG; }
    int rc = check_device(device);
    if (rc) return rc;
    DeviceGuard guard(device);
    if (!guard.ok) { g_err = "hipSetDevice failed"; return PK_E_HIP; }
    if (m == 0) return PK_OK;
    if (launch_eval7(hands_d, m, out_d, distinct) != PK_OK || hipDeviceSynchronize() != hipSuccess) { g_err = "pk_eval7_d: launch failed"; return PK_E_HIP; }
    return PK_OK;
}

int pk_make_hands_d(int device, uint64_t seed, size_t m, uint64_t *hands_d) {
    if (!hands_d) { g_err = "pk_make_hands_d: NULL buffer"; return PK_E_INVALID_ARG; }
    int rc = check_device(device);
    if (rc) return rc;
    DeviceGuard guard(device);
    if (!guard.ok) { g_err = "hipSetDevice failed"; return PK_E_HIP; }
    if (m == 0) return PK_OK;
    Hot H{};
    H.key0 = (uint32_t)seed; H.key1 = (uint32_t)(seed >> 32);
    hipLaunchKernelGGL(k_make_hands, dim3(256 * 16), dim3(256), 0, 0, H, m, hands_d);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { g_err = "pk_make_hands_d: launch failed"; return PK_E_HIP; }
    return PK_OK;
}

int pk_time_eval7_d(int device, const uint64_t *hands_d, size_t m, uint32_t *out_d, int distinct, int reps, double *ms_per_pass) {
    if (!hands_d || !out_d || !ms_per_pass || reps < 1) { g_err = "pk_time_eval7_d: bad argument"; return PK_E_INVALID_ARG; }
    int rc = check_device(device);
    if (rc) return rc;
    DeviceGuard guard(device);
    if (!guard.ok) { g_err = "hipSetDevice failed"; return PK_E_HIP; }
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { g_err = "hipEventCreate failed"; return PK_E_HIP; }
    launch_eval7(hands_d, m, out_d, distinct);  // warm (instruction cache, clocks)
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) launch_eval7(hands_d, m, out_d, distinct);
    (void)hipEventRecord(e1, 0);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (e != hipSuccess) { g_err = std::string("pk_time_eval7_d: ") + hipGetErrorString(e); return PK_E_HIP; }
    *ms_per_pass = (double)ms / reps;
    return PK_OK;
}

// Test hook (declared in pokerl_hip.h as part of the judger surface): values rank<<20|kick of all 7-card hands whose
// two lowest canonical indices are (a, b), lexicographic order.  out holds C(51-b, 5) words.
int pk_eval7_prefix(int device, int a, int b, int fast, uint32_t *out, size_t *count_out) {
    if (!out || a < 0 || b <= a || b > 51) { g_err = "pk_eval7_prefix: bad argument"; return PK_E_INVALID_ARG; }
    int rc = check_device(device);
    if (rc) return rc;
    if (device >= PK_MAX_DEVICES) { g_err = "pk_eval7_prefix: device index beyond PK_MAX_DEVICES"; return PK_E_INVALID_ARG; }
    DeviceGuard guard(device);
    if (!guard.ok) { g_err = "hipSetDevice failed"; return PK_E_HIP; }
    int n = 51 - b;
    size_t count = n >= 5 ? (size_t)n * (n - 1) * (n - 2) * (n - 3) * (n - 4) / 120 : 0;
    if (count_out) *count_out = count;
    if (!count) return PK_OK;
    tmp_handle th;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    uint32_t *d = (uint32_t *)scratch(device, count * 4);
    if (!d) { g_err = "pk_eval7_prefix: out of device memory"; return PK_E_OOM; }
    hipLaunchKernelGGL(k_eval7_prefix, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, 0, a, b, fast, (uint32_t)count, d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(out, d, count * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return th.fail(PK_E_HIP, "pk_eval7_prefix", e);
    return PK_OK;
}

}  // extern "C"
