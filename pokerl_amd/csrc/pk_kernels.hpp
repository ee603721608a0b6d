// pk_kernels.hpp -- the __global__ kernels of libpokerl_hip.so (included by pk_api.hip, the host side of the C ABI; the
// per-table state machine they run is pk_device.hpp).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include "pk_device.hpp"

using namespace pk;

// ================================================================================================ kernels
// One table per lane, one wavefront per workgroup.  At the headline size (65 536 tables) that is exactly one wave per
// SIMD (256 CUs x 4), so occupancy cannot hide anything and the register budget is the whole 512-entry file:
// __launch_bounds__(64) lets the compiler keep a table's full state in VGPRs instead of spilling to scratch.
#define PK_TABLE_BLOCK 64
static_assert(PK_TABLE_BLOCK == PK_WAVE, "one wavefront per workgroup: what PK_QSYNC (pk_device.hpp) and every wave-level ballot of the step machine assume");

template <typename ST>
__device__ __forceinline__ void wave_add_counters(const ST &S, uint32_t steps, uint32_t hands, uint32_t evals, uint32_t games) {
    // 64-wide butterfly reduction in registers, then lane 0 adds to the slot this wavefront owns (plain RMW, no atomics).
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        steps += __shfl_down(steps, off, 64); hands += __shfl_down(hands, off, 64);
        evals += __shfl_down(evals, off, 64); games += __shfl_down(games, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        const auto slot = as_global(S.counters) + (size_t)blockIdx.x * PK_NUM_COUNTERS;
        slot[PK_C_STEPS] += steps; slot[PK_C_HANDS] += hands; slot[PK_C_EVALS] += evals; slot[PK_C_GAMES] += games;
    }
}

#ifndef PK_TABLES_ONLY   // (the per-seat-count translation units, pk_tables.hip, hold the table kernel templates only)
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

#endif

// Game.StateView(player to act), game.py:117-131 -- what `game.active_state` builds right after every step (game.py:323-332) -- written
// from the REGISTERS of the kernel that made the step.  The two row forms as arrays of compile-time-indexed 64-bit words (they alias the table's
// own registers): the dense f64 row (layout: pokerl_hip.h PK_OBS_DIM; the row k_obs builds from HBM) ...
// First 8 bytes of a packed observation row (pokerl_hip.h PK_OBS_PACKED_BYTES): seat, turn, valid mask, hole cards, flop.
__device__ __forceinline__ uint64_t obs_packed_header0(uint32_t who, uint32_t turn, uint32_t vmask, uint32_t h0, uint32_t h1, uint32_t c0, uint32_t c1, uint32_t c2) {
    return (uint64_t)(who | (turn << 8) | ((vmask & 0x7fu) << 16) | (h0 << 24)) | ((uint64_t)(h1 | (c0 << 8) | (c1 << 16) | (c2 << 24)) << 32);
}
template <int N>
__device__ __forceinline__ void obs_row_words(const Table<N> &tb, uint32_t vmask, double (&o)[PK_OBS_DIM(N)]) {
    const int who = tb.active;
    o[0] = who; o[1] = tb.turn; o[2] = tb.min_raise;
    PK_FOR(a, PK_NUM_MOVES) o[3 + a] = (vmask >> a) & 1; PK_END
    double h0 = 0.0, h1 = 0.0;
    PK_FOR(p, N) h0 = (who == p) ? (double)tb.card(5 + 2 * p) : h0; h1 = (who == p) ? (double)tb.card(6 + 2 * p) : h1; PK_END
    o[10] = h0; o[11] = h1;                                                    // game.py:385-389
    PK_FOR(c, 5) o[12 + c] = (tb.turn != 0 && c < tb.turn + 2) ? (double)tb.card(c) : -1.0; PK_END   // game.py:278
    PK_FOR(p, N) o[17 + p] = tb.credits[p]; o[17 + N + p] = tb.bets[p]; o[17 + 2 * N + p] = tb.pending[p]; PK_END
}
// ... and the compact one (layout: pokerl_hip.h PK_OBS_PACKED_BYTES; k_obs_packed's row): 16 header bytes + (3N+1) f64 = 3(N+1) words
template <int N>
__device__ __forceinline__ void obs_packed_words(const Table<N> &tb, uint32_t vmask, uint64_t (&o)[3 * N + 3]) {
    const uint32_t who = (uint32_t)tb.active;
    uint32_t h0 = 0, h1 = 0;
    PK_FOR(p, N) h0 = (who == (uint32_t)p) ? tb.card(5 + 2 * p) : h0; h1 = (who == (uint32_t)p) ? tb.card(6 + 2 * p) : h1; PK_END
    uint32_t cc[5];
    PK_FOR(c, 5) cc[c] = (tb.turn != 0 && c < tb.turn + 2) ? tb.card(c) : 0xffu; PK_END
    o[0] = obs_packed_header0(who, (uint32_t)tb.turn, vmask, h0, h1, cc[0], cc[1], cc[2]);
    o[1] = (uint64_t)cc[3] | ((uint64_t)cc[4] << 8);
    o[2] = (uint64_t)__double_as_longlong(tb.min_raise);
    PK_FOR(p, N)
        o[3 + p] = (uint64_t)__double_as_longlong(tb.credits[p]); o[3 + N + p] = (uint64_t)__double_as_longlong(tb.bets[p]);
        o[3 + 2 * N + p] = (uint64_t)__double_as_longlong(tb.pending[p]);
    PK_END
}
// Each lane stores its own row: 64 lanes x 8 bytes at a stride of one row (168 / 280 bytes at six seats) per instruction.  The PokerGameEnv kernels' form.
template <int N>
__device__ __forceinline__ void write_obs_row(const Table<N> &tb, uint32_t vmask, PK_GLOBAL double *o) {
    double w[PK_OBS_DIM(N)];
    obs_row_words<N>(tb, vmask, w);
    PK_FOR(k, PK_OBS_DIM(N)) o[k] = w[k]; PK_END
}
template <int N>
__device__ __forceinline__ void write_obs_packed_row(const Table<N> &tb, uint32_t vmask, PK_GLOBAL uint64_t *o) {
    uint64_t w[3 * N + 3];
    obs_packed_words<N>(tb, vmask, w);
    PK_FOR(k, 3 * N + 3) o[k] = w[k]; PK_END
}
// (Tried in round 6: the wave's 64 rows -- one contiguous piece of the output -- staged through the idle showdown queue in LDS in chunks of N+1
//  words and stored with consecutive lanes on consecutive words, 56-byte runs instead of 64 scattered 8-byte pieces per instruction.  No faster:
//  Game.step + packed row at 65 536 x 6 21.7 us against 21.1 us with the plain per-lane stores, at 1 M tables 204 against 191 us, the dense row 269
//  against 194 us -- L2 merges the scattered pieces of a line anyway, and the LDS round trip with its barriers is pure extra issue time:
//  profiles/r06_step_obs_ab.txt.)

template <int N>
__global__ void __launch_bounds__(PK_TABLE_BLOCK) k_reset(State S, Hot H, const uint8_t *mask, int mask_bits, int dealer) {  // Game.reset, game.py:397-412
    int t = blockIdx.x * H.tpb + threadIdx.x;
    if ((int)threadIdx.x >= H.tpb || t >= S.T) return;
    if (mask && !(mask[t] & mask_bits)) return;       // (pk_reset: any non-zero byte; pk_reset_d: the caller's bits, e.g. PK_FLAG_GAME_OVER of pk_step_d's flags)
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
// POLICY == PK_POLICY_EXTERNAL is Game.step itself (k_step): ONE step per table with the caller's action, read once;
// an invalid one leaves the table untouched (game.py:649-651).  PASSES = betting passes between two looks at the parked
// lanes: four for the multi-step kernels, one for the single-step ones (a lane makes one step; three more passes
// would run empty).
// k_step's formal parameter = the layout of its kernarg segment: the two output pointers are read through the kernarg
// segment pointer after the loop (see EnvArgs below for why).
struct StepKernArgs { const State *Sp; Hot H; const int32_t *actions; uint8_t *flags, *terr; int park, auto_reset; uint8_t *ready; int max_end;
                      double *obs; uint8_t *obs_packed; };   // pk_set_step_obs: the StateView row of the player to act, from registers (NULL: off)
// BOUNDED (pk_step_async_d, only with the external policy): a launch runs at most `max_end` end_blocks.  A Game.step that rolls on through
// further hands (the reference's next_player loop plays whole hands nobody can act in, game.py:607-611: ~1 table in 10 000 per step, yet
// the SLOWEST table of 65 536 needs ~4 hands and every end_hand is ~3 us of serial work) stays IN FLIGHT -- its machine state is in the
// table's cursor bits and State::mid, as for a deferred rollout launch -- and the next launch carries on with it, ignoring actions[t];
// ready[t] says whose step has returned.  Per table the sequence of steps, flags and RNG draws is the synchronous one.
#ifndef PK_STEP_ROLLING
#define PK_STEP_ROLLING 16   // hands rolled inside one Game.step from which a bounded launch carries that step to its end (as PK_ENV_ROLLING)
#endif
// TAB (k_rollout_tab, up to six seats): the showdown hands are ranked by the table-driven evaluator, whose 32 KB table the wave stages in its LDS.
template <int N, bool ONE_PASS, int POLICY, int PASSES = 0, bool BOUNDED = false, bool TAB = false>
__device__ __forceinline__ void rollout_body(const State *__restrict__ Sp, const Hot &H, int K, int auto_reset, int park, int slack, int clear_terr,
                                             const int32_t *actions = nullptr, const StepKernArgs *ext = nullptr) {
    // Array bases by pointer (loaded only where the table is loaded / stored), loop scalars by value: see pk::Hot.
    constexpr int policy = POLICY;
    const auto &S = *as_global(Sp);
    using LDS = std::conditional_t<TAB, LdsTab<N, POLICY == PK_POLICY_RANDOM>, Lds<N>>;
    __shared__ LDS lds;
    if constexpr (TAB) {     // the workgroup's copy of the evaluator's table: 32 x (16-byte load + 16-byte LDS write) per lane, eight loads in flight
        static_assert(ONE_PASS && POLICY != PK_POLICY_EXTERNAL, "k_rollout_tab is a fused-rollout kernel");
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const auto g_tab = (PK_GLOBAL const u32x4 *)as_global(S.evtab);
        const int lane0 = threadIdx.x & (PK_WAVE - 1);
#pragma unroll 1
        for (int i = lane0; i < EVAL7_TAB_WORDS / 4; i += 8 * PK_WAVE) {
            u32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = g_tab[i + j * PK_WAVE];
#pragma unroll
            for (int j = 0; j < 8; ++j) reinterpret_cast<u32x4 *>(lds.evtab)[i + j * PK_WAVE] = v[j];
        }
    }
    const int t = blockIdx.x * H.tpb + threadIdx.x;
    const bool live = (int)threadIdx.x < H.tpb && t < S.T;
    const uint32_t table_id = H.table_id_base + (uint32_t)t;
    Table<N> tb;
    uint32_t owed = 0;
    Table<N>::stage_fresh(lds, H.fresh);
    if constexpr (!TAB || POLICY == PK_POLICY_RANDOM) stage_nth(lds);      // (the table variant of the all-in agents has no such table: LdsTab<N, false>)
    constexpr bool EXTERNAL = POLICY == PK_POLICY_EXTERNAL;
    constexpr bool PAY = PASSES != 1;        // the single-step kernels (k_step, k_rollout_single) move the payoffs only where a hand ended
    if (live) { tb.template load<PAY>(S, t); tb.hands_this_step = (int)as_global(S.mid)[t]; owed = as_global(S.owed)[t] + (uint32_t)K; } else tb.blank();
    uint32_t steps = 0;
    bool alive = live;
    ActionRing ring;
    double high_bet;
    int ext_action = -1;
    bool ext_ok = true;
    static_assert(!BOUNDED || POLICY == PK_POLICY_EXTERNAL, "bounded launches are pk_step_async_d's");
    if (EXTERNAL) {                                                                // game.py:648-651
        const uint32_t vm = tb.valid_mask(high_bet);
        if constexpr (BOUNDED) ext_action = (live && actions) ? as_global(actions)[t] : -1;    // (actions == NULL: a drain that steps nothing, pk_step_async_d)
        else ext_action = live ? as_global(actions)[t] : -1;
        const bool carried = BOUNDED && live && tb.stepped;                        // a step of an earlier launch is in flight: its action was taken then
        ext_ok = carried || (live && ext_action >= 0 && ext_action < PK_NUM_MOVES && ((vm >> ext_action) & 1));
        owed = ext_ok ? 1u : 0u;                                                   // (the host has flushed: nothing was owed)
        alive = ext_ok;
    }
    int nend = 0;
    uint32_t late = (live ? 1u : 0u) | (ext_ok ? 2u : 0u);   // the epilogue's lane predicates travel through the loop in ONE VGPR
    if (EXTERNAL) asm volatile("" : "+v"(late));               // (as lane masks: an SGPR pair each, live across the loop; see env_step_body)
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
    constexpr int NPASS = PASSES > 0 ? PASSES : PK_BET_PASSES;
    constexpr int TAILS = PASSES > 0 ? (1 << (PASSES - 1)) | (PK_TAIL_MASK & ((1 << PASSES) - 1)) : PK_TAIL_MASK;
    for (;;) {
        // Nothing is in flight at the top of an iteration.  Without this the compiler cannot rule out that a table
        // register still waits for the global loads before the loop or for end_block's LDS reads (both sit in
        // conditionally executed blocks), and parks a full s_waitcnt right behind the first LDS read of every betting
        // pass: the action ring's latency was exposed three passes out of four.
        if (policy == PK_POLICY_RANDOM) __builtin_amdgcn_s_waitcnt(0);   // (the all-in kernel has no LDS read in its passes)
        if constexpr (policy == PK_POLICY_RANDOM) ring.ensure(lds, H, table_id, tb.step_serial, alive && owed > 0, NPASS);   // wave-uniform
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const bool go = alive && tb.lstate == LS_DONE && owed > 0;
            uint32_t word = 0;
            if constexpr (policy == PK_POLICY_RANDOM) word = ActionRing::peek(lds, tb.step_serial);
            if (go) {
                uint32_t mask = tb.valid_mask(high_bet);
                int act;
                if constexpr (EXTERNAL) act = ext_action;
                else if constexpr (policy == PK_POLICY_ALLIN) act = (int)MV_ALL_IN;
                else if constexpr (policy == PK_POLICY_CALL) act = call_action(mask);
                else act = action_from_draw_lds(lds, ActionRing::half_of(word, tb.step_serial), mask);
                tb.begin_step(H, act, high_bet);
            }
            PK_PROF(tb.prof.lap(PF_ACTION);)
            tb.scan_first();      // every lane in LS_SCAN: the steps just begun and the ones end_block carried into a new hand
            if ((TAILS >> pass) & 1) tb.cursor_tail();
            PK_PROF(tb.prof.count(PF_N_CURSOR);)
            retire();
            PK_PROF(tb.prof.lap(PF_CURSOR);)
        }
        // showdowns waiting in their side-pot loop (LS_POT) count towards `park` like arrivals (weights 0 and 1/2 measured no better)
        const int parked = __popcll(__ballot(tb.parked()));
        const int runnable = __popcll(__ballot(alive && tb.lstate == LS_DONE && owed > 0));
        if (parked + runnable < quit) break;
        if (parked >= park || runnable == 0) {
            if constexpr (BOUNDED) {   // the budget is used up: what is parked stays in flight -- unless a step is rolling hand after hand
                if (ext->max_end > 0 && nend >= ext->max_end && !__any(tb.stepped && tb.hands_this_step >= PK_STEP_ROLLING)) break;
                ++nend;
            }
            tb.template end_block<ONE_PASS>(H, t, table_id, lds, auto_reset != 0);
            retire();
        }
    }
    // LS_POT never survives a kernel.  A launch that ends early (deferred work) simply takes such a showdown back to
    // LS_END: end_hand up to there is idempotent (the pending bets are committed and zero, payoffs are re-zeroed, the
    // side pots restart from the unchanged committed bets), so the next launch redoes it together with its own arrivals.
    if (tb.lstate == LS_POT) { tb.lstate = LS_END; tb.evals -= (uint32_t)__popc((tb.st_called | tb.st_allin) & Table<N>::FULL); }
    if (EXTERNAL) {
        asm volatile("" : "+v"(late));
        const bool ext_ok_l = late & 2u;
        if (late & 1u) {
            const auto flags_out = as_global(ext->flags), terr_out = as_global(ext->terr);
            const bool returned = !BOUNDED || !ext_ok_l || tb.lstate == LS_DONE;   // (an invalid action returns at once, table untouched)
            if (ext_ok_l) {
                tb.template store<PAY>(S, t);                                      // (a step in flight: with its machine state in the cursor bits)
                tb.store_show(S.show, S.T, t, lds);
                as_global(S.mid)[t] = returned ? 0u : (uint32_t)tb.hands_this_step;
                if (returned) as_global(S.valid)[t] = (uint8_t)tb.valid_mask(high_bet);
            }
            if (returned) {
                const uint8_t te = ext_ok_l ? (uint8_t)(tb.terr | tb.seen) : (uint8_t)PK_TERR_INVALID_ACTION;   // (seen: the error bits of a table pk_step_auto_d reset on the spot)
                flags_out[t] = ext_ok_l ? (uint8_t)tb.flags : (uint8_t)0;          // :649-651: no mutation
                as_global(S.terr)[t] = te;
                if (terr_out) terr_out[t] = te;
            }
            if constexpr (BOUNDED) as_global(ext->ready)[t] = returned ? 1 : 0;
            // game.active_state (game.py:323-332: the StateView of the player to act, built right after every step) for a learner that drives
            // Game.step itself: the row of every table whose step has returned -- also of one whose action was refused (its table is as it
            // was), and after an auto-reset the first view of the new game -- from the registers the step left, not by a second launch that
            // re-reads the table this one has just stored (k_obs / k_obs_packed write the same rows).  A step still in flight: row untouched.
            if (returned) {
                const auto obs = as_global(ext->obs);
                const auto obs_packed = as_global(ext->obs_packed);
                if (obs || obs_packed) {
                    const uint32_t vmask = tb.valid_mask(high_bet);
                    if (obs) write_obs_row<N>(tb, vmask, obs + (size_t)t * PK_OBS_DIM(N));
                    if (obs_packed) write_obs_packed_row<N>(tb, vmask, (PK_GLOBAL uint64_t *)(obs_packed + (size_t)t * PK_OBS_PACKED_BYTES(N)));
                }
            }
        }
        PK_PROF(tb.prof.flush(S.prof);)
        return;                                                                    // Game.step is not counted as rollout work
    }
    if (live) {
        tb.template store<PAY>(S, t);
        tb.store_show(S.show, S.T, t, lds);
        as_global(S.owed)[t] = owed; as_global(S.mid)[t] = (uint32_t)tb.hands_this_step;
        as_global(S.valid)[t] = (uint8_t)tb.valid_mask(high_bet);
        const auto g_terr = as_global(S.terr);
        g_terr[t] = (uint8_t)((clear_terr ? 0 : g_terr[t]) | tb.terr | tb.seen);
    }
    wave_add_counters(S, steps, tb.hands, tb.evals, tb.games);  // every lane takes part in the shuffles
    PK_PROF(tb.prof.flush(S.prof);)
}

// From 13 seats on the table state no longer fits 256 VGPRs (k_rollout<15> spilled 131 registers to scratch under the two-wave
// cap in round 3; running the whole side-pot loop per call -- no pot_wb / pot_hv across the betting passes -- did not cure it:
// the pressure sits inside end_block).  Those instantiations give up the second wave per SIMD instead: one wave may use the
// unified 512-register file, the compiler parks what does not fit the 256 architectural VGPRs in AGPRs (one v_accvgpr move per
// access) and nothing goes to scratch memory.  At 65 536 tables a SIMD holds one wave anyway.
#define PK_WAVES_PER_SIMD_N(N) ((N) <= 12 ? 2 : 1)
// Batches of up to two waves per SIMD: registers capped at 256 (no instantiation needs more; N = 10 uses 245), which
// also steers the max-ILP scheduler to a slightly better schedule than an unlimited budget (24.1 vs 23.4 G at 65 536 x 6 when it was introduced) ...
template <int N>
__global__ void __launch_bounds__(PK_TABLE_BLOCK, PK_WAVES_PER_SIMD_N(N)) PK_ROLLOUT_ATTR k_rollout(const State *__restrict__ Sp, Hot H, int K, int auto_reset, int park, int slack, int clear_terr) {
    rollout_body<N, true, PK_POLICY_RANDOM>(Sp, H, K, auto_reset, park, slack, clear_terr);
}
// ... with the showdown hands ranked by the table-driven evaluator (rollout_body<..., TAB>): up to six seats, batches of at most one wave per SIMD
// (40 752 bytes of LDS per wave: a second wave per SIMD would not fit the CU), launches long enough to pay for staging the table (pk_api.hip).
template <int N>
__global__ void __launch_bounds__(PK_TABLE_BLOCK, PK_WAVES_PER_SIMD_N(N)) PK_ROLLOUT_ATTR k_rollout_tab(const State *__restrict__ Sp, Hot H, int K, int auto_reset, int park, int slack, int clear_terr) {
    rollout_body<N, true, PK_POLICY_RANDOM, 0, false, true>(Sp, H, K, auto_reset, park, slack, clear_terr);
}
template <int N>     // ... the all-in agents (every hand a showdown: BASELINE configs[4]) with the table evaluator: no action ring in LDS, up to ten seats
__global__ void __launch_bounds__(PK_TABLE_BLOCK, PK_WAVES_PER_SIMD_N(N)) PK_ROLLOUT_ATTR k_rollout_allin_tab(const State *__restrict__ Sp, Hot H, int K, int auto_reset, int park, int slack, int clear_terr) {
    rollout_body<N, true, PK_POLICY_ALLIN, 0, false, true>(Sp, H, K, auto_reset, park, slack, clear_terr);
}
template <int N>
__global__ void __launch_bounds__(PK_TABLE_BLOCK, PK_WAVES_PER_SIMD_N(N)) PK_ROLLOUT_ATTR k_rollout_allin(const State *__restrict__ Sp, Hot H, int K, int auto_reset, int park, int slack, int clear_terr) {
    rollout_body<N, true, PK_POLICY_ALLIN>(Sp, H, K, auto_reset, park, slack, clear_terr);
}
// Game.step (game.py:621-700) with the caller's actions, and the single-step form of the random-agent rollout (pk_rollout
// with fused == 0: the state round-trips HBM every step): the same body with one betting pass per look at the parked lanes.
template <int N>
__global__ void __launch_bounds__(PK_TABLE_BLOCK, PK_WAVES_PER_SIMD_N(N)) k_step(StepKernArgs) {
    const StepKernArgs *ka = (const StepKernArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    const Hot H = ka->H;
    rollout_body<N, false, PK_POLICY_EXTERNAL, 1>(ka->Sp, H, 0, ka->auto_reset, ka->park, PK_WAVE, 1, ka->actions, ka);
}
template <int N>
__global__ void __launch_bounds__(PK_TABLE_BLOCK, PK_WAVES_PER_SIMD_N(N)) k_step_async(StepKernArgs) {   // pk_step_async_d: bounded launches
    const StepKernArgs *ka = (const StepKernArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    Hot H = ka->H;
    // The first 32 bytes of the argument block (Sp, H.fresh, the blinds) arrive as ONE s_load_dwordx8; the blinds are live across the whole loop, and
    // with six and fourteen seats the allocator spilled the eight-register tuple as a unit -- then rematerialised the load instead, leaving a dead
    // 36-byte stack object behind that made every dispatch set up scratch memory (.amdhsa_private_segment_fixed_size 36 with not one
    // scratch instruction in the kernel).  Passing the two doubles through an empty asm makes them registers of their own; the tuple dies at once.
    asm volatile("" : "+s"(H.big_blind), "+s"(H.small_blind));
    rollout_body<N, false, PK_POLICY_EXTERNAL, 1, true>(ka->Sp, H, 0, ka->auto_reset, ka->park, PK_WAVE, 1, ka->actions, ka);
}
template <int N>
__global__ void __launch_bounds__(PK_TABLE_BLOCK, PK_WAVES_PER_SIMD_N(N)) k_rollout_single(const State *__restrict__ Sp, Hot H, int K, int auto_reset, int park, int slack, int clear_terr) {
    rollout_body<N, false, PK_POLICY_RANDOM, 1>(Sp, H, K, auto_reset, park, slack, clear_terr);
}
template <int N>
__global__ void __launch_bounds__(PK_TABLE_BLOCK, PK_WAVES_PER_SIMD_N(N)) PK_ROLLOUT_ATTR k_rollout_call(const State *__restrict__ Sp, Hot H, int K, int auto_reset, int park, int slack, int clear_terr) {
    rollout_body<N, true, PK_POLICY_CALL>(Sp, H, K, auto_reset, park, slack, clear_terr);
}
// ... the same capped at 168 registers for three waves per SIMD.  Up to six seats k_rollout is below the cap anyway
// (three resident waves are what give 49 G at 1 M x 6); the cap pays at seven seats (175 -> 168 without a spill: 44.3 vs
// 40.3 G at 1 M x 7) and at eight (39.5 vs 36.8 G at 1 M x 8), and costs 10..35 % at nine and ten seats, where it spills
// (pk_create picks; profiles/r02_occ3_sweep.txt).  A cap of 128 (four waves) spills too much at every seat count.
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
struct EnvResetKernArgs { const State *Sp; Hot H; const uint8_t *mask; uint64_t seatpol; int park; };   // = the kernarg segment (see EnvArgs)
template <int N>
__global__ void __launch_bounds__(PK_TABLE_BLOCK) k_env_reset(EnvResetKernArgs) {
    const EnvResetKernArgs *ka = (const EnvResetKernArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    const auto &S = *as_global(ka->Sp);
    const Hot H = ka->H;
    __shared__ Lds<N> lds;
    const int t = blockIdx.x * H.tpb + threadIdx.x;
    const auto mask = as_global(ka->mask);
    const bool live = (int)threadIdx.x < H.tpb && t < S.T && (!mask || mask[t < S.T ? t : 0]);
    const uint32_t table_id = H.table_id_base + (uint32_t)t;
    Table<N> tb;
    if (live) tb.load(S, t); else tb.blank();
    ActionRng rng;
    double high_bet = 0.0;
    bool more = live, due_reset = live;
    uint32_t late = live ? 1u : 0u;                      // (`live` for the epilogue, through the loop in a VGPR: see env_step_body)
    asm volatile("" : "+v"(late));
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
            if (due_reset) { tb.reset_state(ka->H, 0); tb.deal(H, table_id); due_reset = false; }   // :23 / :27 (start credits: read from the argument block here)
            more = tb.active != 0;                                                 // :24
            if (more) {
                uint32_t vm = tb.valid_mask(high_bet);
                tb.begin_step(H, pick_action(H, rng, table_id, tb.step_serial, vm, seat_policy(ka->seatpol, tb.active)), high_bet);  // :25-26 (the
                                                                            // policy word and `park`: read from the argument block where used)
            }
        }
        tb.cursor();
        retire();
        const int parked = __popcll(__ballot(tb.parked()));
        const int runnable = __popcll(__ballot(more && tb.lstate == LS_DONE));
        if (parked == 0 && runnable == 0) break;
        if (parked >= ka->park || runnable == 0) {
            tb.template end_block<false>(H, t, table_id, lds, false);   // (the whole side-pot loop per call: a reset's tail is a few lanes)
            retire();
        }
    }
    asm volatile("" : "+v"(late));
    if (late) {
        tb.store(S, t);
        tb.store_show(S.show, S.T, t, lds);
        as_global(S.valid)[t] = (uint8_t)tb.valid_mask(high_bet);
        as_global(S.terr)[t] = (uint8_t)tb.terr;
    }
}

// What PokerGameEnv does between two Game.steps (game_env.py:24-27, :35-52) as a TABLE: index = phase | over << 3 |
// hand_over << 4 | err << 5 | (seat 0 is to act) << 6 | (seat 0 is BROKEN) << 7, entry = next phase (bits 0..2, before the
// "step has returned" rule), done / hand updates (bit 3 set + bit 4 value, bit 5 set + bit 6 value), reward = payoffs[0]
// (bit 7), PokerGameEnv.step has returned (bit 8).  The control flow itself, as the reference writes it, is env_transition()
// below -- evaluated at compile time for all 256 inputs; the kernels stage the table in LDS and look a returned lane's
// entry up (as ~45 lane-mask operations and a dozen selects per betting pass this logic was 13 % of the asynchronous
// kernel: profiles/r03_blockprof_env_async.txt).
enum : int { PH_SEAT0 = 0, PH_HAND = 1, PH_TURN = 2, PH_RESET = 3, PH_RESET_PLAY = 4, PH_END = 5 };
constexpr uint16_t env_transition(int idx) {
    const int phase = idx & 7;
    const bool over = (idx >> 3) & 1, hand_now = (idx >> 4) & 1, err = (idx >> 5) & 1, seat0 = (idx >> 6) & 1, broken0 = (idx >> 7) & 1;
    const bool rp = phase == PH_RESET_PLAY;                                    // a step of PokerGameEnv.reset()'s loop (:24-27)
    const bool rn = !rp;                                                       // a step of PokerGameEnv.step (:35-52)
    int ph = phase;
    if (rp) ph = err ? (int)PH_END : (over ? (int)PH_RESET : (seat0 ? (int)PH_END : (int)PH_RESET_PLAY));   // :27 / :24
    const bool s0 = rn && !err && phase == PH_SEAT0, sh = rn && !err && phase == PH_HAND, st = rn && !err && phase == PH_TURN;
    bool done_set = s0 || sh || st, done_val = over;                           // :35 / :44 / :52
    bool hand_set = s0 || sh, hand_val = hand_now;
    const bool bust = s0 && (over || broken0);                                 // :37-39
    if (bust) { done_set = true; done_val = true; hand_set = true; hand_val = true; }
    const bool hand = hand_val;                                                // (only read where hand_set holds)
    const bool to_hand = s0 && !bust && !hand && !seat0;                       // :41
    const bool leave = (s0 && !bust && !to_hand) || (sh && (hand || seat0));   // :41's loop is over (or never entered)
    const bool take_rew = bust || (leave && hand);                             // :39 / :47
    const bool to_turn = leave && !over && !seat0;                             // :49 (done == over wherever `leave` can hold)
    const bool fin = (rn && err) || bust || (leave && !to_turn) || (st && (over || seat0));
    if (to_hand) ph = PH_HAND;
    if (to_turn) ph = PH_TURN;
    const bool live_phase = phase == PH_SEAT0 || phase == PH_HAND || phase == PH_TURN || phase == PH_RESET_PLAY;
    if (!live_phase) return (uint16_t)phase;                                   // (no Game.step can return in PH_RESET / PH_END)
    return (uint16_t)(ph | (done_set << 3) | (done_val << 4) | (hand_set << 5) | (hand_val << 6) | (take_rew << 7) | (fin << 8));
}
struct alignas(16) EnvTransitions {   // (copied into LDS as 32-bit words)
    uint16_t e[256];
    constexpr EnvTransitions() : e{} { for (int i = 0; i < 256; ++i) e[i] = env_transition(i); }
};
__device__ __constant__ const EnvTransitions g_env_transitions{};

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
//
// MULTI (pk_env_step_multi_d): one agent PER SEAT (PokerGameEnv's `agents` list, game_env.py:13-18): seat p plays the
// in-kernel policy in nibble p of `seatpol`, or -- PK_POLICY_EXTERNAL -- is played by the CALLER: when such a seat is
// to act inside a PokerGameEnv.step / .reset, the table YIELDS (ready[t] = 2, who[t] = the seat, obs row = that seat's
// StateView), its env call stays in flight exactly like a step that ran out of passes, and the next launch takes
// actions[t] as that seat's action.  reset_req[t] != 0 starts PokerGameEnv.reset() on that table instead of a step.

// The arguments of the PokerGameEnv.step kernels as ONE block, read through the kernarg segment pointer WHERE A FIELD IS USED:
// formal kernel parameters are all s_load-ed in the entry block and then live across the whole step loop -- the seven output
// pointers, needed only after it, cost k_env_step<6> 14 SGPRs there (31 spilled SGPRs in round 3).  A load the kernel writes
// itself stays where it is written.
struct EnvArgs {
    const int32_t *actions;      // [T] seat 0's action per table (or the yielded seat's, MULTI); NULL: seat 0 plays `seat0_policy`
    const uint8_t *reset_req;    // [T] MULTI: != 0 starts PokerGameEnv.reset() on that table; may be NULL
    double *reward; uint8_t *done, *hand, *terr;   // [T] outputs of a returned PokerGameEnv.step (game_env.py:53)
    double *obs;                 // [T][PK_OBS_DIM] or NULL
    uint8_t *obs_packed;         // [T][PK_OBS_PACKED_BYTES] or NULL (pk_get_obs_packed's row, written from registers)
    uint8_t *ready, *who;        // [T] ASYNC / MULTI status bytes
    uint64_t seatpol;            // one policy nibble per seat
    int seat0_policy, auto_reset, park, max_passes, abandon, t0, tend;   // [t0, tend): the sub-range of the handle this launch serves
};
struct EnvKernArgs { const State *Sp; Hot H; EnvArgs A; };   // the kernels' single formal parameter = the layout of their kernarg segment
static_assert(std::is_trivially_copyable<EnvKernArgs>::value, "kernel argument block");

template <int N, bool ASYNC, bool MULTI>
__device__ __forceinline__ void env_step_body() {
    static_assert(ASYNC || !MULTI, "yielding to the caller needs the in-flight context of the asynchronous form");
    const EnvKernArgs *ka = (const EnvKernArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    const EnvArgs &A = ka->A;                                    // (fields are loaded at their uses)
    const auto &S = *as_global(ka->Sp);
    const Hot H = ka->H;                                         // the loop's scalars: loaded once, here
    const auto actions = as_global(A.actions);
    const int seat0_policy = A.seat0_policy, auto_reset = A.auto_reset, park = A.park, max_passes = A.max_passes;
    const uint64_t seatpol = A.seatpol;
    __shared__ Lds<N> lds;
    const int t = A.t0 + blockIdx.x * H.tpb + threadIdx.x;      // [t0, tend): the sub-range of the handle this launch serves
    const bool live = (int)threadIdx.x < H.tpb && t < S.T && t < A.tend;
    const uint32_t table_id = H.table_id_base + (uint32_t)t;
    Table<N> tb;
    PK_PROF(tb.prof.start();)
    if (live) tb.load(S, t); else tb.blank();
    uint64_t ctx = 0;                                          // != 0: a PokerGameEnv.step of this table is in flight
    if (ASYNC && live) ctx = as_global(S.env_ctx)[t];
    const auto reset_req = as_global(A.reset_req);
    const bool want_reset = MULTI && live && reset_req && reset_req[t];
    if (want_reset) { ctx = 0; tb.idle(); }                    // PokerGameEnv.reset(): whatever was in flight is dropped
    const bool carried = ctx != 0;
    bool yielded = MULTI && carried && ((ctx >> 6) & 1);       // waiting for the caller's action for seat tb.active
    ActionRing ring;
    __shared__ alignas(16) uint16_t trans[256];                // env_transition() of every input (see above); copied as 32-bit words
    for (int i = threadIdx.x & (PK_WAVE - 1); i < 128; i += PK_WAVE)
        reinterpret_cast<uint32_t *>(trans)[i] = reinterpret_cast<const uint32_t *>(g_env_transitions.e)[i];
    stage_nth(lds);                                            // (its barrier covers the table above)
    double high_bet;
    const uint32_t vm0 = tb.valid_mask(high_bet);
    // the action this call supplies: seat 0's for a new PokerGameEnv.step (checked here, game.py:648-651), or the
    // yielded seat's; ignored for a step that is simply still in flight
    const int action = (!live || !actions || want_reset || (carried && !yielded)) ? -1 : actions[t];
    const bool action_ok = action >= 0 && action < PK_NUM_MOVES && ((vm0 >> action) & 1);
    const bool skip = MULTI && live && !carried && !want_reset && action == PK_ACTION_SKIP;   // an idle table the caller leaves alone
    const bool ok = carried || want_reset || (live && !skip && (seat0_policy >= 0 || action_ok));
    int phase = ok ? (want_reset ? PH_RESET : PH_SEAT0) : PH_END;
    double rew = 0.0;                                                              // :34
    uint32_t done = 0, hand = 0;
    uint32_t terr_step = 0;
    int budget = PK_ENV_STEP_CAP, budget_reset = PK_ENV_STEP_CAP;
    if (ASYNC && carried) {
        phase = (int)(ctx >> 1) & 7; done = (ctx >> 4) & 1; hand = (ctx >> 5) & 1; terr_step = (uint32_t)(ctx >> 8) & 0xff;
        budget = (int)((ctx >> 16) & 0xffff) - 1; budget_reset = (int)((ctx >> 32) & 0xffff) - 1;
        rew = as_global(S.env_rew)[t];
        tb.hands_this_step = (int)as_global(S.mid)[t];
    }
    bool have_ext = MULTI && yielded && action_ok;             // the yielded seat's action has arrived and is valid
    const bool ext_invalid = MULTI && yielded && !action_ok;   // ... is invalid: the table is untouched and keeps waiting
    // What only the epilogue needs of the lane predicates above travels through the loop as bits of ONE VGPR: kept as lane masks
    // they are an SGPR pair each, live across the whole loop (a dozen of k_env_step<6>'s 31 spilled SGPRs in round 3).  The empty
    // asm keeps the optimiser from seeing through the packing.
    uint32_t late = (live ? 1u : 0u) | (ok ? 2u : 0u) | (carried ? 4u : 0u) | (skip ? 8u : 0u) | (ext_invalid ? 16u : 0u);
    asm volatile("" : "+v"(late));
    yielded = yielded && !have_ext;
    int passes = 0;
    const uint32_t caps = PK_TERR_HAND_CAP | PK_TERR_ENV_CAP;
    // A Game.step() of this lane has returned: the reference's control flow between two steps (game_env.py:24-27, :37-52),
    // as SELECTS over the predicates of all its branches -- written as nested ifs (round 2) every pass carried ~25
    // divergent branches (s_and_saveexec / s_cbranch / s_or each) against ~3 in k_rollout's pass -- behind ONE wave-uniform
    // test: in the second half of a bounded launch most passes retire nothing, and unguarded selects cost the asynchronous
    // kernel 10 % (synchronous +5 % either way; guarded: +7.6 % synchronous, +2.6 % asynchronous).
    auto retire = [&]() {
        const bool r = tb.stepped && tb.lstate == LS_DONE;
        if (!__any(r)) return;                                                     // (wave-uniform: most passes of a bounded launch's second half)
        tb.step_serial += (r && !(tb.terr & PK_TERR_NO_WINNER)) ? 1u : 0u;          // finish_step()
        tb.stepped = r ? 0u : tb.stepped;
        const bool rp = r && phase == PH_RESET_PLAY;                               // a step of PokerGameEnv.reset()'s loop (:24-27)
        const bool opp = r && (phase == PH_HAND || phase == PH_TURN);              // a step of PokerGameEnv.step played by an opponent
        budget_reset -= rp ? 1 : 0; budget -= opp ? 1 : 0;
        tb.terr |= ((rp && budget_reset < 0) || (opp && budget < 0)) ? (uint32_t)PK_TERR_ENV_CAP : 0u;
        const uint32_t idx = (uint32_t)phase | ((tb.flags & 3u) << 3) | (tb.terr != 0 ? 32u : 0u) | (tb.active == 0 ? 64u : 0u) | ((tb.st_broken & 1u) << 7);
        static_assert(PK_FLAG_GAME_OVER == 1 && PK_FLAG_HAND_OVER == 2, "env_transition()'s index takes the two flags as they are");
        const uint32_t w = r ? (uint32_t)trans[idx] : (uint32_t)phase;             // (a lane that did not return: no bit set, its own phase)
        done = (w & 8u) ? (w >> 4) & 1u : done;                                    // :35 / :44 / :52, :38
        hand = (w & 32u) ? (w >> 6) & 1u : hand;
        rew = (w & 128u) ? tb.payoffs[0] : rew;                                    // :39 / :47
        const bool fin = (w & 256u) != 0;
        // PokerGameEnv.step has returned: its outputs are final; maybe reset the episode
        terr_step = fin ? tb.terr : terr_step;
        phase = fin ? ((auto_reset && (done || (tb.terr & caps))) ? (int)PH_RESET : (int)PH_END) : (int)(w & 7u);
    };
#ifndef PK_ENV_PASSES
#define PK_ENV_PASSES 4   // betting passes between two looks at the parked lanes, as in k_rollout (asynchronous: 2 -> 2.97 G, 4 -> 3.20 G, 8 -> 2.71 G)
#endif
#ifndef PK_ENV_ROLLING
#define PK_ENV_ROLLING 16   // hands rolled inside one Game.step from which a bounded launch carries that step to its end
#endif
#ifndef PK_ENV_PASSES_SYNC
#define PK_ENV_PASSES_SYNC 8   // ... of the synchronous kernel, whose tail is a few lanes per wave (0.226 / 0.228 / 0.233 G at 2 / 4 / 8; the
#endif                         // action ring holds draws for eight passes at most)
    constexpr int EP = ASYNC ? PK_ENV_PASSES : PK_ENV_PASSES_SYNC;
    bool draws = seat0_policy == PK_POLICY_RANDOM;                                 // wave-uniform: some agent draws from the ring
    PK_FOR(p, N) if (p > 0) draws = draws || seat_policy(seatpol, p) == PK_POLICY_RANDOM; PK_END
    for (;;) {
        // ASYNC, pass budget used up: the launch ends.  Whatever has not returned stays IN FLIGHT as it is -- also a hand parked
        // at end_hand, which the next launch's first end_block serves together with that launch's own arrivals.  (Rounds 2 and
        // 3 brought those hands to their end first, "or a parked lane would wait for `park` neighbours launch after launch": it
        // cannot -- end_block also runs whenever no lane of the wave can begin a step -- and the drain's end_blocks, each for a
        // handful of lanes, cost a bounded launch a third of its time: 1.00 -> 1.46 G env.step/s with one batch, 3.19 -> 3.59 G
        // with four.)  One more round, in which no lane may begin a step: a seat walk, so that no lane is left between a deal and
        // its first seat (LS_SCAN).  (As code of its own in front of the loop's exit it cost the kernel its third wave: 87
        // spilled VGPRs.)
        // Exception: a Game.step that is rolling hand after hand (blinds far above the stacks: every new hand ends before
        // anybody can act -- 1 300 hands inside one step seen) is carried on to its end as before; at two or three end_blocks
        // per launch such a step would take hundreds of launches.  The threshold: short stacks posting all-in blinds roll a few
        // hands per step in ordinary late games too (3: 1.44 -> 1.20 G env.step/s with one batch; 16: no cost).
        const bool closing = ASYNC && max_passes > 0 && passes >= max_passes;    // no lane may begin a Game.step any more
        const bool last = closing && !__any(tb.stepped && tb.hands_this_step >= PK_ENV_ROLLING);   // (parked, or between a deal and its first seat)
        PK_PROF(tb.prof.lap(PF_CURSOR);)
        if (!closing && phase == PH_RESET && tb.lstate == LS_DONE) {               // game_env.py:23 / :27
            tb.reset_state(ka->H, 0); tb.deal(H, table_id);   // (the start credits: read from the argument block here, not kept in SGPRs)
            phase = tb.active != 0 ? PH_RESET_PLAY : PH_END;                       // :24
        }
        PK_PROF(tb.prof.lap(14);)                   // (diagnostic build: slot 14 = the episode-reset branch, 15 = the action draws,
        if (draws) __builtin_amdgcn_s_waitcnt(0);   // as in k_rollout: no stray full wait behind the passes' LDS reads
        if (draws && !closing) ring.ensure(lds, H, table_id, tb.step_serial, phase != PH_END && !yielded, EP);   // (a lane without a table: PH_END)
        PK_PROF(tb.prof.lap(15);)                   //  PF_CURSOR = load + census between the rounds)
        int made = 0;
#pragma unroll
        for (int pass = 0; pass < EP; ++pass) {
            const bool open = !(ASYNC && max_passes > 0 && passes + pass >= max_passes);
            const uint32_t word = draws ? ActionRing::peek(lds, tb.step_serial) : 0u;
            if (open && phase != PH_END && phase != PH_RESET && tb.lstate == LS_DONE && !yielded) {   // begin this lane's next Game.step()
                const uint32_t vm = tb.valid_mask(high_bet);
                // self.agents[active_player] (:25, :43, :51); seat 0's own step (:35) takes the caller's action when supplied
                const int pol = phase == PH_SEAT0 ? seat0_policy : seat_policy(seatpol, tb.active);
                const bool supplied = (phase == PH_SEAT0 && pol < 0) || (MULTI && phase != PH_SEAT0 && pol == PK_POLICY_EXTERNAL);
                const bool begin = !supplied || phase == PH_SEAT0 || have_ext;     // an external seat without an action: yield
                // every agent's move computed, then selected: as a chain of conditionals around the LDS lookup of the random
                // agent's move the compiler made this four nested branches per pass
                const int a_rand = draws ? action_from_draw_lds(lds, ActionRing::half_of(word, tb.step_serial), vm) : 0;
                const int a_call = call_action(vm);
                int a = pol == PK_POLICY_ALLIN ? (int)MV_ALL_IN : (pol == PK_POLICY_CALL ? a_call : a_rand);
                a = supplied ? action : a;
                if (MULTI && supplied && phase != PH_SEAT0) { yielded = !have_ext; have_ext = false; }
                if (begin) tb.begin_step(H, a, high_bet);                          // :35 / :43-44 / :51-52 / :25-26
            }
            tb.cursor();          // (k_rollout's scan_first / cursor_tail split measured 1.7 % slower here)
            retire();
            ++made;
            // The tail of an env step is a handful of lanes (a busted seat 0 waits for the end of its game): once no lane can
            // begin another Game.step in this round of passes -- all parked at end_hand, returned, or waiting for a reset /
            // the caller -- the remaining passes would run empty; go and serve the parked lanes at once.
            if (pass + 1 < EP &&
                !__any(phase != PH_END && phase != PH_RESET && tb.lstate == LS_DONE && !yielded)) break;
            // ... and so would they once the pass budget of a bounded launch is used up
            if (ASYNC && max_passes > 0 && passes + pass + 1 >= max_passes) break;
        }
        PK_PROF(tb.prof.lap(PF_ACTION); tb.prof.count(PF_N_CURSOR, (unsigned)made);)
        const int parked = __popcll(__ballot(tb.parked()));
        const int runnable = closing ? 0 : __popcll(__ballot(phase != PH_END && tb.lstate == LS_DONE && !yielded));
        if (last || (parked == 0 && runnable == 0)) break;
        passes += made;
        // (Serving a parked table at once while only 1 / 2 / 4 / 8 tables of the wave are still inside their env step, instead of waiting
        //  for the others to park too: 0.2367 / 0.2375 / 0.2374 / 0.236 G against 0.2383 G -- the wait is free, the wave is busy with the
        //  others' passes anyway, and merged end_blocks are fewer end_blocks.  docs/history.md section 12.)
        if (parked >= park || runnable == 0) {
            tb.template end_block<false, !ASYNC>(H, t, table_id, lds, false);   // the side-pot loop runs to its end inside the call (the step's tail is a few lanes); lone-table paths: synchronous only
            retire();
        }
    }
    PK_PROF(tb.prof.lap(PF_OTHER); tb.prof.flush(S.prof);)
    asm volatile("" : "+v"(late));
    const bool live_l = late & 1u, ok_l = late & 2u, carried_l = late & 4u, skip_l = late & 8u, ext_invalid_l = late & 16u;
    if (!live_l) return;
    // ---- outputs: the pointers are loaded from the argument block only now
    const auto ready = as_global(ASYNC ? A.ready : nullptr), who_out = as_global(MULTI ? A.who : nullptr);
    const auto terr = as_global(A.terr);
    if (MULTI && skip_l) {                                     // nothing ran, nothing is written but the two status bytes
        if (ready) ready[t] = 3;
        if (who_out) who_out[t] = (uint8_t)tb.active;
        return;
    }
    if (ok_l) {
        tb.store(S, t);
        tb.store_show(S.show, S.T, t, lds);
    }
    const uint32_t vmask = tb.valid_mask(high_bet);
    const bool returned = !ASYNC || phase == PH_END;
    if (ASYNC) {
        // pk_env_end_multi_d: after a drain only yielded tables (between two Game.steps) are still in flight; their env
        // call is abandoned and the table is an ordinary idle table again
        const bool keep = !returned && !(MULTI && A.abandon);
        as_global(S.env_ctx)[t] = keep ? (1ull | ((uint64_t)phase << 1) | ((uint64_t)done << 4) | ((uint64_t)hand << 5) | ((uint64_t)(yielded ? 1 : 0) << 6) |
                                          ((uint64_t)(terr_step & 0xff) << 8) | ((uint64_t)(budget + 1) << 16) | ((uint64_t)(budget_reset + 1) << 32))
                                       : 0ull;
        as_global(S.mid)[t] = keep ? (uint32_t)tb.hands_this_step : 0u;
        as_global(S.valid)[t] = (uint8_t)vmask;
        if (ready) ready[t] = returned ? 1 : ((MULTI && yielded) ? 2 : 0);
        if (MULTI && who_out) who_out[t] = (uint8_t)tb.active;
        if (!returned) {
            as_global(S.env_rew)[t] = rew;
            if (!(MULTI && yielded) || !terr) return;
            terr[t] = ext_invalid_l ? (uint8_t)PK_TERR_INVALID_ACTION : (uint8_t)0;   // the yielded seat's action was refused / is awaited
        }
    }
    if (returned) {
        const uint32_t te = ok_l ? (terr_step | tb.terr) : (uint32_t)PK_TERR_INVALID_ACTION;
        as_global(A.reward)[t] = ok_l ? rew : 0.0; as_global(A.done)[t] = ok_l && done; as_global(A.hand)[t] = ok_l && hand;  // :53
        if (ok_l) as_global(S.valid)[t] = (uint8_t)vmask;
        // (an idle table the abandoning launch of pk_env_end_multi_d merely passes over keeps its error byte: it made no call)
        if (!(MULTI && A.abandon && !carried_l)) as_global(S.terr)[t] = (uint8_t)te;
        terr[t] = (uint8_t)te;
    }
    const auto obs = as_global(A.obs);
    if (obs) write_obs_row<N>(tb, vmask, obs + (size_t)t * PK_OBS_DIM(N));      // Game.StateView(player to act), from registers
    const auto obs_packed = as_global(A.obs_packed);
    if (obs_packed) write_obs_packed_row<N>(tb, vmask, (PK_GLOBAL uint64_t *)(obs_packed + (size_t)t * PK_OBS_PACKED_BYTES(N)));
}
template <int N>
__global__ void __launch_bounds__(PK_TABLE_BLOCK) k_env_step(EnvKernArgs) { env_step_body<N, false, false>(); }
template <int N>
__global__ void __launch_bounds__(PK_TABLE_BLOCK, (N <= 6 ? 3 : PK_WAVES_PER_SIMD_N(N))) k_env_step_async(EnvKernArgs) { env_step_body<N, true, false>(); }
template <int N>
__global__ void __launch_bounds__(PK_TABLE_BLOCK, PK_WAVES_PER_SIMD_N(N)) k_env_step_multi(EnvKernArgs) { env_step_body<N, true, true>(); }

#ifndef PK_TABLES_ONLY
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
// The per-seat money of a table is LOADED FIRST, all of it (3 x 16 predicated loads in flight; runtime N: the export kernels are not templated),
// then stored: with one load -> store pair per seat in a loop (rounds 1-5) the compiler could not move a load above the previous seat's store
// (`out` may alias the state for all it knows) and the kernel ran at the latency of 3N dependent round trips -- 1.1 TB/s at 1 M tables.
struct SeatMoney { double credits[PK_MAX_PLAYERS], bets[PK_MAX_PLAYERS], pending[PK_MAX_PLAYERS]; };
__device__ __forceinline__ void load_seat_money(const State &S, int t, int N, SeatMoney &m) {
    const size_t T = (size_t)S.T;
#pragma unroll
    for (int p = 0; p < PK_MAX_PLAYERS; ++p) {
        const size_t i = (size_t)(p < N ? p : 0) * T + t;
        m.credits[p] = S.credits[i]; m.bets[p] = S.bets[i]; m.pending[p] = S.pending[i];
    }
}
__global__ void __launch_bounds__(64) k_obs(State S, int N, int player, double *__restrict__ out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S.T) return;
    const int T = S.T, D = PK_OBS_DIM(N);
    double *o = out + (size_t)t * D;
    SeatMoney m;
    load_seat_money(S, t, N, m);
    uint32_t cur = S.cursors[t];
    int active = cur & 0xf, turn = (cur >> 16) & 0xf;
    const int who = player < 0 ? active : player;
    const uint32_t vm = player < 0 ? S.valid[t] : valid_bits_of(S, t, N, who);
    auto card = [&](int c) { return (double)((S.cards[(size_t)(c >> 2) * T + t] >> (8 * (c & 3))) & 0xff); };
    o[0] = who; o[1] = turn; o[2] = S.min_raise[t];
    for (int a = 0; a < PK_NUM_MOVES; ++a) o[3 + a] = (vm >> a) & 1;
    o[10] = card(5 + 2 * who); o[11] = card(6 + 2 * who);                          // game.py:385-389
    for (int c = 0; c < 5; ++c) o[12 + c] = (turn != 0 && c < turn + 2) ? card(c) : -1.0;  // game.py:278
#pragma unroll
    for (int p = 0; p < PK_MAX_PLAYERS; ++p)
        if (p < N) { o[17 + p] = m.credits[p]; o[17 + N + p] = m.bets[p]; o[17 + 2 * N + p] = m.pending[p]; }
}
// The same row as k_obs, compact: 16 header bytes (seat, turn, valid-mask bits, 2 hole cards, 5 community cards with 0xFF for a
// card not yet visible, 6 zero bytes) + (3N+1) f64 (minimum_raise_value, credits, bets, pending_bets): pokerl_hip.h.
__global__ void __launch_bounds__(64) k_obs_packed(State S, int N, int player, uint8_t *__restrict__ out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S.T) return;
    const int T = S.T;
    uint64_t *o = reinterpret_cast<uint64_t *>(out + (size_t)t * PK_OBS_PACKED_BYTES(N));
    SeatMoney sm;
    load_seat_money(S, t, N, sm);
    uint32_t cur = S.cursors[t];
    int active = cur & 0xf, turn = (cur >> 16) & 0xf;
    const int who = player < 0 ? active : player;
    const uint32_t vm = player < 0 ? S.valid[t] : valid_bits_of(S, t, N, who);
    auto card = [&](int c) { return (uint32_t)((S.cards[(size_t)(c >> 2) * T + t] >> (8 * (c & 3))) & 0xff); };
    auto comm = [&](int c) { return (turn != 0 && c < turn + 2) ? card(c) : 0xffu; };
    o[0] = obs_packed_header0((uint32_t)who, (uint32_t)turn, vm, card(5 + 2 * who), card(6 + 2 * who), comm(0), comm(1), comm(2));
    o[1] = (uint64_t)comm(3) | ((uint64_t)comm(4) << 8);
    double *m = reinterpret_cast<double *>(o + 2);
    m[0] = S.min_raise[t];
#pragma unroll
    for (int p = 0; p < PK_MAX_PLAYERS; ++p)
        if (p < N) { m[1 + p] = sm.credits[p]; m[1 + N + p] = sm.bets[p]; m[1 + 2 * N + p] = sm.pending[p]; }
}
// Game.step's precondition (game.py:648-651) over a batch: the lowest table index whose action is not in its active player's mask.
__global__ void k_check_actions(const uint8_t *valid, const int32_t *actions, int T, int32_t *first_bad) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const int a = actions[t];
    const bool ok = a >= 0 && a < PK_NUM_MOVES && ((valid[t] >> a) & 1);
    if (!ok) atomicMin(first_bad, t);
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
        } else {   // numpy's pairwise_sum: eight partial sums over whole blocks of eight, the tree, then the tail
            double r8[8];
            for (int j = 0; j < 8; ++j) r8[j] = a[(size_t)j * T + t];
            int p = 8;
            for (; p + 8 <= N; p += 8)
                for (int j = 0; j < 8; ++j) r8[j] = r8[j] + a[(size_t)(p + j) * T + t];
            r = ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
            for (; p < N; ++p) r = r + a[(size_t)p * T + t];
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
    uint32_t v = eval_hand_any(c, n, nk);     // 3..7 distinct cards: the bitmask fast path; a repeated card, 0..2 cards: the literal scan
    rank[i] = (uint8_t)(v >> 20);
    kick[i] = v & 0xFFFFF;
    if (nkick) nkick[i] = (uint8_t)nk;
}
// The same op on the TABLE path (eval_tab_bits: ~two thirds of the instructions of the register evaluator, checks included):
// EVAL_TAB_BLOCK-thread workgroups with the 32 KB rank-mask table of the streaming evaluator in LDS, grid-stride, one hand per lane per
// iteration, nothing shared between lanes after the table copy (no barrier in the loop; eight waves per SIMD hide the lookups).  A hand's
// seven card bytes start at ANY byte offset: ONE unaligned 8-byte load per hand (gfx950 runs in unaligned-access mode; the eighth byte
// belongs to the next hand and is ignored -- the LAST hand of the buffer is read byte by byte instead, so nothing past cards[7m) is
// touched), the next iteration's load in flight while this one is evaluated.  3..7 distinct real cards: the table; 0..2 cards: the
// reference's first lines as selects (eval_small); a repeated card or a byte that is no card: the literal scan, executed by a wave only if
// one of its lanes needs it.  HAS_N == false: ncards == NULL, every hand holds seven cards.
// (Tried: two hands per lane per iteration as in the streaming kernel, 16-byte loads and paired stores -- no faster on seven-card hands,
// the kernel is bound by VALU issue, not by latency or memory instructions, and slower on mixed batches, where one short hand sends its
// partner down the slow branch too: profiles/r05_eval_hands_bench.txt.)
#define EVAL_TAB_BLOCK 512
template <bool HAS_N>
__global__ void __launch_bounds__(EVAL_TAB_BLOCK, 8) k_eval_hands_tab(const uint8_t *__restrict__ cards, const uint8_t *__restrict__ ncards, size_t m,
                                                                      uint8_t *__restrict__ rank, uint32_t *__restrict__ kick, uint8_t *__restrict__ nkick,
                                                                      const uint32_t *__restrict__ tab) {
    __shared__ uint32_t T[EVAL7_TAB_WORDS];
    for (int i = threadIdx.x; i < EVAL7_TAB_WORDS / 4; i += EVAL_TAB_BLOCK) reinterpret_cast<uint4 *>(T)[i] = reinterpret_cast<const uint4 *>(tab)[i];
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * EVAL_TAB_BLOCK;
    auto fetch = [&](size_t i, uint64_t &w, int &n) {
        if (i + 1 < m) __builtin_memcpy(&w, cards + 7 * i, 8);            // global_load_dwordx2 at a byte address
        else { w = 0; for (int j = 0; j < 7; ++j) w |= (uint64_t)cards[7 * i + j] << (8 * j); }
        n = HAS_N ? ncards[i] : 7;
    };
    size_t i = (size_t)blockIdx.x * EVAL_TAB_BLOCK + threadIdx.x;
    uint64_t w = 0; int n = 0;
    if (i < m) fetch(i, w, n);
    for (; i < m; i += stride) {
        uint64_t wn = 0; int nn = 0;
        if (i + stride < m) fetch(i + stride, wn, nn);
        n = n > 7 ? 7 : n;                                                  // (u8: never negative)
        int nk = 0;
        uint32_t v;
        uint64_t bits;
        if (tab_bits_of<!HAS_N>(w, n, bits)) v = eval_tab_bits<!HAS_N>(bits, T, nk);
        else if (HAS_N && n < 3) v = eval_small(w, n, nk);
        else {
            const uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32);
            const uint32_t c[7] = {lo & 0xff, (lo >> 8) & 0xff, (lo >> 16) & 0xff, lo >> 24, hi & 0xff, (hi >> 8) & 0xff, (hi >> 16) & 0xff};
            v = eval_hand(c, n, nk);
        }
        rank[i] = (uint8_t)(v >> 20);
        kick[i] = v & 0xFFFFF;
        if (nkick) nkick[i] = (uint8_t)nk;
        w = wn; n = nn;
    }
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
        return DISTINCT ? eval7_distinct(c) : eval_hand_any(c, 7, nk);
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
// The table of eval7_tab (pk_device.hpp), built once per device into global memory; every workgroup of the streaming
// kernel below copies it into its LDS.
__global__ void k_make_eval7_tab(uint32_t *tab) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (uint32_t)EVAL7_TAB_WORDS) tab[i] = eval7_tab_entry(i);
}
// Streaming evaluator for 7 DISTINCT cards, table-driven (eval7_tab): 512-thread workgroups, four per CU (4 x 32 KB of LDS),
// eight waves per SIMD under the 64-register cap; same two-hands-per-lane 16-byte loads / 8-byte stores as above.
template <bool VEC, int VARIANT>
__global__ void __launch_bounds__(512, 8) k_eval7_tab_stream(const uint64_t *__restrict__ hands, size_t m, uint32_t *__restrict__ out, const uint32_t *__restrict__ tab) {
    __shared__ uint32_t T[EVAL7_TAB_WORDS];
    for (int i = threadIdx.x; i < EVAL7_TAB_WORDS / 4; i += 512) reinterpret_cast<uint4 *>(T)[i] = reinterpret_cast<const uint4 *>(tab)[i];
    __syncthreads();
    const size_t pairs = VEC ? m / 2 : 0, stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    constexpr bool PREFETCH = (VARIANT & 1) == 0, TWO_PHASE = (VARIANT & 2) == 0;
    uint4 w = i < pairs ? reinterpret_cast<const uint4 *>(hands)[i] : uint4{0, 0, 0, 0};
    for (; i < pairs; i += stride) {
        const size_t nx = i + stride;                       // the next iteration's hands are in flight while these are evaluated
        uint4 wn = uint4{0, 0, 0, 0};
        if (PREFETCH) { if (nx < pairs) wn = reinterpret_cast<const uint4 *>(hands)[nx]; }
        uint2 r;
        if (TWO_PHASE) {
            const Eval7Front f0 = eval7_tab_front(w.x, w.y, T), f1 = eval7_tab_front(w.z, w.w, T);   // ten lookups issued ...
            r.x = eval7_tab_back(f0, T); r.y = eval7_tab_back(f1, T);                                // ... before the first is used
        } else { r.x = eval7_tab(w.x, w.y, T); r.y = eval7_tab(w.z, w.w, T); }
        reinterpret_cast<uint2 *>(out)[i] = r;
        if (PREFETCH) w = wn;
        else if (nx < pairs) w = reinterpret_cast<const uint4 *>(hands)[nx];
    }
    if constexpr (VEC) {
        if ((m & 1) && blockIdx.x == 0 && threadIdx.x == 0) out[m - 1] = eval7_tab((uint32_t)hands[m - 1], (uint32_t)(hands[m - 1] >> 32), T);
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) out[i] = eval7_tab((uint32_t)hands[i], (uint32_t)(hands[i] >> 32), T);
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
__global__ void k_eval7_prefix(int a, int b, int fast, uint32_t count, uint32_t *out, const uint32_t *tab) {
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
    // fast 1: the in-game evaluator; 0: the general (multiset) evaluator; 2: the table-driven evaluator of the streaming
    // kernel (cards rotated by the hand index so that every byte position of the packed word is exercised)
    if (fast == 2) {
        uint32_t r[7];
        for (int j = 0; j < 7; ++j) r[j] = h[(j + i) % 7];
        out[i] = eval7_tab(r[0] | (r[1] << 8) | (r[2] << 16) | (r[3] << 24), r[4] | (r[5] << 8) | (r[6] << 16) | 0xAB000000u, tab);
    } else if (fast == 4) {                                  // the table path of pk_eval_hands(_d), cards rotated likewise
        uint32_t r[7];
        for (int j = 0; j < 7; ++j) r[j] = h[(j + i) % 7];
        out[i] = eval_tab_n(r, 7, tab, nk);
    } else if (fast == 3) out[i] = eval_hand_any(h, 7, nk);   // pk_eval_hands' register dispatch (its fast path: eval_distinct_n)
    else out[i] = fast ? eval7_distinct(h) : eval_hand(h, 7, nk);
}

#endif  // PK_TABLES_ONLY

// ================================================================================================ instantiation lists
// The table kernels are compiled once per seat count in translation units of their own (pk_tables.hip with -DPK_SEATS=N:
// explicit instantiations), in parallel; pk_api.hip declares them `extern template` and launches them.
#define PK_ROLLOUT_SIG (const State *__restrict__, Hot, int, int, int, int, int)
#define PK_TABLE_KERNELS(X, N)                                               \
    X(N, k_reset, (State, Hot, const uint8_t *, int, int))                      \
    X(N, k_make_fresh, (Hot, Fresh *))                                       \
    X(N, k_pick, (State, Hot, int, int32_t *))                               \
    X(N, k_rollout, PK_ROLLOUT_SIG)                                          \
    X(N, k_rollout_allin, PK_ROLLOUT_SIG)                                    \
    X(N, k_rollout_call, PK_ROLLOUT_SIG)                                     \
    X(N, k_rollout_single, PK_ROLLOUT_SIG)                                   \
    X(N, k_step, (StepKernArgs))                                             \
    X(N, k_step_async, (StepKernArgs))                                       \
    X(N, k_env_reset, (EnvResetKernArgs))                                    \
    X(N, k_env_step, (EnvKernArgs))                                          \
    X(N, k_env_step_async, (EnvKernArgs))                                    \
    X(N, k_env_step_multi, (EnvKernArgs))
// ... and the ones that exist up to ten seats only (the 168-register variants: beyond ten seats they would spill)
#define PK_TABLE_KERNELS_LE10(X, N)                                          \
    X(N, k_rollout_occ3, PK_ROLLOUT_SIG)                                     \
    X(N, k_rollout_occ3_allin, PK_ROLLOUT_SIG)                               \
    X(N, k_rollout_allin_tab, PK_ROLLOUT_SIG)
// ... and up to six (the table evaluator's LDS layout: LdsTab)
#define PK_TABLE_KERNELS_LE6(X, N)                                           \
    X(N, k_rollout_tab, PK_ROLLOUT_SIG)
#define PK_INSTANTIATE_KERNEL(N, name, sig) template __global__ void name<N> sig;
#define PK_DECLARE_KERNEL(N, name, sig) extern template __global__ void name<N> sig;
