// Diagnostic (GPU box): the side-pot pass of end_hand (pokerl/game.py:498-525) in TWO formulations over the same synthetic
// showdowns -- (A) one table per lane, as Table::end_block runs it today (the ~27 of 64 lanes that hold a showdown active,
// N-unrolled f64 select chains), and (B) SEAT-PARALLEL on compacted showdown tables: the showdown lanes put their table's per-seat
// values into LDS, then every pass of the wave serves EIGHT tables with eight lanes each (lane = seat), cross-seat reductions
// through ballots / shuffles, np.sum in numpy's seat order -- the "one structural candidate not yet tried" of round 3's review.
// Both produce payoffs / working bets / rankings / todo / npw / pot_over of one call (one general pass + the closing pass, the
// ONE_PASS form of end_block); the host compares them bit for bit and the kernels report wave cycles per call (s_memtime).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -fno-honor-nans -mllvm -amdgpu-sched-strategy=max-ilp \
//              -I. -o /tmp/sidepot_variants tools/microbench/sidepot_variants.hip
// run:   /tmp/sidepot_variants [showdown lanes per wave = 27]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pokerl_amd/csrc/pk_device.hpp"

using namespace pk;
constexpr int N = 6;

struct In {   // one table (lane): a showdown that arrives at the side-pot loop
    double bets[N];
    uint32_t hv[N];
    uint32_t todo;   // showdown seats
    int npw;         // potential winners
    int sd;          // this lane holds a showdown
};
struct Out {
    double payoffs[N], wb[N];
    uint32_t hv[N];
    uint32_t todo;
    int npw, over;
};

// ---------------------------------------------------------------------------------------------- (A) one table per lane
__device__ __forceinline__ void pass_table_parallel(const double (&bets)[N], double (&payoffs)[N], double (&pot_wb)[N], uint32_t (&pot_hv)[N],
                                                    uint32_t &pot_todo, int &pot_npw, bool &pot_over) {
    bool left = false;
    PK_FOR(p, N) left = left || !(pot_wb[p] <= 0.0); PK_END
    if (pot_todo != 0 && left && pot_npw != 1) {
        int player = 0; double best = 0.0, max_bet = 0.0; bool have = false;
        PK_FOR(p, N)
            bool cand = (pot_todo >> p) & 1;
            bool better = cand && (!have || bets[p] < best);
            player = better ? p : player; best = better ? bets[p] : best; max_bet = better ? pot_wb[p] : max_bet;
            have = have || cand;
        PK_END
        pot_todo &= ~(1u << player);
        double mb[N];
        PK_FOR(p, N) double x = pot_wb[p]; x = (x < 0.0) ? 0.0 : x; x = (x > max_bet) ? max_bet : x; mb[p] = x; PK_END
        int nw;
        uint32_t win = compare_rankings<N>(pot_hv, nw);
        double s = np_sum<N>(mb);
        double share = (nw == 2) ? s * 0.5 : s;
        if (nw > 2) share = s / (double)nw;
        left = false;
        PK_FOR(p, N)
            payoffs[p] = ((win >> p) & 1) ? payoffs[p] + share : payoffs[p];
            pot_hv[p] = (p == player) ? NONE_V : pot_hv[p];
            pot_wb[p] = pot_wb[p] - mb[p];
            left = left || !(pot_wb[p] <= 0.0);
        PK_END
        pot_npw -= 1;
    }
    {
        const bool stop = pot_todo == 0 || !left;
        const int player = (!stop && pot_npw == 1) ? __ffs(pot_todo) - 1 : -1;
        const double s = np_sum<N>(pot_wb);
        PK_FOR(p, N) payoffs[p] = (p == player) ? payoffs[p] + s : payoffs[p]; PK_END
        pot_over = stop || pot_npw == 1;
    }
}

__global__ void __launch_bounds__(64) k_table_parallel(const In *in, Out *out, int reps, unsigned long long *cycles) {
    const int lane = threadIdx.x, t = blockIdx.x * 64 + lane;
    const In I = in[t];
    Out O{};
    unsigned long long c0 = 0, acc = 0;
    for (int r = 0; r < reps; ++r) {
        double bets[N], payoffs[N], wb[N];
        uint32_t hv[N];
        PK_FOR(p, N) bets[p] = I.bets[p]; payoffs[p] = 0.0; wb[p] = I.bets[p]; hv[p] = I.hv[p]; PK_END
        uint32_t todo = I.todo; int npw = I.npw; bool over = false;
        asm volatile("" : "+v"(todo));                    // every repetition really recomputes
        c0 = __builtin_readcyclecounter();
        if (I.sd) pass_table_parallel(bets, payoffs, wb, hv, todo, npw, over);
        acc += __builtin_readcyclecounter() - c0;
        PK_FOR(p, N) O.payoffs[p] = payoffs[p]; O.wb[p] = wb[p]; O.hv[p] = hv[p]; PK_END
        O.todo = todo; O.npw = npw; O.over = over;
    }
    out[t] = O;
    if (lane == 0) cycles[blockIdx.x] = acc;
}

// ---------------------------------------------------------------------------------------------- (B) seat-parallel, 8 lanes per table
struct SeatLds {
    double wb[64][8], bets[64][8], pay[64][8];   // [compact table slot][seat]
    uint32_t hv[64][8];
    uint32_t todo[64];
    int npw[64], over[64];
};
// byte of the 64-bit ballot that belongs to this lane's group of eight
__device__ __forceinline__ uint32_t group_byte(unsigned long long bal, int lane) { return (uint32_t)(bal >> (lane & 56)) & 0xffu; }

__global__ void __launch_bounds__(64) k_seat_parallel(const In *in, Out *out, int reps, unsigned long long *cycles) {
    __shared__ SeatLds L;
    const int lane = threadIdx.x, t = blockIdx.x * 64 + lane;
    const In I = in[t];
    Out O{};
    unsigned long long c0 = 0, acc = 0;
    const int s = lane & 7, gbase = lane & 56;
    for (int r = 0; r < reps; ++r) {
        uint32_t todo_in = I.todo;
        asm volatile("" : "+v"(todo_in));
        c0 = __builtin_readcyclecounter();
        // ---- compaction: the showdown lanes take consecutive slots and lay their table out seat by seat
        const unsigned long long bal = __ballot(I.sd != 0);
        const int total = __popcll(bal);
        const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (I.sd) {
            PK_FOR(p, N) L.wb[slot][p] = I.bets[p]; L.bets[slot][p] = I.bets[p]; L.pay[slot][p] = 0.0; L.hv[slot][p] = I.hv[p]; PK_END
            L.wb[slot][6] = 0.0; L.wb[slot][7] = 0.0; L.hv[slot][6] = NONE_V; L.hv[slot][7] = NONE_V;
            L.todo[slot] = todo_in; L.npw[slot] = I.npw;
        }
        __syncthreads();
        for (int base = 0; base < total; base += 8) {            // eight tables per pass, lane = (table, seat)
            const int q = base + (lane >> 3);
            const bool tv = q < total;                            // this group holds a table
            const bool sv = tv && s < N;
            const int qq = tv ? q : 0;
            double wb = L.wb[qq][s];
            const double bet = L.bets[qq][s];
            double pay = L.pay[qq][s];
            uint32_t hv = L.hv[qq][s];
            uint32_t todo = L.todo[qq];
            int npw = L.npw[qq];
            if (!sv) { wb = 0.0; hv = NONE_V; }
            bool left = group_byte(__ballot(!(wb <= 0.0)), lane) != 0;
            const bool go = tv && todo != 0 && left && npw != 1;
            // next seat in ascending ORIGINAL-bet order, lowest seat among equals (game.py:495-496, stable argsort)
            const bool cand = sv && ((todo >> s) & 1);
            double key = cand ? bet : __builtin_huge_val();
            double gmin = key;
            gmin = fmin(gmin, __shfl_xor(gmin, 1, 64)); gmin = fmin(gmin, __shfl_xor(gmin, 2, 64)); gmin = fmin(gmin, __shfl_xor(gmin, 4, 64));
            const uint32_t eqb = group_byte(__ballot(cand && key == gmin), lane);
            const int player = __ffs(eqb) - 1;
            const double max_bet = __shfl(wb, gbase + (player < 0 ? 0 : player), 64);                        // :508
            double mb = wb; mb = (mb < 0.0) ? 0.0 : mb; mb = (mb > max_bet) ? max_bet : mb;                  // :509
            // judger.compare_rankings incl. its line 148 (pk_device.hpp compare_rankings), across the group's lanes
            uint32_t rank = hv >> 20, kick = hv & 0xFFFFF;
            uint32_t best = rank;
            best = min(best, (uint32_t)__shfl_xor((int)best, 1, 64)); best = min(best, (uint32_t)__shfl_xor((int)best, 2, 64));
            best = min(best, (uint32_t)__shfl_xor((int)best, 4, 64));
            best = min(best, (uint32_t)HR_NONE);
            const bool br = rank == best;
            const uint32_t brb = group_byte(__ballot(br), lane);
            const int first = __ffs(brb) - 1;
            const uint32_t k0 = (uint32_t)__shfl((int)kick, gbase + (first < 0 ? 0 : first), 64);
            const uint32_t E = group_byte(__ballot(br && kick == k0), lane), G = group_byte(__ballot(br && kick > k0), lane);
            const uint32_t g = 31 - __clz((int)G);
            const uint32_t win = (G ? ((1u << g) | (E & ~((2u << g) - 1))) : E) & ((1u << N) - 1);
            const int nw = __popc(win);
            // np.sum(max_bets) in numpy's order: seats 0..5 left to right
            double sum = __shfl(mb, gbase + 0, 64);
#pragma unroll
            for (int p = 1; p < N; ++p) sum = sum + __shfl(mb, gbase + p, 64);
            double share = (nw == 2) ? sum * 0.5 : sum;
            if (nw > 2) share = sum / (double)nw;
            if (go) {
                pay = ((win >> s) & 1) ? pay + share : pay;
                hv = (s == player) ? NONE_V : hv;
                wb = wb - mb;
                todo &= ~(1u << player);
                npw -= 1;
            }
            if (go) left = false;
            const bool left2 = group_byte(__ballot(go && sv && !(wb <= 0.0)), lane) != 0;
            left = go ? left2 : left;
            // closing pass (:498-505)
            const bool stop = todo == 0 || !left;
            const int cplayer = (!stop && npw == 1) ? __ffs(todo) - 1 : -1;
            double sum2 = __shfl(wb, gbase + 0, 64);
#pragma unroll
            for (int p = 1; p < N; ++p) sum2 = sum2 + __shfl(wb, gbase + p, 64);
            pay = (s == cplayer) ? pay + sum2 : pay;
            const int over = stop || npw == 1;
            if (sv) { L.wb[q][s] = wb; L.pay[q][s] = pay; L.hv[q][s] = hv; }
            if (tv && s == 0) { L.todo[q] = todo; L.npw[q] = npw; L.over[q] = over; }
        }
        __syncthreads();
        if (I.sd) {
            PK_FOR(p, N) O.payoffs[p] = L.pay[slot][p]; O.wb[p] = L.wb[slot][p]; O.hv[p] = L.hv[slot][p]; PK_END
            O.todo = L.todo[slot]; O.npw = L.npw[slot]; O.over = L.over[slot];
        } else {
            PK_FOR(p, N) O.payoffs[p] = 0.0; O.wb[p] = I.bets[p]; O.hv[p] = I.hv[p]; PK_END
            O.todo = todo_in; O.npw = I.npw; O.over = 0;
        }
        __syncthreads();
        acc += __builtin_readcyclecounter() - c0;
    }
    out[t] = O;
    if (lane == 0) cycles[blockIdx.x] = acc;
}

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static unsigned rnd() { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(rng_state >> 33); }

int main(int argc, char **argv) {
    const int sd_lanes = argc > 1 ? atoi(argv[1]) : 27, waves = 1024, reps = 2000;
    const int T = waves * 64;
    std::vector<In> in(T);
    const double chips[8] = {2, 4, 10, 10, 20, 37.5, 50, 100};
    for (int t = 0; t < T; ++t) {
        In &I = in[t];
        memset(&I, 0, sizeof(I));
        I.sd = (int)(rnd() % 64) < sd_lanes;
        int shown = 0;
        for (int p = 0; p < N; ++p) {
            const unsigned k = rnd() % 10;
            const bool folded = k < 3, in_show = !folded;
            I.bets[p] = chips[rnd() % 8] * (folded ? 0.25 : 1.0);
            I.hv[p] = in_show ? ((1 + rnd() % 9) << 20) | (rnd() % 4 ? (rnd() & 0xFFFFF) : 0x12345u) : NONE_V;
            if (in_show) { I.todo |= 1u << p; ++shown; }
        }
        if (shown < 2) { I.todo |= 3; I.hv[0] = (9u << 20) | 5; I.hv[1] = (9u << 20) | 5; shown = __builtin_popcount(I.todo); }
        I.npw = shown;
        if (!I.sd) { }
    }
    In *d_in; Out *d_a, *d_b; unsigned long long *d_c;
    hipMalloc(&d_in, T * sizeof(In)); hipMalloc(&d_a, T * sizeof(Out)); hipMalloc(&d_b, T * sizeof(Out)); hipMalloc(&d_c, waves * 8);
    hipMemcpy(d_in, in.data(), T * sizeof(In), hipMemcpyHostToDevice);
    std::vector<Out> a(T), b(T);
    std::vector<unsigned long long> c(waves);
    double cyc[2];
    for (int v = 0; v < 2; ++v) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int w = 0; w < 2; ++w) {   // second run timed
            hipEventRecord(e0);
            if (v == 0) hipLaunchKernelGGL(k_table_parallel, dim3(waves), dim3(64), 0, 0, d_in, d_a, reps, d_c);
            else hipLaunchKernelGGL(k_seat_parallel, dim3(waves), dim3(64), 0, 0, d_in, d_b, reps, d_c);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(c.data(), d_c, waves * 8, hipMemcpyDeviceToHost);
        double sum = 0; for (auto x : c) sum += (double)x;
        cyc[v] = sum / waves / reps;
        printf("%-44s %8.1f wave-cycles per call (s_memtime, 100 MHz ticks x clock ratio not applied: compare the two), kernel %.3f ms for %d calls per wave\n",
               v == 0 ? "(A) one table per lane (end_block today):" : "(B) seat-parallel, 8 tables per pass via LDS:", cyc[v], ms, reps);
    }
    hipMemcpy(a.data(), d_a, T * sizeof(Out), hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), d_b, T * sizeof(Out), hipMemcpyDeviceToHost);
    long bad = 0, shown = 0;
    for (int t = 0; t < T; ++t) {
        shown += in[t].sd;
        if (memcmp(&a[t], &b[t], sizeof(Out)) != 0) {
            if (bad < 3) printf("MISMATCH table %d (sd %d todo %x npw %d): A pay0 %g todo %x npw %d over %d | B pay0 %g todo %x npw %d over %d\n", t, in[t].sd, in[t].todo, in[t].npw,
                                a[t].payoffs[0], a[t].todo, a[t].npw, a[t].over, b[t].payoffs[0], b[t].todo, b[t].npw, b[t].over);
            ++bad;
        }
    }
    printf("showdown lanes per wave %.1f; outputs of the two formulations: %s (%ld of %d tables differ)\n", (double)shown / waves, bad ? "DIFFER" : "bit-identical", bad, T);
    printf("seat-parallel / table-parallel = %.2fx the wave cycles\n", cyc[1] / cyc[0]);
    return bad != 0;
}
