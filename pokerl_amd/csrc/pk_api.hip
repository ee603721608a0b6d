// pk_api.hip -- host side of the C ABI (include/pokerl_hip.h) of libpokerl_hip.so; kernels: pk_kernels.hpp.  gfx950 only; plain HIP runtime,
// no torch types.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC (see pokerl_amd/build.py).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "pk_device.hpp"
#include "pk_kernels.hpp"

using namespace pk;

// ================================================================================================ host side
static thread_local std::string g_err;

struct pk_handle {
    int device = 0, T = 0, N = 0, block = 64, dealer = 0;
    int tpb = 64;   // tables per wavefront (Hot::tpb)
    int env_tpb = 64;  // ... of the PokerGameEnv kernels (hot_env): see pk_create; knob PK_ENV_TPB
    Hot hot_env{};
    bool occ3 = false;  // k_rollout_occ3 (registers capped for 3 waves per SIMD) vs k_rollout: see pk_create; knob PK_OCC3
    int tab_min_steps = 16;  // k_rollout_tab for launches of at least this many steps (0: never; 20-step launches: no gain, no loss); knob PK_ROLLOUT_TAB
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
    // Host-side coalescing of asynchronous pk_rollout calls (counters == NULL, auto_reset): while the two most recent
    // launches are still running, a call only ADDS its steps to `acc`; they are launched as ONE kernel once a launch slot
    // frees up, at `coalesce` accumulated steps, or by the flush every observer issues.  A stream of 20-step calls thus
    // runs as launches of up to `coalesce` steps: the fixed cost of a launch (load + store of every table, ramp and
    // tail: ~8 us of a 60 us 20-step launch) is paid once per launch, not per call.  0: every call launches.
    int coalesce = 1024;
    int acc = 0;
    hipEvent_t ev_ring[2] = {nullptr, nullptr};
    bool ev_used[2] = {false, false};
    int ev_idx = 0;
    // launches of the fused rollout kernel since pk_get_launch_stats(reset): count, steps summed, min / max steps per launch
    uint64_t st_launches = 0, st_steps = 0, st_min = 0, st_max = 0;
    int env_seat0 = 0, env_opp = 0, env_auto = 0;   // agents / auto_reset of the PokerGameEnv.steps in flight (env_pending)
    // Sub-batches of pk_env_step_async_d inside ONE handle (pk_set_env_batches): contiguous table ranges, each on its own
    // internal stream, launched round robin -- a call launches one range and delivers the range launched longest ago, so that
    // the launches of the ranges overlap like those of separate handles do (a launch ends with a tail; launches on one stream
    // are serialised).
    int env_batches = 1, env_range = 0;             // number of ranges, tables per range (a multiple of 64)
    hipStream_t env_streams[PK_MAX_ENV_BATCHES] = {};
    hipEvent_t env_done[PK_MAX_ENV_BATCHES] = {};
    bool env_launched[PK_MAX_ENV_BATCHES] = {};
    hipEvent_t env_in = nullptr;
    int env_next = 0, env_last_begin = 0, env_last_end = 0, env_last_fresh = 0;
    bool env_multi = false;                         // ... they belong to pk_env_step_multi_d, with these per-seat agents:
    uint64_t env_seats = 0;
    // PokerGameEnv.steps left in flight by pk_env_step_async_d (State::env_ctx): every other entry point that touches
    // table state refuses to run until a draining call (max_passes <= 0) has completed them.
    bool env_pending = false;
    bool step_pending = false;   // Game.steps left in flight by pk_step_async_d (their machine state is in the tables' cursor bits): as env_pending
    int step_auto = 0;
    bool host_step = false;   // between pk_env_step_begin and pk_env_step_end: the caller's host buffers are the targets of queued copies
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
    uint8_t *d_obs_packed = nullptr;     // [T][PK_OBS_PACKED_BYTES(N)]: device staging of pk_env_step_begin (allocated on first use)
    uint8_t *env_obs_packed = nullptr;   // pk_set_env_obs_packed: device buffer [T][PK_OBS_PACKED_BYTES(N)] or NULL
    double *step_obs = nullptr;          // pk_set_step_obs: device buffers the Game.step kernels write the row of the player to act into, or NULL
    uint8_t *step_obs_packed = nullptr;
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

// One instantiation of every table kernel per seat count, PK_MIN_PLAYERS .. PK_MAX_PLAYERS, each seat count compiled in a
// translation unit of its own (pk_tables.hip).  A development build may hold ONE seat count (-DPK_ONLY_SEATS=N: seconds
// instead of minutes); pk_create refuses the others there.
#ifdef PK_ONLY_SEATS
#define PK_SEAT_ENABLED(N) ((N) == (PK_ONLY_SEATS))
#else
#define PK_SEAT_ENABLED(N) 1
#endif
#define PK_FOR_SEATS_LE10(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10)
#define PK_FOR_SEATS_GT10(X) X(11) X(12) X(13) X(14) X(15) X(16)
static_assert(PK_MIN_PLAYERS == 2 && PK_MAX_PLAYERS == 16, "the lists above name the seat counts");
#define PK_DECLARE_ALL(N) PK_TABLE_KERNELS(PK_DECLARE_KERNEL, N)
#define PK_DECLARE_ALL_LE10(N) PK_TABLE_KERNELS_LE10(PK_DECLARE_KERNEL, N)
#define PK_DECLARE_ALL_LE6(N) PK_TABLE_KERNELS_LE6(PK_DECLARE_KERNEL, N)
#define PK_FOR_SEATS_LE6(X) X(2) X(3) X(4) X(5) X(6)
PK_FOR_SEATS_LE10(PK_DECLARE_ALL)
PK_FOR_SEATS_GT10(PK_DECLARE_ALL)
PK_FOR_SEATS_LE10(PK_DECLARE_ALL_LE10)
PK_FOR_SEATS_LE6(PK_DECLARE_ALL_LE6)
static inline bool seat_count_built(int n) {
#define PK_SEAT_CASE(N) if (n == N) return PK_SEAT_ENABLED(N);
    PK_FOR_SEATS_LE10(PK_SEAT_CASE)
    PK_FOR_SEATS_GT10(PK_SEAT_CASE)
#undef PK_SEAT_CASE
    return false;
}
#define DISPATCH_N_ON(h, strm, KERNEL, grid, ...) \
    do { \
        dim3 g_((grid)), b_((h)->block); \
        hipStream_t strm_ = (strm); \
        switch ((h)->N) { \
            case 2: if constexpr (PK_SEAT_ENABLED(2)) hipLaunchKernelGGL(KERNEL<2>, g_, b_, 0, strm_, __VA_ARGS__); break; \
            case 3: if constexpr (PK_SEAT_ENABLED(3)) hipLaunchKernelGGL(KERNEL<3>, g_, b_, 0, strm_, __VA_ARGS__); break; \
            case 4: if constexpr (PK_SEAT_ENABLED(4)) hipLaunchKernelGGL(KERNEL<4>, g_, b_, 0, strm_, __VA_ARGS__); break; \
            case 5: if constexpr (PK_SEAT_ENABLED(5)) hipLaunchKernelGGL(KERNEL<5>, g_, b_, 0, strm_, __VA_ARGS__); break; \
            case 6: if constexpr (PK_SEAT_ENABLED(6)) hipLaunchKernelGGL(KERNEL<6>, g_, b_, 0, strm_, __VA_ARGS__); break; \
            case 7: if constexpr (PK_SEAT_ENABLED(7)) hipLaunchKernelGGL(KERNEL<7>, g_, b_, 0, strm_, __VA_ARGS__); break; \
            case 8: if constexpr (PK_SEAT_ENABLED(8)) hipLaunchKernelGGL(KERNEL<8>, g_, b_, 0, strm_, __VA_ARGS__); break; \
            case 9: if constexpr (PK_SEAT_ENABLED(9)) hipLaunchKernelGGL(KERNEL<9>, g_, b_, 0, strm_, __VA_ARGS__); break; \
            case 10: if constexpr (PK_SEAT_ENABLED(10)) hipLaunchKernelGGL(KERNEL<10>, g_, b_, 0, strm_, __VA_ARGS__); break; \
            case 11: if constexpr (PK_SEAT_ENABLED(11)) hipLaunchKernelGGL(KERNEL<11>, g_, b_, 0, strm_, __VA_ARGS__); break; \
            case 12: if constexpr (PK_SEAT_ENABLED(12)) hipLaunchKernelGGL(KERNEL<12>, g_, b_, 0, strm_, __VA_ARGS__); break; \
            case 13: if constexpr (PK_SEAT_ENABLED(13)) hipLaunchKernelGGL(KERNEL<13>, g_, b_, 0, strm_, __VA_ARGS__); break; \
            case 14: if constexpr (PK_SEAT_ENABLED(14)) hipLaunchKernelGGL(KERNEL<14>, g_, b_, 0, strm_, __VA_ARGS__); break; \
            case 15: if constexpr (PK_SEAT_ENABLED(15)) hipLaunchKernelGGL(KERNEL<15>, g_, b_, 0, strm_, __VA_ARGS__); break; \
            case 16: if constexpr (PK_SEAT_ENABLED(16)) hipLaunchKernelGGL(KERNEL<16>, g_, b_, 0, strm_, __VA_ARGS__); break; \
        } \
    } while (0)
#define DISPATCH_N(h, KERNEL, grid, ...) DISPATCH_N_ON(h, (h)->stream, KERNEL, grid, __VA_ARGS__)
// ... for the kernels that exist for up to ten seats only (the 168-register variants: beyond ten seats they would spill)
#define DISPATCH_N_LE10(h, KERNEL, grid, ...) \
    do { \
        dim3 g_((grid)), b_((h)->block); \
        switch ((h)->N) { \
            case 2: if constexpr (PK_SEAT_ENABLED(2)) hipLaunchKernelGGL(KERNEL<2>, g_, b_, 0, (h)->stream, __VA_ARGS__); break; \
            case 3: if constexpr (PK_SEAT_ENABLED(3)) hipLaunchKernelGGL(KERNEL<3>, g_, b_, 0, (h)->stream, __VA_ARGS__); break; \
            case 4: if constexpr (PK_SEAT_ENABLED(4)) hipLaunchKernelGGL(KERNEL<4>, g_, b_, 0, (h)->stream, __VA_ARGS__); break; \
            case 5: if constexpr (PK_SEAT_ENABLED(5)) hipLaunchKernelGGL(KERNEL<5>, g_, b_, 0, (h)->stream, __VA_ARGS__); break; \
            case 6: if constexpr (PK_SEAT_ENABLED(6)) hipLaunchKernelGGL(KERNEL<6>, g_, b_, 0, (h)->stream, __VA_ARGS__); break; \
            case 7: if constexpr (PK_SEAT_ENABLED(7)) hipLaunchKernelGGL(KERNEL<7>, g_, b_, 0, (h)->stream, __VA_ARGS__); break; \
            case 8: if constexpr (PK_SEAT_ENABLED(8)) hipLaunchKernelGGL(KERNEL<8>, g_, b_, 0, (h)->stream, __VA_ARGS__); break; \
            case 9: if constexpr (PK_SEAT_ENABLED(9)) hipLaunchKernelGGL(KERNEL<9>, g_, b_, 0, (h)->stream, __VA_ARGS__); break; \
            case 10: if constexpr (PK_SEAT_ENABLED(10)) hipLaunchKernelGGL(KERNEL<10>, g_, b_, 0, (h)->stream, __VA_ARGS__); break; \
        } \
    } while (0)

// ... and up to six (k_rollout_tab)
#define DISPATCH_N_LE6(h, KERNEL, grid, ...) \
    do { \
        dim3 g_((grid)), b_((h)->block); \
        switch ((h)->N) { \
            case 2: if constexpr (PK_SEAT_ENABLED(2)) hipLaunchKernelGGL(KERNEL<2>, g_, b_, 0, (h)->stream, __VA_ARGS__); break; \
            case 3: if constexpr (PK_SEAT_ENABLED(3)) hipLaunchKernelGGL(KERNEL<3>, g_, b_, 0, (h)->stream, __VA_ARGS__); break; \
            case 4: if constexpr (PK_SEAT_ENABLED(4)) hipLaunchKernelGGL(KERNEL<4>, g_, b_, 0, (h)->stream, __VA_ARGS__); break; \
            case 5: if constexpr (PK_SEAT_ENABLED(5)) hipLaunchKernelGGL(KERNEL<5>, g_, b_, 0, (h)->stream, __VA_ARGS__); break; \
            case 6: if constexpr (PK_SEAT_ENABLED(6)) hipLaunchKernelGGL(KERNEL<6>, g_, b_, 0, (h)->stream, __VA_ARGS__); break; \
        } \
    } while (0)

#define PK_STR_(x) #x
#define PK_STR(x) PK_STR_(x)
static inline bool bad_policy(int policy) { return policy < 0 || policy >= PK_NUM_POLICIES; }
// every seat plays `policy`: the per-seat word of the entry points that take ONE opponent policy
static inline uint64_t uniform_seats(int policy) { return 0x1111111111111111ull * (uint64_t)(policy & 15); }
static inline int table_grid(const pk_handle *h) { return (h->T + h->tpb - 1) / h->tpb; }
// parking threshold for waves that hold h->tpb tables instead of 64
static inline int scaled_park(const pk_handle *h, int dflt = 32, int tpb = 0) {
    int p = ((h->park > 0 ? h->park : dflt) * (tpb > 0 ? tpb : h->tpb) + 63) / 64;
    return p < 1 ? 1 : p;
}
static inline int env_grid(const pk_handle *h) { return (h->T + h->env_tpb - 1) / h->env_tpb; }
// (16: asynchronous env.step +1.3 % over 32 with the round's final kernels, synchronous unchanged; 12 the same, 24 half of it)
static inline int env_park(const pk_handle *h) { return scaled_park(h, 16, h->env_tpb); }
static inline int flat_grid(size_t n) { return (int)((n + 255) / 256); }
// the observation getters: one table per lane, 64-thread workgroups -- 65 536 tables are 1 024 waves, one per SIMD (256-thread blocks put them on a quarter of the CUs' schedulers' slots)
#define OBS_BLOCK 64
static inline int obs_grid(int T) { return (T + OBS_BLOCK - 1) / OBS_BLOCK; }

// One fused rollout launch: every table owes k_steps more steps; the launch ends once fewer than `endk` lanes of a
// wave have work left (endk == 1: runs to completion).
static int launch_rollout(pk_handle *h, int k_steps, int policy, int auto_reset, int endk) {
    const int slack = endk <= 1 ? PK_WAVE : ((PK_WAVE - endk) * h->tpb) / PK_WAVE;   // lanes allowed to idle before a launch ends
#define ROLLOUT_ARGS (const State *)h->d_S, h->hot, k_steps, auto_reset, scaled_park(h, policy == PK_POLICY_RANDOM ? 28 : 32), slack, h->pending ? 0 : 1
    if (policy == PK_POLICY_RANDOM && k_steps == 1 && endk <= 1 && !h->occ3) DISPATCH_N(h, k_rollout_single, table_grid(h), ROLLOUT_ARGS);
    else if (policy == PK_POLICY_CALL) DISPATCH_N(h, k_rollout_call, table_grid(h), ROLLOUT_ARGS);
    else if (!h->occ3) {
        // k_rollout_tab: the showdown hands ranked by the table-driven evaluator -- where its 40 KB of LDS per wave cost no occupancy (at most one
        // wave per SIMD: 1 024 workgroups) and the launch is long enough to pay for staging the table (~1.5 us); knob PK_ROLLOUT_TAB (min steps, 0 = off)
        const bool tab_ok = h->S.evtab && h->tab_min_steps > 0 && k_steps >= h->tab_min_steps && table_grid(h) <= 1024;
        if (policy == PK_POLICY_RANDOM && h->N <= 6 && tab_ok) DISPATCH_N_LE6(h, k_rollout_tab, table_grid(h), ROLLOUT_ARGS);
        else if (policy == PK_POLICY_ALLIN && h->N <= 10 && tab_ok) DISPATCH_N_LE10(h, k_rollout_allin_tab, table_grid(h), ROLLOUT_ARGS);   // (no action ring in LDS: up to ten seats)
        else if (policy == PK_POLICY_RANDOM) DISPATCH_N(h, k_rollout, table_grid(h), ROLLOUT_ARGS);
        else DISPATCH_N(h, k_rollout_allin, table_grid(h), ROLLOUT_ARGS);
    } else {
        if (policy == PK_POLICY_RANDOM) DISPATCH_N_LE10(h, k_rollout_occ3, table_grid(h), ROLLOUT_ARGS);
        else DISPATCH_N_LE10(h, k_rollout_occ3_allin, table_grid(h), ROLLOUT_ARGS);
    }
#undef ROLLOUT_ARGS
    HIPCHK(h, hipGetLastError());
    h->pending = slack < PK_WAVE;
    h->pend_policy = policy; h->pend_auto = auto_reset;
    h->st_launches += 1; h->st_steps += (uint64_t)k_steps;
    if (k_steps > 0) {
        h->st_min = (h->st_min == 0 || (uint64_t)k_steps < h->st_min) ? (uint64_t)k_steps : h->st_min;
        h->st_max = (uint64_t)k_steps > h->st_max ? (uint64_t)k_steps : h->st_max;
    }
    return PK_OK;
}
// Completes whatever deferred rollout launches left undone.  Called by every entry point that reads or mutates tables.
static int flush_rollout(pk_handle *h) {
    if (!h->pending && h->acc == 0) return PK_OK;
    const int k = h->acc;
    h->acc = 0;
    return launch_rollout(h, k, h->pend_policy, h->pend_auto, 1);
}
// One asynchronous, deferring launch of the accumulated steps; remembers it in the two-slot event ring.
static int launch_coalesced(pk_handle *h, int policy, int auto_reset) {
    const int k = h->acc;
    h->acc = 0;
    int rc = launch_rollout(h, k, policy, auto_reset, h->endk);
    if (rc) return rc;
    HIPCHK(h, hipEventRecord(h->ev_ring[h->ev_idx], h->stream));
    h->ev_used[h->ev_idx] = true;
    h->ev_idx ^= 1;
    return PK_OK;
}
static inline bool in_flight(const pk_handle *h) { return h->env_pending || h->host_step || h->step_pending; }
static int flush(pk_handle *h) {
    if (h->step_pending)
        return h->fail(PK_E_BUSY, "Game.steps are in flight (pk_step_async_d): drain them with max_hands = 0 first");
    if (h->host_step)
        return h->fail(PK_E_BUSY, "a pk_env_step_begin is waiting for its pk_env_step_end: the copies into the caller's buffers are still queued");
    if (h->env_pending)
        return h->fail(PK_E_BUSY, "PokerGameEnv steps are in flight (pk_env_step_async_d / pk_env_step_multi_d): drain them with max_passes = 0 "
                                  "(pk_env_end_multi_d where seats are played by the caller) first");
    return flush_rollout(h);
}
#define FLUSH(h)                   \
    do {                           \
        int rc_ = flush(h);        \
        if (rc_) return rc_;       \
    } while (0)
// The device-resident READERS (pk_pick_actions_d, pk_get_obs_d, pk_get_obs_packed_d, pk_get_valid_actions_d, pk_get_f64_d) also work while
// Game.steps are in flight (pk_step_async_d): a caller needs them to act on the tables that are ready.  What they return for a table whose
// step is still in flight is that table in the middle of its step -- ready_d says which rows to ignore.
static int flush_reader(pk_handle *h) {
    if (h->step_pending && !h->env_pending && !h->host_step) return PK_OK;
    return flush(h);
}
#define FLUSH_READER(h)                \
    do {                               \
        int rc_ = flush_reader(h);     \
        if (rc_) return rc_;           \
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

// Streams are RECYCLED through a per-device pool instead of being destroyed with their handle: a process that opens and closes
// handles (bench.py's legs, a test suite) keeps working on the same few streams -- and therefore on the same hardware queues --
// instead of walking through HIP's stream-to-queue assignment (see the sub-batch notes below and docs/history.md section 5).
static std::mutex g_stream_mu;
static std::vector<hipStream_t> g_stream_pool[PK_MAX_DEVICES][2];   // [device][0: normal priority, 1: the sub-batch streams]
// The internal streams of pk_set_env_batches are created with the HIGHEST stream priority: HIP keeps separate hardware queues per
// priority level, so they do not compete for the four normal-priority queues with the caller's stream, the legacy default stream
// (one hipMemcpy or one torch kernel brings it to life) and whatever else the process has created -- with normal-priority streams a
// process that had merely made one hipMemcpy before creating the handle ran 524 288 x 6 in three sub-batches at 2.57 G env.step/s
// against 3.66 G (tools/env_queue_scenarios.py).  env PK_ENV_STREAM_PRIO=0: off.
// How the pool notices a device reset WITHOUT touching a pooled (then dangling) stream handle: a small device allocation made when the first stream
// of a device is pooled.  hipMemGetAddressRange looks an ADDRESS up in the runtime's allocation map -- no dereference of a dead object -- and
// after a reset the canary is gone from it (or, should the application have re-allocated that very address in the meantime, would have to
// come back with the same odd size: not impossible, so pk_stream_pool_drain before hipDeviceReset remains the rule; this is the seat belt).
static void *g_pool_canary[PK_MAX_DEVICES];
static const size_t POOL_CANARY_BYTES = 4096 + 272;
static bool pool_alive(int device) {   // caller holds g_stream_mu and has the device current
    void *c = g_pool_canary[device];
    if (!c) return false;
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    const hipError_t e = hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)c);
    if (e != hipSuccess) (void)hipGetLastError();
    const bool ok = e == hipSuccess && (void *)base == c && size == POOL_CANARY_BYTES;
    if (!ok) g_pool_canary[device] = nullptr;          // (not freed: it no longer exists)
    return ok;
}
static hipError_t stream_acquire(int device, hipStream_t *out, bool sub_batch = false) {   // the device is current
    static const bool prio = !(getenv("PK_ENV_STREAM_PRIO") && atoi(getenv("PK_ENV_STREAM_PRIO")) == 0);
    const int cls = (sub_batch && prio) ? 1 : 0;
    if (device >= 0 && device < PK_MAX_DEVICES) {
        std::lock_guard<std::mutex> lock(g_stream_mu);
        auto &pool = g_stream_pool[device][cls];
        if (!pool.empty() && !pool_alive(device)) {
            // The application reset the device (hipDeviceReset destroys every stream) without pk_stream_pool_drain: the pooled handles are
            // dangling pointers -- ANY call on one, hipStreamQuery included, segfaults inside the runtime (measured, ROCm 7.2).  Forget them
            // (nothing is left to destroy) and create a fresh stream below.
            for (int c = 0; c < 2; ++c) g_stream_pool[device][c].clear();
        }
        if (!pool.empty()) { *out = pool.back(); pool.pop_back(); return hipSuccess; }
    }
    if (cls == 1) {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least)
            return hipStreamCreateWithPriority(out, hipStreamNonBlocking, greatest);
        (void)hipGetLastError();
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
static void stream_release(int device, hipStream_t s, bool sub_batch = false) {   // the device is current; s is idle or about to be
    if (!s) return;
    if (hipStreamSynchronize(s) != hipSuccess) {      // a stream that cannot even be waited for (stale after a device reset, or broken) is not recycled
        (void)hipGetLastError(); (void)hipStreamDestroy(s); (void)hipGetLastError();
        return;
    }
    static const bool prio = !(getenv("PK_ENV_STREAM_PRIO") && atoi(getenv("PK_ENV_STREAM_PRIO")) == 0);
    if (device >= 0 && device < PK_MAX_DEVICES) {
        std::lock_guard<std::mutex> lock(g_stream_mu);
        if (!g_pool_canary[device]) {                  // the pool's reset detector (pool_alive); without one the stream is not pooled
            for (int c = 0; c < 2; ++c) g_stream_pool[device][c].clear();    // (whatever was pooled under an earlier, vanished canary is stale)
            if (hipMalloc(&g_pool_canary[device], POOL_CANARY_BYTES) != hipSuccess) { (void)hipGetLastError(); g_pool_canary[device] = nullptr; (void)hipStreamDestroy(s); return; }
        }
        g_stream_pool[device][(sub_batch && prio) ? 1 : 0].push_back(s);
    } else (void)hipStreamDestroy(s);
}

// Destroys the pooled (idle) streams of a device; -1: of every device.  For a host application that resets the device (hipDeviceReset
// invalidates every stream, pooled ones included) or wants its hardware queues back.  Streams of live handles are not in the pool.
static int drain_stream_pool(int device) {
    std::lock_guard<std::mutex> lock(g_stream_mu);
    int n = 0;
    for (int d = 0; d < PK_MAX_DEVICES; ++d) {
        if (device >= 0 && d != device) continue;
        for (int cls = 0; cls < 2; ++cls) {
            auto &pool = g_stream_pool[d][cls];
            if (pool.empty()) continue;
            DeviceGuard guard(d);
            const bool alive = guard.ok && pool_alive(d);     // (after a device reset the handles are dangling: dropped, not destroyed)
            for (hipStream_t s : pool) { if (alive) (void)hipStreamDestroy(s); ++n; }
            (void)hipGetLastError();
            pool.clear();
        }
        if (g_pool_canary[d] && g_stream_pool[d][0].empty() && g_stream_pool[d][1].empty()) {
            DeviceGuard guard(d);
            if (guard.ok && pool_alive(d)) (void)hipFree(g_pool_canary[d]);
            (void)hipGetLastError();
            g_pool_canary[d] = nullptr;
        }
    }
    return n;
}

static int check_device_any() {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) { g_err = "no HIP device available; this library has no CPU fallback"; return PK_E_NO_DEVICE; }
    return PK_OK;
}

// The argument block of the PokerGameEnv.step kernels (pk_kernels.hpp EnvKernArgs): common fields; callers set the rest.
static EnvKernArgs env_args(const pk_handle *h, const int32_t *actions_d, int seat0_policy, uint64_t seatpol, int auto_reset,
                            double *reward_d, uint8_t *done_d, uint8_t *hand_d, uint8_t *terr_d, double *obs_d) {
    EnvKernArgs ka{};
    ka.Sp = (const State *)h->d_S; ka.H = h->hot_env;
    ka.A.actions = actions_d; ka.A.seat0_policy = seat0_policy; ka.A.seatpol = seatpol; ka.A.auto_reset = auto_reset;
    ka.A.reward = reward_d; ka.A.done = done_d; ka.A.hand = hand_d; ka.A.terr = terr_d; ka.A.obs = obs_d;
    ka.A.obs_packed = nullptr;             // (pk_set_env_obs_packed's buffer: set by the three entry points the header names, see with_packed)
    ka.A.park = env_park(h); ka.A.t0 = 0; ka.A.tend = h->T;
    return ka;
}

// pk_set_env_obs_packed: pk_env_step_fused_d / _async_d / _multi_d write the compact row of every table they deliver
static inline EnvKernArgs with_packed(const pk_handle *h, EnvKernArgs ka) { ka.A.obs_packed = h->env_obs_packed; return ka; }

// the table-driven evaluator's rank-mask table, one per device (defined with the judger entry points); built on `stream` the first time
static const uint32_t *eval7_table(int device, hipStream_t stream = nullptr);

extern "C" {

int pk_abi_version(void) { return PK_ABI_VERSION; }

#ifndef PK_SOURCE_HASH
#define PK_SOURCE_HASH "unknown"
#endif
const char *pk_build_info(void) { return "abi=" PK_STR(PK_ABI_VERSION) " src=" PK_SOURCE_HASH; }

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
        g_err = "pk_create: need num_tables >= 1 and " PK_STR(PK_MIN_PLAYERS) " <= num_players <= " PK_STR(PK_MAX_PLAYERS)
                " (the reference takes any num_players, game.py:246; a 17th seat does not fit the 16 nibbles of a policy word, and from 18 seats on "
                "numpy's argsort of the bets, game.py:495, is no longer a stable insertion sort, so the side-pot order among equal bets is not a "
                "rule the reference pins: DESIGN.md section 8, docs/history.md section 9)";
        return PK_E_INVALID_ARG;
    }
    if (!seat_count_built(num_players)) { g_err = "pk_create: this development build of the library holds one seat count only (PK_ONLY_SEATS)"; return PK_E_INVALID_ARG; }
    {   // Money must be finite: the library is built with -fno-honor-nans (v_max_f64 / v_min_f64 for np.max / np.minimum), and an
        // infinite stack turns into NaN at the first all-in (inf - inf), where numpy and the device would disagree silently.
        bool finite = std::isfinite(big_blind) && std::isfinite(small_blind);
        if (start_credits) { for (int i = 0; i < num_players; ++i) finite = finite && std::isfinite(start_credits[i]); }
        else finite = finite && std::isfinite(start_credit_scalar);
        if (!finite) { g_err = "pk_create: start_credits, big_blind and small_blind must be finite (no inf / NaN)"; return PK_E_INVALID_ARG; }
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
        // k_rollout_occ3 (registers capped at 168: three waves per SIMD) only where it was measured to pay: six seats
        // (k_rollout<6> needs 171 registers -- profiles/r04_resource_usage.txt --, the capped build 162 without a spill) as soon as a SIMD gets a
        // third wave, i.e. beyond 131 072 tables; seven seats from 262 144 tables (+6..10 %), eight from 524 288 (+5..7 %).
        // Up to five seats k_rollout fits 168 registers by itself; at nine and ten seats the cap spills to scratch and
        // loses 10..35 % at every batch size.
        h->occ3 = (num_players == 6 && num_tables > 131072) || (num_players == 7 && num_tables >= 262144) ||
                  (num_players == 8 && num_tables >= 524288);
        if (const char *pk = getenv("PK_OCC3")) h->occ3 = atoi(pk) != 0 && num_players <= 10;
    }
    {   // The env kernels are bound by the tail of the slowest table of a wave (a busted seat 0 waits for the end of the game),
        // not by issue slots: half-populated waves halve that tail and bring a second wave to each SIMD
        // (profiles/r03_env_tpb_sweep.txt)
        h->env_tpb = h->tpb;
        if (const char *pk = getenv("PK_ENV_TPB")) { int v = atoi(pk); if (v >= 1 && v <= 64 && (v & (v - 1)) == 0) h->env_tpb = v; }
    }
    if (const char *pk = getenv("PK_ROLLOUT_TAB")) { int v = atoi(pk); if (v >= 0) h->tab_min_steps = v; }
    if (const char *pk = getenv("PK_PARK")) { int v = atoi(pk); if (v >= 1 && v <= 64) h->park = v; }
    if (const char *pk = getenv("PK_ENDK")) { int v = atoi(pk); if (v >= 1 && v <= 64) h->endk = v; }
    if (const char *pk = getenv("PK_COALESCE")) { int v = atoi(pk); if (v >= 0 && v <= (1 << 20)) h->coalesce = v; }
    auto bail = [&](int code) { g_err = h->err; pk_destroy(h); return code; };
    DeviceGuard guard(device);
    if (!guard.ok) return bail(h->fail(PK_E_HIP, "hipSetDevice"));
    if (stream_acquire(device, &h->own_stream) != hipSuccess) {  // (stream_acquire has validated what it took from the pool, or created a fresh one)
        (void)hipGetLastError();
        return bail(h->fail(PK_E_HIP, "hipStreamCreate"));
    }
    h->stream = h->own_stream;
    if (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_ring[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_ring[1], hipEventDisableTiming) != hipSuccess) return bail(h->fail(PK_E_HIP, "hipEventCreate"));
    if (hipHostMalloc((void **)&h->h_pinned, (size_t)num_tables, hipHostMallocDefault) != hipSuccess) return bail(h->fail(PK_E_OOM, "hipHostMalloc"));

    const size_t T = (size_t)num_tables, N = (size_t)num_players;
    const size_t K = 5 + 2 * N, W = (K + 3) / 4;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t obs = (size_t)PK_OBS_DIM(N) * 8;
    h->export_bytes = al(T * (obs > N * 8 ? obs : N * 8));
    const size_t nwaves = (T + (size_t)h->tpb - 1) / (size_t)h->tpb;
    State &S = h->S;
    double *d_start = nullptr;
    Fresh *d_fresh = nullptr;
    // ONE description of the arena, run twice: with base == NULL it only measures (the result is the allocation size),
    // then it hands out the pointers -- an array added here cannot be forgotten in a separately kept size expression.
    auto layout = [&](char *base) -> size_t {
        size_t off = 0;
        auto take = [&](size_t bytes) { void *r = base ? (void *)(base + off) : nullptr; off += al(bytes); return r; };
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
        d_start = (double *)take(PK_MAX_PLAYERS * 8);
        d_fresh = (Fresh *)take(sizeof(Fresh));
        h->d_actions = (int32_t *)take(T * 4);
        h->d_flags = (uint8_t *)take(T); h->d_terr = (uint8_t *)take(T); h->d_mask = (uint8_t *)take(T);
        h->d_done = (uint8_t *)take(T); h->d_handf = (uint8_t *)take(T);
        h->d_reward = (double *)take(T * 8);
        h->d_export = take(h->export_bytes);
        return off;
    };
    const size_t total = layout(nullptr);
    e = hipMalloc(&h->arena, total);
    if (e != hipSuccess) return bail(h->fail(PK_E_OOM, "hipMalloc(table state)", e));
    if (hipMemsetAsync(h->arena, 0, total, h->stream) != hipSuccess) return bail(h->fail(PK_E_HIP, "hipMemset"));
    if (layout((char *)h->arena) != total) return bail(h->fail(PK_E_HIP, "arena layout changed between its two passes"));
    for (int i = 0; i < PK_MAX_PLAYERS; ++i)
        S.start_credits[i] = i < num_players ? (start_credits ? start_credits[i] : start_credit_scalar) : 0.0;
    S.big_blind = big_blind; S.small_blind = small_blind;
    S.key0 = (uint32_t)seed; S.key1 = (uint32_t)(seed >> 32);
    S.table_id_base = table_id_base; S.T = num_tables;
    // k_rollout_tab / k_rollout_allin_tab stage it in LDS (NULL -- out of memory -- : k_rollout(_allin) is used).  Built on the HANDLE's stream: the legacy
    // default stream, once brought to life, takes one of the four normal-priority hardware queues for the rest of the process -- four env handles then
    // share three (measured in round 6's first bench refresh: 3.6 -> 2.5 G env.step/s with four handles; docs/history.md section 5 has the mechanism)
    S.evtab = num_players <= 10 ? eval7_table(device, h->stream) : nullptr;

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
        h->hot_env = h->hot; h->hot_env.tpb = h->env_tpb;
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
    if (h->d_obs_packed) (void)hipFree(h->d_obs_packed);
    if (h->h_pinned) (void)hipHostFree(h->h_pinned);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    for (int i = 0; i < 2; ++i) if (h->ev_ring[i]) (void)hipEventDestroy(h->ev_ring[i]);
    for (int b = 0; b < PK_MAX_ENV_BATCHES; ++b) {
        stream_release(h->device, h->env_streams[b], true);
        if (h->env_done[b]) (void)hipEventDestroy(h->env_done[b]);
    }
    if (h->env_in) (void)hipEventDestroy(h->env_in);
    stream_release(h->device, h->own_stream);
    delete h;
    return PK_OK;
}

int pk_stream_pool_drain(int device) {
    if (device < -1 || device >= PK_MAX_DEVICES) { g_err = "pk_stream_pool_drain: device index out of range (-1 = every device)"; return PK_E_INVALID_ARG; }
    return drain_stream_pool(device);
}

int pk_num_tables(const pk_handle *h) { return h ? h->T : PK_E_INVALID_ARG; }
int pk_num_players(const pk_handle *h) { return h ? h->N : PK_E_INVALID_ARG; }

// ---- stream control: how a caller with its own stream (a learner on the same GPU) orders its work against ours
int pk_get_stream(pk_handle *h, void **stream_out) {
    if (!h || !stream_out) return PK_E_INVALID_ARG;
    *stream_out = (void *)h->stream;
    return PK_OK;
}
static int switch_stream(pk_handle *h, hipStream_t to) {
    if (!in_flight(h)) FLUSH(h);                 // host-side accumulated / deferred rollout steps belong to the old stream
    HIPCHK(h, hipStreamSynchronize(h->stream));  // nothing of ours may still be running on the stream we leave
    h->stream = to;
    h->ev_used[0] = h->ev_used[1] = false;
    return PK_OK;
}
int pk_set_stream(pk_handle *h, void *stream) {
    if (!h) return PK_E_INVALID_ARG;
    ON_DEVICE(h);
    if (stream) {   // a stream of another device would make every later launch fail (or, worse, run there)
        hipDevice_t dev = -1;
        if (hipStreamGetDevice((hipStream_t)stream, &dev) != hipSuccess) { (void)hipGetLastError(); return h->fail(PK_E_INVALID_ARG, "pk_set_stream: not a valid hipStream_t"); }
        if ((int)dev != h->device) return h->fail(PK_E_INVALID_ARG, "pk_set_stream: the stream belongs to another device than the handle");
    }
    return switch_stream(h, (hipStream_t)stream);   // NULL = the legacy default stream (what torch's default stream is)
}
int pk_use_own_stream(pk_handle *h) {
    if (!h) return PK_E_INVALID_ARG;
    ON_DEVICE(h);
    return switch_stream(h, h->own_stream);
}
int pk_wait_event(pk_handle *h, void *event) {
    if (!h || !event) return PK_E_INVALID_ARG;
    ON_DEVICE(h);
    // rollout steps the host still holds back (coalescing) were requested BEFORE this wait: launch them now, so that they do not
    // queue up behind the caller's event (and overlap with whatever the caller does until it records it)
    if (!in_flight(h) && h->acc > 0) { int rc = launch_coalesced(h, h->pend_policy, h->pend_auto); if (rc) return rc; }
    HIPCHK(h, hipStreamWaitEvent(h->stream, (hipEvent_t)event, 0));
    return PK_OK;
}
int pk_record_event(pk_handle *h, void *event) {
    if (!h || !event) return PK_E_INVALID_ARG;
    ON_DEVICE(h);
    if (!in_flight(h)) FLUSH(h);    // "everything requested so far" includes deferred rollout steps (env steps in flight stay so)
    // ... and the launches of pk_env_step_async_d's sub-batches on the handle's internal streams (pk_set_env_batches)
    for (int b = 0; b < h->env_batches && h->env_batches > 1; ++b)
        if (h->env_launched[b]) HIPCHK(h, hipStreamWaitEvent(h->stream, h->env_done[b], 0));
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
    DISPATCH_N(h, k_reset, table_grid(h), h->S, h->hot, dmask, 0xff, d);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PK_OK;
}

int pk_reset_d(pk_handle *h, const uint8_t *mask_d, int mask_bits, int dealer) {
    if (!h) return PK_E_INVALID_ARG;
    if (mask_d && !(mask_bits & 0xff)) return h->fail(PK_E_INVALID_ARG, "pk_reset_d: mask_bits selects no bit of a mask byte");
    ON_DEVICE(h);
    FLUSH(h);
    int d = ((dealer % h->N) + h->N) % h->N;
    DISPATCH_N(h, k_reset, table_grid(h), h->S, h->hot, mask_d, mask_bits & 0xff, d);
    HIPCHK(h, hipGetLastError());
    return PK_OK;
}

static int launch_step(pk_handle *h, const int32_t *actions_d, uint8_t *flags_d, uint8_t *terr_d, int auto_reset) {
    ON_DEVICE(h);
    FLUSH(h);
    const StepKernArgs ka{(const State *)h->d_S, h->hot, actions_d, flags_d, terr_d, scaled_park(h, 28), auto_reset ? 1 : 0, nullptr, 0, h->step_obs, h->step_obs_packed};
    DISPATCH_N(h, k_step, table_grid(h), ka);
    HIPCHK(h, hipGetLastError());
    return PK_OK;
}
int pk_step_d(pk_handle *h, const int32_t *actions_d, uint8_t *flags_d, uint8_t *terr_d) {
    if (!h || !actions_d || !flags_d) return h ? h->fail(PK_E_INVALID_ARG, "pk_step_d: NULL buffer") : PK_E_INVALID_ARG;
    return launch_step(h, actions_d, flags_d, terr_d, 0);
}
int pk_step_auto_d(pk_handle *h, const int32_t *actions_d, uint8_t *flags_d, uint8_t *terr_d) {
    if (!h || !actions_d || !flags_d) return h ? h->fail(PK_E_INVALID_ARG, "pk_step_auto_d: NULL buffer") : PK_E_INVALID_ARG;
    return launch_step(h, actions_d, flags_d, terr_d, 1);
}
int pk_step_async_d(pk_handle *h, const int32_t *actions_d, uint8_t *flags_d, uint8_t *terr_d, uint8_t *ready_d, int max_hands, int auto_reset) {
    if (!h || !flags_d || !ready_d) return h ? h->fail(PK_E_INVALID_ARG, "pk_step_async_d: NULL buffer") : PK_E_INVALID_ARG;
    if (!actions_d && max_hands > 0) return h->fail(PK_E_INVALID_ARG, "pk_step_async_d: actions_d may be NULL only for a drain (max_hands <= 0): finish what is in flight, step nothing");
    ON_DEVICE(h);
    if (h->step_pending) {
        if ((auto_reset ? 1 : 0) != h->step_auto)   // steps in flight keep the reset rule they were started with
            return h->fail(PK_E_INVALID_ARG, "pk_step_async_d: Game.steps are in flight with another auto_reset; drain them first (max_hands = 0)");
        if (h->env_pending || h->host_step) return flush(h);
    } else FLUSH(h);
    const StepKernArgs ka{(const State *)h->d_S, h->hot, actions_d, flags_d, terr_d, scaled_park(h, 28), auto_reset ? 1 : 0, ready_d, max_hands > 0 ? max_hands : 0,
                          h->step_obs, h->step_obs_packed};
    DISPATCH_N(h, k_step_async, table_grid(h), ka);
    HIPCHK(h, hipGetLastError());
    h->step_pending = max_hands > 0;
    h->step_auto = auto_reset ? 1 : 0;
    return PK_OK;
}

static int any_terr(const uint8_t *terr, int T) {
    for (int i = 0; i < T; ++i) if (terr[i]) return 1;
    return 0;
}

int pk_step(pk_handle *h, const int32_t *actions, uint8_t *flags, uint8_t *terr) {
    if (!h || !actions || !flags) return h ? h->fail(PK_E_INVALID_ARG, "pk_step: NULL buffer") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    if (in_flight(h)) return flush(h);     // PK_E_BUSY, before anything is queued
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

int pk_get_f64_d(pk_handle *h, int field, double *out_d) {
    if (!h || !out_d || field < 0 || field > 3) return h ? h->fail(PK_E_INVALID_ARG, "pk_get_f64_d: bad argument") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    FLUSH_READER(h);
    const double *src = field == PK_F_CREDITS ? h->S.credits : field == PK_F_BETS ? h->S.bets : field == PK_F_PENDING_BETS ? h->S.pending : h->S.payoffs;
    size_t n = (size_t)h->T * h->N;
    hipLaunchKernelGGL(k_export_f64, dim3(flat_grid(n)), dim3(256), 0, h->stream, src, h->T, h->N, out_d);
    HIPCHK(h, hipGetLastError());
    return PK_OK;
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
        hipLaunchKernelGGL(k_obs, dim3(obs_grid(h->T)), dim3(OBS_BLOCK), 0, h->stream, h->S, h->N, player, (double *)h->d_export);
    });
}

int pk_get_obs_d(pk_handle *h, int player, double *out_d) {
    if (!h || !out_d || bad_player(h, player)) return h ? h->fail(PK_E_INVALID_ARG, "pk_get_obs_d: bad argument") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    FLUSH_READER(h);
    hipLaunchKernelGGL(k_obs, dim3(obs_grid(h->T)), dim3(OBS_BLOCK), 0, h->stream, h->S, h->N, player, out_d);
    HIPCHK(h, hipGetLastError());
    return PK_OK;
}

int pk_get_obs_packed(pk_handle *h, int player, uint8_t *out) {
    if (!h || !out || bad_player(h, player)) return h ? h->fail(PK_E_INVALID_ARG, "pk_get_obs_packed: bad argument") : PK_E_INVALID_ARG;
    size_t bytes = (size_t)h->T * PK_OBS_PACKED_BYTES(h->N);
    return export_to_host(h, out, bytes, [&] {
        hipLaunchKernelGGL(k_obs_packed, dim3(obs_grid(h->T)), dim3(OBS_BLOCK), 0, h->stream, h->S, h->N, player, (uint8_t *)h->d_export);
    });
}

int pk_get_obs_packed_d(pk_handle *h, int player, uint8_t *out_d) {
    if (!h || !out_d || bad_player(h, player) || ((uintptr_t)out_d & 7)) return h ? h->fail(PK_E_INVALID_ARG, "pk_get_obs_packed_d: bad argument (out_d must be 8-byte aligned)") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    FLUSH_READER(h);
    hipLaunchKernelGGL(k_obs_packed, dim3(obs_grid(h->T)), dim3(OBS_BLOCK), 0, h->stream, h->S, h->N, player, out_d);
    HIPCHK(h, hipGetLastError());
    return PK_OK;
}

int pk_set_env_obs_packed(pk_handle *h, uint8_t *obs_packed_d) {
    if (!h || ((uintptr_t)obs_packed_d & 7)) return h ? h->fail(PK_E_INVALID_ARG, "pk_set_env_obs_packed: the buffer must be 8-byte aligned") : PK_E_INVALID_ARG;
    if (in_flight(h)) return h->fail(PK_E_BUSY, "pk_set_env_obs_packed: PokerGameEnv steps are in flight");
    h->env_obs_packed = obs_packed_d;
    return PK_OK;
}

int pk_set_step_obs(pk_handle *h, double *obs_d, uint8_t *obs_packed_d) {
    if (!h || ((uintptr_t)obs_d & 7) || ((uintptr_t)obs_packed_d & 7)) return h ? h->fail(PK_E_INVALID_ARG, "pk_set_step_obs: the buffers must be 8-byte aligned") : PK_E_INVALID_ARG;
    if (in_flight(h)) return h->fail(PK_E_BUSY, "pk_set_step_obs: steps are in flight; drain them first");
    h->step_obs = obs_d; h->step_obs_packed = obs_packed_d;
    return PK_OK;
}

int pk_host_alloc(void **out, size_t bytes) {
    if (!out || bytes == 0) { g_err = "pk_host_alloc: bad argument"; return PK_E_INVALID_ARG; }
    *out = nullptr;
    int rc = check_device_any();
    if (rc) return rc;
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocPortable);
    if (e != hipSuccess) { g_err = std::string("pk_host_alloc: ") + hipGetErrorString(e); *out = nullptr; return PK_E_OOM; }
    return PK_OK;
}

int pk_host_free(void *p) {
    if (!p) return PK_OK;
    return hipHostFree(p) == hipSuccess ? PK_OK : PK_E_HIP;
}

// Game.step's precondition (game.py:648-651) for a whole batch WITHOUT touching a table: index of the first table whose
// action is not valid for its active player, or -1.
int pk_check_actions(pk_handle *h, const int32_t *actions, int32_t *first_bad) {
    if (!h || !actions || !first_bad) return h ? h->fail(PK_E_INVALID_ARG, "pk_check_actions: NULL buffer") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    FLUSH(h);
    const size_t T = (size_t)h->T;
    int32_t *slot = (int32_t *)h->d_totals;                         // (8-byte slot 0 of the counters' output: free between calls)
    const int32_t init = 0x7fffffff;
    HIPCHK(h, hipMemcpyAsync(h->d_actions, actions, T * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(slot, &init, 4, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_check_actions, dim3(flat_grid(T)), dim3(256), 0, h->stream, h->S.valid, h->d_actions, h->T, slot);
    HIPCHK(h, hipGetLastError());
    int32_t got = 0;
    HIPCHK(h, hipMemcpyAsync(&got, slot, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *first_bad = got == init ? -1 : got;
    return PK_OK;
}

int pk_get_valid_actions_d(pk_handle *h, int player, uint8_t *out_d) {
    if (!h || !out_d || bad_player(h, player)) return h ? h->fail(PK_E_INVALID_ARG, "pk_get_valid_actions_d: bad argument") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    FLUSH_READER(h);
    size_t n = (size_t)h->T * PK_NUM_MOVES;
    hipLaunchKernelGGL(k_export_valid, dim3(flat_grid(n)), dim3(256), 0, h->stream, h->S, h->N, player, out_d);
    HIPCHK(h, hipGetLastError());
    return PK_OK;
}

int pk_env_reset_d(pk_handle *h, const uint8_t *mask_d, int opp_policy) {
    if (!h || bad_policy(opp_policy)) return h ? h->fail(PK_E_INVALID_ARG, "pk_env_reset_d: bad argument") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    FLUSH(h);
    const EnvResetKernArgs ka{(const State *)h->d_S, h->hot_env, mask_d, uniform_seats(opp_policy), env_park(h)};
    DISPATCH_N(h, k_env_reset, env_grid(h), ka);
    HIPCHK(h, hipGetLastError());
    return PK_OK;
}

int pk_env_step_d(pk_handle *h, const int32_t *actions_d, int opp_policy, double *reward_d, uint8_t *done_d,
                  uint8_t *hand_d, uint8_t *terr_d) {
    if (!h || !actions_d || !reward_d || !done_d || !hand_d || !terr_d || bad_policy(opp_policy))
        return h ? h->fail(PK_E_INVALID_ARG, "pk_env_step_d: bad argument") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    FLUSH(h);
    const EnvKernArgs ka = env_args(h, actions_d, -1, uniform_seats(opp_policy), 0, reward_d, done_d, hand_d, terr_d, nullptr);
    DISPATCH_N(h, k_env_step, env_grid(h), ka);
    HIPCHK(h, hipGetLastError());
    return PK_OK;
}

int pk_env_step_fused_d(pk_handle *h, const int32_t *actions_d, int seat0_policy, int opp_policy, int auto_reset,
                        double *reward_d, uint8_t *done_d, uint8_t *hand_d, uint8_t *terr_d, double *obs_d) {
    if (!h || !reward_d || !done_d || !hand_d || !terr_d || bad_policy(opp_policy) ||
        (!actions_d && (bad_policy(seat0_policy))))
        return h ? h->fail(PK_E_INVALID_ARG, "pk_env_step_fused_d: bad argument") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    FLUSH(h);
    const EnvKernArgs ka = with_packed(h, env_args(h, actions_d, actions_d ? -1 : seat0_policy, uniform_seats(opp_policy), auto_reset ? 1 : 0, reward_d, done_d, hand_d, terr_d, obs_d));
    DISPATCH_N(h, k_env_step, env_grid(h), ka);
    HIPCHK(h, hipGetLastError());
    return PK_OK;
}

int pk_env_step_async_d(pk_handle *h, const int32_t *actions_d, int seat0_policy, int opp_policy, int auto_reset, int max_passes,
                        double *reward_d, uint8_t *done_d, uint8_t *hand_d, uint8_t *terr_d, double *obs_d, uint8_t *ready_d) {
    if (!h || !reward_d || !done_d || !hand_d || !terr_d || !ready_d || bad_policy(opp_policy) ||
        (!actions_d && (bad_policy(seat0_policy))))
        return h ? h->fail(PK_E_INVALID_ARG, "pk_env_step_async_d: bad argument") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    if (h->host_step || h->step_pending) return flush(h);   // PK_E_BUSY: a host-buffer step awaits its end / Game.steps are in flight
    const int s0 = actions_d ? -1 : seat0_policy, au = auto_reset ? 1 : 0;
    // steps in flight keep THEIR agents and reset rule (as owed rollout steps do): a call that would change them is refused
    if (h->env_pending && (h->env_multi || s0 != h->env_seat0 || opp_policy != h->env_opp || au != h->env_auto))
        return h->fail(PK_E_INVALID_ARG, "pk_env_step_async_d: seat-0 source, opp_policy and auto_reset must stay the same while steps are in flight (drain with max_passes = 0 first)");
    h->env_seat0 = s0; h->env_opp = opp_policy; h->env_auto = au;
    int rc = flush_rollout(h);
    if (rc) return rc;
    const int mp = max_passes > 0 ? max_passes : 0;
    if (h->env_batches <= 1) {
        EnvKernArgs ka = with_packed(h, env_args(h, actions_d, actions_d ? -1 : seat0_policy, uniform_seats(opp_policy), au, reward_d, done_d, hand_d, terr_d, obs_d));
        ka.A.ready = ready_d; ka.A.max_passes = mp;
        DISPATCH_N(h, k_env_step_async, env_grid(h), ka);
        HIPCHK(h, hipGetLastError());
        h->env_last_begin = 0; h->env_last_end = h->T; h->env_last_fresh = 0;
    } else {
        // the caller's writes so far (actions of the range about to be launched) order before the launch(es) below
        // ... unless that stream is IDLE (everything the caller queued there has completed: a host-side learner, bench.py): then
        // there is nothing to order, and no barrier packet is put into any queue (see the note on the delivery wait below)
        hipError_t qs = hipStreamQuery(h->stream);
        if (qs != hipSuccess && qs != hipErrorNotReady) return h->fail(PK_E_HIP, "hipStreamQuery", qs);
        if (qs == hipErrorNotReady) (void)hipGetLastError();
        const bool caller_busy = qs == hipErrorNotReady;
        if (caller_busy) HIPCHK(h, hipEventRecord(h->env_in, h->stream));
        auto launch_range = [&](int b, int passes) -> int {
            const int t0 = b * h->env_range, tend = (t0 + h->env_range < h->T) ? t0 + h->env_range : h->T;
            if (caller_busy) HIPCHK(h, hipStreamWaitEvent(h->env_streams[b], h->env_in, 0));
            EnvKernArgs ka = with_packed(h, env_args(h, actions_d, actions_d ? -1 : seat0_policy, uniform_seats(opp_policy), au, reward_d, done_d, hand_d, terr_d, obs_d));
            ka.A.ready = ready_d; ka.A.max_passes = passes; ka.A.t0 = t0; ka.A.tend = tend;
            DISPATCH_N_ON(h, h->env_streams[b], k_env_step_async, (tend - t0 + h->env_tpb - 1) / h->env_tpb, ka);
            HIPCHK(h, hipGetLastError());
            HIPCHK(h, hipEventRecord(h->env_done[b], h->env_streams[b]));
            return PK_OK;
        };
        if (mp == 0) {   // drain: every range runs to its end; everything is delivered
            for (int b = 0; b < h->env_batches; ++b) { rc = launch_range(b, 0); if (rc) return rc; }
            for (int b = 0; b < h->env_batches; ++b) { HIPCHK(h, hipStreamWaitEvent(h->stream, h->env_done[b], 0)); h->env_launched[b] = false; }
            // everything is delivered; the next bounded call starts the round again with range 0
            h->env_last_begin = 0; h->env_last_end = h->T; h->env_last_fresh = 0; h->env_next = 0;
        } else {         // launch one range, deliver the one launched longest ago (the next in the round)
            const int b = h->env_next, d = (b + 1) % h->env_batches;
            rc = launch_range(b, mp);
            if (rc) return rc;
            h->env_launched[b] = true;
            h->env_last_fresh = h->env_launched[d] ? 0 : 1;
            // Delivery of the range launched longest ago.  The HOST waits for it (it was launched B - 1 calls ago and is done or
            // nearly so), then the caller's stream needs no device-side dependency at all.  A hipStreamWaitEvent on the caller's
            // stream is a barrier packet parked at the head of its hardware queue: when that queue happens to share a pipe of the
            // command processor with one of the range queues (which queues share depends on how many the process created before),
            // the parked barrier holds that range's dispatches back -- the same call sequence ran at 1.5 G env.step/s in a process
            // that had used another PokerGameEnv handle before, 3.6 G in a fresh one.  env PK_ENV_HOST_WAIT=0: the device-side wait.
            static const bool host_wait = !(getenv("PK_ENV_HOST_WAIT") && atoi(getenv("PK_ENV_HOST_WAIT")) == 0);
            if (h->env_launched[d]) {
                if (host_wait) HIPCHK(h, hipEventSynchronize(h->env_done[d]));
                else HIPCHK(h, hipStreamWaitEvent(h->stream, h->env_done[d], 0));
            }
            h->env_last_begin = d * h->env_range;
            h->env_last_end = (h->env_last_begin + h->env_range < h->T) ? h->env_last_begin + h->env_range : h->T;
            h->env_next = d;
        }
    }
    h->env_pending = max_passes > 0;
    h->env_multi = false;
    return PK_OK;
}

int pk_set_env_batches(pk_handle *h, int batches) {
    if (!h || batches < 1 || batches > PK_MAX_ENV_BATCHES) return h ? h->fail(PK_E_INVALID_ARG, "pk_set_env_batches: 1 <= batches <= PK_MAX_ENV_BATCHES") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    FLUSH(h);                                         // PK_E_BUSY while env steps are in flight: drain first
    HIPCHK(h, hipStreamSynchronize(h->stream));
    int range = (h->T + batches - 1) / batches;
    range = (range + 63) / 64 * 64;                   // whole waves per range
    const int nb = (h->T + range - 1) / range;        // (fewer ranges than asked for when the batch is small)
    for (int b = 0; b < nb && nb > 1; ++b) {
        if (!h->env_streams[b]) HIPCHK(h, stream_acquire(h->device, &h->env_streams[b], true));
        if (!h->env_done[b]) HIPCHK(h, hipEventCreateWithFlags(&h->env_done[b], hipEventDisableTiming));
        h->env_launched[b] = false;
    }
    if (nb > 1 && !h->env_in) HIPCHK(h, hipEventCreateWithFlags(&h->env_in, hipEventDisableTiming));
    h->env_batches = nb; h->env_range = range; h->env_next = 0;
    // "the range one call delivers is the range the next call launches": before the first call that is range 0, untouched
    h->env_last_begin = 0; h->env_last_end = nb > 1 ? (range < h->T ? range : h->T) : h->T; h->env_last_fresh = nb > 1 ? 1 : 0;
    return PK_OK;
}

int pk_env_last_range(pk_handle *h, int *begin, int *end, int *fresh) {
    if (!h) return PK_E_INVALID_ARG;
    if (begin) *begin = h->env_last_begin;
    if (end) *end = h->env_last_end;
    if (fresh) *fresh = h->env_last_fresh;
    return PK_OK;
}

static int check_seat_policies(pk_handle *h, uint64_t seat_policies, bool *any_external) {
    *any_external = false;
    for (int p = 0; p < h->N; ++p) {
        const int pol = PK_SEAT_POLICY(seat_policies, p);
        if (pol == PK_POLICY_EXTERNAL) *any_external = true;
        else if (bad_policy(pol)) return h->fail(PK_E_INVALID_ARG, "seat_policies: nibble p must be PK_POLICY_RANDOM / _ALLIN / _CALL or PK_POLICY_EXTERNAL for every seat p < num_players");
    }
    return PK_OK;
}

int pk_env_step_multi_d(pk_handle *h, const int32_t *actions_d, const uint8_t *reset_d, uint64_t seat_policies, int auto_reset,
                        int max_passes, double *reward_d, uint8_t *done_d, uint8_t *hand_d, uint8_t *terr_d, double *obs_d,
                        uint8_t *who_d, uint8_t *ready_d) {
    if (!h || !reward_d || !done_d || !hand_d || !terr_d || !ready_d || !who_d)
        return h ? h->fail(PK_E_INVALID_ARG, "pk_env_step_multi_d: NULL buffer") : PK_E_INVALID_ARG;
    bool ext = false;
    int rc = check_seat_policies(h, seat_policies, &ext);
    if (rc) return rc;
    if (ext && !actions_d) return h->fail(PK_E_INVALID_ARG, "pk_env_step_multi_d: actions_d is required when a seat is PK_POLICY_EXTERNAL");
    ON_DEVICE(h);
    if (h->host_step || h->step_pending) return flush(h);   // PK_E_BUSY
    const int au = auto_reset ? 1 : 0;
    // steps in flight keep THEIR agents and reset rule: a call that would change them is refused
    if (h->env_pending && (!h->env_multi || seat_policies != h->env_seats || au != h->env_auto))
        return h->fail(PK_E_INVALID_ARG, "pk_env_step_multi_d: seat_policies and auto_reset must stay the same while steps are in flight (pk_env_end_multi_d first)");
    rc = flush_rollout(h);
    if (rc) return rc;
    const int pol0 = PK_SEAT_POLICY(seat_policies, 0);
    EnvKernArgs ka = with_packed(h, env_args(h, actions_d, pol0 == PK_POLICY_EXTERNAL ? -1 : pol0, seat_policies, au, reward_d, done_d, hand_d, terr_d, obs_d));
    ka.A.ready = ready_d; ka.A.max_passes = max_passes > 0 ? max_passes : 0; ka.A.reset_req = reset_d; ka.A.who = who_d;
    DISPATCH_N(h, k_env_step_multi, env_grid(h), ka);
    HIPCHK(h, hipGetLastError());
    // a table may be waiting for the caller's action for an external seat even after a drain: in flight until pk_env_end_multi_d
    h->env_pending = max_passes > 0 || ext;
    h->env_multi = true; h->env_seats = seat_policies; h->env_auto = au;
    return PK_OK;
}

int pk_env_end_multi_d(pk_handle *h) {
    if (!h) return PK_E_INVALID_ARG;
    ON_DEVICE(h);
    if (!h->env_pending) return PK_OK;
    if (!h->env_multi) return h->fail(PK_E_INVALID_ARG, "pk_env_end_multi_d: the steps in flight belong to pk_env_step_async_d (drain them with max_passes = 0)");
    const int pol0 = PK_SEAT_POLICY(h->env_seats, 0);
    // drain; what is still in flight afterwards waits for an external seat's action between two Game.steps and is abandoned.
    // Outputs of steps that return during this drain go to the handle's own staging buffers, i.e. are dropped.
    EnvKernArgs ka = env_args(h, nullptr, pol0 == PK_POLICY_EXTERNAL ? -1 : pol0, h->env_seats, h->env_auto, h->d_reward, h->d_done, h->d_handf, h->d_terr, nullptr);
    ka.A.ready = h->d_flags; ka.A.who = h->d_mask; ka.A.abandon = 1; ka.A.obs_packed = nullptr;
    DISPATCH_N(h, k_env_step_multi, env_grid(h), ka);
    HIPCHK(h, hipGetLastError());
    h->env_pending = false; h->env_multi = false;
    return PK_OK;
}

int pk_pick_actions_d(pk_handle *h, int policy, int32_t *actions_d) {
    if (!h || !actions_d || bad_policy(policy)) return h ? h->fail(PK_E_INVALID_ARG, "pk_pick_actions_d: bad argument") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    FLUSH_READER(h);
    DISPATCH_N(h, k_pick, table_grid(h), h->S, h->hot, policy, actions_d);
    HIPCHK(h, hipGetLastError());
    return PK_OK;
}

int pk_pick_actions(pk_handle *h, int policy, int32_t *actions) {
    if (!h || !actions || bad_policy(policy)) return h ? h->fail(PK_E_INVALID_ARG, "pk_pick_actions: bad argument") : PK_E_INVALID_ARG;
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
static int enqueue_rollout(pk_handle *h, int k_steps, int policy, int auto_reset, int fused, bool complete, bool may_coalesce = true) {
    if (in_flight(h)) return flush(h);     // PK_E_BUSY (k_rollout would carry a Game.step of pk_step_async_d on as if it were deferred rollout work)
    if ((h->pending || h->acc) && (policy != h->pend_policy || auto_reset != h->pend_auto)) FLUSH(h);  // owed steps keep THEIR agents
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
    if (complete || !auto_reset) {
        k_steps += h->acc; h->acc = 0;
        return launch_rollout(h, k_steps, policy, auto_reset, 1);
    }
    if (k_steps > (1 << 30) - h->acc) FLUSH(h);
    h->acc += k_steps;
    h->pend_policy = policy; h->pend_auto = auto_reset;
    if (may_coalesce && h->coalesce > 0 && h->acc < h->coalesce && h->ev_used[h->ev_idx]) {
        // the launch before the last one (the ring slot about to be reused) still running: two launches are in flight,
        // the GPU will not idle if this call just leaves its steps with the host
        hipError_t q = hipEventQuery(h->ev_ring[h->ev_idx]);
        if (q == hipErrorNotReady) { (void)hipGetLastError(); return PK_OK; }   // "not ready" is an answer, not an error to keep
        if (q != hipSuccess) return h->fail(PK_E_HIP, "hipEventQuery", q);
    }
    return launch_coalesced(h, policy, auto_reset);
}

int pk_rollout(pk_handle *h, int k_steps, int policy, int auto_reset, int fused, uint64_t *counters) {
    if (!h || k_steps < 0 || bad_policy(policy)) return h ? h->fail(PK_E_INVALID_ARG, "pk_rollout: bad argument") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    int rc = enqueue_rollout(h, k_steps, policy, auto_reset ? 1 : 0, fused, counters != nullptr);
    if (rc) return rc;
    if (counters) return fetch_counters(h, counters);
    return PK_OK;
}

int pk_set_coalesce(pk_handle *h, int max_steps) {
    if (!h || max_steps < 0 || max_steps > (1 << 20)) return h ? h->fail(PK_E_INVALID_ARG, "pk_set_coalesce: 0 <= max_steps <= 2^20") : PK_E_INVALID_ARG;
    h->coalesce = max_steps;
    return PK_OK;
}

int pk_get_launch_stats(pk_handle *h, uint64_t *out, int reset) {
    if (!h || !out) return PK_E_INVALID_ARG;
    out[0] = h->st_launches; out[1] = h->st_steps; out[2] = h->st_min; out[3] = h->st_max;
    if (reset) h->st_launches = h->st_steps = h->st_min = h->st_max = 0;
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
    if (!h || !ms_per_launch || reps < 1 || k_steps < 0 || bad_policy(policy))
        return h ? h->fail(PK_E_INVALID_ARG, "pk_time_rollout: bad argument") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    FLUSH(h);
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    for (int r = 0; r < reps; ++r) {
        // k_steps == 0: empty launches (load the tables, store them) -- the fixed cost of a launch, for diagnostics
        int rc = k_steps ? enqueue_rollout(h, k_steps, policy, auto_reset ? 1 : 0, fused, false, false)
                         : launch_rollout(h, 0, policy, auto_reset ? 1 : 0, 1);
        if (rc) return rc;
    }
    FLUSH(h);  // what the deferred launches left is part of the work that is being timed
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    int launches = reps * (fused ? 1 : k_steps);   // the flush is work of these launches: its time is shared among them
    *ms_per_launch = launches ? (double)ms / launches : 0.0;
    if (counters) return fetch_counters(h, counters);
    return PK_OK;
}

int pk_env_reset(pk_handle *h, const uint8_t *mask, int opp_policy) {
    if (!h || bad_policy(opp_policy)) return h ? h->fail(PK_E_INVALID_ARG, "pk_env_reset: bad argument") : PK_E_INVALID_ARG;
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
    if (!h || !actions || !reward || !done || !hand || bad_policy(opp_policy))
        return h ? h->fail(PK_E_INVALID_ARG, "pk_env_step: bad argument") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    if (in_flight(h)) return flush(h);     // PK_E_BUSY, before anything is queued
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

// PokerGameEnv.step through HOST buffers, in two halves: pk_env_step_begin uploads the actions, launches the step (seat 0 =
// the caller's actions; finished episodes reset on the spot when auto_reset != 0) and queues the copies of every output into
// the caller's buffers; pk_env_step_end waits for them.  With PINNED buffers (pk_host_alloc) nothing in `begin` blocks, the
// copies run at PCIe line rate, and the caller -- or another handle's step -- overlaps with them.  obs / obs_packed may be
// NULL (no row of that kind).  Between begin and end the handle may not be used otherwise.
int pk_env_step_begin(pk_handle *h, const int32_t *actions, int opp_policy, int auto_reset, double *reward, uint8_t *done,
                      uint8_t *hand, uint8_t *terr, double *obs, uint8_t *obs_packed) {
    if (!h || !actions || !reward || !done || !hand || !terr || bad_policy(opp_policy))
        return h ? h->fail(PK_E_INVALID_ARG, "pk_env_step_begin: bad argument") : PK_E_INVALID_ARG;
    ON_DEVICE(h);
    FLUSH(h);
    const size_t T = (size_t)h->T, D = (size_t)PK_OBS_DIM(h->N), PB = (size_t)PK_OBS_PACKED_BYTES(h->N);
    if (obs && T * D * 8 > h->export_bytes) return h->fail(PK_E_INVALID_ARG, "export buffer too small");
    if (obs_packed && !h->d_obs_packed) {
        HIPCHK(h, hipMalloc((void **)&h->d_obs_packed, T * PB));
    }
    HIPCHK(h, hipMemcpyAsync(h->d_actions, actions, T * 4, hipMemcpyHostToDevice, h->stream));
    EnvKernArgs ka = env_args(h, h->d_actions, -1, uniform_seats(opp_policy), auto_reset ? 1 : 0, h->d_reward, h->d_done, h->d_handf, h->d_terr,
                              obs ? (double *)h->d_export : nullptr);
    ka.A.obs_packed = obs_packed ? h->d_obs_packed : nullptr;
    DISPATCH_N(h, k_env_step, env_grid(h), ka);
    HIPCHK(h, hipGetLastError());
    // Busy from HERE on: if one of the copies below fails to queue, the earlier ones are already queued into the caller's buffers, and
    // pk_env_step_end is the caller's sanctioned way to wait for them (round 5 set the flag after the last copy: ADVICE r05).
    h->host_step = true;      // until pk_env_step_end: every entry point that reads or changes tables (another begin included) is PK_E_BUSY
    HIPCHK(h, hipMemcpyAsync(reward, h->d_reward, T * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(done, h->d_done, T, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(hand, h->d_handf, T, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(terr, h->d_terr, T, hipMemcpyDeviceToHost, h->stream));
    if (obs) HIPCHK(h, hipMemcpyAsync(obs, h->d_export, T * D * 8, hipMemcpyDeviceToHost, h->stream));
    if (obs_packed) HIPCHK(h, hipMemcpyAsync(obs_packed, h->d_obs_packed, T * PB, hipMemcpyDeviceToHost, h->stream));
    return PK_OK;
}

int pk_env_step_end(pk_handle *h) {
    if (!h) return PK_E_INVALID_ARG;
    ON_DEVICE(h);
    if (!h->host_step) return h->fail(PK_E_INVALID_ARG, "pk_env_step_end: no pk_env_step_begin is waiting for it");
    h->host_step = false;     // (whatever the wait below returns, nothing is queued into the caller's buffers any more once it has)
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PK_OK;
}

#ifdef PK_PROFILE
// Diagnostic library only: read and clear the per-block cycle sums (slots: pk_device.hpp PF_*).
int pk_prof_read(pk_handle *h, unsigned long long *out) {
    if (!h || !out) return PK_E_INVALID_ARG;
    ON_DEVICE(h);
    if (!h->env_pending) FLUSH(h);   // (the counters themselves can be read while env steps are in flight)
    HIPCHK(h, hipMemcpyAsync(out, h->S.prof, PF_SLOTS * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemsetAsync(h->S.prof, 0, PF_SLOTS * 8, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PK_OK;
}
#endif

int pk_sync(pk_handle *h) {
    if (!h) return PK_E_INVALID_ARG;
    ON_DEVICE(h);
    if (!in_flight(h)) FLUSH(h);     // env steps in flight stay in flight: only wait for the launches made so far
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int b = 0; b < h->env_batches && h->env_batches > 1; ++b) HIPCHK(h, hipStreamSynchronize(h->env_streams[b]));
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
// A per-device cache of device memory (the judger's scratch arena, the evaluator table) outlives its allocation when the application resets the
// device: a kernel launched on such a pointer would fault.  Same seat belt as the stream pool's canary: look the ADDRESS up before trusting it.
// The allocator hands the same addresses out again after a reset (measured: the evaluator table's address came back as the next handle's arena),
// so base AND size must match -- and every cached allocation has an odd size (PK_ODD_BYTES more than it needs) that no caller is likely to ask for.
#define PK_ODD_BYTES 272
static bool dev_alloc_alive(const void *p, size_t bytes) {
    if (!p) return false;
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    const hipError_t e = hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p);
    if (e != hipSuccess) (void)hipGetLastError();
    return e == hipSuccess && (const void *)base == p && size == bytes;
}
static void *scratch(int device, size_t bytes) {  // caller holds g_scratch_mu and has the device current
    auto &s = g_scratch[device];
    if (s.p && !dev_alloc_alive(s.p, s.cap + PK_ODD_BYTES)) { s.p = nullptr; s.cap = 0; }      // (the device was reset: the arena is gone, not freed)
    if (s.cap < bytes) {
        if (s.p) (void)hipFree(s.p);
        s.p = nullptr; s.cap = 0;
        size_t want = bytes < (1u << 20) ? (1u << 20) : bytes + bytes / 2;
        if (hipMalloc(&s.p, want + PK_ODD_BYTES) != hipSuccess) { (void)hipGetLastError(); s.p = nullptr; return nullptr; }
        s.cap = want;
    }
    return s.p;
}

// pk_eval_hands(_d)'s launch: the table path (k_eval_hands_tab); the register evaluator if the table could not be allocated or
// env PK_EVAL_HANDS_TAB=0 (A/B knob).  tab: eval7_table(device), fetched by the caller BEFORE it takes g_scratch_mu.
static void launch_eval_hands(const uint32_t *tab, const uint8_t *cards_d, const uint8_t *ncards_d, size_t m, uint8_t *rank_d, uint32_t *kick_d,
                              uint8_t *nkick_d, hipStream_t stream) {
    static const bool use_tab = !(getenv("PK_EVAL_HANDS_TAB") && atoi(getenv("PK_EVAL_HANDS_TAB")) == 0);
    if (tab && use_tab) {
        static const int grid_env = getenv("PK_EVAL_HANDS_GRID") ? atoi(getenv("PK_EVAL_HANDS_GRID")) : 256 * 8;
        static const int grid_max = grid_env < 1 ? 1 : grid_env;              // (a knob value of 0 or below: one workgroup, never an empty or a wrapped-around grid)
        const size_t chunks = (m + EVAL_TAB_BLOCK - 1) / EVAL_TAB_BLOCK;      // every workgroup copies the 32 KB table: no more of them than have hands
        const unsigned grid = (unsigned)(chunks < (size_t)grid_max ? chunks : (size_t)grid_max);   // four resident per CU (LDS), two rounds; grid-stride over the rest
        if (ncards_d) hipLaunchKernelGGL(k_eval_hands_tab<true>, dim3(grid), dim3(EVAL_TAB_BLOCK), 0, stream, cards_d, ncards_d, m, rank_d, kick_d, nkick_d, tab);
        else hipLaunchKernelGGL(k_eval_hands_tab<false>, dim3(grid), dim3(EVAL_TAB_BLOCK), 0, stream, cards_d, ncards_d, m, rank_d, kick_d, nkick_d, tab);
    } else {
        hipLaunchKernelGGL(k_eval_hands, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream, cards_d, ncards_d, m, rank_d, kick_d, nkick_d);
    }
}

int pk_eval_hands_d(int device, const uint8_t *cards_d, const uint8_t *ncards_d, size_t m, uint8_t *rank_d, uint32_t *kick_d,
                    uint8_t *nkick_d, void *stream) {
    if (!cards_d || !rank_d || !kick_d) { g_err = "pk_eval_hands_d: NULL buffer"; return PK_E_INVALID_ARG; }
    int rc = check_device(device);
    if (rc) return rc;
    DeviceGuard guard(device);
    if (!guard.ok) { g_err = "hipSetDevice failed"; return PK_E_HIP; }
    if (m == 0) return PK_OK;
    launch_eval_hands(eval7_table(device, (hipStream_t)stream), cards_d, ncards_d, m, rank_d, kick_d, nkick_d, (hipStream_t)stream);   // (a first call builds the table on the CALLER's stream)
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
    const uint32_t *tab = eval7_table(device);   // (takes g_scratch_mu itself)
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    size_t off_n = m * 7, off_r = off_n + m, off_nk = off_r + m, off_k = (off_nk + m + 3) & ~(size_t)3, total = off_k + m * 4;
    uint8_t *d = (uint8_t *)scratch(device, total);
    if (!d) { g_err = "pk_eval_hands: out of device memory"; return PK_E_OOM; }
    hipError_t e = hipMemcpyAsync(d, cards, m * 7, hipMemcpyHostToDevice, 0);
    if (e == hipSuccess && ncards) e = hipMemcpyAsync(d + off_n, ncards, m, hipMemcpyHostToDevice, 0);
    if (e == hipSuccess) {
        launch_eval_hands(tab, d, ncards ? d + off_n : nullptr, m, d + off_r, (uint32_t *)(d + off_k), d + off_nk, 0);
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

// The 32 KB table of the table-driven evaluator (eval7_tab), one per device, built on first use.
static uint32_t *g_eval_tab[PK_MAX_DEVICES];
static const uint32_t *eval7_table(int device, hipStream_t stream) {   // the device is current
    if (device < 0 || device >= PK_MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_eval_tab[device] && !dev_alloc_alive(g_eval_tab[device], EVAL7_TAB_WORDS * 4 + PK_ODD_BYTES)) g_eval_tab[device] = nullptr;   // (the device was reset: build it again)
    if (!g_eval_tab[device]) {
        uint32_t *p = nullptr;
        if (hipMalloc((void **)&p, EVAL7_TAB_WORDS * 4 + PK_ODD_BYTES) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        hipLaunchKernelGGL(k_make_eval7_tab, dim3(EVAL7_TAB_WORDS / 256), dim3(256), 0, stream, p);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) { (void)hipFree(p); return nullptr; }
        g_eval_tab[device] = p;
    }
    return g_eval_tab[device];
}

static int launch_eval7(int device, const uint64_t *hands_d, size_t m, uint32_t *out_d, int distinct) {
    const unsigned grid = 256 * 16;  // 16 workgroups per CU, grid-stride over the rest
    const bool vec = (((uintptr_t)hands_d & 15) | ((uintptr_t)out_d & 7)) == 0;  // 16-byte loads / 8-byte stores need it
    static const bool no_tab = getenv("PK_EVAL7_NOTAB") != nullptr;   // A/B knob: the register-only evaluator of the table kernels
    const uint32_t *tab = (distinct && !no_tab) ? eval7_table(device) : nullptr;
    if (tab) {   // four 512-thread workgroups per CU (LDS: 4 x 32 KB), two rounds of them
        // A/B knob (bit 0 clear: software prefetch of the next hands; bit 1 clear: both hands' lookups issued before either is used):
        // all four within 1.5 % of each other at 2^28 hands; the plain form is the default
        static const int variant = getenv("PK_EVAL7_VARIANT") ? atoi(getenv("PK_EVAL7_VARIANT")) : 3;
        static const int grid_env = getenv("PK_EVAL7_GRID") ? atoi(getenv("PK_EVAL7_GRID")) : 256 * 32;
        static const int grid_max = grid_env < 1 ? 1 : grid_env;
        const size_t want = (m / 2 + 511) / 512;     // every workgroup copies the 32 KB table: no more of them than have hands
        const unsigned gridx = (unsigned)(want < 1 ? 1 : (want < (size_t)grid_max ? want : (size_t)grid_max));
        if (vec && variant == 1) hipLaunchKernelGGL((k_eval7_tab_stream<true, 1>), dim3(gridx), dim3(512), 0, 0, hands_d, m, out_d, tab);
        else if (vec && variant == 2) hipLaunchKernelGGL((k_eval7_tab_stream<true, 2>), dim3(gridx), dim3(512), 0, 0, hands_d, m, out_d, tab);
        else if (vec && variant == 0) hipLaunchKernelGGL((k_eval7_tab_stream<true, 0>), dim3(gridx), dim3(512), 0, 0, hands_d, m, out_d, tab);
        else if (vec) hipLaunchKernelGGL((k_eval7_tab_stream<true, 3>), dim3(gridx), dim3(512), 0, 0, hands_d, m, out_d, tab);
        else hipLaunchKernelGGL((k_eval7_tab_stream<false, 3>), dim3(gridx), dim3(512), 0, 0, hands_d, m, out_d, tab);
    } else if (distinct) {
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
    if (launch_eval7(device, hands_d, m, out_d, distinct) != PK_OK || hipDeviceSynchronize() != hipSuccess) { g_err = "pk_eval7_d: launch failed"; return PK_E_HIP; }
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
    launch_eval7(device, hands_d, m, out_d, distinct);  // warm (instruction cache, clocks)
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) launch_eval7(device, hands_d, m, out_d, distinct);
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
    const uint32_t *tab = (fast == 2 || fast == 4) ? eval7_table(device) : nullptr;   // (takes g_scratch_mu itself)
    if ((fast == 2 || fast == 4) && !tab) { g_err = "pk_eval7_prefix: out of device memory"; return PK_E_OOM; }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    uint32_t *d = (uint32_t *)scratch(device, count * 4);
    if (!d) { g_err = "pk_eval7_prefix: out of device memory"; return PK_E_OOM; }
    hipLaunchKernelGGL(k_eval7_prefix, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, 0, a, b, fast, (uint32_t)count, d, tab);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(out, d, count * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return th.fail(PK_E_HIP, "pk_eval7_prefix", e);
    return PK_OK;
}

}  // extern "C"
