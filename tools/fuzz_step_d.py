#!/usr/bin/env python3
"""One-off confidence run (GPU box): Game.step with the caller's actions on DEVICE buffers -- pk_pick_actions_d + pk_step_auto_d
(the reset of a finished game inside the step's launch), on a twin handle pk_step_d + pk_reset_d(flags, GAME_OVER), and on a third one the
BOUNDED form pk_step_async_d (one or two hand ends per launch, reset inside; a table whose step rolls on stays in flight and is delivered by
a later call: its delivered flags / terr must be those the oracle returned when the step started, its state equal after a drain) -- against
the CPU oracle's step + reset over seeded odd configurations (every N, zero / fractional / oversized blinds, per-seat stacks
0.5 .. 1e6, any table-id base and dealer, batches from 65 to 4 097 tables: full and nearly empty waves).  This is the path on which a
step that rolls hand after hand is served by end_block's single-table paths (lone showdown, deck stock).  Every handle also has the step kernels
write the StateView row of the player to act (pk_set_step_obs: dense + packed / packed only / dense only): after every call the rows of the
tables whose step returned must be, byte for byte, what pk_get_obs_d / pk_get_obs_packed_d deliver.
usage: python tools/fuzz_step_d.py [configs] [seed]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import golden_util as GU  # noqa: E402
from hip_backend import HipBackend as HB  # noqa: E402
from oracle import loader as O  # noqa: E402
from pokerl_amd import _lib as L, packed_dtype  # noqa: E402
from pokerl_amd.hipmem import DeviceBuffer  # noqa: E402

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 2027)
stacks = [0.5, 1, 2, 3, 5, 10, 37.5, 100, 1000, 1e6]
blinds = [0, 0.25, 0.5, 1, 2, 3, 7.5, 40]


def same(a, b, where):
    for k in GU.SNAP_FIELDS:
        assert GU.bits_equal(a[k], b[k]), (where, k)


def check_rows(g, T, N, dense, packed, ref, refp, mask, where):
    """The rows the step kernel wrote (pk_set_step_obs) against the getter kernels' rows of the same state, for the tables in `mask`."""
    global rows_checked
    D, P = 17 + 3 * N, packed_dtype(N).itemsize
    lib = L.lib()
    if dense is not None:
        L.check(lib.pk_get_obs_d(g._h, -1, ref.ptr), g._h)
    if packed is not None:
        L.check(lib.pk_get_obs_packed_d(g._h, -1, refp.ptr), g._h)
    g.sync()
    if dense is not None:
        a, b = dense.download(np.uint64, T * D).reshape(T, D), ref.download(np.uint64, T * D).reshape(T, D)
        assert not (mask & (a != b).any(axis=1)).any(), (where, "dense rows")
    if packed is not None:
        a, b = packed.download(np.uint8, T * P).reshape(T, P), refp.download(np.uint8, T * P).reshape(T, P)
        assert not (mask & (a != b).any(axis=1)).any(), (where, "packed rows")
    rows_checked += int(mask.sum()) * ((dense is not None) + (packed is not None))


steps = resets = async_steps = async_inflight = rows_checked = 0
async_off = False
for i in range(n_cfg):
    N = 2 + i % 15
    start = [rng.choice(stacks) for _ in range(N)] if rng.random() < 0.5 else rng.choice(stacks)
    bb, sb = rng.choice(blinds), rng.choice(blinds)
    policy = 1 if rng.random() < 0.25 else 0
    seed, base, dealer = rng.getrandbits(63), rng.getrandbits(32) & 0xFFFFF000, rng.randrange(N)
    T = rng.choice([65, 128, 300, 1000, 4097])
    K = rng.choice([40, 90, 200])
    probe = O.OracleGame(16, N, start, bb, sb, seed=seed, table_id_base=base)
    probe.reset()
    t_probe = time.time()
    probe.rollout(K, policy, True)
    if (time.time() - t_probe) * T / 16 > 20:        # blinds far above the stacks: steps that roll thousands of hands
        T = 65
    where = "cfg %d: T=%d N=%d start=%s bb=%s sb=%s policy=%d K=%d" % (i, T, N, start, bb, sb, policy, K)
    o = O.OracleGame(T, N, start, bb, sb, seed=seed, table_id_base=base)
    ha, hb = HB(T, N, start, bb, sb, seed=seed, table_id_base=base), HB(T, N, start, bb, sb, seed=seed, table_id_base=base)
    hc, oc = HB(T, N, start, bb, sb, seed=seed, table_id_base=base), O.OracleGame(T, N, start, bb, sb, seed=seed, table_id_base=base)
    o.reset(); ha.reset(); hb.reset(); hc.reset(); oc.reset()   # (pk_step_auto_d resets with dealer 0, as Game.reset() does)
    bufs = [DeviceBuffer(T * 4), DeviceBuffer(T), DeviceBuffer(T), DeviceBuffer(T * 4), DeviceBuffer(T), DeviceBuffer(T),
            DeviceBuffer(T * 4), DeviceBuffer(T), DeviceBuffer(T), DeviceBuffer(T)]
    act_a, fl_a, te_a, act_b, fl_b, te_b, act_c, fl_c, te_c, rdy_c = bufs
    D, P = 17 + 3 * N, packed_dtype(N).itemsize
    obs = [DeviceBuffer(T * D * 8), DeviceBuffer(T * P), DeviceBuffer(T * P), DeviceBuffer(T * D * 8), DeviceBuffer(T * D * 8), DeviceBuffer(T * P)]
    dense_a, packed_a, packed_b, dense_c, ref_d, ref_p = obs
    bufs = bufs + obs
    ha.g.set_step_obs(dense_a, packed_a); hb.g.set_step_obs(None, packed_b); hc.g.set_step_obs(dense_c, None)
    all_t = np.ones(T, bool)
    hands_c = 1 + i % 2
    idle = np.ones(T, bool); want_f = np.zeros(T, np.uint8); want_e = np.zeros(T, np.uint8)

    def async_call(actions_for_idle, budget):
        """One pk_step_async_d call of the third handle against its own oracle `oc` (which makes a step at the call that starts it)."""
        global want_f, want_e, idle, async_steps, async_inflight, async_off
        fo2, eo2 = oc.step(np.where(idle, actions_for_idle, -1).astype(np.int32))
        want_f = np.where(idle, fo2, want_f); want_e = np.where(idle, eo2, want_e)
        hc.g.step_async_d(act_c, fl_c, te_c, rdy_c, max_hands=budget, auto_reset=True); hc.g.sync()
        r = rdy_c.download(np.uint8, T) != 0
        check_rows(hc.g, T, N, dense_c, None, ref_d, ref_p, r, where + " (async)")
        ov = ((want_f & 1) | ((want_e & 4) >> 2)).astype(np.uint8)
        exp = ((want_f & 6) | ov).astype(np.uint8)
        assert np.array_equal(fl_c.download(np.uint8, T)[r], exp[r]) and np.array_equal(te_c.download(np.uint8, T)[r], want_e[r]), (where, "async")
        if (r & ((want_e & 2) != 0)).any():               # game.py:473 on a delivered step (the caller would drain and reset it): the twin
            async_off = True                              # stops here for this configuration
        m2 = (r & (ov != 0)).astype(np.uint8)
        if m2.any():
            oc.reset(mask=m2)
        async_steps += int(r.sum()); async_inflight += int((~r).sum())
        idle = r.copy()

    for s in range(K):
        a = o.pick_actions(policy)
        fo, eo = o.step(a)
        over = ((fo & 1) | ((eo & 4) >> 2)).astype(np.uint8)
        ha.g.pick_actions_d(act_a, policy); ha.g.step_d(act_a, fl_a, te_a, auto_reset=True)
        hb.g.pick_actions_d(act_b, policy); hb.g.step_d(act_b, fl_b, te_b)
        ha.g.sync(); hb.g.sync()
        assert np.array_equal(act_a.download(np.int32, T), a) and np.array_equal(act_b.download(np.int32, T), a), (where, s)
        fa, ea = fl_a.download(np.uint8, T), te_a.download(np.uint8, T)
        fb, eb = fl_b.download(np.uint8, T), te_b.download(np.uint8, T)
        assert np.array_equal(fb, fo) and np.array_equal(eb, eo), (where, s)
        assert np.array_equal(fa & 6, fo & 6) and np.array_equal(fa & 1, over) and np.array_equal(ea, eo), (where, s)
        check_rows(ha.g, T, N, dense_a, packed_a, ref_d, ref_p, all_t, where + " step %d (auto)" % s)
        check_rows(hb.g, T, N, None, packed_b, ref_d, ref_p, all_t, where + " step %d (step_d)" % s)
        hb.g.reset_d(fl_b, L.FLAG_GAME_OVER)
        if (eo & 4).any():
            hb.g.reset_d(te_b, L.TERR_HAND_CAP)
        if (eo & 2).any():                            # the reference's assertion (game.py:473): the table stays as it is; reset it everywhere
            m = ((eo & 2) != 0).astype(np.uint8)
            o.reset(mask=m); ha.reset(mask=m); hb.reset(mask=m)
        if over.any():
            o.reset(mask=over)
            resets += int(over.sum())
        if not async_off and not (eo & 2).any():
            hc.g.pick_actions_d(act_c, policy); hc.g.sync()
            async_call(act_c.download(np.int32, T), hands_c)
        else:
            async_off = True                              # (the twin stops at the first game.py:473 table of this configuration)
        if s % 40 == 39 or s == K - 1:
            snap = o.snapshot()
            same(snap, ha.snapshot(), where + " step %d (auto)" % s)
            same(snap, hb.snapshot(), where + " step %d (step_d + reset_d)" % s)
    if not async_off:
        act_c.upload(np.full(T, -1, np.int32))            # drain: idle tables get "no step"
        async_call(np.full(T, -1, np.int32), 0)
        if not async_off:
            assert idle.all()
            same(oc.snapshot(), hc.snapshot(), where + " (async, drained)")
    async_off = False
    if hc.g._lib.pk_set_step_obs(hc.g._h, None, None) != L.PK_OK:      # (the twin stopped with steps in flight: drain first)
        act_c.upload(np.full(T, -1, np.int32))
        hc.g.step_async_d(act_c, fl_c, te_c, rdy_c, max_hands=0, auto_reset=True); hc.g.sync()
        hc.g.set_step_obs(None, None)
    ha.g.set_step_obs(None, None); hb.g.set_step_obs(None, None)
    hc.g.close()
    steps += 2 * T * K
    for b in bufs:
        b.free()
    ha.g.close(); hb.g.close()
    if i % 25 == 24:
        print("%d configurations bit-exact so far" % (i + 1), flush=True)
print("fuzz: %d configurations, %d device-resident Game.steps, %d games reset inside a step's launch, all bit-exact vs the oracle; "
      "bounded launches: %d steps delivered, %d times a table's step was left in flight, every delivery and every drained state equal; "
      "%d observation rows written by the step kernels (pk_set_step_obs) equal to the getter kernels' rows"
      % (n_cfg, steps, resets, async_steps, async_inflight, rows_checked))
