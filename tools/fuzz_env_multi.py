#!/usr/bin/env python3
"""One-off confidence run (GPU box): pk_env_step_multi_d -- one agent per seat, some seats played by the caller -- against
the CPU oracle over seeded odd configurations (every N, odd blinds / stacks, random per-seat policies random / all-in /
call, a random subset of the opponent seats external and played on the host by the policy's own rule, bounded launches of
1..9 passes with auto-reset): per table the delivered (reward, done, hand, terr) sequence must equal the oracle's.
usage: python tools/fuzz_env_multi.py [configs] [seed] [only this configuration index]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import golden_util as GU  # noqa: E402
import pokerl_amd  # noqa: E402
from pokerl_amd import _lib as L  # noqa: E402
from pokerl_amd.hipmem import DeviceBuffer  # noqa: E402
from oracle import loader as O  # noqa: E402
from oracle import rng_spec as R  # noqa: E402

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
only = int(sys.argv[3]) if len(sys.argv) > 3 else None
cap = int(os.environ.get("PK_FUZZ_LAUNCH_CAP", "6000"))   # (calling stations at one pass per launch: 3 300 launches per env.step seen)
stacks = [2, 5, 10, 37.5, 100, 1000]
blinds = [0.5, 1, 2, 3, 7.5, 40]
delivered = yields = 0
for i in range(n_cfg):
    N = 2 + i % 15   # 2 ... 16 seats
    start = [rng.choice(stacks) for _ in range(N)] if rng.random() < 0.5 else rng.choice(stacks)
    bb, sb = rng.choice(blinds), rng.choice(blinds)
    # shoving or random opponents for the caller's seats: calling stations would play endless games once seat 0 is broke
    pols = [rng.choice([0, 0, 1, 2]) for _ in range(N - 1)]
    external = [s for s in range(1, N) if pols[s - 1] != 2 and rng.random() < 0.5]
    seed, base = rng.getrandbits(63), rng.getrandbits(32) & 0xFFFFF000
    T, K, passes = rng.choice([65, 300, 700]), rng.choice([8, 15, 25]), rng.randrange(1, 10)
    cfg = dict(num_tables=T, num_players=N, start_credits=start, big_blind=bb, small_blind=sb, seed=seed, table_id_base=base)
    where = "cfg %d: %s pols=%s external=%s K=%d passes=%d" % (i, cfg, pols, external, K, passes)
    if only is not None and i != only:
        continue
    o = O.OracleGame(T, N, start, bb, sb, seed=seed, table_id_base=base)
    o.env_reset(None, pols)
    want = []
    for k in range(K):
        a = o.pick_actions(0)
        ro, do, ho, eo = o.env_step(a, pols)
        m = ((do != 0) | ((eo & 12) != 0)).astype(np.uint8)      # done, or PK_TERR_HAND_CAP / _ENV_CAP: auto-reset
        if m.any():
            o.env_reset(m, pols)
            eo = eo | (o.errs() * m)                                 # ... whose own error bits the fused call reports with the step's
        want.append((ro, do, ho, eo))
    agents = [(lambda st: 0) if s in external else [pokerl_amd.RandomAgent(), pokerl_amd.AllInAgent(), pokerl_amd.CallAgent()][pols[s - 1]]
              for s in range(1, N)]
    env = pokerl_amd.VecPokerGameEnv(agents, **cfg)
    g = env.game
    D = 17 + 3 * N
    rew, done, hand, terr, obs, who, ready, act, rst = (DeviceBuffer(T * 8), DeviceBuffer(T), DeviceBuffer(T), DeviceBuffer(T),
                                                        DeviceBuffer(T * D * 8), DeviceBuffer(T), DeviceBuffer(T), DeviceBuffer(T * 4), DeviceBuffer(T))
    count = np.full(T, -1, np.int64)
    rst.upload(np.ones(T, np.uint8))
    a = np.full(T, L.ACTION_SKIP, np.int32)
    launches, first = 0, True
    while count.min() < K:
        launches += 1
        assert launches < cap * K, where
        act.upload(a)
        env.step_multi_d(act.ptr, rst.ptr if first else None, rew.ptr, done.ptr, hand.ptr, terr.ptr, obs.ptr, who.ptr, ready.ptr,
                         max_passes=passes, auto_reset=True)
        first = False
        g.sync()
        r, w = ready.download(np.uint8, T), who.download(np.uint8, T)
        rows = obs.download(np.float64, T * D).reshape(T, D)
        te = terr.download(np.uint8, T)
        ret = r == 1
        idx = np.nonzero(ret & (count >= 0) & (count < K))[0]
        if len(idx):
            rw, dn, hd = rew.download(np.float64, T), done.download(np.uint8, T), hand.download(np.uint8, T)
            for t in idx:
                k = count[t]
                ro, do, ho, eo = want[k]
                assert rw[t:t + 1].tobytes() == ro[t:t + 1].tobytes() and dn[t] == do[t] and hd[t] == ho[t] and te[t] == eo[t], (where, t, k)
            delivered += len(idx)
        count[ret] += 1
        assert not te[r == 2].any() and np.isin(w[r == 2], external).all() and (w[ret] == 0).all(), where
        yields += int((r == 2).sum())
        # the caller's moves: seat 0 by the random agent's rule, its opponent seats by their policy's rule, from the delivered
        # row's valid mask and the table's step serial (readable while env calls are in flight)
        a = np.full(T, -1, np.int32)
        serial = g.step_serial
        bits = (rows[:, 3:10] > 0).astype(np.uint32) @ (1 << np.arange(7, dtype=np.uint32))
        for t in np.nonzero((r == 1) | (r == 2))[0]:
            pol = 0 if r[t] == 1 else pols[int(w[t]) - 1]
            a[t] = R.pick_action(seed, base + int(t), int(serial[t]), int(bits[t]), pol)
    env.end_multi()
    env.close()
    if only is not None:
        print("%s: %d launches" % (where, launches))
    if i % 10 == 9:
        print("%d configurations bit-exact so far" % (i + 1), flush=True)
print("fuzz: %d configurations, %d env.steps delivered, %d yields to caller-played seats, all equal to the oracle's sequences"
      % (n_cfg, delivered, yields))
