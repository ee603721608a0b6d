#!/usr/bin/env python3
"""One-off confidence run (GPU box): pk_env_step_async_d against pk_env_step_fused_d and the CPU oracle over seeded odd
configurations (every N, odd blinds / stacks, both opponent policies, pass budgets 1..9, half of them with the handle split
into 2..8 sub-batches by pk_set_env_batches): per table the delivered (reward, done, hand, obs row) sequence must equal the
synchronous one bit for bit.
usage: python tools/fuzz_env_async.py [configs] [seed]"""
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

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
stacks = [1, 2, 5, 10, 37.5, 100, 1000]
blinds = [0.5, 1, 2, 3, 7.5, 40]
lib = L.lib()
delivered = sub = 0
for i in range(n_cfg):
    N = 2 + i % 15   # 2 ... 16 seats
    start = [rng.choice(stacks) for _ in range(N)] if rng.random() < 0.5 else rng.choice(stacks)
    bb, sb = rng.choice(blinds), rng.choice(blinds)
    opp = 1 if rng.random() < 0.25 else 0
    seed, base = rng.getrandbits(63), rng.getrandbits(32) & 0xFFFFF000
    T, K, passes = rng.choice([65, 300, 1000]), rng.choice([8, 15, 25]), rng.randrange(1, 10)
    B = rng.choice([1, 1, 1, 2, 3, 4, 8])
    cfg = dict(num_tables=T, num_players=N, start_credits=start, big_blind=bb, small_blind=sb, seed=seed, table_id_base=base)
    if os.environ.get("PK_FUZZ_ONLY") and i != int(os.environ["PK_FUZZ_ONLY"]):
        continue
    where = "cfg %d: %s opp=%d K=%d passes=%d sub-batches=%d" % (i, cfg, opp, K, passes, B)
    D = 17 + 3 * N
    rew, done, hand, terr, obs, ready = (DeviceBuffer(T * 8), DeviceBuffer(T), DeviceBuffer(T), DeviceBuffer(T),
                                         DeviceBuffer(T * D * 8), DeviceBuffer(T))
    out = lambda: (rew.download(np.float64, T), done.download(np.uint8, T), hand.download(np.uint8, T),
                   obs.download(np.float64, T * D).reshape(T, D), terr.download(np.uint8, T))
    env = pokerl_amd.VecPokerGameEnv(opp, **cfg)
    g = env.game
    o = O.OracleGame(T, N, start, bb, sb, seed=seed, table_id_base=base)
    env.reset(); o.env_reset(None, opp)
    want = []
    for k in range(K):
        a = o.pick_actions(0)
        ro, do, ho, eo = o.env_step(a, opp)
        caps = (eo & 12) != 0                                    # PK_TERR_HAND_CAP | PK_TERR_ENV_CAP: reset like `done`
        m = ((do != 0) | caps).astype(np.uint8)
        if m.any():
            o.env_reset(m, opp)
            eo = eo | (o.errs() * m)                             # the fused call also reports an error raised by the reset that followed
        L.check(lib.pk_env_step_fused_d(g._h, None, 0, opp, 1, rew.ptr, done.ptr, hand.ptr, terr.ptr, obs.ptr), g._h)
        g.sync()
        w = out()
        assert GU.bits_equal(ro, w[0]) and np.array_equal(do, w[1]) and np.array_equal(ho, w[2]) and np.array_equal(eo, w[4]), where
        want.append(w)
    g.close()
    env = pokerl_amd.VecPokerGameEnv(opp, **cfg)
    g = env.game
    env.reset()
    nb = env.set_env_batches(B) if B > 1 else 1
    sub += int(nb > 1)
    count = np.zeros(T, np.int64)
    launches = 0
    while count.min() < K:
        launches += 1
        # (a table whose seat 0 is broke with the game not over plays up to PK_ENV_STEP_CAP = 8 192 opponent steps per env.step: with a
        #  budget of `passes` Game.steps per launch that is ~8 192 / passes launches for ONE env.step -- slow, not stuck)
        assert launches < max(200, 2 * 8192 // passes + 50) * K * nb, where
        env.step_async_d(None, rew.ptr, done.ptr, hand.ptr, terr.ptr, obs.ptr, ready.ptr, max_passes=passes)
        g.sync()
        r = ready.download(np.uint8, T) != 0
        if nb > 1:                                   # one range was launched; outputs are complete inside the DELIVERED range only
            db, de, fresh = env.last_range()
            if fresh:
                continue
            inside = np.zeros(T, bool)
            inside[db:de] = True
            r &= inside
        w = out()
        for t in np.nonzero(r & (count < K))[0]:
            k = count[t]
            for x, y in zip(w, want[k]):
                assert GU.bits_equal(np.ascontiguousarray(x[t:t + 1]), np.ascontiguousarray(y[t:t + 1])), (where, t, k)
        delivered += int((r & (count < K)).sum())
        count[r] += 1
    env.step_async_d(None, rew.ptr, done.ptr, hand.ptr, terr.ptr, obs.ptr, ready.ptr, max_passes=0)
    g.sync()
    g.close()
    if i % 20 == 19:
        print("%d configurations bit-exact so far" % (i + 1), flush=True)
print("fuzz: %d configurations (%d with sub-batches inside the handle), %d env.steps delivered asynchronously, all equal to the synchronous sequences"
      % (n_cfg, sub, delivered))
