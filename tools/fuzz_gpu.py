#!/usr/bin/env python3
"""One-off confidence run (GPU box): tests/test_hip_parity.py::test_random_configurations_vs_oracle over many more seeded
odd configurations (every N, zero / fractional / oversized blinds, per-seat stacks 0.5 .. 1e6, any table-id base and
dealer): fused rollout (deferred launches of mixed lengths) + lockstep steps against the CPU oracle.
usage: python tools/fuzz_gpu.py [configs] [seed]"""
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

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
only = int(sys.argv[3]) if len(sys.argv) > 3 else None   # run this configuration index alone
stacks = [0.5, 1, 2, 3, 5, 10, 37.5, 100, 1000, 1e6]
blinds = [0, 0.25, 0.5, 1, 2, 3, 7.5, 40]


def same(a, b, where):
    for k in GU.SNAP_FIELDS:
        assert GU.bits_equal(a[k], b[k]), (where, k)


steps = 0
for i in range(n_cfg):
    N = 2 + i % 15   # 2 ... 16 seats
    start = [rng.choice(stacks) for _ in range(N)] if rng.random() < 0.5 else rng.choice(stacks)
    bb, sb = rng.choice(blinds), rng.choice(blinds)
    policy = 1 if rng.random() < 0.25 else 0
    seed, base, dealer = rng.getrandbits(63), rng.getrandbits(32) & 0xFFFFF000, rng.randrange(N)
    T = rng.choice([65, 128, 300, 1000, 4097])
    K = rng.choice([60, 120, 333])
    if (only is not None and i != only) or i < int(os.environ.get("PK_FUZZ_FROM", "0")):
        continue
    if os.environ.get("PK_FUZZ_VERBOSE"):
        print("config %d: T=%d N=%d start=%s bb=%s sb=%s policy=%d K=%d" % (i, T, N, start, bb, sb, policy, K), flush=True)
    # blinds far above the stacks make single steps roll thousands of hands: the single-thread oracle would need a quarter of an
    # hour for 4 097 such tables (seed 77, configuration 85): probe 16 tables and shrink the batch if so
    probe = O.OracleGame(16, N, start, bb, sb, seed=seed, table_id_base=base)
    probe.reset(dealer=dealer)
    t_probe = time.time()
    probe.rollout(K, policy, True)
    if (time.time() - t_probe) * T / 16 > 30:
        T = 65
    o = O.OracleGame(T, N, start, bb, sb, seed=seed, table_id_base=base)
    h = HB(T, N, start, bb, sb, seed=seed, table_id_base=base)
    o.reset(dealer=dealer); h.reset(dealer=dealer)
    where = "cfg %d: T=%d N=%d start=%s bb=%s sb=%s policy=%d K=%d" % (i, T, N, start, bb, sb, policy, K)
    t_cfg = time.time()
    co, _ = o.rollout(K, policy, True)
    t_oracle = time.time() - t_cfg
    k1 = K // 3
    h.g.rollout(k1, policy, True, True, counters=False)          # deferred launches of mixed lengths ...
    h.g.rollout(K - k1 - 7, policy, True, True, counters=False)
    ch = h.rollout(7, policy, True)                              # ... and a completing one
    assert co.tolist() == ch.tolist(), where
    same(o.snapshot(), h.snapshot(), where + " rollout")
    for s in range(6):
        a = o.pick_actions(policy)
        fo, eo = o.step(a)
        fh, eh = h.step(a)
        assert np.array_equal(fo, fh) and np.array_equal(eo, eh), where
        bad = ((fo & 1) | (eo != 0)).astype(np.uint8)
        if bad.any():
            o.reset(mask=bad); h.reset(mask=bad)
    same(o.snapshot(), h.snapshot(), where + " lockstep")
    steps += T * (K + 6)
    if time.time() - t_cfg > 20:
        print("slow: %s took %.0f s (%.0f s of it the CPU oracle's rollout)" % (where, time.time() - t_cfg, t_oracle), flush=True)
    h.g.close()
    if i % 50 == 49:
        print("%d configurations bit-exact so far" % (i + 1), flush=True)
print("fuzz: %d configurations, %d env-steps, all bit-exact vs the oracle" % (n_cfg, steps))
