#!/usr/bin/env python3
"""Soak parity (run on the GPU box; not part of the pytest suite): the fused rollout at full batch size for thousands of
steps against the CPU oracle run on all host cores (tables split into per-thread chunks by global table id)."""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402
import golden_util as GU  # noqa: E402
from hip_backend import HipBackend  # noqa: E402
from oracle import loader as O  # noqa: E402

SCALE = int(os.environ.get("PK_SOAK_SCALE", "1"))   # PK_SOAK_SCALE=8: 5.8 G env-steps, about four minutes
CASES = [(65536, 6, 0, 4096 * SCALE), (65536, 9, 1, 2048 * SCALE), (65536, 2, 0, 4096 * SCALE), (16384, 10, 0, 3072 * SCALE),
         (16384, 13, 0, 2048 * SCALE), (8192, 15, 1, 2048 * SCALE), (8192, 16, 0, 2048 * SCALE), (8192, 16, 1, 1024 * SCALE)]
THREADS = 16
for T, N, policy, K in CASES:
    t0 = time.time()
    h = HipBackend(T, N)
    h.reset()
    # asynchronous launches of mixed lengths that defer their stragglers (State::owed); one completing call at the end
    for _ in range(K // 512):
        for _ in range(16):
            h.g.rollout(20, policy, True, True, counters=False)
        h.g.rollout(192, policy, True, True, counters=False)
    ch = h.rollout(0, policy, True, fused=True)
    chunk = T // THREADS
    games = [O.OracleGame(chunk, N, table_id_base=i * chunk) for i in range(THREADS)]

    def work(g):
        g.reset()
        c, e = g.rollout(K, policy, True)
        return c, g.snapshot()

    with ThreadPoolExecutor(THREADS) as ex:
        parts = list(ex.map(work, games))
    co = sum(p[0] for p in parts)
    assert co.tolist() == ch.tolist(), (co, ch)
    snap = h.snapshot()
    for k in GU.SNAP_FIELDS:
        exp = np.concatenate([p[1][k] for p in parts])
        assert GU.bits_equal(exp, snap[k]), (T, N, k)
    print("soak T=%d N=%d policy=%d K=%d: %d env-steps, %d hands, %d evals, %d games bit-exact (%.0f s)"
          % (T, N, policy, K, ch[0], ch[1], ch[2], ch[3], time.time() - t0), flush=True)
