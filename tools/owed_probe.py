#!/usr/bin/env python3
"""Diagnostic (GPU box): distribution of the steps tables still owe after n deferred launches of K steps."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import pokerl_amd  # noqa: E402

T, N, K = 65536, 6, int(sys.argv[1]) if len(sys.argv) > 1 else 20
for endk in (64, 62, 56, 48):
    g = pokerl_amd.VecGame(T, num_players=N)
    g.reset()
    g.rollout(512, 0)
    g.set_tuning(0, endk)
    done = 0
    for n in (1, 4, 16, 64, 256):
        t0 = time.perf_counter()
        for _ in range(n - done):
            g.rollout(K, 0, True, True, counters=False)
        done = n
        ow = g.owed
        dt = time.perf_counter() - t0
        w = ow.reshape(-1, 64)
        print("endk=%d after %4d launches of %d: owed mean %.1f  max %d  per-wave min: mean %.1f  per-wave max: mean %.1f  (%.2f ms)" % (
            endk, n, K, ow.mean(), ow.max(), w.min(axis=1).mean(), w.max(axis=1).mean(), dt * 1e3), flush=True)
    t0 = time.perf_counter()
    g.sync()
    print("   flush: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
    g.close()
