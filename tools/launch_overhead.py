#!/usr/bin/env python3
"""Diagnostic (GPU box): fixed vs marginal cost of a fused rollout launch.  usage: python tools/launch_overhead.py [N] [T]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pokerl_amd  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
T = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
g = pokerl_amd.VecGame(T, num_players=N)
g.reset()
g.rollout(2048, 0)
print("N=%d T=%d" % (N, T))
ms, _ = g.time_rollout(0, 0, True, True, 2000)
print("empty launch (load + store all tables, no step): %.2f us" % (ms * 1e3))
for endk in (48, 1):
    g.set_tuning(0, endk)
    for K in (1, 2, 5, 10, 20, 40, 80, 160, 512, 2048):
        reps = max(2, 8192 // K)
        g.time_rollout(K, 0, True, True, max(2, reps // 4))
        ms, c = g.time_rollout(K, 0, True, True, reps)
        print("endk=%2d K=%5d reps=%5d  %9.2f us/launch  %7.3f us/step  %6.2f G env-steps/s" % (
            endk, K, reps, ms * 1e3, ms * 1e3 / K, T * K / ms / 1e6), flush=True)
