#!/usr/bin/env python3
"""Diagnostic (GPU box): env-steps/s of deferred K-step launches over the (park, endk) tuning grid.
usage: python tools/tune_sweep.py [K] [N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pokerl_amd  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = int(sys.argv[2]) if len(sys.argv) > 2 else 6
policy = int(sys.argv[3]) if len(sys.argv) > 3 else 0
T = 65536
g = pokerl_amd.VecGame(T, num_players=N)
g.reset()
g.rollout(1024, policy)
reps = max(4, 8192 // K)
print("K=%d N=%d policy=%d: G env-steps/s" % (K, N, policy))
endks = (1, 32, 40, 44, 48, 52, 56, 60, 62, 64)
print("park\\endk " + " ".join("%6d" % e for e in endks))
for park in (24, 28, 32, 36, 40, 44, 48):
    row = []
    for endk in endks:
        g.set_tuning(park, endk)
        g.time_rollout(K, policy, True, True, max(2, reps // 4))
        ms, _ = g.time_rollout(K, policy, True, True, reps)
        row.append(T * K / ms / 1e6)
    print("%9d " % park + " ".join("%6.2f" % x for x in row), flush=True)
