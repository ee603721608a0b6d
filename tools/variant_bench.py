#!/usr/bin/env python3
"""Diagnostic (GPU box): G env-steps/s of one library build (POKERL_HIP_LIB) at K = 2048 and K = 20."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pokerl_amd  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
policy = int(sys.argv[2]) if len(sys.argv) > 2 else 0
T = 65536
g = pokerl_amd.VecGame(T, num_players=N)
g.reset()
g.rollout(2048, policy)
out = []
for K, reps in ((2048, 6), (20, 400)):
    g.time_rollout(K, policy, True, True, max(2, reps // 3))
    ms, _ = g.time_rollout(K, policy, True, True, reps)
    out.append("K=%d %.2f G" % (K, T * K / ms / 1e6))
print(os.path.basename(os.environ.get("POKERL_HIP_LIB", "default")), "N=%d policy=%d:" % (N, policy), "  ".join(out))
