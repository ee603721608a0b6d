#!/usr/bin/env python3
"""A host-side learner's loop at PCIe line rate: PokerGameEnv.step over two batches of tables whose copies and launches
overlap (VecPokerGameEnvPool.step_pipelined = VecPokerGameEnv.send / .recv = pk_env_step_begin / pk_env_step_end), outputs in
PINNED arrays the pool keeps and reuses, observations as PACKED rows (state_view.packed_dtype: 168 B per table at six seats
against 280 for the f64 rows; the money fields stay binary64).  The reference's loop (examples/q_learning.py:49-92) calls
env.step(action) per table and reads a StateView; here `obs` is a structured array with the same fields for every table.

    python examples/host_pinned_pipeline.py [tables per batch] [steps]
On a node, VecPokerGameEnvPool(devices=[0, 1, ..., 7]) puts one batch on each GPU (one Python thread per device)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pokerl_amd
from pokerl_amd import Policy, PokerMoves

T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
pool = pokerl_amd.VecPokerGameEnvPool(Policy.RANDOM, num_tables=2 * T, num_batches=2, num_players=6)
pool.reset()
# the first observation of every table, through the same packed rows
obs = [e.game.observations_packed_of(None) for e in pool.envs]


def policy(rows):
    """A host policy on the packed rows: call when allowed, else check, else fold -- field access is zero-copy."""
    valid = rows['valid_bits']
    return np.where(valid & (1 << PokerMoves.CALL), PokerMoves.CALL,
                    np.where(valid & (1 << PokerMoves.CHECK), PokerMoves.CHECK, PokerMoves.FOLD)).astype(np.int32)


t0 = time.perf_counter()
total_reward, episodes = 0.0, 0
for _ in range(steps):
    actions = np.concatenate([policy(o) for o in obs])
    outs = pool.step_pipelined(actions, obs='packed', auto_reset=True)   # views of pinned arrays, overwritten by the next call
    obs = [o[0] for o in outs]
    total_reward += sum(float(o[1].sum()) for o in outs)
    episodes += sum(int(o[2].sum()) for o in outs)
dt = time.perf_counter() - t0
print("%d tables x %d steps: %.1f M env.step/s through host arrays; %d episodes ended, mean reward per step %.4f"
      % (2 * T, steps, 2 * T * steps / dt / 1e6, episodes, total_reward / (2 * T * steps)))
pool.close()
