#!/usr/bin/env python3
"""The reference's examples/random_game.py (4 random agents, 1000/40/20) for 4 096 tables at once on one MI355X.

    python examples/random_rollout.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import pokerl_amd  # noqa: E402

game = pokerl_amd.VecGame(4096, num_players=4, start_credits=1000, big_blind=40, small_blind=20)
game.reset()

# (a) the reference loop, vectorised: host picks a uniformly random valid action per table
rng = np.random.default_rng(0)
done = np.zeros(game.num_tables, bool)
for _ in range(50):
    onehot, _ = game.get_valid_actions()                       # [T, 7]
    actions = (rng.random(onehot.shape) * onehot).argmax(axis=1)
    over, hand, turn = game.step(actions)
    done |= over
    if over.any():
        game.reset(mask=over.astype(np.uint8))
print("host-driven: %d tables finished a game in 50 steps; mean pot now %.1f" % (done.sum(), game.pot.mean()))

# (b) the same workload with the agents in-kernel (what bench.py measures)
stats = game.rollout(2000, policy=pokerl_amd.Policy.RANDOM, auto_reset=True)
print("in-kernel: %(steps)d env-steps, %(hands)d hands, %(evals)d showdown evaluations, %(games)d games" % stats)
rank, kick = game.hand_rankings
names = pokerl_amd.HandRanking.as_string
print("last showdown of table 0:", [names[r] for r in rank[0]])
