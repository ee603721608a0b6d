#!/usr/bin/env python3
"""Code written against the reference's single-table objects runs on pokerl_amd.Game / PokerGameEnv unchanged: one
table with the reference's own shapes (scalars, per-seat vectors, Card lists, StateView), every value produced by the
HIP kernels.  (For throughput use VecGame / VecPokerGameEnv: one table per launch is latency-bound.)

    python examples/single_table.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from pokerl_amd import CallAgent, Game, HandRanking, PokerGameEnv, PokerMoves, RandomAgent, eval_hand  # noqa: E402

# ---- a whole game between four uniformly random players (the configuration of the reference's examples/random_game.py)
game = Game(num_players=4, start_credits=1000, big_blind=40, small_blind=20)
game.reset()
rng = np.random.default_rng(7)
steps = 0
while not game.game_over:
    valid, _ = game.get_valid_actions()
    over, hand_over, turn_over = game.step(int(rng.choice(len(valid), p=valid / valid.sum())))
    steps += 1
    if hand_over and not over:
        print("hand %3d over after step %4d: payoffs %s" % (game.hand - 1, steps, np.round(game.payoffs, 1)))
print("game over after %d steps, credits %s" % (steps, np.round(game.credits, 1)))
game.close()

# ---- the RL-facing wrapper: seat 0 against [calling station, random, random]
env = PokerGameEnv([CallAgent(), RandomAgent(), RandomAgent()], num_players=4)
state = env.reset()
total = 0.0
for _ in range(200):
    rank, _ = eval_hand(state.player_hand)                      # the feature of the reference's examples/q_learning.py
    valid = state.valid_actions
    action = PokerMoves.CALL if (rank <= HandRanking.PAIR and valid[PokerMoves.CALL]) else int(np.flatnonzero(valid)[0])
    state, reward, done, hand_over = env.step(int(action))
    total += reward
    if done:
        state = env.reset()
print("seat 0 (calls with a pair or better, else folds / first valid move): total reward %.1f over 200 steps" % total)
env.close()
