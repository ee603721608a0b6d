#!/usr/bin/env python3
"""PokerGameEnv with the reference's list of agents (pokerl/envs/game_env.py:13-18), some of them played by the caller.

    python examples/self_play_opponents.py

Seat 0 is the learner's seat; seat 1 is a host agent written like the reference's agents (agent(state) -> action with a
StateView, pokerl/agents/agent.py:11-13), seat 2 a batched host agent (one call for all tables that wait for it: what a
network forward pass wants), seat 3 the in-kernel random agent.  Whenever a host agent's seat is to act inside env.step /
env.reset the tables concerned yield to the host (pk_env_step_multi_d, ready == 2), everything else stays on the GPU.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import pokerl_amd  # noqa: E402
from pokerl_amd import PokerAgent, PokerMoves, RandomAgent, VecPokerGameEnv  # noqa: E402


class TightAgent(PokerAgent):
    """Reference-style agent: folds weak hole cards pre-flop, calls otherwise."""

    def __call__(self, state):
        ranks = sorted(c.rank for c in state.player_cards)
        if state.turn == 0 and ranks[1] < 9 and ranks[0] != ranks[1] and state.valid_actions[PokerMoves.FOLD]:
            return PokerMoves.FOLD
        for a in (PokerMoves.CALL, PokerMoves.CHECK):
            if state.valid_actions[a]:
                return a
        return PokerMoves.ALL_IN


class BatchedManiac(PokerAgent):
    """Batched agent: gets the dense observation rows [k, PK_OBS_DIM] of all tables that wait for it."""
    batched = True

    def __call__(self, rows, tables):
        valid = rows[:, 3:10] > 0
        raise_half = valid[:, PokerMoves.RAISE_HALF]
        return np.where(raise_half, PokerMoves.RAISE_HALF, PokerMoves.ALL_IN).astype(np.int32)


T = 2048
env = VecPokerGameEnv([TightAgent(), BatchedManiac(), RandomAgent()], num_tables=T, num_players=4)
obs = env.reset()
rng = np.random.default_rng(0)
total, episodes = np.zeros(T), 0
for step in range(100):
    valid = obs[:, 3:10]
    actions = (rng.random(valid.shape) * valid).argmax(axis=1)          # seat 0: uniformly random valid action
    obs, reward, done, hand = env.step(actions)
    total += reward
    if done.any():
        episodes += int(done.sum())
        obs = env.reset(done.astype(np.uint8))
print("seat 0 against [tight host agent, batched host maniac, in-kernel random]: %d episodes, mean reward per step %.3f"
      % (episodes, total.mean() / 100))
env.close()
