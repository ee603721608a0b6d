"""Agents for VecPokerGameEnv's `agents` list -- the vectorised counterpart of the reference's pokerl/agents package
(agent.py:3-15 `PokerAgent`, random.py:4-18 `RandomAgent`).

Two kinds of entry are accepted wherever the reference takes an agent callable (envs/game_env.py:13-18):

  * IN-KERNEL agents: `RandomAgent()`, `AllInAgent()`, `CallAgent()` (or the bare `Policy` value).  The seat is played
    inside the HIP kernel under the RNG / agent spec; no host round trip.
  * HOST agents: any other callable.  The seat is PK_POLICY_EXTERNAL: whenever it is to act the tables concerned yield
    to the host, which calls the agent.  A plain callable gets what the reference gives it -- one `StateView` per call
    (agent.py:11-13) -- and is called once per table; a callable with `batched = True` gets the dense observation rows
    of all those tables at once, `agent(obs[k, PK_OBS_DIM], tables[k]) -> actions[k]`.
"""
import numpy as np

from .enums import Policy


class PokerAgent:
    """Interface of the reference's agents (agents/agent.py:3-15): `__call__(state) -> action`."""
    batched = False

    def __call__(self, state):
        raise NotImplementedError


class _KernelAgent(PokerAgent):
    """An agent the kernels play themselves; calling it on the host applies the same RULE to a StateView (the random
    agent then draws from numpy's generator, as the reference's does -- only in-kernel play follows the Philox spec)."""
    policy = None


class RandomAgent(_KernelAgent):
    """agents/random.py:12-18: uniform over the valid actions."""
    policy = Policy.RANDOM

    def __call__(self, state):
        vu = np.asarray(state.valid_actions, np.float64)
        return int(np.random.choice(len(vu), p=vu / np.sum(vu)))


class AllInAgent(_KernelAgent):
    policy = Policy.ALL_IN

    def __call__(self, state):
        return 6


class CallAgent(_KernelAgent):
    """CALL if valid, else CHECK if valid, else ALL_IN."""
    policy = Policy.CALL

    def __call__(self, state):
        vu = state.valid_actions
        return 2 if vu[2] else (1 if vu[1] else 6)


def kernel_policy(agent):
    """Policy value if `agent` is played in-kernel, else None (a host agent)."""
    if isinstance(agent, _KernelAgent):
        return int(agent.policy)
    if isinstance(agent, (int, np.integer)) and not isinstance(agent, bool):
        return int(Policy(int(agent)))
    return None
