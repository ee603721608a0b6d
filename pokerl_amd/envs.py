"""VecPokerGameEnv: the reference's PokerGameEnv (pokerl/envs/game_env.py:6-53) for T tables on one MI355X.

Seat 0 is the controlled seat (game_env.py:17-18); the other seats are played in-kernel by a synthetic policy
(enums.Policy) -- the vectorised stand-in for the reference's list of agent callables.
"""
import numpy as np

from . import _lib as L
from .enums import Policy
from .game import VecGame


class VecPokerGameEnv:
    def __init__(self, agents=Policy.RANDOM, num_tables=1, **game_config):
        self.game = VecGame(num_tables, **game_config)  # game_env.py:16
        self.opp_policy = int(agents)
        self.player_agent = 0                           # game_env.py:18

    @property
    def num_tables(self):
        return self.game.num_tables

    def reset(self, mask=None):
        """game_env.py:20-29 on all tables (or where mask != 0); returns the observation rows (StateView fields)."""
        g = self.game
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        L.check(g._lib.pk_env_reset(g._h, L.ptr(m), self.opp_policy), g._h)
        return g.observations

    def step(self, actions):
        """game_env.py:31-53: returns (obs, reward f64[T], done bool[T], hand bool[T]) -- the reference's 4-tuple."""
        g = self.game
        a = g._actions(actions)
        T = g.num_tables
        reward = np.zeros(T, np.float64)
        done = np.zeros(T, np.uint8)
        hand = np.zeros(T, np.uint8)
        terr = np.zeros(T, np.uint8)
        rc = g._lib.pk_env_step(g._h, L.ptr(a), self.opp_policy, L.ptr(reward), L.ptr(done), L.ptr(hand), L.ptr(terr))
        L.check(rc, g._h, allow_table_errors=True)
        if (terr & L.TERR_INVALID_ACTION).any():
            t = int(np.argmax(terr & L.TERR_INVALID_ACTION))
            raise ValueError('Player 0 invalid move (table %d); tables with a valid action were stepped' % t)
        if terr.any():
            raise L.PokerlHipError('table error bits %s' % np.unique(terr))
        return g.observations, reward, done != 0, hand != 0
