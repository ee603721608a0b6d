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

    def close(self):
        """Frees the device buffers of step_async and the game's handle."""
        for b in getattr(self, '_async_buf', {}).values():
            b.free()
        self._async_buf = {}
        self.game.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, mask=None):
        """game_env.py:20-29 on all tables (or where mask != 0); returns the observation rows (StateView fields)."""
        g = self.game
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        if m is not None and m.shape != (g.num_tables,):
            raise ValueError('mask must have shape (num_tables,)')
        L.check(g._lib.pk_env_reset(g._h, L.ptr(m), self.opp_policy), g._h)
        return g.observations

    def step(self, actions, strict=True):
        """game_env.py:31-53: returns (obs, reward f64[T], done bool[T], hand bool[T]) -- the reference's 4-tuple.

        strict=True mirrors the reference for the batch: seat 0's action is checked against the valid mask first and a
        ValueError (game.py:649-651) is raised BEFORE any table is mutated.  strict=False steps the tables whose action
        is valid, leaves the others untouched and returns per-table error bits as a 5th array."""
        g = self.game
        a = g._actions(actions)
        T = g.num_tables
        if strict:
            valid = g.get_valid_actions()[0]
            ok = (a >= 0) & (a < valid.shape[1])
            ok[ok] = valid[np.nonzero(ok)[0], a[ok]] != 0
            if not ok.all():
                t = int(np.argmin(ok))
                raise ValueError('Player %d invalid move: `%d` (table %d)' % (int(g.active_player[t]), int(a[t]), t))
        reward = np.zeros(T, np.float64)
        done = np.zeros(T, np.uint8)
        hand = np.zeros(T, np.uint8)
        terr = np.zeros(T, np.uint8)
        rc = g._lib.pk_env_step(g._h, L.ptr(a), self.opp_policy, L.ptr(reward), L.ptr(done), L.ptr(hand), L.ptr(terr))
        L.check(rc, g._h, allow_table_errors=True)
        if not strict:
            return g.observations, reward, done != 0, hand != 0, terr
        if terr.any():
            raise L.PokerlHipError('table error bits %s' % np.unique(terr))
        return g.observations, reward, done != 0, hand != 0

    def step_async_d(self, actions_d, reward_d, done_d, hand_d, terr_d, obs_d, ready_d, max_passes=8, seat0_policy=Policy.RANDOM,
                     auto_reset=True):
        """pk_env_step_async_d on DEVICE pointers (ints / c_void_p; actions_d None = seat 0 played by `seat0_policy`
        in-kernel): a bounded launch that delivers the tables whose PokerGameEnv.step returned (ready_d[t] = 1) and keeps
        the others in flight.  max_passes <= 0 drains.  See include/pokerl_hip.h."""
        g = self.game
        vp = lambda x: x if x is None or isinstance(x, L.C.c_void_p) else L.C.c_void_p(int(x))
        L.check(g._lib.pk_env_step_async_d(g._h, vp(actions_d), int(seat0_policy), self.opp_policy, 1 if auto_reset else 0,
                                           int(max_passes), vp(reward_d), vp(done_d), vp(hand_d), vp(terr_d), vp(obs_d),
                                           vp(ready_d)), g._h)

    def step_async(self, actions=None, max_passes=8, seat0_policy=Policy.RANDOM, auto_reset=True):
        """Host-array convenience over step_async_d: returns (ready bool[T], obs f64[T, D], reward f64[T], done bool[T],
        hand bool[T], terr u8[T]); rows are meaningful where `ready`.  `actions` (int32[T], None = seat 0 played by
        `seat0_policy` in-kernel) is read for the tables that were ready after the previous call only."""
        from .hipmem import DeviceBuffer
        g = self.game
        T, D = g.num_tables, 17 + 3 * g.num_players
        if not getattr(self, '_async_buf', None):
            dev = g.device
            self._async_buf = dict(act=DeviceBuffer(T * 4, dev), rew=DeviceBuffer(T * 8, dev), done=DeviceBuffer(T, dev),
                                   hand=DeviceBuffer(T, dev), terr=DeviceBuffer(T, dev), obs=DeviceBuffer(T * D * 8, dev),
                                   ready=DeviceBuffer(T, dev))
        b = self._async_buf
        if actions is not None:
            b['act'].upload(g._actions(actions))
        self.step_async_d(b['act'].ptr if actions is not None else None, b['rew'].ptr, b['done'].ptr, b['hand'].ptr,
                          b['terr'].ptr, b['obs'].ptr, b['ready'].ptr, max_passes, seat0_policy, auto_reset)
        g.sync()
        return (b['ready'].download(np.uint8, T) != 0, b['obs'].download(np.float64, T * D).reshape(T, D),
                b['rew'].download(np.float64, T), b['done'].download(np.uint8, T) != 0,
                b['hand'].download(np.uint8, T) != 0, b['terr'].download(np.uint8, T))
