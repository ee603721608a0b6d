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
        """`PokerGameEnv(agents, **game_config)` (game_env.py:13-18) for num_tables tables.  agents: the reference's list
        with one agent per OPPONENT seat (seat 1 first) -- in-kernel agents (pokerl_amd.agents.RandomAgent / AllInAgent /
        CallAgent or a Policy value) and / or host callables, see pokerl_amd/agents.py -- or, as a shorthand, ONE Policy
        that every opponent plays."""
        self.game = VecGame(num_tables, **game_config)  # game_env.py:16
        self.player_agent = 0                           # game_env.py:18
        self._pin, self._sent, self._in_flight = {}, None, False    # send() / recv(): pinned arrays, what send() asked for, a send awaits its recv
        self.set_agents(agents)

    def set_agents(self, agents):
        from .agents import kernel_policy
        n = self.game.num_players
        if isinstance(agents, (list, tuple)):
            if len(agents) != n - 1:
                raise ValueError('agents: one per opponent seat, %d expected' % (n - 1))
            agents = list(agents)
        else:
            agents = [Policy(int(agents))] * (n - 1)
        self.agents = [None, *agents]                   # game_env.py:17
        pols = [kernel_policy(a) for a in agents]
        self._external = [p + 1 for p, pol in enumerate(pols) if pol is None]
        for p in self._external:
            if not callable(self.agents[p]):
                raise TypeError('agent of seat %d is neither an in-kernel agent / Policy nor callable' % p)
        # seat 0 is the caller's (PK_POLICY_EXTERNAL = 15); an opponent is its in-kernel policy or external as well
        self.seat_policies = 15 | sum((15 if pol is None else pol) << (4 * (p + 1)) for p, pol in enumerate(pols))
        uniform = not self._external and len(set(pols)) == 1
        self._opp_policy = pols[0] if uniform else None  # one policy for every opponent: the single-launch entry points

    @property
    def opp_policy(self):
        """The one in-kernel policy every opponent plays, or None (per-seat agents: the pk_env_step_multi_d path)."""
        return self._opp_policy

    @opp_policy.setter
    def opp_policy(self, agents):
        self.set_agents(agents)

    @property
    def num_tables(self):
        return self.game.num_tables

    def close(self):
        """Frees the device buffers of step_async / the multi-agent path and the game's handle."""
        for name in ('_async_buf', '_multi_buf'):
            for b in (getattr(self, name, None) or {}).values():
                b.free()
            setattr(self, name, {})
        self.game.close()
        for b in (getattr(self, '_pin', None) or {}).values():
            b.free()
        self._pin = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ per-seat agents (pk_env_step_multi_d)
    def _call_agent(self, agent, rows, tables):
        if getattr(agent, 'batched', False):
            return np.asarray(agent(rows, tables), np.int32)
        from .state_view import StateView
        n = self.game.num_players
        return np.array([int(agent(StateView(row, n))) for row in rows], np.int32)   # agent(state), agents/agent.py:11-13

    def _multi_run(self, actions=None, reset_mask=None):
        """One PokerGameEnv.step (or .reset where reset_mask != 0) on every table with one agent per seat: launches run
        until every table has returned or waits for a host agent; the host agents are called for the tables that wait
        for them, and so on until every table has returned.  Returns (reward, done, hand, terr)."""
        from .hipmem import DeviceBuffer
        g = self.game
        T, D, lib, h = g.num_tables, 17 + 3 * g.num_players, g._lib, g._h
        if not getattr(self, '_multi_buf', None):
            dev = g.device
            self._multi_buf = dict(act=DeviceBuffer(T * 4, dev), reset=DeviceBuffer(T, dev), rew=DeviceBuffer(T * 8, dev),
                                   done=DeviceBuffer(T, dev), hand=DeviceBuffer(T, dev), terr=DeviceBuffer(T, dev),
                                   obs=DeviceBuffer(T * D * 8, dev), who=DeviceBuffer(T, dev), ready=DeviceBuffer(T, dev))
        b = self._multi_buf
        acts = np.full(T, L.ACTION_SKIP, np.int32) if actions is None else actions
        if reset_mask is not None:
            b['reset'].upload(np.ascontiguousarray(reset_mask, np.uint8))
        reset_ptr = b['reset'].ptr if reset_mask is not None else None
        try:
            while True:
                b['act'].upload(acts)
                L.check(lib.pk_env_step_multi_d(h, b['act'].ptr, reset_ptr, self.seat_policies, 0, 0, b['rew'].ptr, b['done'].ptr,
                                                b['hand'].ptr, b['terr'].ptr, b['obs'].ptr, b['who'].ptr, b['ready'].ptr), h)
                g.sync()
                ready = b['ready'].download(np.uint8, T)
                waiting = ready == 2
                if not waiting.any():
                    break
                who = b['who'].download(np.uint8, T)
                bad = waiting & (b['terr'].download(np.uint8, T) != 0)
                if bad.any():      # a host agent returned an action Game.step refuses (game.py:649-651)
                    t = int(np.argmax(bad))
                    raise ValueError('Player %d invalid move: `%d` (table %d)' % (int(who[t]), int(acts[t]), t))
                obs = b['obs'].download(np.float64, T * D).reshape(T, D)
                acts = np.full(T, L.ACTION_SKIP, np.int32)      # tables that have returned are left alone
                for seat in self._external:
                    idx = np.nonzero(waiting & (who == seat))[0]
                    if len(idx):
                        acts[idx] = self._call_agent(self.agents[seat], obs[idx], idx)
                reset_ptr = None
        except BaseException:
            # (a host agent's invalid action, or an exception it raised: the env calls still in flight are abandoned -- those
            #  tables stand between two Game.steps of an unfinished PokerGameEnv.step, whose reward / done are lost: reset() the
            #  env before using it again.  The original exception must not be masked by a failure of the clean-up.)
            try:
                L.check(lib.pk_env_end_multi_d(h), h)
            except Exception:
                pass
            raise
        L.check(lib.pk_env_end_multi_d(h), h)         # nothing is left in flight: getters and the other entry points work again
        return (b['rew'].download(np.float64, T), b['done'].download(np.uint8, T), b['hand'].download(np.uint8, T),
                b['terr'].download(np.uint8, T))

    def reset(self, mask=None):
        """game_env.py:20-29 on all tables (or where mask != 0); returns the observation rows (StateView fields)."""
        g = self.game
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        if m is not None and m.shape != (g.num_tables,):
            raise ValueError('mask must have shape (num_tables,)')
        if self._opp_policy is None:
            self._multi_run(reset_mask=np.ones(g.num_tables, np.uint8) if m is None else m)
        else:
            L.check(g._lib.pk_env_reset(g._h, L.ptr(m), self._opp_policy), g._h)
        return g.observations

    def check_actions(self, actions, table_offset=0):
        """Raises the reference's ValueError (game.py:649-651) if any table's action is not valid for its active player;
        nothing is mutated.  Checked on the device (pk_check_actions): one upload, one small kernel, four bytes back."""
        self.game.check_actions(actions, table_offset)

    # ------------------------------------------------------------------ host-array fast path (pinned buffers, two halves)
    def send(self, actions, obs='packed', auto_reset=False, strict=False):
        """First half of a PokerGameEnv.step through host arrays at PCIe line rate (pk_env_step_begin): uploads the actions,
        launches the step and queues the copies of reward / done / hand / terr and of the observation rows -- obs='packed'
        (state_view.packed_dtype rows, 168 B per table at six seats), 'dense' (f64 [T, PK_OBS_DIM], 280 B) or None -- into
        PINNED arrays this env keeps and reuses (so nothing here blocks, and another env's send / the caller's own work
        overlaps with the copies).  recv() completes it.  Needs ONE in-kernel policy for all opponents.
        Between send() and recv() the env must not be used: a second send(), step(), reset() or a getter raises (PK_E_BUSY in the
        library; here a RuntimeError before anything is touched), because the views an earlier recv() handed out are being written.
        Invalid actions: strict=True checks them first and raises the reference's ValueError before any table is mutated (as step()
        does); with the default strict=False a table whose action is invalid is left UNSTEPPED and recv()'s terr[t] says
        PK_TERR_INVALID_ACTION (1) -- the caller must look at terr (step_pipelined does not)."""
        from .hipmem import pinned_empty
        from .state_view import packed_dtype
        g = self.game
        if self._opp_policy is None:
            raise ValueError('send / recv need ONE in-kernel policy for all opponents; per-seat / host agents: step()')
        if obs not in ('packed', 'dense', None):
            raise ValueError("obs: 'packed', 'dense' or None")
        if self._in_flight:
            raise RuntimeError('send(): the previous send() has not been completed by recv()')
        T, n = g.num_tables, g.num_players
        if not self._pin:
            self._pin = dict(act=pinned_empty(T, np.int32), rew=pinned_empty(T, np.float64), done=pinned_empty(T, np.uint8),
                             hand=pinned_empty(T, np.uint8), terr=pinned_empty(T, np.uint8))
        b = self._pin
        if obs == 'packed' and 'packed' not in b:
            b['packed'] = pinned_empty(T, packed_dtype(n))
        if obs == 'dense' and 'dense' not in b:
            b['dense'] = pinned_empty((T, 17 + 3 * n), np.float64)
        a = g._actions(actions)
        if strict:
            self.check_actions(a)
        b['act'].array[:] = a
        L.check(g._lib.pk_env_step_begin(g._h, L.ptr(b['act'].array), self._opp_policy, 1 if auto_reset else 0, L.ptr(b['rew'].array),
                                         L.ptr(b['done'].array), L.ptr(b['hand'].array), L.ptr(b['terr'].array),
                                         L.ptr(b['dense'].array) if obs == 'dense' else None,
                                         L.ptr(b['packed'].array) if obs == 'packed' else None), g._h)
        self._sent, self._in_flight = obs, True

    def recv(self):
        """Second half: waits for the copies send() queued and returns (obs, reward, done, hand, terr) -- VIEWS of the env's
        pinned arrays, overwritten by the next send(); copy what must outlive it.  obs is None / packed rows / dense rows.
        terr[t] != 0: that table's step did not happen as asked (1 = invalid action, table untouched; see send())."""
        if not self._in_flight:
            raise RuntimeError('recv() without a send()')
        g, b = self.game, self._pin
        self._in_flight = False
        L.check(g._lib.pk_env_step_end(g._h), g._h)
        obs = None if self._sent is None else b[self._sent].array
        return obs, b['rew'].array, b['done'].array.view(np.bool_), b['hand'].array.view(np.bool_), b['terr'].array

    def step(self, actions, strict=True):
        """game_env.py:31-53: returns (obs, reward f64[T], done bool[T], hand bool[T]) -- the reference's 4-tuple.

        strict=True mirrors the reference for the batch: seat 0's action is checked against the valid mask first and a
        ValueError (game.py:649-651) is raised BEFORE any table is mutated.  strict=False steps the tables whose action
        is valid, leaves the others untouched and returns per-table error bits as a 5th array."""
        g = self.game
        a = g._actions(actions)
        T = g.num_tables
        if strict:
            self.check_actions(a)
        if self._opp_policy is None:
            reward, done, hand, terr = self._multi_run(actions=a)
        else:
            reward = np.zeros(T, np.float64)
            done = np.zeros(T, np.uint8)
            hand = np.zeros(T, np.uint8)
            terr = np.zeros(T, np.uint8)
            rc = g._lib.pk_env_step(g._h, L.ptr(a), self._opp_policy, L.ptr(reward), L.ptr(done), L.ptr(hand), L.ptr(terr))
            L.check(rc, g._h, allow_table_errors=True)
        if not strict:
            return g.observations, reward, done != 0, hand != 0, terr
        if terr.any():
            raise L.PokerlHipError('table error bits %s' % np.unique(terr))
        return g.observations, reward, done != 0, hand != 0

    def step_multi_d(self, actions_d, reset_d, reward_d, done_d, hand_d, terr_d, obs_d, who_d, ready_d, max_passes=0, auto_reset=True):
        """pk_env_step_multi_d on DEVICE pointers with this env's per-seat agents (self.seat_policies): tables yield
        (ready 2, who = seat) where a host agent's seat is to act; see include/pokerl_hip.h.  end_multi() leaves the mode."""
        g = self.game
        vp = lambda x: x if x is None or isinstance(x, L.C.c_void_p) else L.C.c_void_p(int(x))
        L.check(g._lib.pk_env_step_multi_d(g._h, vp(actions_d), vp(reset_d), self.seat_policies, 1 if auto_reset else 0, int(max_passes),
                                           vp(reward_d), vp(done_d), vp(hand_d), vp(terr_d), vp(obs_d), vp(who_d), vp(ready_d)), g._h)

    def end_multi(self):
        L.check(self.game._lib.pk_env_end_multi_d(self.game._h), self.game._h)

    def set_env_batches(self, batches):
        """pk_set_env_batches: split this env's tables into `batches` contiguous ranges with internal streams; every bounded
        step_async_d call then launches one range and delivers the range launched longest ago (see last_range).  Returns
        the number of ranges actually made (fewer for a small batch)."""
        g = self.game
        L.check(g._lib.pk_set_env_batches(g._h, int(batches)), g._h)
        return -(-g.num_tables // max(1, self.last_range()[1]))

    def last_range(self):
        """(begin, end, fresh) of the tables the last step_async_d call delivered: outputs are complete inside [begin, end);
        fresh: that range has not been stepped yet (nothing written, every table awaits its first action)."""
        g = self.game
        b, e, f = L.C.c_int(0), L.C.c_int(0), L.C.c_int(0)
        L.check(g._lib.pk_env_last_range(g._h, L.C.byref(b), L.C.byref(e), L.C.byref(f)), g._h)
        return b.value, e.value, bool(f.value)

    def step_async_d(self, actions_d, reward_d, done_d, hand_d, terr_d, obs_d, ready_d, max_passes=8, seat0_policy=Policy.RANDOM,
                     auto_reset=True):
        """pk_env_step_async_d on DEVICE pointers (ints / c_void_p; actions_d None = seat 0 played by `seat0_policy`
        in-kernel): a bounded launch that delivers the tables whose PokerGameEnv.step returned (ready_d[t] = 1) and keeps
        the others in flight.  max_passes <= 0 drains.  See include/pokerl_hip.h."""
        g = self.game
        vp = lambda x: x if x is None or isinstance(x, L.C.c_void_p) else L.C.c_void_p(int(x))
        if self._opp_policy is None:
            raise ValueError('step_async needs ONE in-kernel policy for all opponents; per-seat / host agents: step_multi_d')
        L.check(g._lib.pk_env_step_async_d(g._h, vp(actions_d), int(seat0_policy), self._opp_policy, 1 if auto_reset else 0,
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


def _check_or_error(env, actions, table_offset):
    """env.check_actions() as a value (for running the checks of several batches on the pool's threads): the ValueError it would raise, or None."""
    try:
        env.check_actions(actions, table_offset=table_offset)
    except ValueError as e:
        return e
    return None


class VecPokerGameEnvPool:
    """ONE environment object over `num_batches` independent batches of tables, each with its own handle and HIP stream.

    Why: a bounded launch of pk_env_step_async_d ends with a tail (the last waves run alone), and launches of one handle are
    serialised on its stream, so ONE batch of 65 536 tables delivers 0.88 G env.step/s while FOUR such batches whose
    launches overlap deliver 3.6 G (docs/history.md section 5) -- a learner works on the batch whose launch has finished while the
    others run.  Tables keep their GLOBAL ids (table_id_base), so the pool's tables play exactly the trajectories of one
    handle holding all of them (RNG spec: streams are keyed by the global table id).

    `envs[b]` is an ordinary VecPokerGameEnv over tables `slices[b]`; reset() / step() below are the synchronous
    convenience forms over the whole pool, the asynchronous device-pointer calls are made per batch (envs[b].step_async_d)."""

    def __init__(self, agents=Policy.RANDOM, num_tables=1, num_batches=4, devices=None, **game_config):
        """devices: None -- every batch on game_config['device'] (default 0) -- or a list with one device index per batch
        (then num_batches = len(devices)): ONE process driving several GPUs, SURVEY 8e's single-process form.  The batches'
        calls are made from one Python thread per batch (ctypes releases the GIL for the duration of a library call), so
        the devices -- or the streams of one device -- work at the same time.  Tables are sharded as pokerl_amd.shard_tables
        does for ranks: contiguous blocks, global table ids."""
        base = int(game_config.pop('table_id_base', 0))
        if devices is not None:
            game_config.pop('device', None)
        self.slices, self.devices = self.plan(num_tables, num_batches, devices, int(game_config.get('device', 0)))
        self.envs = [VecPokerGameEnv(agents, num_tables=s.stop - s.start, table_id_base=base + s.start,
                                     **dict(game_config, device=d))
                     for s, d in zip(self.slices, self.devices)]
        self.num_tables = int(num_tables)
        self._pool = None

    @staticmethod
    def plan(num_tables, num_batches=4, devices=None, default_device=0):
        """(slices, devices) of the pool's batches: contiguous blocks of the tables as pokerl_amd.shard_tables cuts them for
        ranks, one per entry of `devices` (or num_batches of them on default_device); empty blocks are dropped."""
        from .sharding import shard_tables
        if devices is not None:
            devices = [int(d) for d in devices]
            num_batches = len(devices)
        num_batches = max(1, min(int(num_batches), int(num_tables)))
        shards = [shard_tables(int(num_tables), b, num_batches) for b in range(num_batches)]
        slices = [slice(start, start + n) for n, start in shards if n > 0]
        devs = list(devices[:len(slices)]) if devices is not None else [int(default_device)] * len(slices)
        return slices, devs

    def __len__(self):
        return len(self.envs)

    def _map(self, fn, *iterables):
        """fn over the batches, one thread per batch (a single batch: inline)."""
        if len(self.envs) == 1:
            return [fn(*args) for args in zip(*iterables)]
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(len(self.envs))
        return list(self._pool.map(fn, *iterables))

    def reset(self):
        return np.concatenate(self._map(lambda e: e.reset(), self.envs))

    def step(self, actions, strict=True):
        """PokerGameEnv.step on every table of the pool.  strict: every batch is checked BEFORE any is stepped, so that an
        invalid action raises with no table of the pool mutated (as VecPokerGameEnv.step does for its batch)."""
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(actions), (self.num_tables,)))
        if strict:
            for e, s in zip(self.envs, self.slices):
                e.check_actions(a[s], table_offset=s.start)
        outs = self._map(lambda e, s: e.step(a[s], strict=False), self.envs, self.slices)
        cat = tuple(np.concatenate([o[i] for o in outs]) for i in range(5))
        if not strict:
            return cat
        if cat[4].any():
            raise L.PokerlHipError('table error bits %s' % np.unique(cat[4]))
        return cat[:4]

    def step_pipelined(self, actions, obs='packed', auto_reset=False, strict=False):
        """The host-array fast path over the pool: send() on every batch, then recv() on every batch -- batch b+1's launch
        and the other devices' work overlap with batch b's device-to-host copies.  Returns one (obs, reward, done, hand,
        terr) tuple of pinned VIEWS per batch (see VecPokerGameEnv.recv): no concatenation, no copy.
        strict=False (the default, as send()): nothing blocks before the launches; a table whose action is invalid is left unstepped and its
        terr is TERR_INVALID_ACTION -- CHECK terr, an invalid action is not raised.  strict=True: every batch's actions are checked first (the
        pk_check_actions round trips of all batches, in parallel on the pool's threads) and the reference's ValueError is raised before any table of
        any batch is mutated -- the price is one blocking round trip per call, which the committed figures of this path
        (profiles/r05_measure_api.txt) do not include."""
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(actions), (self.num_tables,)))
        if strict:
            errs = self._map(lambda e, s: _check_or_error(e, a[s], s.start), self.envs, self.slices)
            for err in errs:                          # the lowest batch's error first: the reference names the first offending table
                if err is not None:
                    raise err
        for e, s in zip(self.envs, self.slices):
            e.send(a[s], obs=obs, auto_reset=auto_reset)
        return [e.recv() for e in self.envs]

    def sync(self):
        for e in self.envs:
            e.game.sync()

    def close(self):
        for e in self.envs:
            e.close()
        if self._pool is not None:
            self._pool.shutdown()
            self._pool = None
