"""VecGame: T independent reference `Game`s stepped in lockstep on one MI355X.

Host-side mirror of the reference's pokerl/game.py `Game` API for the data-parallel hot path: same method and
attribute names, same argument meaning and error behaviour, with a leading table axis on everything.  All game
logic runs in hand-written HIP kernels behind the C ABI of include/pokerl_hip.h (ctypes; no torch, no CPU fallback).
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .enums import PokerMoves

DEFAULT_SEED = 0x706F6B65726C  # 'pokerl'


class VecGame:
    """`Game(**config)` (pokerl/game.py:242-264) for `num_tables` tables.

    config: num_players=4, start_credits=100 (int or per-seat list/array), big_blind=2, small_blind=1, dealer=0 -- as
    the reference -- plus num_tables, seed, device, table_id_base (global id of table 0: RNG streams are keyed by the
    global table id, so a table's trajectory does not depend on which GPU/shard hosts it).
    """

    def __init__(self, num_tables=1, **config):
        self.num_tables = int(num_tables)
        self.num_players = int(config.get('num_players', 4))
        self.start_credits = config.get('start_credits', 100)
        self.big_blind = config.get('big_blind', 2)
        self.small_blind = config.get('small_blind', 1)
        self.dealer = int(config.get('dealer', 0))
        self.seed = int(config.get('seed', DEFAULT_SEED))
        self.device = int(config.get('device', 0))
        self.table_id_base = int(config.get('table_id_base', 0))
        if not (L.MIN_PLAYERS <= self.num_players <= L.MAX_PLAYERS):
            raise ValueError('num_players must be in [%d, %d]' % (L.MIN_PLAYERS, L.MAX_PLAYERS))
        self._lib = L.lib()
        self._h = C.c_void_p()
        if isinstance(self.start_credits, (int, float, np.integer, np.floating)):
            sc, scalar = None, float(self.start_credits)
        else:
            sc = np.ascontiguousarray(self.start_credits, np.float64)
            if sc.shape != (self.num_players,):
                raise ValueError('start_credits must be a scalar or one value per player')
            scalar = 0.0
        rc = self._lib.pk_create(C.byref(self._h), self.device, self.num_tables, self.num_players, L.ptr(sc), scalar,
                                 float(self.big_blind), float(self.small_blind), self.dealer, self.seed,
                                 self.table_id_base)
        if rc != L.PK_OK:
            self._h = C.c_void_p()
            L.check(rc)

    def close(self):
        if getattr(self, '_h', None) and self._h.value:
            self._lib.pk_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ reset / step
    def reset(self, mask=None, **config):
        """`Game.reset(**config)` (game.py:397-412) on every table, or on tables where mask != 0.  Only `dealer` is
        re-read from config (game.py:403)."""
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        if m is not None and m.shape != (self.num_tables,):
            raise ValueError('mask must have shape (num_tables,)')
        L.check(self._lib.pk_reset(self._h, L.ptr(m), int(config.get('dealer', 0))), self._h)

    def _actions(self, actions):
        a = np.asarray(actions)
        if a.dtype.kind not in 'iu':          # game.py:646,700: only int actions are implemented
            raise NotImplementedError
        a = np.ascontiguousarray(np.broadcast_to(a, (self.num_tables,)), np.int32)
        return a

    def step(self, actions, strict=True):
        """`Game.step(action)` (game.py:621-700), one action per table.

        Returns (game_over, hand_over, turn_over) as bool arrays [T] -- the reference's tuple, vectorised.
        strict=True mirrors the reference's error contract for the batch: if ANY action is invalid a ValueError
        ('Player %d invalid move: `%s`', game.py:651) is raised before any table is mutated.  strict=False steps the
        tables whose action is valid, leaves the others untouched and returns a 4th array of per-table error bits.
        """
        a = self._actions(actions)
        if strict:
            self.check_actions(a)
        flags = np.zeros(self.num_tables, np.uint8)
        terr = np.zeros(self.num_tables, np.uint8)
        rc = self._lib.pk_step(self._h, L.ptr(a), L.ptr(flags), L.ptr(terr))
        L.check(rc, self._h, allow_table_errors=True)
        out = ((flags & L.FLAG_GAME_OVER) != 0, (flags & L.FLAG_HAND_OVER) != 0, (flags & L.FLAG_TURN_OVER) != 0)
        if strict:
            if (terr & L.TERR_NO_WINNER).any():   # game.py:473
                raise AssertionError('Invalid state: no potential winner (table %d)' % int(np.argmax(terr & L.TERR_NO_WINNER)))
            if terr.any():
                raise L.PokerlHipError('table error bits %s' % np.unique(terr))
            return out
        return out + (terr,)

    # ---- device-resident forms (a learner on the same GPU: no host round trip; asynchronous on the handle's stream)
    @staticmethod
    def _dptr(p):
        """A device pointer as c_void_p: an int (tensor.data_ptr()), a c_void_p, or an object with .ptr (hipmem.DeviceBuffer)."""
        if p is None:
            return None
        if hasattr(p, 'ptr'):
            p = p.ptr
        return p if isinstance(p, C.c_void_p) else C.c_void_p(int(p))

    def step_d(self, actions_d, flags_d, terr_d=None, auto_reset=False):
        """`Game.step` (game.py:621-700) on DEVICE buffers (pk_step_d): actions_d i32[T] in, flags_d u8[T] (PK_FLAG_* bits =
        the reference's (game_over, hand_over, turn_over)) and terr_d u8[T] (optional) out.  A table whose action is not
        valid is left untouched and gets PK_TERR_INVALID_ACTION (game.py:649-651); nothing is raised -- the caller reads
        terr_d.  Asynchronous: order it with your stream through set_stream / wait_event / record_event.
        auto_reset=True (pk_step_auto_d): a table whose step ends its game is `Game.reset()` in the same launch (flags_d still says
        game_over) -- the rollout loop of examples/random_game.py:8-12 without a separate reset launch."""
        fn = self._lib.pk_step_auto_d if auto_reset else self._lib.pk_step_d
        L.check(fn(self._h, self._dptr(actions_d), self._dptr(flags_d), self._dptr(terr_d)), self._h)

    def step_async_d(self, actions_d, flags_d, terr_d, ready_d, max_hands=1, auto_reset=False):
        """`Game.step` as a bounded launch (pk_step_async_d): tables whose step has returned get ready_d[t] = 1 and their flags_d /
        terr_d; a table whose step rolls on through further hands (game.py:607-611) stays in flight on the device, ready_d[t] = 0,
        and the next call carries on with it, ignoring actions_d[t].  max_hands <= 0 drains (every table ready); until then every
        other method that reads or changes tables raises (PK_E_BUSY).  Per table the steps, flags and RNG draws are the synchronous ones.
        A drain is a full step call -- idle tables are stepped with actions_d[t] -- unless actions_d is None (drain only: nobody steps, idle
        tables come back untouched with TERR_INVALID_ACTION in terr_d)."""
        L.check(self._lib.pk_step_async_d(self._h, self._dptr(actions_d), self._dptr(flags_d), self._dptr(terr_d), self._dptr(ready_d),
                                          int(max_hands), int(bool(auto_reset))), self._h)

    def set_step_obs(self, obs_d=None, obs_packed_d=None):
        """`game.active_state` (game.py:323-332) from the step kernels themselves (pk_set_step_obs): from now on step_d / step_async_d (and
        step) write the StateView row of the player to act of every table whose step returned into obs_d (f64 [T, PK_OBS_DIM(N)], the row of
        observations_of(None)) and / or obs_packed_d ([T] rows of state_view.packed_dtype(N)) -- device buffers the caller keeps alive;
        None, None switches it off.  One launch per step instead of step + observation."""
        L.check(self._lib.pk_set_step_obs(self._h, self._dptr(obs_d), self._dptr(obs_packed_d)), self._h)

    def pick_actions_d(self, actions_d, policy=0):
        """The action the in-kernel agent `policy` takes on every table -> actions_d i32[T] (pk_pick_actions_d)."""
        L.check(self._lib.pk_pick_actions_d(self._h, int(policy), self._dptr(actions_d)), self._h)

    def reset_d(self, mask_d=None, mask_bits=0xff, dealer=0):
        """`Game.reset` on the tables with (mask_d[t] & mask_bits) != 0 (pk_reset_d; mask_d None = all): with step_d's flags_d
        and mask_bits = FLAG_GAME_OVER this is `if game_over: game.reset()` of examples/random_game.py:8-12 on the device."""
        L.check(self._lib.pk_reset_d(self._h, self._dptr(mask_d), int(mask_bits), int(dealer)), self._h)

    def check_actions(self, actions, table_offset=0):
        """Game.step's precondition (game.py:648-651) for the whole batch, checked on the device (pk_check_actions): raises
        the reference's ValueError naming the first offending table; nothing is mutated."""
        a = self._actions(actions)
        bad = C.c_int32(-1)
        L.check(self._lib.pk_check_actions(self._h, L.ptr(a), C.byref(bad)), self._h)
        if bad.value >= 0:
            t = bad.value
            name = PokerMoves.as_string[a[t]] if 0 <= a[t] < PokerMoves.NUM_MOVES else str(int(a[t]))
            raise ValueError('Player %d invalid move: `%s` (table %d)' % (int(self.active_player[t]), name, t + table_offset))

    def _seat(self, player):
        """None -> -1 (each table's active player); else a seat index valid for every table."""
        if player is None:
            return -1
        p = int(player)
        if not (0 <= p < self.num_players):
            raise IndexError('player %d out of range' % p)   # the reference's credits[player] would raise the same way
        return p

    def get_valid_actions(self, player=None):
        """`Game.get_valid_actions(player=None)` (game.py:339-383): (onehot f64 [T,7], generator of index arrays).
        player=None: each table's active player; an int: that seat on every table (0 IS seat 0 here, game.py:363)."""
        out = np.zeros((self.num_tables, PokerMoves.NUM_MOVES), np.uint8)
        L.check(self._lib.pk_get_valid_actions(self._h, self._seat(player), L.ptr(out)), self._h)
        onehot = out.astype(np.float64)
        return onehot, (np.nonzero(row)[0] for row in out)

    # ------------------------------------------------------------------ throughput / agents
    def pick_actions(self, policy=0):
        a = np.zeros(self.num_tables, np.int32)
        L.check(self._lib.pk_pick_actions(self._h, int(policy), L.ptr(a)), self._h)
        return a

    def rollout(self, steps, policy=0, auto_reset=True, fused=True, counters=True):
        """`steps` lockstep Game.step()s per table with in-kernel agents (examples/random_game.py:8-12 as a kernel).
        Returns dict(steps, hands, evals, games) when counters=True (synchronises), else None (asynchronous)."""
        c = np.zeros(L.NUM_COUNTERS, np.uint64) if counters else None
        L.check(self._lib.pk_rollout(self._h, int(steps), int(policy), int(bool(auto_reset)), int(bool(fused)), L.ptr(c)),
                self._h)
        if c is not None:
            return dict(steps=int(c[0]), hands=int(c[1]), evals=int(c[2]), games=int(c[3]))

    def time_rollout(self, steps, policy=0, auto_reset=True, fused=True, reps=1):
        """Average device milliseconds per kernel launch (HIP events on the handle's stream) + counters."""
        ms = C.c_double(0.0)
        c = np.zeros(L.NUM_COUNTERS, np.uint64)
        L.check(self._lib.pk_time_rollout(self._h, int(steps), int(policy), int(bool(auto_reset)), int(bool(fused)),
                                          int(reps), C.byref(ms), L.ptr(c)), self._h)
        return ms.value, dict(steps=int(c[0]), hands=int(c[1]), evals=int(c[2]), games=int(c[3]))

    def sync(self):
        """Completes deferred rollout steps and waits for the handle's stream."""
        L.check(self._lib.pk_sync(self._h), self._h)

    def flush(self):
        L.check(self._lib.pk_flush(self._h), self._h)

    @property
    def owed(self):
        """Diagnostic: steps each table still owes after the launches queued so far (deferred work is NOT completed)."""
        out = np.zeros(self.num_tables, np.uint32)
        L.check(self._lib.pk_get_owed(self._h, L.ptr(out)), self._h)
        return out

    def set_tuning(self, park=0, endk=0):
        L.check(self._lib.pk_set_tuning(self._h, int(park), int(endk)), self._h)

    def set_coalesce(self, max_steps):
        """Asynchronous rollout calls that arrive while two launches are in flight are merged on the host into launches
        of up to `max_steps` steps (0: every call launches; default 1024)."""
        L.check(self._lib.pk_set_coalesce(self._h, int(max_steps)), self._h)

    def launch_stats(self, reset=False):
        """dict(launches, steps, min, max) of the fused rollout launches since the last reset."""
        out = np.zeros(4, np.uint64)
        L.check(self._lib.pk_get_launch_stats(self._h, L.ptr(out), int(bool(reset))), self._h)
        return dict(launches=int(out[0]), steps=int(out[1]), min=int(out[2]), max=int(out[3]))

    # ------------------------------------------------------------------ streams (callers with their own HIP stream)
    @property
    def stream(self):
        s = C.c_void_p()
        L.check(self._lib.pk_get_stream(self._h, C.byref(s)), self._h)
        return s.value

    def set_stream(self, stream):
        """Run on the caller's hipStream_t (int / c_void_p).  0 / None IS a stream -- the legacy default stream, which is
        what torch.cuda.current_stream().cuda_stream returns outside a torch.cuda.Stream context; use_own_stream() goes
        back to the handle's own non-blocking stream."""
        L.check(self._lib.pk_set_stream(self._h, C.c_void_p(stream) if isinstance(stream, int) else stream), self._h)

    def use_own_stream(self):
        L.check(self._lib.pk_use_own_stream(self._h), self._h)

    def wait_event(self, event):
        L.check(self._lib.pk_wait_event(self._h, C.c_void_p(event) if isinstance(event, int) else event), self._h)

    def record_event(self, event):
        L.check(self._lib.pk_record_event(self._h, C.c_void_p(event) if isinstance(event, int) else event), self._h)

    # ------------------------------------------------------------------ RNG-spec serials (checkpoint / resume)
    def _serials(self):
        hs = np.zeros(self.num_tables, np.uint64)
        ss = np.zeros(self.num_tables, np.uint64)
        L.check(self._lib.pk_get_serials(self._h, L.ptr(hs), L.ptr(ss)), self._h)
        return hs, ss

    hand_serial = property(lambda self: self._serials()[0])
    step_serial = property(lambda self: self._serials()[1])

    def set_serials(self, hand_serial=None, step_serial=None):
        def arr(v):
            return None if v is None else np.ascontiguousarray(np.broadcast_to(np.asarray(v, np.uint64), (self.num_tables,)))
        hs, ss = arr(hand_serial), arr(step_serial)
        L.check(self._lib.pk_set_serials(self._h, L.ptr(hs), L.ptr(ss)), self._h)

    # ------------------------------------------------------------------ state reads (Game attributes)
    def _f64(self, field):
        out = np.zeros((self.num_tables, self.num_players), np.float64)
        L.check(self._lib.pk_get_f64(self._h, field, L.ptr(out)), self._h)
        return out

    def _i32(self, field):
        out = np.zeros(self.num_tables, np.int32)
        L.check(self._lib.pk_get_i32(self._h, field, L.ptr(out)), self._h)
        return out

    credits = property(lambda self: self._f64(L.F_CREDITS))
    bets = property(lambda self: self._f64(L.F_BETS))
    pending_bets = property(lambda self: self._f64(L.F_PENDING_BETS))
    payoffs = property(lambda self: self._f64(L.F_PAYOFFS))
    active_player = property(lambda self: self._i32(L.I_ACTIVE_PLAYER))
    turn = property(lambda self: self._i32(L.I_TURN))
    dealer_idx = property(lambda self: self._i32(L.I_DEALER_IDX))
    small_blind_idx = property(lambda self: self._i32(L.I_SMALL_BLIND_IDX))
    big_blind_idx = property(lambda self: self._i32(L.I_BIG_BLIND_IDX))
    hand = property(lambda self: self._i32(L.I_HAND))

    def _table_f64(self, field):
        out = np.zeros(self.num_tables, np.float64)
        L.check(self._lib.pk_get_table_f64(self._h, field, L.ptr(out)), self._h)
        return out

    minimum_raise_value = property(lambda self: self._table_f64(L.TF_MIN_RAISE))
    pot = property(lambda self: self._table_f64(L.TF_POT))            # game.py:281-284, np.sum order kept on the device
    high_bet = property(lambda self: self._table_f64(L.TF_HIGH_BET))  # game.py:287-290

    @property
    def player_states(self):
        out = np.zeros((self.num_tables, self.num_players), np.uint8)
        L.check(self._lib.pk_get_player_states(self._h, L.ptr(out)), self._h)
        return out

    @property
    def deck(self):
        """deck[:, 0:5+2N] as Card.value bytes -- the only part of the deck the game ever reads (game.py:278,385-395)."""
        out = np.zeros((self.num_tables, 5 + 2 * self.num_players), np.uint8)
        L.check(self._lib.pk_get_cards(self._h, L.ptr(out)), self._h)
        return out

    @property
    def community_cards(self):
        """[T,5] card values, -1 where not yet visible (game.py:266-278: [] at turn 0, deck[:turn+2] after)."""
        cards = self.deck[:, :5].astype(np.int16)
        turn = self.turn
        visible = (turn[:, None] != 0) & (np.arange(5)[None, :] < (turn[:, None] + 2))
        cards[~visible] = -1
        return cards

    def get_cards_of(self, player):
        """game.py:385-389.  player: int or int array [T]."""
        p = np.broadcast_to(np.asarray(player), (self.num_tables,))
        d = self.deck
        t = np.arange(self.num_tables)
        return np.stack([d[t, 5 + 2 * p], d[t, 6 + 2 * p]], axis=1)

    def get_hand_for(self, player):
        """game.py:391-395: the 5 community cards + the player's 2 hole cards, [T,7]."""
        return np.concatenate([self.deck[:, :5], self.get_cards_of(player)], axis=1)

    @property
    def hand_rankings(self):
        """Rankings of each table's last showdown (game.py:488-489): (rank [T,N] HandRanking, kickers value [T,N])."""
        rank = np.zeros((self.num_tables, self.num_players), np.uint8)
        kick = np.zeros((self.num_tables, self.num_players), np.uint32)
        L.check(self._lib.pk_get_hand_ranks(self._h, L.ptr(rank), L.ptr(kick)), self._h)
        return rank, kick

    @property
    def game_over(self):
        out = np.zeros(self.num_tables, np.uint8)                     # game.py:317-320
        L.check(self._lib.pk_get_game_over(self._h, L.ptr(out)), self._h)
        return out != 0

    def observations_of(self, player=None, out=None):
        """Dense `StateView(game, player)` rows (game.py:117-131), f64 [T, PK_OBS_DIM(N)]; layout in pokerl_hip.h.
        player=None or 0: each table's active player (`player or game.active_player`, game.py:122).
        out: a C-contiguous f64 [T, PK_OBS_DIM] array to fill (a pinned one -- pokerl_amd.pinned_empty -- copies at PCIe rate)."""
        seat = -1 if not player else self._seat(player)
        shape = (self.num_tables, 17 + 3 * self.num_players)
        if out is None:
            out = np.empty(shape, np.float64)
        elif out.shape != shape or out.dtype != np.float64 or not out.flags.c_contiguous:
            raise ValueError('out must be a C-contiguous f64 array of shape %s' % (shape,))
        L.check(self._lib.pk_get_obs(self._h, seat, L.ptr(out)), self._h)
        return out

    def observations_packed_of(self, player=None, out=None):
        """The same rows in the compact form (pk_get_obs_packed: 16 header bytes + (3N+1) f64 per table, 168 B against 280 at six
        seats) as a structured array [T] of state_view.packed_dtype(N); state_view.unpack_obs() gives the dense rows back."""
        from .state_view import packed_dtype
        seat = -1 if not player else self._seat(player)
        dt = packed_dtype(self.num_players)
        if out is None:
            out = np.empty(self.num_tables, dt)
        elif out.shape != (self.num_tables,) or out.dtype != dt or not out.flags.c_contiguous:
            raise ValueError('out must be a C-contiguous array of packed_dtype(N) with one row per table')
        L.check(self._lib.pk_get_obs_packed(self._h, seat, L.ptr(out)), self._h)
        return out

    observations = property(lambda self: self.observations_of(None))

    def state_views(self, player=None):
        """One `StateView` (the reference's observation object, game.py:39-240) per table, for host-side policies."""
        from .state_view import StateView
        return [StateView(row, self.num_players) for row in self.observations_of(player)]

    def state_view(self, table=0, player=None):
        from .state_view import StateView
        return StateView(self.observations_of(player)[table], self.num_players)

    active_state = observations
