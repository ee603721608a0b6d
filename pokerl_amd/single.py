"""`Game` and `PokerGameEnv` with the reference's OWN shapes: one table, scalars and per-seat vectors, `Card` lists and
`StateView` objects -- for code written against pokerl/game.py and pokerl/envs/game_env.py that is moved over unchanged
(examples/random_game.py, a learner's evaluation loop, a debugger session).  Thin views over VecGame / VecPokerGameEnv with
`num_tables = 1`: every value still comes from the HIP kernels, nothing is computed here.  (One table per launch is a
latency-bound way to use a GPU: batch with VecGame when throughput matters.)
"""
import numpy as np

from .enums import PlayerState
from .envs import VecPokerGameEnv
from .game import VecGame
from .state_view import Card, StateView


class Game:
    """`pokerl.game.Game(**config)` (game.py:242-264): same config keys, attributes, properties and methods."""

    StateView = StateView    # game.py:39: the reference nests the class in Game

    def __init__(self, **config):
        self.logger = config.pop('logger', None)       # accepted and ignored: no result depends on it (game.py:250)
        self._v = VecGame(1, **config)
        self.num_players = self._v.num_players
        self.start_credits = self._v.start_credits
        self.big_blind, self.small_blind = self._v.big_blind, self._v.small_blind

    def close(self):
        self._v.close()

    # ---- attributes of game.py:251-264 (read fresh from the device on every access)
    credits = property(lambda self: self._v.credits[0])
    bets = property(lambda self: self._v.bets[0])
    pending_bets = property(lambda self: self._v.pending_bets[0])
    payoffs = property(lambda self: self._v.payoffs[0])
    player_states = property(lambda self: self._v.player_states[0])
    minimum_raise_value = property(lambda self: float(self._v.minimum_raise_value[0]))
    active_player = property(lambda self: int(self._v.active_player[0]))
    turn = property(lambda self: int(self._v.turn[0]))
    hand = property(lambda self: int(self._v.hand[0]))
    dealer_idx = property(lambda self: int(self._v.dealer_idx[0]))
    big_blind_idx = property(lambda self: int(self._v.big_blind_idx[0]))
    small_blind_idx = property(lambda self: int(self._v.small_blind_idx[0]))

    @property
    def deck(self):
        """deck[0 : 5 + 2N] as Card objects -- the part of the deck the game ever reads (game.py:278, :385-395)."""
        return [Card(int(v)) for v in self._v.deck[0]]

    # ---- properties of game.py:266-332
    @property
    def community_cards(self):
        return [] if self.turn == 0 else self.deck[:self.turn + 2]                 # :266-278

    pot = property(lambda self: float(self._v.pot[0]))                            # :281-284
    high_bet = property(lambda self: float(self._v.high_bet[0]))                  # :287-290
    high_bidders = property(lambda self: self.bets == self.high_bet)              # :293-296
    pending_credits = property(lambda self: self.credits - self.pending_bets)     # :299-302
    blind_idx = property(lambda self: [self.big_blind_idx, self.small_blind_idx])  # :305-308
    blind_value = property(lambda self: [self.big_blind, self.small_blind])       # :311-314
    game_over = property(lambda self: bool(self._v.game_over[0]))                 # :317-320
    active_state = property(lambda self: self._v.state_view(0))                   # :323-332

    def get_first_playing(self, idx):                                             # :334-337
        alive = np.roll(self.player_states, -idx) != PlayerState.BROKEN
        return int((idx + np.argmax(alive)) % self.num_players)

    def get_valid_actions(self, player=None):                                     # :339-383
        onehot, gen = self._v.get_valid_actions(player)
        return onehot[0], (int(i) for i in next(gen))

    def get_cards_of(self, player):                                               # :385-389
        i = 5 + player * 2
        return self.deck[i:i + 2]

    def get_hand_for(self, player):                                               # :391-395
        i = 5 + player * 2
        d = self.deck
        return d[:5] + d[i:i + 2]

    def reset(self, **config):                                                    # :397-412
        self._v.reset(**config)

    def step(self, action):                                                       # :621-700
        """Returns (game_over, hand_over, turn_over); raises the reference's ValueError / NotImplementedError /
        AssertionError in the reference's situations."""
        if not isinstance(action, (int, np.integer)) or isinstance(action, bool):
            raise NotImplementedError                                             # :646, :700
        over, hand, turn = self._v.step(np.array([action], np.int64))
        return bool(over[0]), bool(hand[0]), bool(turn[0])


class PokerGameEnv:
    """`pokerl.envs.PokerGameEnv(agents, **game_config)` (envs/game_env.py:6-53) for one table: reset() -> StateView,
    step(action) -> (StateView, reward, done, hand).  agents: in-kernel agents and / or callables, see pokerl_amd.agents."""

    def __init__(self, agents=None, **game_config):
        self._e = VecPokerGameEnv(agents, num_tables=1, **game_config)
        self.agents = self._e.agents                       # [None, *agents], game_env.py:17
        self.player_agent = 0
        self.game = _GameView(self._e.game)

    def reset(self):
        return StateView(self._e.reset()[0], self._e.game.num_players)           # game_env.py:20-29

    def step(self, action):
        if not isinstance(action, (int, np.integer)) or isinstance(action, bool):
            raise NotImplementedError
        obs, reward, done, hand = self._e.step(np.array([action], np.int64))      # game_env.py:31-53
        return StateView(obs[0], self._e.game.num_players), float(reward[0]), bool(done[0]), bool(hand[0])

    def close(self):
        self._e.close()


class _GameView(Game):
    """`env.game` of the single-table env: the Game view over the env's own table."""

    def __init__(self, vec_game):                 # no second handle: shares the env's
        self.logger = None
        self._v = vec_game
        self.num_players = vec_game.num_players
        self.start_credits = vec_game.start_credits
        self.big_blind, self.small_blind = vec_game.big_blind, vec_game.small_blind

    def close(self):
        pass
