"""Host-side mirror of the reference's observation object (pokerl/game.py:39-240 `Game.StateView`) and of `Card`
(pokerl/cards.py:4-72), built from the dense observation rows the device writes (include/pokerl_hip.h PK_OBS_DIM).
Pure glue: no game logic; same attribute / property names and the same `__getstate__` tuple order as the reference,
so host policies written against `game.active_state` work on `VecGame.state_view(t)` unchanged."""
import numpy as np

from .cards import card_id, card_rank, card_suit, card_value


def packed_dtype(num_players):
    """numpy structured dtype of ONE packed observation row (include/pokerl_hip.h PK_OBS_PACKED_BYTES): zero-copy field access
    to what pk_get_obs_packed / the PokerGameEnv kernels write -- rows['credits'] is an f64 [T, N] view, and so on."""
    n = int(num_players)
    dt = np.dtype([('player', np.uint8), ('turn', np.uint8), ('valid_bits', np.uint8), ('player_cards', np.uint8, (2,)),
                   ('community_cards', np.uint8, (5,)), ('_pad', np.uint8, (6,)), ('minimum_raise_value', np.float64),
                   ('credits', np.float64, (n,)), ('bets', np.float64, (n,)), ('pending_bets', np.float64, (n,))])
    assert dt.itemsize == 16 + 8 * (3 * n + 1)
    return dt


def unpack_obs(rows, num_players):
    """Packed rows (structured array of packed_dtype, or raw bytes [T, PK_OBS_PACKED_BYTES]) -> the dense f64 rows
    [T, PK_OBS_DIM] of pk_get_obs, value for value (0xFF community bytes become -1)."""
    n = int(num_players)
    r = np.asarray(rows)
    if r.dtype != packed_dtype(n):
        r = np.ascontiguousarray(r, np.uint8).reshape(-1, 16 + 8 * (3 * n + 1)).view(packed_dtype(n)).reshape(-1)
    out = np.empty((r.shape[0], 17 + 3 * n), np.float64)
    out[:, 0] = r['player']; out[:, 1] = r['turn']; out[:, 2] = r['minimum_raise_value']
    out[:, 3:10] = (r['valid_bits'][:, None] >> np.arange(7, dtype=np.uint8)[None, :]) & 1
    out[:, 10:12] = r['player_cards']
    cc = r['community_cards'].astype(np.float64)
    cc[r['community_cards'] == 0xFF] = -1.0
    out[:, 12:17] = cc
    out[:, 17:17 + n] = r['credits']; out[:, 17 + n:17 + 2 * n] = r['bets']; out[:, 17 + 2 * n:] = r['pending_bets']
    return out


class Card:
    """A playing card: `value` = (suit << 4) | rank0 (pokerl/cards.py:28-62)."""
    __slots__ = ("value",)

    def __init__(self, value):
        self.value = card_value(value)

    rank = property(lambda self: card_rank(self.value))   # ace-high 1..13, cards.py:8-14
    suit = property(lambda self: card_suit(self.value))   # cards.py:17-20
    id = property(lambda self: card_id(self.value))       # cards.py:23-26

    def __eq__(self, other):
        return isinstance(other, Card) and other.value == self.value

    def __hash__(self):
        return hash(self.value)

    def __repr__(self):  # the reference prints the Unicode playing-card glyph (cards.py:64-72)
        return chr((0x1f0a2 if (self.value & 0xf) > 10 else 0x1f0a1) + self.value)


class StateView:
    """The game as seen by one table's active player (game.py:117-131) -- as a SNAPSHOT of the step it was made after.

    The one documented difference from the reference: there `self.credits = game.credits` (game.py:128-130) stores the game's own arrays, so a
    view held across a later Game.step changes under its holder (credits / bets for ever, pending_bets until the next setup_hand rebinds the
    game's attribute, game.py:445); device memory cannot be aliased by a host object, so the arrays here are copies.  Code that relied on the
    aliasing reads `game.credits / .bets / .pending_bets` (fresh on every access).  Pinned against the reference's behaviour by
    tests/golden/views_alias_*.json ("StateView: snapshot, not alias")."""

    def __init__(self, row, num_players):
        n = int(num_players)
        row = np.asarray(row, np.float64)
        self.player = int(row[0])                                    # game.py:122
        self.valid_actions = row[3:10].copy()                        # :123 one-hot f64[7]
        self.num_players = n                                         # :124
        self.turn = int(row[1])                                      # :125
        self.player_cards = [Card(int(c)) for c in row[10:12]]       # :126
        self.community_cards = [Card(int(c)) for c in row[12:17] if c >= 0]   # :127 ([] / 3 / 4 / 5 cards, game.py:278)
        self.credits = row[17:17 + n].copy()                         # :128
        self.bets = row[17 + n:17 + 2 * n].copy()                    # :129
        self.pending_bets = row[17 + 2 * n:17 + 3 * n].copy()        # :130
        self.minimum_raise_value = float(row[2])                     # :131

    @property
    def valid_action_indices(self):                                  # game.py:133-137
        return (action for action, valid in enumerate(self.valid_actions) if valid)

    @property
    def player_hand(self):                                           # :139-143
        return self.player_cards + self.community_cards

    @property
    def pot(self):                                                   # :145-149
        return np.sum(self.bets)

    @property
    def high_bet(self):                                              # :151-155
        return np.max(self.pending_bets)

    @property
    def credit(self):                                                # :157-161
        return self.credits[self.player]

    def __getstate__(self):                                          # :208-223, same tuple order
        return (self.player, self.valid_actions, self.num_players, self.turn, self.player_cards, self.community_cards,
                self.credits, self.bets, self.pending_bets, self.minimum_raise_value)

    def __setstate__(self, state):                                   # :225-240
        (self.player, self.valid_actions, self.num_players, self.turn, self.player_cards, self.community_cards,
         self.credits, self.bets, self.pending_bets, self.minimum_raise_value) = state

    def __repr__(self):
        return "StateView(player=%d, turn=%d, credit=%.2f, pot=%.2f, valid=%s)" % (
            self.player, self.turn, self.credit, self.pot, list(self.valid_action_indices))
