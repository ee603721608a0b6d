"""Integer codes of the hot path.  The numeric VALUES are part of the parity contract (they appear in flags, masks, state
arrays and rankings exchanged with the kernels) and equal the reference's pokerl/enums.py:9-18, :38-41, :59-72,
:104-114, :130-136; everything else here (IntEnum types, derived display names, symbol helpers) is this package's own.
"""
import enum


class HandRanking(enum.IntEnum):
    """Category returned by eval_hand; a LOWER number is a stronger hand (judger.py:140)."""
    STRAIGHT_FLUSH = 1
    POKER = 2          # four of a kind
    FULL = 3           # full house
    FLUSH = 4
    STRAIGHT = 5
    TRIS = 6           # three of a kind
    TWO_PAIR = 7
    PAIR = 8
    HIGH = 9
    NONE = 10          # "no hand": seats that do not reach the showdown

    @property
    def label(self):
        return _HAND_LABELS[self]


_HAND_LABELS = {HandRanking.STRAIGHT_FLUSH: 'Straight Flush', HandRanking.POKER: 'Four of a Kind',
                HandRanking.FULL: 'Full House', HandRanking.FLUSH: 'Flush', HandRanking.STRAIGHT: 'Straight',
                HandRanking.TRIS: 'Three of a Kind', HandRanking.TWO_PAIR: 'Two Pair', HandRanking.PAIR: 'One Pair',
                HandRanking.HIGH: 'High Card', HandRanking.NONE: 'Nothing'}
# index-by-value table (slot 0 unused by eval_hand), for code written against the reference's `as_string`
HandRanking.as_string = ['Five of a Kind'] + [_HAND_LABELS[HandRanking(v)] for v in range(1, 11)]


class PokerMoves(enum.IntEnum):
    """Action codes accepted by step(); bit a of a valid-action mask refers to move a."""
    FOLD = 0
    CHECK = 1
    CALL = 2
    RAISE_TEN = 3      # raise by 10 % of what would be left after calling
    RAISE_QUARTER = 4
    RAISE_HALF = 5
    ALL_IN = 6


PokerMoves.RAISE_ANY = 3          # first raise code
PokerMoves.NUM_MOVES = 7
PokerMoves.NUM_RAISE_MOVES = 3
PokerMoves.RAISE_FRACTIONS = (0.1, 0.25, 0.5)   # game.py:370, :676
PokerMoves.as_string = ['Fold', 'Check', 'Call', 'Raise 10%', 'Raise 25%', 'Raise half', 'All-in']


class PlayerState(enum.IntEnum):
    """Per-seat state stored in player_states."""
    FOLDED = 0
    ACTIVE = 1         # still to act in this betting round
    CALLED = 2         # matched (or made) the high bet
    ALL_IN = 3
    BROKEN = 4         # out of the game (credits <= 0 at the end of a hand)


PlayerState.NUM_STATES = 5
PlayerState.as_string = [m.name.replace('_', '-').capitalize() for m in PlayerState]


class CardSuit(enum.IntEnum):
    """High nibble of a card byte."""
    SPADES = 0
    HEARTS = 1
    DIAMONDS = 2
    CLUBS = 3


CardSuit.NUM_SUITS = 4
CardSuit.from_symbol = {m.name[0]: int(m) for m in CardSuit}


class CardRank(enum.IntEnum):
    """Low nibble of a card byte is ONE..KING (0..12); ACE (13) is the ace-high rank eval_hand reports as a kicker."""
    ONE = 0
    TWO = 1
    THREE = 2
    FOUR = 3
    FIVE = 4
    SIX = 5
    SEVEN = 6
    EIGHT = 7
    NINE = 8
    TEN = 9
    JACK = 10
    QUEEN = 11
    KING = 12
    ACE = 13


CardRank.NUM_RANKS = 13
CardRank.as_symbol = list('123456789TJQKA')
CardRank.from_symbol = {s: i for i, s in enumerate(CardRank.as_symbol)}


class Policy(enum.IntEnum):
    """In-kernel synthetic agents (include/pokerl_hip.h PK_POLICY_*)."""
    RANDOM = 0   # uniform over the valid actions, the reference's RandomAgent (agents/random.py:12-16)
    ALL_IN = 1   # always PokerMoves.ALL_IN
    CALL = 2     # CALL if valid, else CHECK if valid, else ALL_IN
