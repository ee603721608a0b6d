"""Integer constants of the path, bit-for-bit those of the reference's pokerl/enums.py:1-138."""


class HandRanking:  # pokerl/enums.py:9-18
    STRAIGHT_FLUSH, POKER, FULL, FLUSH, STRAIGHT, TRIS, TWO_PAIR, PAIR, HIGH, NONE = range(1, 11)
    as_string = ['Five of a Kind', 'Straight Flush', 'Four of a Kind', 'Full House', 'Flush', 'Straight',
                 'Three of a Kind', 'Two Pair', 'One Pair', 'High Card', 'Nothing']


class CardSuit:  # pokerl/enums.py:35-53
    SPADES, HEARTS, DIAMONDS, CLUBS = range(4)
    NUM_SUITS = 4
    from_symbol = {'S': 0, 'H': 1, 'D': 2, 'C': 3}


class CardRank:  # pokerl/enums.py:56-96
    ONE, TWO, THREE, FOUR, FIVE, SIX, SEVEN, EIGHT, NINE, TEN, JACK, QUEEN, KING, ACE = range(14)
    NUM_RANKS = 13
    as_symbol = ['1', '2', '3', '4', '5', '6', '7', '8', '9', 'T', 'J', 'Q', 'K', 'A']
    from_symbol = {s: i for i, s in enumerate(as_symbol)}


class PokerMoves:  # pokerl/enums.py:98-116
    FOLD, CHECK, CALL, RAISE_ANY, RAISE_QUARTER, RAISE_HALF, ALL_IN = range(7)
    RAISE_TEN = 3
    NUM_MOVES = 7
    NUM_RAISE_MOVES = 3
    as_string = ['Fold', 'Check', 'Call', 'Raise 10%', 'Raise 25%', 'Raise half', 'All-in']


class PlayerState:  # pokerl/enums.py:118-138
    FOLDED, ACTIVE, CALLED, ALL_IN, BROKEN = range(5)
    NUM_STATES = 5
    as_string = ['Folded', 'Active', 'Called', 'All-in', 'Broken']


class Policy:
    """In-kernel synthetic agents (include/pokerl_hip.h PK_POLICY_*)."""
    RANDOM = 0  # uniform over valid actions: RandomAgent, pokerl/agents/random.py:12-16
    ALL_IN = 1
