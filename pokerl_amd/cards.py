"""Card encoding of the path: one byte per card, value = (suit << 4) | rank0 (reference pokerl/cards.py:4-77).

Host-side data helpers only (string <-> byte); no game logic lives here.
"""
import numpy as np

from .enums import CardRank, CardSuit


def card_value(card) -> int:
    """int -> itself; 'RS' string ('AD', '1D', 'TS', ...) or (rank, suit) tuple -> value (pokerl/cards.py:28-62)."""
    if isinstance(card, (int, np.integer)):
        value = int(card)
    elif isinstance(card, tuple):
        rank, suit = card
        if rank == CardRank.ACE:
            rank = CardRank.ONE
        value = (suit << 4) | rank
    elif isinstance(card, str):
        rank, suit = CardRank.from_symbol[card[0]], CardSuit.from_symbol[card[1]]
        if rank == CardRank.ACE:
            rank = CardRank.ONE
        value = (suit << 4) | rank
    else:
        value = int(card.value)  # anything Card-like
    assert (value & 0xf) < CardRank.NUM_RANKS, 'Invalid card rank'
    assert (value >> 4) < CardSuit.NUM_SUITS, 'Invalid card suit'
    return value


def card_rank(value):
    """Ace-high rank 1..13 (pokerl/cards.py:8-14)."""
    return (value & 0xf) or 13


def card_suit(value):
    return value >> 4  # pokerl/cards.py:17-20


def card_id(value):
    return (value & 0xf) + (value >> 4) * 13  # pokerl/cards.py:23-26


def default_deck_values() -> np.ndarray:
    """Card values of create_default_deck() in order (pokerl/cards.py:74-77): rank-major, suits inner."""
    i = np.arange(52)
    return (((i % 4) << 4) | (i // 4)).astype(np.uint8)
