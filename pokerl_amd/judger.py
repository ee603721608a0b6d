"""pokerl.judger on the device: eval_hand / compare_rankings / compare_hands (reference pokerl/judger.py:7-189),
batched through the C ABI.  Cards are Card.value bytes (or 'RS' strings / Card-likes via cards.card_value)."""
import numpy as np

from . import _lib as L
from .cards import card_value
from .enums import HandRanking


def _unpack(kick, nk):
    return [int((kick >> (4 * (nk - 1 - i))) & 0xf) for i in range(nk)]


def get_kickers_value(kickers) -> int:
    """judger.py:101-109: kickers packed big-endian in nibbles."""
    v = 0
    for k in kickers:
        v = (v << 4) | int(k)
    return v


def eval_hands(cards, ncards=None, device=0):
    """cards: uint8 [M,7] Card.value (slots >= ncards ignored), ncards: [M] in 0..7 or None (=7).
    Returns (rank uint8[M], kickers_value uint32[M], num_kickers uint8[M])."""
    cards = np.ascontiguousarray(cards, np.uint8)
    if cards.ndim != 2 or cards.shape[1] != 7:
        raise ValueError('cards must have shape [M, 7]')
    m = cards.shape[0]
    nc = None if ncards is None else np.ascontiguousarray(ncards, np.uint8)
    rank = np.zeros(m, np.uint8)
    kick = np.zeros(m, np.uint32)
    nk = np.zeros(m, np.uint8)
    L.check(L.lib().pk_eval_hands(int(device), L.ptr(cards), L.ptr(nc), m, L.ptr(rank), L.ptr(kick), L.ptr(nk)))
    return rank, kick, nk


def eval_hands_d(cards_d, ncards_d, m, rank_d, kick_d, nkick_d=None, device=0, stream=None):
    """pk_eval_hands_d: the same op on device-resident buffers (device pointers as ints / c_void_p), asynchronous on
    `stream` -- e.g. the rank feature of examples/q_learning.py:29-33 for a learner that lives on the GPU."""
    L.check(L.lib().pk_eval_hands_d(int(device), cards_d, ncards_d, int(m), rank_d, kick_d, nkick_d, stream))


def eval_hand(hand, device=0):
    """judger.eval_hand(hand) -> (HandRanking, [kickers]) for one hand of 0..7 cards (judger.py:7-99)."""
    vals = [card_value(c) for c in hand]
    if len(vals) > 7:
        raise ValueError('at most seven cards')
    row = np.zeros((1, 7), np.uint8)
    row[0, :len(vals)] = vals
    rank, kick, nk = eval_hands(row, np.array([len(vals)], np.uint8), device)
    return int(rank[0]), _unpack(int(kick[0]), int(nk[0]))


def compare_rankings_batch(rank, kick, device=0):
    """rank uint8 [M,n], kick uint32 [M,n] -> onehot uint8 [M,n] (judger.py:111-158, incl. its line-148 behaviour)."""
    rank = np.ascontiguousarray(rank, np.uint8)
    kick = np.ascontiguousarray(kick, np.uint32)
    if rank.ndim != 2 or rank.shape != kick.shape:
        raise ValueError('rank and kick must both have shape [M, n]')
    m, n = rank.shape
    onehot = np.zeros((m, n), np.uint8)
    L.check(L.lib().pk_compare_rankings(int(device), L.ptr(rank), L.ptr(kick), n, m, L.ptr(onehot)))
    return onehot


def compare_rankings(rankings, device=0):
    """judger.compare_rankings(rankings) -> (onehot list, winners list); rankings = [(rank, [kickers]), ...]."""
    rank = np.array([[r for r, _ in rankings]], np.uint8)
    kick = np.array([[get_kickers_value(k) for _, k in rankings]], np.uint32)
    onehot = compare_rankings_batch(rank, kick, device)[0]
    return [int(x) for x in onehot], [i for i, x in enumerate(onehot) if x]


def compare_hands(hands, device=0):
    """judger.compare_hands(hands) -> (onehot, winners, rankings) (judger.py:160-189)."""
    rankings = [eval_hand(h, device) for h in hands]
    return compare_rankings(rankings, device) + (rankings,)


def eval7_prefix(a, b, fast=True, device=0):
    """Values rank<<20|kick of all 7-card hands whose two lowest canonical indices are (a, b) (exhaustive checks).
    fast=True / 1: the distinct-card evaluator used by the showdown kernels; False / 0: the general evaluator; 3: what
    pk_eval_hands applies (fast path for 3..7 distinct cards);
    2: the table-driven distinct-card evaluator of the streaming kernel (eval7_stream)."""
    import ctypes as C
    import math
    n = math.comb(51 - b, 5)
    out = np.zeros(max(n, 1), np.uint32)
    cnt = C.c_size_t(0)
    L.check(L.lib().pk_eval7_prefix(int(device), int(a), int(b), int(fast), L.ptr(out), C.byref(cnt)))
    return out[:cnt.value]


NONE_RANKING = (HandRanking.NONE, [])


def eval7_stream(hands_d, m, out_d, distinct=True, device=0):
    """pk_eval7_d: m 7-card hands resident in HBM (one per 64-bit word, card i = byte i) -> rank<<20|kickers words."""
    L.check(L.lib().pk_eval7_d(int(device), hands_d, int(m), out_d, int(bool(distinct))))


def make_hands(hands_d, m, seed=0x706F6B65726C, device=0):
    """pk_make_hands_d: synthetic distinct 7-card hands (first 7 cards of the RNG-spec deck of table_id = i)."""
    L.check(L.lib().pk_make_hands_d(int(device), int(seed), int(m), hands_d))


def time_eval7_stream(hands_d, m, out_d, distinct=True, reps=5, device=0):
    import ctypes as C
    ms = C.c_double(0.0)
    L.check(L.lib().pk_time_eval7_d(int(device), hands_d, int(m), out_d, int(bool(distinct)), int(reps), C.byref(ms)))
    return ms.value
