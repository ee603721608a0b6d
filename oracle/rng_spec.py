"""RNG specification of the build (TEST INFRASTRUCTURE -- never imported by the product).

The reference draws decks from the process-global stdlib MT19937
(`random.shuffle(self.deck)`, reference pokerl/game.py:424) and random-agent
actions from numpy's global legacy RNG (pokerl/agents/random.py:12-16,
examples/random_game.py:8).  Neither is reproducible per table, so "identical
RNG seeds" is defined by THIS counter-based spec instead:

  Philox4x32-10 (Salmon et al., Random123 constants), key = (seed_lo, seed_hi).

  Serials are 64-bit (a table never repeats a stream: 2^64 hands / steps), table ids 32-bit.

  deck of a table's `hand_serial`-th setup_hand() call (0-based, never reset):
      block b = philox(counter = (table_id, hand_serial mod 2^32, STREAM_DECK + b, hand_serial >> 32)) -> words w0..w3
      64-bit words X[2b] = w0 | w1<<32,  X[2b+1] = w2 | w3<<32
      draw i (i = 0,1,...) takes x = X[i // 9] when i % 9 == 0, then
          p = x * (52 - i)  (128-bit);  c_i = p >> 64;  x = p mod 2^64     (chained multiply-high,
          9 bounded draws per 64-bit word; relative bias <= 52!/43!/2^64 = 7.3e-5)
      card i of the deck = the c_i-th (0-based) not-yet-dealt card of the canonical deck
      (reference pokerl/cards.py:74-77: value[k] = ((k%4)<<4)|(k//4)); cards never dealt keep canonical order.
      Only deck[0 : 5+2N] is ever read by the game (game.py:278,388-389,394-395), so device code
      stops after draw 4+2N; draw i depends on nothing but (seed, table_id, hand_serial, i).

  random-agent action of a table's `step_serial`-th Game.step() (0-based, never reset):
      one block serves EIGHT consecutive steps: q = step_serial >> 3, j = step_serial & 7,
      w = philox(counter = (table_id, q mod 2^32, STREAM_ACTION, q >> 32));  r = 16-bit half (j & 1) of word w[j >> 1]
      n = popcount(valid_mask);  k = (r * n) >> 16;  action = k-th set bit (ascending)
      (n <= 7, so the bias of the bounded draw is at most 7/65536 per action.)

  deterministic agents (no random draw; a step played by one still advances step_serial):
      all-in agent (policy 1): always ALL_IN (6).
      call agent   (policy 2): CALL (2) if it is valid, else CHECK (1) if it is valid, else ALL_IN (6) -- the passive
      "calling station"; CALL is invalid exactly when high_bet >= credit (game.py:376), CHECK when high_bet != 0 (:375).

  per-seat agents (PokerGameEnv with a list of agents, envs/game_env.py:13-18): seat p plays policy nibble
  (seat_policies >> 4p) & 15 of a 64-bit word; 15 = the caller supplies that seat's actions.

Pure-Python ints here (small cases only); the C restatement lives in pokerl_oracle.c.
"""

M0 = 0xD2511F53
M1 = 0xCD9E8D57
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK32 = 0xFFFFFFFF

STREAM_DECK = 0x4445434B    # 'DECK'
STREAM_ACTION = 0x41435432  # 'ACT2'

DEFAULT_SEED = 0x706F6B65726C  # 'pokerl'

POLICY_RANDOM = 0
POLICY_ALLIN = 1
POLICY_CALL = 2
POLICY_EXTERNAL = 15


def seat_policies(policies):
    """Packs one policy per seat (seat 0 first) into the 64-bit word the oracle / the C ABI take."""
    w = 0
    for p, pol in enumerate(policies):
        w |= (int(pol) & 15) << (4 * p)
    return w


def philox4x32_10(ctr, key):
    """One Philox4x32-10 block. ctr: 4 u32, key: 2 u32 -> 4 u32."""
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> 32, p0 & MASK32
        hi1, lo1 = p1 >> 32, p1 & MASK32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & MASK32, lo1, (hi0 ^ c3 ^ k1) & MASK32, lo0
        k0 = (k0 + W0) & MASK32
        k1 = (k1 + W1) & MASK32
    return c0, c1, c2, c3


def seed_key(seed):
    return seed & MASK32, (seed >> 32) & MASK32


def canonical_deck_values():
    """Card.value of the reference's create_default_deck(), in order (cards.py:77)."""
    return [((i % 4) << 4) | (i // 4) for i in range(52)]


def deck_draws(seed, table_id, hand_serial, ndraws=52):
    """The bounded draws c_i in [0, 52-i)."""
    key = seed_key(seed)
    c = []
    x = 0
    words = []
    for i in range(ndraws):
        if i % 18 == 0:
            w = philox4x32_10((table_id & MASK32, hand_serial & MASK32, (STREAM_DECK + i // 18) & MASK32,
                               (hand_serial >> 32) & MASK32), key)
            words = [w[0] | (w[1] << 32), w[2] | (w[3] << 32)]
        if i % 9 == 0:
            x = words[(i // 9) % 2]
        p = x * (52 - i)
        c.append(p >> 64)
        x = p & 0xFFFFFFFFFFFFFFFF
    return c


def deck_permutation(seed, table_id, hand_serial, ndraws=52):
    """Index permutation p such that deck[i] = canonical[p[i]]."""
    remaining = list(range(52))
    perm = [remaining.pop(c) for c in deck_draws(seed, table_id, hand_serial, ndraws)]
    return perm + remaining


def pick_action(seed, table_id, step_serial, valid_mask_bits, policy=POLICY_RANDOM):
    """Action of the synthetic agents. valid_mask_bits: bit a set iff action a valid."""
    if policy == POLICY_ALLIN:
        return 6
    if policy == POLICY_CALL:
        return 2 if (valid_mask_bits >> 2) & 1 else (1 if (valid_mask_bits >> 1) & 1 else 6)
    q, j = step_serial >> 3, step_serial & 7
    w = philox4x32_10((table_id & MASK32, q & MASK32, STREAM_ACTION, (q >> 32) & MASK32), seed_key(seed))
    r = (w[j >> 1] >> (16 * (j & 1))) & 0xFFFF
    n = bin(valid_mask_bits).count("1")
    k = (r * n) >> 16
    for a in range(7):
        if (valid_mask_bits >> a) & 1:
            if k == 0:
                return a
            k -= 1
    raise AssertionError("empty valid mask")
