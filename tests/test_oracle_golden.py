"""CPU oracle (oracle/pokerl_oracle.c) pinned against vectors captured from the imported reference."""
import json
import os

import numpy as np
import pytest

import golden_util as GU
from oracle import loader as O
from oracle import rng_spec as R


def make_oracle(meta):
    cfg = meta["cfg"]
    return O.OracleGame(meta["tables"], meta["n"], cfg["start_credits"], cfg["big_blind"], cfg["small_blind"],
                        seed=meta["seed"], table_id_base=meta["table_id_base"])


def test_philox_known_answers():
    # Random123 kat_vectors: philox4x32 10 rounds
    L = O.lib()
    out = np.zeros(4, np.uint32)
    for ctr, key, exp in [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]:
        L.orc_philox4x32_10(np.array(ctr, np.uint32), np.array(key, np.uint32), out)
        assert tuple(int(x) for x in out) == exp
        assert R.philox4x32_10(ctr, key) == exp


def test_deck_spec_c_vs_python():
    L = O.lib()
    canon = R.canonical_deck_values()
    out = np.zeros(52, np.uint8)
    for seed, tid, hs in [(R.DEFAULT_SEED, 0, 0), (R.DEFAULT_SEED, 65535, 17), (1, 524287, 4000000000), (2**63 + 5, 3, 9)]:
        L.orc_deck(seed, tid, hs, out)
        perm = R.deck_permutation(seed, tid, hs)
        assert sorted(perm) == list(range(52))
        assert [canon[i] for i in perm] == out.tolist()
        # the prefix the device computes (5+2N draws) equals the full deck's prefix
        for n in (2, 6, 9):
            k = 5 + 2 * n
            assert R.deck_permutation(seed, tid, hs, ndraws=k)[:k] == perm[:k]


def test_np_sum_order_matches_numpy():
    L = O.lib()
    rng = np.random.default_rng(5)
    for n in range(1, 17):
        for _ in range(400):
            a = rng.random(n) * rng.choice([1.0, 1e3, 1e-3, 1e7], n)
            assert L.orc_np_sum(a, n) == np.sum(a)


def test_judger_known_answers_of_reference_tests():
    with open(os.path.join(GU.GOLDEN, "judger_kat.json")) as f:
        kat = json.load(f)
    for group in ("kat", "quirks"):
        for case in kat[group]:
            cards = np.array(case["values"], np.uint8)
            rank, kick, nk = O.eval_hands(cards.reshape(1, 7))
            exp_kick = 0
            for k in case["kickers"]:
                exp_kick = (exp_kick << 4) | k
            assert (int(rank[0]), int(kick[0]), int(nk[0])) == (case["rank"], exp_kick, len(case["kickers"])), case
    for case in kat["compare_kat"]:
        vals = np.array(case["values"], np.uint8)
        rank, kick, _ = O.eval_hands(vals)
        assert O.compare_rankings(rank, kick).tolist() == case["onehot"]


def test_judger_vectors():
    z = np.load(os.path.join(GU.GOLDEN, "judger_vectors.npz"))
    rank, kick, nk = O.eval_hands(z["eval_cards"], z["eval_ncards"])
    assert np.array_equal(rank, z["eval_rank"])
    assert np.array_equal(kick, z["eval_kick"])
    assert np.array_equal(nk, z["eval_nkick"])
    for i in range(len(z["cr_n"])):
        n = int(z["cr_n"][i])
        got = O.compare_rankings(z["cr_rank"][i, :n], z["cr_kick"][i, :n])
        assert np.array_equal(got, z["cr_onehot"][i, :n]), i


@pytest.mark.parametrize("name", GU.GAME_SETS)
def test_game_trajectories(name):
    GU.replay_game(make_oracle, name)


@pytest.mark.parametrize("name", GU.DIGEST_SETS)
def test_game_digests(name):
    GU.replay_digest(make_oracle, name)


@pytest.mark.parametrize("name", GU.ENV_SETS)
def test_env_trajectories(name):
    GU.replay_env(make_oracle, name)


@pytest.mark.parametrize("name", GU.VIEW_SETS)
def test_reference_state_views(name):
    """The reference's own StateView.__getstate__() tuples, get_valid_actions(p) of every seat, pot / high_bet /
    game_over (tests/golden/views_*.json, recorded from the imported reference after every step) against the oracle's
    state: pins the observation contract on the CPU side."""
    meta = GU.load_json(name)
    o = make_oracle(meta)
    o.reset(dealer=meta.get("dealer", 0))
    n, T = meta["n"], meta["tables"]
    fh = float.fromhex
    for s in range(meta["steps"]):
        acts = np.array(meta["actions"][s], np.int32)
        assert np.array_equal(o.pick_actions(meta["policy"]), acts)
        flags, err = o.step(acts)
        assert not err.any() and flags.tolist() == meta["flags"][s]
        snap = o.snapshot()
        masks = [o.valid_actions_for(p) for p in range(n)]
        for t in range(T):
            rec = meta["views"][s][t]
            act = rec["active"]
            assert act["player"] == snap["active"][t] and act["turn"] == snap["turn"][t] and act["num_players"] == n
            assert [fh(x) for x in act["credits"]] == snap["credits"][t].tolist()
            assert [fh(x) for x in act["bets"]] == snap["bets"][t].tolist()
            assert [fh(x) for x in act["pending_bets"]] == snap["pending"][t].tolist()
            assert fh(act["minimum_raise_value"]) == snap["min_raise"][t]
            nvis = 0 if snap["turn"][t] == 0 else snap["turn"][t] + 2
            assert act["community_cards"] == snap["cards"][t, :nvis].tolist()                       # game.py:266-278
            a = int(snap["active"][t])
            assert act["player_cards"] == snap["cards"][t, 5 + 2 * a:7 + 2 * a].tolist()          # game.py:385-389
            assert [int(fh(x)) for x in act["valid_actions"]] == [(int(snap["valid"][t]) >> k) & 1 for k in range(7)]
            for p in range(n):
                assert [int(fh(x)) for x in rec["valid_for"][p]] == [(int(masks[p][t]) >> k) & 1 for k in range(7)], (s, t, p)
                who = p if p else a                                                             # `player or active_player`, game.py:122
                assert rec["per_player"][p]["player"] == who
                assert rec["per_player"][p]["player_cards"] == snap["cards"][t, 5 + 2 * who:7 + 2 * who].tolist()
            assert fh(rec["pot"]) == np.sum(snap["bets"][t]) and fh(rec["high_bet"]) == np.max(snap["pending"][t])
            assert rec["game_over"] == bool((snap["states"][t] != 4).sum() == 1)
        over = (flags & 1).astype(np.uint8)
        if over.any():
            o.reset(mask=over)


def test_invalid_action_leaves_state_untouched():
    g = O.OracleGame(4, 3)
    g.reset()
    before = g.snapshot()
    flags, err = g.step(np.array([1, 7, -1, 1], np.int32))  # CHECK invalid preflop (high_bet=2); 7/-1 out of range
    assert err.tolist() == [O.ERR_INVALID_ACTION] * 4
    after = g.snapshot()
    for k in GU.SNAP_FIELDS:
        assert GU.bits_equal(before[k], after[k])


def test_eval7_exhaustive_digest_vs_reference():
    """All C(52,7) hands: the C oracle reproduces the digest computed from the imported reference
    (tests/golden/eval7_digest.json, made by tests/golden/make_eval_digest.py)."""
    from concurrent.futures import ThreadPoolExecutor
    gold = GU.load_json("eval7_digest")
    O.lib()
    chunks = [(0, 2), (2, 4), (4, 7), (7, 11), (11, 16), (16, 24), (24, 52)]
    with ThreadPoolExecutor(len(chunks)) as ex:  # ctypes releases the GIL
        parts = list(ex.map(lambda c: O.eval7_digest(*c), chunks))
    per_first = np.zeros(52, np.uint64)
    counts = np.zeros(11, np.uint64)
    for pf, c in parts:
        per_first += pf
        counts += c
    assert int(counts.sum()) == gold["hands"]
    assert counts.tolist() == gold["category_counts"]
    assert ["%016x" % int(x) for x in per_first] == gold["per_first_card"]
    assert "%016x" % (int(per_first.astype(object).sum()) % (1 << 64)) == gold["digest"]


@pytest.mark.parametrize("name", GU.ALIAS_SETS)
def test_reference_state_view_alias_rule(name):
    """What a StateView HELD across a Game.step shows in the reference (game.py:128-130 `self.credits = game.credits` ...), recorded from the
    imported reference by tests/golden/make_golden.py (ALIAS_SETS): credits and bets are the game's live arrays (only ever mutated in place);
    pending_bets is live until the next setup_hand rebinds the game's attribute (game.py:445), after which the view keeps the old array as
    game.py:438-440 left it -- zeros + the new hand's blinds, UNCLIPPED; every other field is a value or a fresh object and stays as it was.
    pokerl_amd.StateView is a snapshot instead (the GPU twin of this test pins that and the replacement rule)."""
    m = GU.load_json(name)
    n, cfg = m["n"], m["cfg"]
    across = 0
    for row in m["held"]:
        for h in row:
            a, b, live = h["at_creation"], h["after_step"], h["live"]
            assert b["credits"] == live["credits"] and b["bets"] == live["bets"]
            for k in ("player", "valid_actions", "num_players", "turn", "player_cards", "community_cards", "minimum_raise_value"):
                assert a[k] == b[k], k
            if h["setup_hands"] == 0:
                assert b["pending_bets"] == live["pending_bets"]
            elif h["setup_hands"] == 1:
                frozen = [0.0] * n
                frozen[live["bb"]] = float(cfg["big_blind"]); frozen[live["sb"]] = float(cfg["small_blind"])     # (:440: the small blind is written last)
                assert b["pending_bets"] == [x.hex() for x in frozen]
                across += 1
    assert across > 50
