"""Backend-agnostic replay of the golden fixtures (tests/golden/, captured from the imported reference).

A backend exposes: reset(mask=None), step(actions)->(flags, err), snapshot()->dict of
arrays stacked over tables, pick_actions(policy), env_reset(mask, opp_policy),
env_step(actions, opp_policy)->(reward, done, hand, err).  Both the CPU oracle
(oracle/loader.OracleGame) and the HIP product adapter implement it.
"""
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SNAP_FIELDS = ("active", "turn", "dealer", "sb", "bb", "hand", "states", "credits", "bets",
               "pending", "payoffs", "min_raise", "cards", "srank", "skick", "valid",
               "hand_serial", "step_serial")
F64_FIELDS = ("credits", "bets", "pending", "payoffs", "min_raise")


def load_npz(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    return z, meta


def load_json(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


def bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    if a.dtype != b.dtype:
        b = b.astype(a.dtype)
    if a.shape != b.shape:
        return False
    return a.tobytes() == b.tobytes()


def assert_snap(snap, gold, where, rows=None):
    """snap: backend snapshot (all tables); gold: dict field -> array over the same tables."""
    for k in SNAP_FIELDS:
        got = snap[k] if rows is None else snap[k][rows]
        exp = gold[k]
        if not bits_equal(exp, got):
            got = np.asarray(got).astype(exp.dtype)
            bad = np.argwhere(np.asarray(exp).reshape(got.shape) != got)
            raise AssertionError("%s: field %s differs (bit-exact compare)\n first bad idx %s\n expected %r\n got      %r"
                                 % (where, k, bad[:1].tolist(), np.asarray(exp)[tuple(bad[0])] if len(bad) else exp,
                                    got[tuple(bad[0])] if len(bad) else got))


def gold_snap(z, prefix, idx=None):
    return {k: (z[prefix + k] if idx is None else z[prefix + k][idx]) for k in SNAP_FIELDS}


def replay_game(make_backend, name, loaded=None):
    """name: fixture under tests/golden/; or loaded=(dict of arrays, meta) straight from the generator."""
    z, meta = loaded if loaded is not None else load_npz(name)
    b = make_backend(meta)
    sbase = meta.get("serial_base", [0, 0])
    if any(sbase):
        b.set_serials(sbase[0], sbase[1])   # a resumed RNG stream (64-bit serials)
    b.reset(dealer=meta.get("dealer", 0))
    assert_snap(b.snapshot(), gold_snap(z, "init_"), name + " init")
    ridx = z["reset_idx"]
    rpos = 0
    for s in range(meta["steps"]):
        acts = z["actions"][s].astype(np.int32)
        picked = b.pick_actions(meta["policy"])
        assert np.array_equal(picked, acts), "%s step %d: agent actions differ %s vs %s" % (name, s, picked, acts)
        flags, err = b.step(acts)
        assert np.array_equal(err, z["errs"][s]), "%s step %d: err %s vs %s" % (name, s, err, z["errs"][s])
        ok = err == 0
        assert np.array_equal(flags[ok], z["flags"][s][ok]), "%s step %d: flags %s vs %s" % (name, s, flags, z["flags"][s])
        assert_snap(b.snapshot(), gold_snap(z, "post_", s), "%s step %d" % (name, s))
        over = ((flags & 1) if meta.get("auto_reset", True) else np.zeros_like(flags)).astype(np.uint8)
        over[err != 0] = 1
        if over.any():
            b.reset(mask=over)
            snap = b.snapshot()
            for t in np.nonzero(over)[0]:
                assert tuple(ridx[rpos]) == (s, t)
                assert_snap(snap, gold_snap(z, "reset_", rpos), "%s reset after step %d table %d" % (name, s, t), rows=t)
                rpos += 1
    assert rpos == len(ridx)
    return meta


def snap_digest(snap):
    h = hashlib.sha256()
    for k in SNAP_FIELDS:
        h.update(np.ascontiguousarray(snap[k]).tobytes())
    return h.hexdigest()


def replay_digest(make_backend, name, use_rollout=False):
    meta = load_json(name)
    b = make_backend(meta)
    b.reset()
    every = meta["digest_every"]
    digests = []
    all_flags, all_actions = [], []
    for s in range(meta["steps"]):
        acts = b.pick_actions(meta["policy"])
        flags, err = b.step(acts)
        assert not err.any()
        all_flags.append(flags.copy())
        all_actions.append(acts.astype(np.int8))
        if (s + 1) % every == 0:
            digests.append(snap_digest(b.snapshot()))
        over = (flags & 1).astype(np.uint8)
        if over.any():
            b.reset(mask=over)
    assert digests == meta["digests"], "%s: first differing digest at block %d" % (
        name, next(i for i, (x, y) in enumerate(zip(digests, meta["digests"])) if x != y))
    assert hashlib.sha256(np.array(all_flags, np.uint8).tobytes()).hexdigest() == meta["flags_sha256"]
    assert hashlib.sha256(np.array(all_actions, np.int8).tobytes()).hexdigest() == meta["actions_sha256"]
    return meta


def replay_env(make_backend, name, loaded=None):
    z, meta = loaded if loaded is not None else load_npz(name)
    b = make_backend(meta)
    opp = meta["opp_policy"]
    b.env_reset(None, opp)
    assert_snap(b.snapshot(), gold_snap(z, "init_"), name + " init")
    ridx = z["reset_idx"]
    rpos = 0
    for s in range(meta["steps"]):
        acts = z["actions"][s].astype(np.int32)
        picked = b.pick_actions(meta["policy"])
        assert np.array_equal(picked, acts), "%s step %d: actions differ" % (name, s)
        reward, done, hand, err = b.env_step(acts, opp)
        assert not err.any()
        assert bits_equal(z["reward"][s], reward), "%s step %d reward %s vs %s" % (name, s, reward, z["reward"][s])
        assert np.array_equal(done, z["done"][s]), "%s step %d done" % (name, s)
        assert np.array_equal(hand, z["hand_over"][s]), "%s step %d hand" % (name, s)
        assert_snap(b.snapshot(), gold_snap(z, "post_", s), "%s step %d" % (name, s))
        if done.any():
            b.env_reset(done, opp)
            snap = b.snapshot()
            for t in np.nonzero(done)[0]:
                assert tuple(ridx[rpos]) == (s, t)
                assert_snap(snap, gold_snap(z, "reset_", rpos), "%s env reset after step %d table %d" % (name, s, t), rows=t)
                rpos += 1
    assert rpos == len(ridx)
    return meta


GAME_SETS = ["game_n2_random", "game_n6_random", "game_n9_random", "game_n6_allin", "game_n9_allin",
             "game_n4_example_cfg", "game_n3_percredits", "game_n10_random", "game_n2_noreset", "game_n3_noreset",
             "game_n5_zero_blinds", "game_n4_sb_gt_bb_fractional", "game_n7_blinds_gt_stacks", "game_n8_mixed_allin",
             "game_n6_serial_hi", "game_n8_argsort_tie", "game_n9_argsort_tie", "game_n6_hand_cap",
             "game_n12_random", "game_n15_random", "game_n14_mixed_allin",
             "game_n16_random", "game_n16_mixed_allin", "game_n16_argsort_tie"]
DIGEST_SETS = ["digest_n2_random", "digest_n6_random", "digest_n9_random", "digest_n9_allin", "digest_n6_shard1",
               "digest_n6_shard7", "digest_n15_random", "digest_n16_random"]
VIEW_SETS = ["views_n6_random", "views_n3_percredits", "views_n13_random", "views_n16_random"]
ALIAS_SETS = ["views_alias_n6_random", "views_alias_n3_percredits"]   # a StateView held across a step: what the reference's aliasing shows (game.py:128-130)
ENV_SETS = ["env_n4_random", "env_n6_random", "env_n6_vs_allin", "env_n2_random", "env_n5_percredits",
            "env_n9_random_hi_base", "env_n3_big_blinds_vs_allin", "env_n3_vs_call", "env_n4_mixed_opponents",
            "env_n6_mixed_percredits", "env_n2_call_vs_call", "env_n11_random", "env_n12_mixed_opponents",
            "env_n16_mixed_opponents"]
