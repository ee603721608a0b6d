"""GPU parity tests: the HIP path, called through the C ABI (ctypes), against (a) the golden vectors captured from the
imported reference and (b) the CPU oracle on seeded inputs.  Bit-exact everywhere (bytes of every f64 compared)."""
import math
import os

import numpy as np
import pytest

import golden_util as GU

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def HB():
    import pokerl_amd
    assert pokerl_amd.device_count() >= 1, "no MI355X visible: the HIP path cannot run (there is no fallback)"
    from hip_backend import HipBackend
    return HipBackend


@pytest.fixture(scope="module")
def O():
    from oracle import loader
    loader.lib()
    return loader


def assert_same(a, b, where):
    for k in GU.SNAP_FIELDS:
        if not GU.bits_equal(a[k], b[k]):
            x, y = np.asarray(a[k]), np.asarray(b[k]).astype(np.asarray(a[k]).dtype)
            bad = np.argwhere(x != y)
            raise AssertionError("%s: field %s differs at %s: oracle %r hip %r" % (
                where, k, bad[:1].tolist(), x[tuple(bad[0])] if len(bad) else x, y[tuple(bad[0])] if len(bad) else y))


# ------------------------------------------------------------------ golden vectors (reference-generated)
@pytest.mark.parametrize("name", GU.GAME_SETS)
def test_golden_game_trajectories(HB, name):
    GU.replay_game(HB.from_meta, name)


@pytest.mark.parametrize("name", GU.DIGEST_SETS)
def test_golden_game_digests(HB, name):
    GU.replay_digest(HB.from_meta, name)


@pytest.mark.parametrize("name", GU.ENV_SETS)
def test_golden_env_trajectories(HB, name):
    GU.replay_env(HB.from_meta, name)


def test_golden_judger_vectors(HB):
    import json
    import os
    import pokerl_amd
    from pokerl_amd import judger
    z = np.load(os.path.join(GU.GOLDEN, "judger_vectors.npz"))
    rank, kick, nk = judger.eval_hands(z["eval_cards"], z["eval_ncards"])
    assert np.array_equal(rank, z["eval_rank"])
    assert np.array_equal(kick, z["eval_kick"])
    assert np.array_equal(nk, z["eval_nkick"])
    for n in range(1, 11):
        rows = np.nonzero(z["cr_n"] == n)[0]
        got = judger.compare_rankings_batch(z["cr_rank"][rows, :n], z["cr_kick"][rows, :n])
        assert np.array_equal(got, z["cr_onehot"][rows, :n]), n
    with open(os.path.join(GU.GOLDEN, "judger_kat.json")) as f:
        kat = json.load(f)
    for group in ("kat", "quirks"):   # the reference's own tests/pokerl/test_judger.py cases, via the mirrored API
        for case in kat[group]:
            assert pokerl_amd.eval_hand(case["cards"].split()) == (case["rank"], case["kickers"]), case
    for case in kat["compare_kat"]:
        out = pokerl_amd.compare_hands([h.split() for h in case["hands"]])
        assert out[0] == case["onehot"] and out[1] == case["winners"]
        assert [[r, k] for r, k in out[2]] == case["rankings"]


@pytest.mark.parametrize("fast", [1, 0, 2, 3, 4], ids=["showdown_evaluator", "general_evaluator", "table_evaluator_of_the_streaming_kernel",
                                                      "eval_hands_dispatch_fast_path", "eval_hands_table_path"])
def test_eval7_exhaustive_digest(HB, fast):
    """All C(52,7) = 133 784 560 hands on the GPU against the digest computed from the imported reference, for both
    device evaluators (the bitmask one the showdown kernels use, the general multiset one of pk_eval_hands, and the
    LDS-table one of the streaming kernel pk_eval7_d)."""
    from pokerl_amd import judger
    gold = GU.load_json("eval7_digest")
    GOLD = np.uint64(0x9E3779B97F4A7C15)

    def mix64(z):
        z ^= z >> np.uint64(30); z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27); z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
        return z

    per_first = [0] * 52
    counts = np.zeros(11, np.int64)
    idx0 = 0
    with np.errstate(over="ignore"):
        for a in range(52):
            for b in range(a + 1, 52):
                n = math.comb(51 - b, 5)
                if n == 0:
                    continue
                v = judger.eval7_prefix(a, b, fast).astype(np.uint64)
                assert len(v) == n
                idx = np.arange(n, dtype=np.uint64) + np.uint64(idx0)
                per_first[a] = (per_first[a] + int(np.sum(mix64(v ^ (idx * GOLD)), dtype=np.uint64))) % (1 << 64)
                counts += np.bincount((v >> np.uint64(20)).astype(np.int64), minlength=11)
                idx0 += n
    assert idx0 == gold["hands"]
    assert counts.tolist() == gold["category_counts"]
    assert ["%016x" % x for x in per_first] == gold["per_first_card"]
    assert "%016x" % (sum(per_first) % (1 << 64)) == gold["digest"]


# ------------------------------------------------------------------ HIP vs CPU oracle, seeded, larger
CASES = [  # (tables, players, policy, steps, seed, table_id_base, start_credits, bb, sb)
    (4096, 2, 0, 300, 0x706F6B65726C, 0, 100, 2, 1),       # BASELINE config 2
    (2048, 6, 0, 300, 0x706F6B65726C, 0, 100, 2, 1),
    (1024, 9, 1, 150, 0x706F6B65726C, 0, 100, 2, 1),       # config 5 policy (all-in), smaller T
    (777, 3, 0, 300, 42, 123456, [30, 100, 5], 4, 2),       # ragged T, per-seat credits
    (512, 10, 0, 200, 7, 4000000000, 50, 3, 2),            # max seats, table ids near 2^32
    (64, 4, 0, 400, 9, 0, 1000, 40, 20),                   # examples/random_game.py:9 config
    (1, 2, 0, 500, 3, 0, 100, 2, 1),                       # BASELINE config 1 shape (single table)
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "T%d_N%d_p%d" % (c[0], c[1], c[2]))
def test_lockstep_vs_oracle(HB, O, case):
    T, N, policy, steps, seed, base, sc, bb, sb = case
    o = O.OracleGame(T, N, sc, bb, sb, seed=seed, table_id_base=base)
    h = HB(T, N, sc, bb, sb, seed=seed, table_id_base=base)
    o.reset(); h.reset()
    assert_same(o.snapshot(), h.snapshot(), "after reset")
    for s in range(steps):
        a = o.pick_actions(policy)
        assert np.array_equal(a, h.pick_actions(policy)), "step %d: in-kernel agent picks differ" % s
        fo, eo = o.step(a)
        fh, eh = h.step(a)
        assert np.array_equal(fo, fh), "step %d flags" % s
        assert np.array_equal(eo, eh), "step %d terr" % s
        if s % 7 == 0 or s == steps - 1:
            assert_same(o.snapshot(), h.snapshot(), "step %d" % s)
        over = (fo & 1).astype(np.uint8)
        if over.any():
            o.reset(mask=over); h.reset(mask=over)
    assert_same(o.snapshot(), h.snapshot(), "final")


@pytest.mark.parametrize("T,N,policy,K", [(65536, 6, 0, 96), (4096, 2, 0, 400), (65536, 9, 1, 40), (8192, 6, 1, 100)])
def test_fused_rollout_vs_oracle(HB, O, T, N, policy, K):
    """BASELINE configs 2/3/5 at full table count: the fused K-step kernel (in-kernel agents, auto-reset) leaves every
    table in exactly the state the scalar oracle reaches, and counts the same steps/hands/evals/games."""
    o = O.OracleGame(T, N)
    h = HB(T, N)
    o.reset(); h.reset()
    co, eo = o.rollout(K, policy, True)
    ch = h.rollout(K, policy, True, fused=True)
    assert eo == 0
    assert co.tolist() == ch.tolist(), "counters (steps, hands, evals, games)"
    assert_same(o.snapshot(), h.snapshot(), "after fused rollout")
    # second launch continues from HBM state; unfused (K launches) must match fused bit for bit
    h2 = HB(T, N)
    h2.reset()
    c2 = h2.rollout(K, policy, True, fused=False)
    assert c2.tolist() == ch.tolist()
    assert_same(h.snapshot(), h2.snapshot(), "fused vs unfused")


def test_config3_all_eight_shards(HB, O):
    """BASELINE configs[3] (524 288 tables x 6 seats over 8 GPUs) on one GPU: each rank's full 65 536-table shard
    (table_id_base = r * 65 536) against the oracle, then the whole 524 288-table batch in ONE handle: its counters are
    the sum of the shards' and its state is their concatenation (placement invariance of the global table ids)."""
    T, N, K = 65536, 6, 32
    total = np.zeros(4, np.uint64)
    digests = []
    for r in range(8):
        o = O.OracleGame(T, N, table_id_base=r * T)
        h = HB(T, N, table_id_base=r * T)
        o.reset(); h.reset()
        co, eo = o.rollout(K, 0, True)
        ch = h.rollout(K, 0, True)
        assert eo == 0 and co.tolist() == ch.tolist(), "shard %d counters" % r
        snap = h.snapshot()
        assert_same(o.snapshot(), snap, "shard %d" % r)
        total += ch
        digests.append({k: np.ascontiguousarray(snap[k]).copy() for k in ("credits", "payoffs", "cards", "states", "step_serial", "hand_serial")})
        h.g.close()
    whole = HB(8 * T, N)
    whole.reset()
    cw = whole.rollout(K, 0, True)
    assert cw.tolist() == total.tolist() and cw[0] == 8 * T * K
    g = whole.g
    for k, got in (("credits", g.credits), ("payoffs", g.payoffs), ("cards", g.deck), ("states", g.player_states),
                   ("step_serial", g.step_serial), ("hand_serial", g.hand_serial)):
        assert GU.bits_equal(np.concatenate([d[k] for d in digests]), got), k
    whole.g.close()


def test_deferred_launches_are_invisible(HB, O):
    """Asynchronous fused rollouts may defer their stragglers to later launches (State::owed, steps in flight across
    launches).  Whatever the split and the tuning, an observer sees exactly the requested steps: many short deferred
    launches == one complete launch == the oracle, at BASELINE's headline size and on a ragged batch."""
    for T, N, policy, auto in [(65536, 6, 0, True), (1000, 9, 1, True), (3000, 3, 0, True), (700, 2, 0, False)]:
        # coalesce: asynchronous calls that arrive while two launches are in flight are merged on the host into launches
        # of up to that many steps (0: never; -1: the library default)
        for endk, park, plan, coalesce in [(64, 40, [20] * 12, 0), (48, 40, [1, 7, 32, 200], -1), (33, 20, [60] * 4, 0), (1, 40, [240], -1),
                                           (64, 64, [3] * 80, 0), (48, 28, [6] * 40, 48), (48, 28, [20] * 12, 512), (40, 32, [3] * 80, 7)]:
            o = O.OracleGame(T, N, seed=5)
            o.reset()
            co = np.zeros(4, np.uint64)
            # with auto-reset the split is invisible; without it a table that reports an error stops for the rest of
            # ITS call only, so the oracle makes the same calls
            for k in ([240] if auto else plan):
                co += o.rollout(k, policy, auto)[0]
            h = HB(T, N, seed=5)
            h.g.set_tuning(park, endk)
            if coalesce >= 0:
                h.g.set_coalesce(coalesce)
            h.reset()
            h.g.launch_stats(reset=True)
            for k in plan:
                h.g.rollout(k, policy, auto, True, counters=False)      # asynchronous, may defer, may be held by the host
            c = h.rollout(0, policy, auto)                              # completes everything, fetches the counters
            assert c.tolist() == co.tolist(), (T, N, endk, park, coalesce)
            st = h.g.launch_stats()
            assert st["steps"] == sum(plan) and st["launches"] <= len(plan) + 1, st
            if coalesce == 0 or not auto:
                assert st["max"] == max(plan) and st["launches"] >= len(plan), st       # one launch per call
            elif T == 65536 and coalesce > plan[0]:
                # a launch of the whole chip lasts far longer than a ctypes call: calls pile up behind the two launches in flight
                assert st["max"] > plan[0] and st["launches"] < len(plan), st
                assert st["max"] < coalesce + plan[0], st
            snap = h.snapshot()
            assert_same(o.snapshot(), snap, "T=%d N=%d endk=%d park=%d coalesce=%d" % (T, N, endk, park, coalesce))
            if auto:
                assert (snap["step_serial"] == 240).all()
            h.g.close()
    # deferred work is completed by ANY observer, not only by counters: a getter right after asynchronous launches
    h = HB(4096, 6); h.reset()
    o = O.OracleGame(4096, 6); o.reset()
    for _ in range(10):
        h.g.rollout(17, 0, True, True, counters=False)
    o.rollout(170, 0, True)
    assert GU.bits_equal(o.f64(0), h.g.credits)        # first call after the launches is a getter
    # ... and pk_wait_event launches the steps the host still holds back (they were requested before the wait), without changing
    # what any observer sees
    from pokerl_amd.hipmem import DeviceEvent
    h2 = HB(65536, 6, seed=8); h2.reset()
    o2 = O.OracleGame(65536, 6, seed=8); o2.reset()
    ev = DeviceEvent()
    for r in range(6):
        for _ in range(8):
            h2.g.rollout(20, 0, True, True, counters=False)             # piles up behind the two launches in flight
        held_before = h2.g.launch_stats()["steps"]
        h.g.record_event(ev.handle)                                     # an event of ANOTHER handle's stream (complete or about to be)
        h2.g.wait_event(ev.handle)
        assert h2.g.launch_stats()["steps"] >= held_before              # (nothing is lost; what was held has been launched)
    c2 = h2.rollout(0, 0, True)
    co2 = o2.rollout(6 * 8 * 20, 0, True)[0]
    assert c2.tolist() == co2.tolist()
    assert_same(o2.snapshot(), h2.snapshot(), "held steps launched by pk_wait_event")
    h2.g.close()
    h.g.rollout(5, 0, True, True, counters=False)
    o.rollout(5, 0, True)
    acts = o.pick_actions(0)
    assert np.array_equal(acts, h.pick_actions(0))     # ... or pick / step
    fo, eo = o.step(acts); fh, eh = h.step(acts)
    assert np.array_equal(fo, fh) and np.array_equal(eo, eh)
    # a change of agents flushes first: owed steps keep the policy they were requested with
    o2 = O.OracleGame(4096, 6); h2 = HB(4096, 6)
    o2.reset(); h2.reset()
    h2.g.rollout(30, 0, True, True, counters=False); h2.g.rollout(30, 1, True, True, counters=False)
    o2.rollout(30, 0, True); o2.rollout(30, 1, True)
    assert_same(o2.snapshot(), h2.snapshot(), "policy switch between deferred launches")


def test_no_autoreset_and_assert_path(HB, O):
    """Without auto-reset the lone survivor keeps acting; its FOLD trips game.py:473 -> PK_TERR_NO_WINNER."""
    T, N = 512, 2
    o = O.OracleGame(T, N, 20, 2, 1, seed=5)
    h = HB(T, N, 20, 2, 1, seed=5)
    o.reset(); h.reset()
    seen = 0
    for s in range(200):
        a = o.pick_actions(0)
        fo, eo = o.step(a)
        fh, eh = h.step(a)
        assert np.array_equal(fo[eo == 0], fh[eo == 0]) and np.array_equal(eo, eh)
        seen += int((eo == 2).sum())
        assert_same(o.snapshot(), h.snapshot(), "step %d" % s)
        bad = (eo != 0).astype(np.uint8)
        if bad.any():
            o.reset(mask=bad); h.reset(mask=bad)
    assert seen > 0


def test_env_vs_oracle(HB, O):
    for T, N, opp in [(1024, 6, 0), (512, 4, 1), (256, 2, 0)]:
        o = O.OracleGame(T, N, seed=77)
        h = HB(T, N, seed=77)
        o.env_reset(None, opp); h.env_reset(None, opp)
        assert_same(o.snapshot(), h.snapshot(), "env reset")
        for s in range(120):
            a = o.pick_actions(0)
            ro, do, ho, eo = o.env_step(a, opp)
            rh, dh, hh, eh = h.env_step(a, opp)
            assert GU.bits_equal(ro, rh) and np.array_equal(do, dh) and np.array_equal(ho, hh) and np.array_equal(eo, eh)
            if s % 5 == 0:
                assert_same(o.snapshot(), h.snapshot(), "env step %d" % s)
            if do.any():
                o.env_reset(do, opp); h.env_reset(do, opp)
        assert_same(o.snapshot(), h.snapshot(), "env final")


def test_sharding_invariance(HB):
    """Tables split over two handles (as two GPUs would) reproduce the single-handle result exactly."""
    from pokerl_amd import shard_tables
    T, N, K = 4096, 6, 120
    whole = HB(T, N)
    whole.reset()
    cw = whole.rollout(K, 0)
    sw = whole.snapshot()
    parts, counters = [], np.zeros(4, np.uint64)
    for r in range(2):
        n, base = shard_tables(T, r, 2)
        h = HB(n, N, table_id_base=base)
        h.reset()
        counters += h.rollout(K, 0)
        parts.append(h.snapshot())
    assert counters.tolist() == cw.tolist()
    for k in GU.SNAP_FIELDS:
        assert GU.bits_equal(sw[k], np.concatenate([p[k] for p in parts])), k


def test_api_error_contract(HB):
    import pokerl_amd
    g = pokerl_amd.VecGame(8, num_players=3)
    g.reset()
    creds = g.credits.copy()
    with pytest.raises(ValueError, match="invalid move"):   # CHECK is invalid pre-flop (high_bet == 2), game.py:649-651
        g.step(np.full(8, 1, np.int32))
    with pytest.raises(NotImplementedError):                # game.py:700
        g.step(np.full(8, 2.0))
    assert GU.bits_equal(creds, g.credits)
    assert (g.step_serial == 0).all()
    over, hand, turn, terr = g.step(np.array([2, 1, 2, 9, -1, 2, 2, 2], np.int32), strict=False)
    assert terr.tolist() == [0, 1, 0, 1, 1, 0, 0, 0]
    assert g.step_serial.tolist() == [1, 0, 1, 0, 0, 1, 1, 1]
    onehot, gens = g.get_valid_actions()
    assert onehot.shape == (8, 7) and onehot.dtype == np.float64
    assert [list(x) for x in gens][0] == list(np.nonzero(onehot[0])[0])
    obs = g.observations
    assert obs.shape == (8, 17 + 9)
    assert np.array_equal(obs[:, 0], g.active_player) and np.array_equal(obs[:, 3:10], onehot)
    assert (obs[:, 12:17] == -1).all()   # turn 0: no community card visible (game.py:278)
    assert np.array_equal(obs[:, 17:20], g.credits)


def test_observation_and_cards(HB, O):
    import pokerl_amd
    T, N = 256, 6
    g = pokerl_amd.VecGame(T, num_players=N)
    g.reset()
    g.rollout(37, 0)
    deck, turn, active = g.deck, g.turn, g.active_player
    obs = g.observations
    for t in range(0, T, 17):
        nvis = 0 if turn[t] == 0 else turn[t] + 2
        assert obs[t, 12:12 + nvis].tolist() == deck[t, :nvis].tolist()
        assert (obs[t, 12 + nvis:17] == -1).all()
        assert obs[t, 10:12].tolist() == deck[t, 5 + 2 * active[t]:7 + 2 * active[t]].tolist()
    assert np.array_equal(g.get_hand_for(active)[:, 5:], g.get_cards_of(active))
    views = g.state_views()                                   # the reference's observation object, per table
    credits = g.credits
    for t in range(0, T, 31):
        sv = views[t]
        assert sv.player == active[t] and sv.turn == turn[t] and sv.credit == credits[t, active[t]]
        assert [c.value for c in sv.player_cards] == deck[t, 5 + 2 * active[t]:7 + 2 * active[t]].tolist()
        assert len(sv.community_cards) == (0 if turn[t] == 0 else turn[t] + 2)
    # every dealt prefix holds distinct valid cards
    assert all(len(set(row)) == len(row) for row in deck.tolist())
    assert ((deck & 0xf) < 13).all() and ((deck >> 4) < 4).all()


def test_all_player_counts_fused_vs_oracle(HB, O):
    """Every template instantiation (N = 2..16), ragged table count, both policies."""
    for N in range(2, 17):
        for policy in (0, 1):
            T, K = 1000 + N, 150
            o = O.OracleGame(T, N, seed=N * 17 + policy)
            h = HB(T, N, seed=N * 17 + policy)
            o.reset(); h.reset()
            co, eo = o.rollout(K, policy, True)
            ch = h.rollout(K, policy, True, fused=True)
            assert co.tolist() == ch.tolist(), (N, policy)
            assert_same(o.snapshot(), h.snapshot(), "N=%d policy=%d" % (N, policy))


@pytest.mark.parametrize("occ3", ["0", "1"])
def test_both_rollout_kernels_vs_oracle(HB, O, monkeypatch, occ3):
    """k_rollout (registers capped at 256) and k_rollout_occ3 (capped for three waves per SIMD; spills to scratch at N >= 8)
    are picked by batch shape; PK_OCC3 forces either, and both must be bit-exact for every seat count."""
    monkeypatch.setenv("PK_OCC3", occ3)
    for N, T, policy, K in [(6, 5000, 0, 150), (9, 3000, 1, 60), (10, 2000, 0, 120), (8, 140000, 0, 10), (2, 70000, 0, 40)]:
        o = O.OracleGame(T, N, seed=1000 + N)
        h = HB(T, N, seed=1000 + N)
        o.reset(); h.reset()
        for k in (K // 3, K - K // 3):
            h.g.rollout(k, policy, True, True, counters=False)        # deferred launches too
        co, _ = o.rollout(K, policy, True)
        assert co.tolist() == h.rollout(0, policy, True).tolist(), (N, T, occ3)
        assert_same(o.snapshot(), h.snapshot(), "N=%d T=%d occ3=%s" % (N, T, occ3))
        h.g.close()


@pytest.mark.parametrize("tab", ["0", "1"])
def test_rollout_with_the_table_evaluator_vs_oracle(HB, O, monkeypatch, tab):
    """k_rollout_tab (round 6: up to six seats, batches of at most one wave per SIMD, launches of >= 16 steps -- the showdown hands ranked by the
    table-driven evaluator out of the wave's own LDS copy of the table, rankings returned through the queue slots) against k_rollout
    (PK_ROLLOUT_TAB=0: never) and the oracle: every state byte incl. the rankings of the last showdown, full-width ragged waves, per-seat stacks,
    deferred / split launches, launches below and above the step threshold, all seat counts the kernel exists for; the all-in agents (every hand a
    showdown: BASELINE configs[4]) take k_rollout_allin_tab up to ten seats (no action ring in LDS, so the table fits beside a wider queue)."""
    monkeypatch.setenv("PK_ROLLOUT_TAB", tab)
    for N, T, policy, K, stacks in [(6, 65536, 0, 96, 100), (6, 65536 - 37, 0, 64, [50, 100, 20, 100, 80, 100]), (2, 65536, 0, 80, 100), (3, 40000, 0, 90, 100),
                                    (4, 5000, 0, 150, [3, 100, 5, 40]), (5, 1500, 0, 300, 100), (6, 8192, 1, 100, 100), (6, 70000, 0, 40, 100),
                                    (9, 65536, 1, 40, 100), (10, 20000, 1, 60, 100), (7, 3000, 1, 80, [5, 100, 30, 100, 100, 2, 60]), (2, 65536, 1, 50, 100)]:
        o = O.OracleGame(T, N, stacks, seed=4400 + N)
        h = HB(T, N, stacks, seed=4400 + N)
        o.reset(); h.reset()
        for k in (K // 3, 7, K - K // 3 - 7):                        # (a 7-step launch in between: below the kernel's threshold)
            h.g.rollout(k, policy, True, True, counters=False)
        co, _ = o.rollout(K, policy, True)
        assert co.tolist() == h.rollout(0, policy, True).tolist(), (N, T, tab)
        assert_same(o.snapshot(), h.snapshot(), "N=%d T=%d PK_ROLLOUT_TAB=%s" % (N, T, tab))
        h.g.close()


def test_hand_cap_rule(HB, O):
    """start_credits = 0: every seat is re-dealt all-in with no chips for ever -- the reference's Game.step would never
    return (docs/history.md section 2).  step() reports PK_TERR_HAND_CAP; a rollout with auto_reset treats it as a finished game."""
    T, N = 256, 3
    o = O.OracleGame(T, N, 0, 2, 1)
    h = HB(T, N, 0, 2, 1)
    o.reset(); h.reset()
    a = o.pick_actions(1)
    fo, eo = o.step(a)
    fh, eh = h.step(a)
    assert (eo == 4).all() and np.array_equal(eo, eh)
    assert_same(o.snapshot(), h.snapshot(), "capped step")
    o.reset(); h.reset()
    co, _ = o.rollout(5, 1, True)
    ch = h.rollout(5, 1, True)
    assert co.tolist() == ch.tolist() and ch[0] == 5 * T and ch[3] == 5 * T   # 5 steps, each a "finished game"
    assert_same(o.snapshot(), h.snapshot(), "capped rollout")


def test_device_pointer_step(HB, O):
    """pk_step_d: actions / flags / terr resident in HBM (buffers from hipMalloc through ctypes), asynchronous."""
    import ctypes as C
    import pokerl_amd
    from pokerl_amd import _lib as L
    hip = C.CDLL("libamdhip64.so")
    T, N = 4096, 6
    g = pokerl_amd.VecGame(T, num_players=N)
    o = O.OracleGame(T, N)
    g.reset(); o.reset()
    d_act, d_flags, d_terr = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(d_act), T * 4) == 0 and hip.hipMalloc(C.byref(d_flags), T) == 0 and hip.hipMalloc(C.byref(d_terr), T) == 0
    lib = L.lib()
    for s in range(40):
        a = o.pick_actions(0)
        fo, eo = o.step(a)
        assert hip.hipMemcpy(d_act, a.ctypes.data_as(C.c_void_p), T * 4, 1) == 0          # H2D
        L.check(lib.pk_step_d(g._h, d_act, d_flags, d_terr), g._h)
        g.sync()
        fl = np.zeros(T, np.uint8); te = np.zeros(T, np.uint8)
        assert hip.hipMemcpy(fl.ctypes.data_as(C.c_void_p), d_flags, T, 2) == 0            # D2H
        assert hip.hipMemcpy(te.ctypes.data_as(C.c_void_p), d_terr, T, 2) == 0
        assert np.array_equal(fl, fo) and np.array_equal(te, eo)
        over = (fo & 1).astype(np.uint8)
        if over.any():
            o.reset(mask=over); g.reset(mask=over)
    assert GU.bits_equal(o.f64(0), g.credits) and GU.bits_equal(o.f64(3), g.payoffs)
    for p in (d_act, d_flags, d_terr):
        hip.hipFree(p)


def test_device_resident_game_loop(HB, O):
    """The rollout loop of examples/random_game.py:8-12 with every buffer in HBM, through the host mirror: VecGame.pick_actions_d +
    step_d + reset_d(flags, GAME_OVER) (pk_pick_actions_d / pk_step_d / pk_reset_d: the step's own flags are the reset mask), and
    step_d(auto_reset=True) (pk_step_auto_d: the reset inside the step's launch) -- both against the oracle's step + reset, every
    state byte, at full-width waves with a ragged tail and at a small batch."""
    import pokerl_amd
    from pokerl_amd import _lib as L
    from pokerl_amd.hipmem import DeviceBuffer
    for T, N, steps in ((65536 + 19, 6, 60), (1500, 9, 260), (4096, 2, 200)):
        hb = HB(T, N, seed=977)
        hb2 = HB(T, N, seed=977)
        g, g2 = hb.g, hb2.g
        o = O.OracleGame(T, N, seed=977)
        o.reset(); g.reset(); g2.reset()
        act, flags, terr = DeviceBuffer(T * 4), DeviceBuffer(T), DeviceBuffer(T)
        act2, flags2, terr2 = DeviceBuffer(T * 4), DeviceBuffer(T), DeviceBuffer(T)
        resets = 0
        for s in range(steps):
            a = o.pick_actions(0)
            fo, eo = o.step(a)
            over = ((fo & 1) | ((eo & 4) >> 2)).astype(np.uint8)      # game over, or a step the reference would never leave
            g.pick_actions_d(act, 0)
            g.step_d(act, flags, terr)
            g.sync()                                                  # (the handle's stream is non-blocking: hipMemcpy does not wait for it)
            assert np.array_equal(act.download(np.int32, T), a), s
            fl, te = flags.download(np.uint8, T), terr.download(np.uint8, T)
            assert np.array_equal(fl, fo) and np.array_equal(te, eo), s
            g.reset_d(flags, L.FLAG_GAME_OVER)                        # HAND_OVER / TURN_OVER bits alone must not reset a table
            if (eo & 4).any():
                g.reset_d(terr, L.TERR_HAND_CAP)
            g2.pick_actions_d(act2, 0)
            g2.step_d(act2, flags2, terr2, auto_reset=True)
            g2.sync()
            fl2, te2 = flags2.download(np.uint8, T), terr2.download(np.uint8, T)
            assert np.array_equal(fl2 & 6, fo & 6) and np.array_equal(fl2 & 1, over) and np.array_equal(te2, eo), s
            if over.any():
                o.reset(mask=over)
                resets += int(over.sum())
            if s % 20 == 19 or s == steps - 1:
                assert_same(o.snapshot(), hb.snapshot(), "T=%d step %d (step_d + reset_d)" % (T, s))
                assert_same(o.snapshot(), hb2.snapshot(), "T=%d step %d (step_d auto_reset)" % (T, s))
        assert resets > 0 or T > 60000        # (the small batches run long enough for games to end)
        g.reset_d()                           # mask None = every table
        o.reset()
        assert_same(o.snapshot(), hb.snapshot(), "reset_d(all)")
        for b in (act, flags, terr, act2, flags2, terr2):
            b.free()
    lib = L.lib()
    g = pokerl_amd.VecGame(64, num_players=3)
    m = DeviceBuffer(64)
    assert lib.pk_reset_d(g._h, m.ptr, 0, 0) == L.PK_E_INVALID_ARG              # a mask that selects no bit
    assert lib.pk_step_auto_d(g._h, None, None, None) == L.PK_E_INVALID_ARG


def test_stream_pool_drain(HB):
    """Handles recycle their streams through a per-device pool; pk_stream_pool_drain destroys the pooled (idle) ones -- what a host
    application calls before hipDeviceReset or to give the hardware queues back -- and handles created afterwards work."""
    import pokerl_amd
    from pokerl_amd import _lib as L
    lib = L.lib()
    g = pokerl_amd.VecGame(256, num_players=3)
    g.reset(); g.rollout(10, 0)
    first = g.stream
    g.close()                                           # its stream goes to the pool
    g = pokerl_amd.VecGame(64, num_players=2)           # ... and comes back out of it (the pool's reset detector -- a canary allocation whose
    assert g.stream == first                            #     address range is looked up -- says the device has not been reset)
    g.close()
    n = lib.pk_stream_pool_drain(0)
    assert n >= 1
    assert lib.pk_stream_pool_drain(0) == 0 and lib.pk_stream_pool_drain(-1) == 0
    assert lib.pk_stream_pool_drain(1000) == L.PK_E_INVALID_ARG
    g = pokerl_amd.VecGame(256, num_players=3)          # a fresh stream
    g.reset()
    assert g.rollout(5, 0)["steps"] == 256 * 5
    g.close()


def test_create_survives_a_device_reset_with_pooled_streams(tmp_path):
    """a host application that calls hipDeviceReset WITHOUT draining the stream pool leaves dead handles in it.  Any call on such a handle
    -- a query included -- crashes inside the runtime, so the pool keeps a canary allocation per device and looks its address up before handing a
    stream out: canary gone = device reset = the pool (and the judger's per-device table / scratch arena) is forgotten, not used.  In a process of
    its own (a device reset takes every allocation of the process down)."""
    import subprocess
    import sys
    script = tmp_path / "reset_pool.py"
    script.write_text("""
import ctypes as C, sys
sys.path.insert(0, %r)
import numpy as np
import pokerl_amd
hip = C.CDLL("libamdhip64.so")
hands = [["AS", "KS", "QS", "JS", "TS", "2D", "3C"], ["2S", "2H", "5D", "9C", "KD", "3S", "7H"]]
before = [pokerl_amd.eval_hand(h) for h in hands]          # builds the per-device evaluator table and the judger's scratch arena
g = pokerl_amd.VecGame(256, num_players=3); g.reset(); assert g.rollout(5, 0)["steps"] == 1280
g.close()                                   # its stream goes to the pool ...
assert hip.hipDeviceReset() == 0            # ... and dies there, with the table and the arena
for _ in range(2):                          # the first create finds the canary gone and forgets the pool; the second one a healthy pool again
    g = pokerl_amd.VecGame(256, num_players=3); g.reset()
    assert g.rollout(7, 0)["steps"] == 256 * 7
    g.close()
assert [pokerl_amd.eval_hand(h) for h in hands] == before   # the table was rebuilt, not used stale
print("ok")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.returncode, out.stdout[-500:], out.stderr[-2000:])


def test_async_game_step(HB, O):
    """pk_step_async_d: Game.step as a bounded launch.  Tables whose step returned are delivered (ready 1, flags, terr); a table whose
    step rolls on through further hands stays in flight and is delivered by a later call, which ignores its action.  Per table the
    DELIVERED (flags, terr) must equal what the oracle returned for that step -- it made the step at the call that started it -- and
    after a drain every state byte.  Full-width ragged waves, small batches, blinds far above the stacks (every step rolls hands), both
    reset rules, one and two hand ends per launch, invalid actions in between."""
    from pokerl_amd import _lib as L
    from pokerl_amd.hipmem import DeviceBuffer
    lib = L.lib()
    cases = [(65536 + 19, 6, 100, 2, 1, 40, 1, True), (4096, 6, 100, 2, 1, 120, 1, False), (1500, 9, 100, 2, 1, 150, 2, True),
             (900, 5, [3, 100, 5, 40, 7.5], 40, 7.5, 120, 1, True), (2048, 3, 2, 40, 20, 60, 1, False), (65, 16, 10, 3, 1, 200, 1, True)]
    for T, N, stacks, bb, sb, calls, max_hands, auto in cases:
        hb = HB(T, N, stacks, bb, sb, seed=4711)
        g = hb.g
        o = O.OracleGame(T, N, stacks, bb, sb, seed=4711)
        o.reset(); g.reset()
        act, flags, terr, ready = DeviceBuffer(T * 4), DeviceBuffer(T), DeviceBuffer(T), DeviceBuffer(T)
        where = "T=%d N=%d bb=%s max_hands=%d auto=%d" % (T, N, bb, max_hands, auto)
        idle = np.ones(T, bool)                                          # no step of that table is in flight on the device
        want_f, want_e = np.zeros(T, np.uint8), np.zeros(T, np.uint8)    # what the oracle returned for the step in flight / just made
        inflight_seen = delivered = 0

        def over_of(f, e):                                               # game over, or a step the reference would never leave
            return ((f & 1) | ((e & 4) >> 2)).astype(np.uint8)

        def check(mask, what):                                           # the rows delivered for the tables in `mask`
            fl, te = flags.download(np.uint8, T), terr.download(np.uint8, T)
            exp_f = ((want_f & 6) | over_of(want_f, want_e)).astype(np.uint8) if auto else want_f
            badrow = mask & ((fl != exp_f) | (te != want_e))
            assert not badrow.any(), (where, what, [(int(t), int(exp_f[t]), int(fl[t]), int(want_e[t]), int(te[t])) for t in np.nonzero(badrow)[0][:6]], int(badrow.sum()))

        drains = [0]

        def drain():                                                     # finish what is in flight; idle tables: "no step" -- an invalid action (-1), or
            drains[0] += 1                                               # no actions buffer at all (NULL: only a drain takes that), alternately
            if drains[0] % 2:
                act.upload(np.full(T, -1, np.int32))
                g.step_async_d(act, flags, terr, ready, max_hands=0, auto_reset=auto)
            else:
                g.step_async_d(None, flags, terr, ready, max_hands=0, auto_reset=auto)
            g.sync()
            assert (ready.download(np.uint8, T) != 0).all(), where
            fin = ~idle
            check(fin, "drain")
            assert (terr.download(np.uint8, T)[idle] == L.TERR_INVALID_ACTION).all()
            idle[:] = True
            return fin

        for c in range(calls):
            g.pick_actions_d(act, 0)                                     # (a device reader: works while steps are in flight)
            g.sync()                                                     # (the handle's stream is non-blocking: hipMemcpy does not wait for it)
            a = act.download(np.int32, T)
            if c % 7 == 3:                                               # some tables get an invalid action: returned at once, untouched
                a = np.where(np.arange(T) % 11 == c % 11, 9, a).astype(np.int32)
                act.upload(a)
            fo, eo = o.step(np.where(idle, a, -1).astype(np.int32))      # the oracle makes the steps that START in this call
            want_f = np.where(idle, fo, want_f); want_e = np.where(idle, eo, want_e)
            g.step_async_d(act, flags, terr, ready, max_hands=max_hands, auto_reset=auto)
            g.sync()
            r = ready.download(np.uint8, T) != 0
            check(r, "call %d" % c)
            inflight_seen += int((~r).sum()); delivered += int(r.sum())
            done_now = r.copy()
            idle = r.copy()
            if not auto and ((~idle).any() or (r & (over_of(want_f, want_e) != 0)).any()):
                done_now |= drain()                                      # the caller's own reset needs a drained handle (PK_E_BUSY otherwise)
            m = (done_now & (over_of(want_f, want_e) != 0)).astype(np.uint8)
            if m.any():                                                  # auto: the launch has reset them already
                o.reset(mask=m)
                if not auto:
                    g.reset(mask=m)
        fin = drain()
        m = (fin & (over_of(want_f, want_e) != 0)).astype(np.uint8)
        if m.any():
            o.reset(mask=m)
            if not auto:
                g.reset(mask=m)
        assert_same(o.snapshot(), hb.snapshot(), where)
        if bb != 40:
            assert delivered > 0.9 * T * calls, where                    # (blinds far above the stacks: half of the steps roll on and are delivered by the drains)
        if bb == 40 or T > 60000:
            assert inflight_seen > 0, where                              # the bound was exercised: steps did stay in flight
        # while steps may be in flight every other entry point is busy -- except the device readers
        g.pick_actions_d(act, 0)
        g.step_async_d(act, flags, terr, ready, max_hands=1, auto_reset=auto)
        assert lib.pk_get_i32(g._h, 0, L.ptr(np.zeros(T, np.int32))) == L.PK_E_BUSY and lib.pk_reset(g._h, None, 0) == L.PK_E_BUSY
        assert lib.pk_step_d(g._h, act.ptr, flags.ptr, terr.ptr) == L.PK_E_BUSY
        assert lib.pk_rollout(g._h, 5, 0, 1, 1, None) == L.PK_E_BUSY and lib.pk_env_reset_d(g._h, None, 0) == L.PK_E_BUSY
        assert lib.pk_env_step_async_d(g._h, None, 0, 0, 1, 4, act.ptr, flags.ptr, flags.ptr, terr.ptr, None, ready.ptr) == L.PK_E_BUSY
        assert lib.pk_step_async_d(g._h, act.ptr, flags.ptr, terr.ptr, ready.ptr, 1, 0 if auto else 1) == L.PK_E_INVALID_ARG
        assert lib.pk_step_async_d(g._h, None, flags.ptr, terr.ptr, ready.ptr, 1, 1 if auto else 0) == L.PK_E_INVALID_ARG      # NULL actions: a drain only
        assert lib.pk_pick_actions_d(g._h, 0, act.ptr) == L.PK_OK and lib.pk_sync(g._h) == L.PK_OK
        act.upload(np.full(T, -1, np.int32))
        g.step_async_d(act, flags, terr, ready, max_hands=0, auto_reset=auto); g.sync()
        assert lib.pk_get_i32(g._h, 0, L.ptr(np.zeros(T, np.int32))) == L.PK_OK
        for b in (act, flags, terr, ready):
            b.free()
        g.close()


def test_bad_arguments_are_reported_not_fatal(HB):
    import ctypes as C
    import pokerl_amd
    from pokerl_amd import _lib as L
    lib = L.lib()
    h = C.c_void_p()
    assert lib.pk_create(C.byref(h), 0, 16, 17, None, 100.0, 2.0, 1.0, 0, 1, 0) == L.PK_E_INVALID_ARG   # N > 16
    assert lib.pk_create(C.byref(h), 0, 16, 6, None, float('inf'), 2.0, 1.0, 0, 1, 0) == L.PK_E_INVALID_ARG and b"finite" in lib.pk_last_error(None)
    assert lib.pk_create(C.byref(h), 0, 0, 4, None, 100.0, 2.0, 1.0, 0, 1, 0) == L.PK_E_INVALID_ARG     # T < 1
    assert lib.pk_create(C.byref(h), 99, 16, 4, None, 100.0, 2.0, 1.0, 0, 1, 0) == L.PK_E_INVALID_ARG   # no such device
    assert b"device" in lib.pk_last_error(None)
    with pytest.raises(ValueError):
        pokerl_amd.VecGame(4, num_players=1)
    g = pokerl_amd.VecGame(4, num_players=2)
    assert lib.pk_get_f64(g._h, 9, None) == L.PK_E_INVALID_ARG
    assert lib.pk_rollout(g._h, 5, 7, 1, 1, None) == L.PK_E_INVALID_ARG     # unknown policy
    with pytest.raises(ValueError):
        g.reset(mask=np.ones(3, np.uint8))


def test_bench_two_ranks_rehearsal(tmp_path):
    """bench.py under torch.distributed.run with 2 ranks (both on GPU 0, gloo -- RCCL refuses a shared device): the N>1
    code path end to end with real kernels: sharding by global table id, barrier, SUM of units / MAX of seconds."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PK_BENCH_BACKEND="gloo", PK_BENCH_SAME_DEVICE="1", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(root, "bench.py"),
                          "--gpus", "2", "--tables", "8192", "--steps", "256", "--warmup", "64", "--chunk", "64", "--min-steps", "1024"],
                         capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert len(line) < 4096
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["steps"] == 256
    assert r["value"] > 0 and abs(r["value"] - 2 * 8192 * 256 / (r["ms_per_step"] * 256 / 1e3)) / r["value"] < 1e-3
    assert "cpu_baseline" not in r and r["roofline"]["frac"] > 0
    # the `dist` block: the collective library saw both ranks, each with its shard (value = units of ALL ranks / MAX seconds, above)
    d = r["dist"]
    assert d["backend"] == "gloo" and d["world"] == 2 and d["devices"] == [0, 0] and d["tables_per_rank"] == [8192, 8192]
    assert d["collectives_on_step_path"] == 0


def test_device_resident_env_loop(HB, O):
    """A learner on the same GPU: pk_env_reset_d / pk_get_obs_d / pk_get_valid_actions_d / pk_env_step_d with every
    buffer in HBM, checked against the host-buffer entry points and the oracle."""
    import ctypes as C
    import pokerl_amd
    from pokerl_amd import _lib as L
    hip = C.CDLL("libamdhip64.so")
    T, N = 2048, 6
    D = 17 + 3 * N
    env = pokerl_amd.VecPokerGameEnv(0, num_tables=T, num_players=N, seed=31)
    g = env.game
    o = O.OracleGame(T, N, seed=31)
    lib = L.lib()

    def dmalloc(nbytes):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), nbytes) == 0
        return p

    d_obs, d_valid, d_act = dmalloc(T * D * 8), dmalloc(T * 7), dmalloc(T * 4)
    d_rew, d_done, d_hand, d_terr, d_mask = dmalloc(T * 8), dmalloc(T), dmalloc(T), dmalloc(T), dmalloc(T)

    def d2h(ptr, arr):
        assert hip.hipMemcpy(arr.ctypes.data_as(C.c_void_p), ptr, arr.nbytes, 2) == 0
        return arr

    L.check(lib.pk_env_reset_d(g._h, None, 0), g._h)
    o.env_reset(None, 0)
    for s in range(60):
        L.check(lib.pk_get_obs_d(g._h, -1, d_obs), g._h)
        L.check(lib.pk_get_valid_actions_d(g._h, -1, d_valid), g._h)
        g.sync()
        obs = d2h(d_obs, np.zeros((T, D), np.float64))
        valid = d2h(d_valid, np.zeros((T, 7), np.uint8))
        assert GU.bits_equal(obs, g.observations) and np.array_equal(valid, g.get_valid_actions()[0].astype(np.uint8))
        a = o.pick_actions(0)
        assert all(valid[np.arange(T), a] == 1)
        ro, do, ho, eo = o.env_step(a, 0)
        assert hip.hipMemcpy(d_act, a.ctypes.data_as(C.c_void_p), T * 4, 1) == 0
        L.check(lib.pk_env_step_d(g._h, d_act, 0, d_rew, d_done, d_hand, d_terr), g._h)
        g.sync()
        assert GU.bits_equal(ro, d2h(d_rew, np.zeros(T, np.float64)))
        done = d2h(d_done, np.zeros(T, np.uint8))
        assert np.array_equal(do, done) and np.array_equal(ho, d2h(d_hand, np.zeros(T, np.uint8)))
        assert not d2h(d_terr, np.zeros(T, np.uint8)).any()
        if done.any():
            assert hip.hipMemcpy(d_mask, done.ctypes.data_as(C.c_void_p), T, 1) == 0
            L.check(lib.pk_env_reset_d(g._h, d_mask, 0), g._h)
            o.env_reset(done, 0)
    g.sync()
    assert GU.bits_equal(o.f64(0), g.credits) and GU.bits_equal(o.f64(3), g.payoffs)
    for p in (d_obs, d_valid, d_act, d_rew, d_done, d_hand, d_terr, d_mask):
        hip.hipFree(p)


def test_env_loops_are_bounded(HB, O):
    """Seat 0 starts with no chips: once it is broke the reference's PokerGameEnv.reset loop (envs/game_env.py:24-27)
    never sees seat 0 active again and spins for ever.  The kernel must not hang the GPU: PK_TERR_ENV_CAP after
    PK_ENV_STEP_CAP auto-played steps, with the same state the (equally capped) oracle reaches."""
    T, N = 64, 3
    o = O.OracleGame(T, N, [0, 100, 100], 2, 1, seed=3)
    h = HB(T, N, [0, 100, 100], 2, 1, seed=3)
    o.env_reset(None, 1)
    h.env_reset(None, 1)
    assert_same(o.snapshot(), h.snapshot(), "capped env reset")
    import pokerl_amd
    assert (h.g.step_serial >= 8192).all()   # every table ran into the cap


def test_launch_splitting_and_determinism(HB):
    """Size-independent properties at BASELINE's full size (65 536 x 6): a rollout split into launches of any length
    reaches the same state (run-ahead scheduling is invisible), and two handles with the same seed agree bit for bit."""
    T, N = 65536, 6
    a, b, c = HB(T, N), HB(T, N), HB(T, N)
    for h in (a, b, c):
        h.reset()
    ca = a.rollout(600, 0)
    cb = b.rollout(1, 0) + b.rollout(299, 0) + b.rollout(300, 0)
    cc = c.rollout(37, 0) + c.rollout(563, 0)
    assert ca.tolist() == cb.tolist() == cc.tolist()
    sa, sb, sc = a.snapshot(), b.snapshot(), c.snapshot()
    for k in GU.SNAP_FIELDS:
        assert GU.bits_equal(sa[k], sb[k]) and GU.bits_equal(sa[k], sc[k]), k
    assert (sa["step_serial"] == 600).all() and (sa["hand_serial"] >= 1).all()
    # a different seed gives different decks (the streams really are keyed by the seed)
    d = HB(T, N, seed=12345)
    d.reset()
    assert not np.array_equal(d.snapshot()["cards"][:64], sa["cards"][:64])


def test_random_configurations_vs_oracle(HB, O):
    """Seeded fuzz over odd configurations (zero / fractional blinds, small blind above big blind, blinds larger than
    the stacks, per-seat stacks from 0.5 to 1e6, every N, table ids anywhere in 2^32, any first dealer): fused rollout
    and a few lockstep steps against the oracle (which tests/golden/fuzz_oracle_vs_reference.py checks against the
    imported reference on the same kind of configurations)."""
    import random
    rng = random.Random(99)
    stacks = [0.5, 1, 2, 3, 5, 10, 37.5, 100, 1000, 1e6]
    blinds = [0, 0.25, 0.5, 1, 2, 3, 7.5, 40]
    for i in range(42):
        N = 2 + i % 14
        start = [rng.choice(stacks) for _ in range(N)] if rng.random() < 0.5 else rng.choice(stacks)
        bb, sb = rng.choice(blinds), rng.choice(blinds)
        policy = 1 if rng.random() < 0.25 else 0
        seed, base, dealer = rng.getrandbits(63), rng.getrandbits(32) & 0xFFFFF000, rng.randrange(N)
        T = rng.choice([65, 128, 300])
        o = O.OracleGame(T, N, start, bb, sb, seed=seed, table_id_base=base)
        h = HB(T, N, start, bb, sb, seed=seed, table_id_base=base)
        o.reset(dealer=dealer); h.reset(dealer=dealer)
        where = "cfg %d: N=%d start=%s bb=%s sb=%s policy=%d" % (i, N, start, bb, sb, policy)
        assert_same(o.snapshot(), h.snapshot(), where + " reset")
        co, _ = o.rollout(120, policy, True)
        ch = h.rollout(120, policy, True)
        assert co.tolist() == ch.tolist(), where
        assert_same(o.snapshot(), h.snapshot(), where + " rollout")
        for s in range(10):
            a = o.pick_actions(policy)
            fo, eo = o.step(a)
            fh, eh = h.step(a)
            assert np.array_equal(fo, fh) and np.array_equal(eo, eh), where
            bad = ((fo & 1) | (eo != 0)).astype(np.uint8)
            if bad.any():
                o.reset(mask=bad); h.reset(mask=bad)
        assert_same(o.snapshot(), h.snapshot(), where + " lockstep")


def test_streaming_evaluator(HB, O):
    """pk_make_hands_d + pk_eval7_d (device-resident, 12 B per evaluation) against the RNG spec and the oracle."""
    from pokerl_amd import judger
    from pokerl_amd.hipmem import DeviceBuffer
    from oracle import rng_spec as R
    m = (1 << 20) + 3            # odd count: exercises the tail hand
    hands_d, out_d = DeviceBuffer(m * 8), DeviceBuffer(m * 4)
    judger.make_hands(hands_d.ptr, m, seed=R.DEFAULT_SEED)
    words = hands_d.download(np.uint64, m)
    cards = words.view(np.uint8).reshape(m, 8)[:, :7]
    canon = R.canonical_deck_values()
    for i in (0, 1, 77, 65535, m - 1):
        assert cards[i].tolist() == [canon[j] for j in R.deck_permutation(R.DEFAULT_SEED, i, 0, ndraws=7)[:7]]
    assert (np.sort(cards, axis=1)[:, 1:] != np.sort(cards, axis=1)[:, :-1]).all()    # 7 distinct cards in every hand
    rank, kick, _ = O.eval_hands(np.ascontiguousarray(cards))
    expect = (rank.astype(np.uint32) << 20) | kick
    for distinct in (True, False):
        judger.eval7_stream(hands_d.ptr, m, out_d.ptr, distinct)
        assert np.array_equal(out_d.download(np.uint32, m), expect), distinct
    # multiset input (duplicate cards) through the general evaluator
    dup = cards[:4096].copy()
    dup[:, 6] = dup[:, 0]
    w = np.zeros((4096, 8), np.uint8); w[:, :7] = dup
    hands_d.upload(w)
    judger.eval7_stream(hands_d.ptr, 4096, out_d.ptr, distinct=False)
    r2, k2, _ = O.eval_hands(dup)
    assert np.array_equal(out_d.download(np.uint32, 4096), (r2.astype(np.uint32) << 20) | k2)
    assert judger.time_eval7_stream(hands_d.ptr, 4096, out_d.ptr, True, reps=2) > 0
    hands_d.free(); out_d.free()
    # pk_eval_hands: PARTIAL hands of distinct cards (the fast path for 3..7 cards; every subset of the deck is checked on the CPU
    # build, tools/host_sim evaln), hands that repeat a card and 0..2-card hands (the reference's scan), mixed in one batch
    rng = np.random.default_rng(12)
    mm = 300000
    part = cards[:mm].copy()
    nc = rng.integers(0, 8, mm).astype(np.uint8)
    rep = rng.random(mm) < 0.2
    part[rep, 1] = part[rep, 0]                       # a repeated card (inside the hand wherever ncards >= 2)
    ro, ko, no = O.eval_hands(np.ascontiguousarray(part), nc)
    rg, kg, ng = judger.eval_hands(part, nc)
    assert np.array_equal(ro, rg) and np.array_equal(ko, kg) and np.array_equal(no, ng)
    # the TABLE path (k_eval_hands_tab: one unaligned 8-byte load per hand, the buffer's last hand byte by byte) at odd sizes, with the
    # card bytes starting at byte offset 0 and 1 of the allocation (canary bytes around them)
    import ctypes as C
    for m2 in (1, 2, 511, 512, 513, 1023, 4097, 300000 - 1):
        d_c, d_n, d_r, d_k, d_nk = DeviceBuffer(m2 * 7 + 8), DeviceBuffer(m2), DeviceBuffer(m2), DeviceBuffer(m2 * 4), DeviceBuffer(m2)
        for off in (0, 1):
            d_c.upload(np.concatenate([np.full(off, 0xEE, np.uint8), part[:m2].reshape(-1), np.full(8 - off, 0xEE, np.uint8)]))
            d_n.upload(nc[:m2])
            judger.eval_hands_d(C.c_void_p(d_c.ptr.value + off), d_n.ptr, m2, d_r.ptr, d_k.ptr, d_nk.ptr)
            C.CDLL("libamdhip64.so").hipDeviceSynchronize()
            assert np.array_equal(d_r.download(np.uint8, m2), ro[:m2]) and np.array_equal(d_k.download(np.uint32, m2), ko[:m2]), (m2, off)
            assert np.array_equal(d_nk.download(np.uint8, m2), no[:m2]), (m2, off)
        for b in (d_c, d_n, d_r, d_k, d_nk):
            b.free()
    # bytes that are no card (suit > 3, rank nibble 13..15) must not reach a bitmask / table evaluator, where they would alias onto real
    # cards: such hands go to the scan wherever the bytes start
    odd = cards[:2048].copy()
    odd[::3, 2] |= 0x40; odd[1::3, 4] = (odd[1::3, 4] & 0xf0) | 0x0d; odd[2::3, 0] = 0xff
    d_c, d_r, d_k, d_nk = DeviceBuffer(2048 * 7 + 8), DeviceBuffer(2048), DeviceBuffer(2048 * 4), DeviceBuffer(2048)
    got = []
    for off in (0, 1):
        d_c.upload(np.concatenate([np.zeros(off, np.uint8), odd.reshape(-1), np.zeros(8 - off, np.uint8)]))
        judger.eval_hands_d(C.c_void_p(d_c.ptr.value + off), None, 2048, d_r.ptr, d_k.ptr, d_nk.ptr)
        C.CDLL("libamdhip64.so").hipDeviceSynchronize()
        got.append((d_r.download(np.uint8, 2048), d_k.download(np.uint32, 2048), d_nk.download(np.uint8, 2048)))
    assert all(np.array_equal(a, b) for a, b in zip(*got))


def test_empty_and_minimal_inputs(HB, O):
    """Empty batches and the smallest shapes: zero hands, zero-card hands, one table."""
    from pokerl_amd import judger
    r, k, n = judger.eval_hands(np.zeros((0, 7), np.uint8))
    assert len(r) == len(k) == len(n) == 0
    assert judger.compare_rankings_batch(np.zeros((0, 3), np.uint8), np.zeros((0, 3), np.uint32)).shape == (0, 3)
    r, k, n = judger.eval_hands(np.zeros((5, 7), np.uint8), np.zeros(5, np.uint8))      # five empty hands
    assert r.tolist() == [10] * 5 and k.tolist() == [0] * 5 and n.tolist() == [0] * 5   # (NONE, []) judger.py:30
    import pokerl_amd
    assert pokerl_amd.eval_hand([]) == (10, [])
    assert pokerl_amd.eval_hand(["KS"]) == (9, [12])                                     # judger.py:31
    assert pokerl_amd.eval_hand(["KS", "KD"]) == (8, [12]) and pokerl_amd.eval_hand(["2S", "AD"]) == (9, [13, 1])
    h = HB(1, 2)
    h.reset()
    assert h.rollout(0, 0).tolist() == [0, 0, 0, 0]
    assert h.rollout(300, 0)[0] == 300


def test_bench_json_contract():
    """`python bench.py` prints exactly ONE JSON line on stdout -- under 4 KB, so that it fits the tail of stdout the driver keeps --
    with every field the driver's contract names; everything else (notes, per-sample arrays, the full roofline block of every leg) is
    in bench_detail.json next to bench.py."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PK_BENCH_CPU_BUDGET="1", PK_BENCH_EVAL_LOG2="22")
    detail_path = os.path.join(root, "bench_detail.json")
    if os.path.exists(detail_path):
        os.remove(detail_path)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "256", "--warmup", "64"],
                         capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    assert len(lines[0]) < 4096, len(lines[0])
    c = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "evaluator", "extra_workloads", "detail"):
        assert k in c, k
    assert c["metric"] == "env-steps/sec (whole node) + showdown hand-evals/sec, 65 536 tables 6-max"      # BASELINE.json's metric
    assert c["n_gpus"] == 1 and c["steps"] == 256 and c["warmup"] == 64 and c["higher_is_better"] is True
    assert c["scaling"] == "weak" and c["vs_baseline"] is None and c["dtype"] == "f64" and c["data"] == "synthetic"
    assert "workload" in c["config"] and "model" not in c["config"] and "BASELINE configs[2]" in c["config"]["workload"]
    assert c["config"]["kernel"].startswith("k_rollout_tab<6>") and "dist" not in c      # (256-step launches: the table-evaluator variant)
    assert abs(c["value"] - 65536 * 256 / (c["ms_per_step"] * 256 / 1e3)) / c["value"] < 1e-3     # (five significant digits in the line)
    crf = c["roofline"]
    assert crf["bound"] == "valu-issue" and crf["unit"] == "wave-instr/s" and 0.0 < crf["frac"] <= 0.5 and crf["traffic"] > 0
    assert abs(crf["frac"] - crf["achieved"] / crf["peak"]) < 1e-4 and crf["source"].endswith("_summary.json") and crf["hbm_algorithmic"]["frac"] > 0
    assert c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["cores"] == 1 and c["cpu_baseline"]["value"] > 0 and c["cpu_baseline"]["sample"]
    assert c["evaluator"]["hand_evals_per_s"] > 0 and 0.0 < c["evaluator"]["frac"] <= 1.0
    cx = c["extra_workloads"]
    assert len(cx) == 12 and all(len(x["name"]) <= 40 for x in cx)
    for x in cx:
        assert set(x) >= {"name", "value", "unit", "kernel_ms", "bound", "frac", "hbm_frac"} and x["value"] > 0 and 0.0 < x["frac"] <= 1.0, x
    assert [x["name"].split()[0] for x in cx] == ["cfg1", "cfg4", "cfg2", "Game.step", "Game.step", "Game.step+obs_packed", "Game.step+obs_packed", "Game.step", "Game.step",
                                                  "env.step", "env.step", "env.step"]
    assert all(x["bound"] == "hbm" for x in cx[3:9]) and cx[3]["unit"] == "env-steps/s" and cx[4]["value"] >= cx[3]["value"] * 0.95
    assert cx[8]["frac"] > cx[4]["frac"]                  # a batch that fills the chip is bound by its bytes, 65 536 tables by the slowest table's chain
    assert cx[7]["value"] > cx[3]["value"]                # bounded launches: the tables whose step rolls on do not hold the launch
    # SURVEY f2 on the Game.step path: the observation row from the step kernel's registers (one launch) beats step + getter kernel (two)
    assert "2 launches" in cx[5]["name"] and "fused" in cx[6]["name"] and cx[6]["value"] > 1.2 * cx[5]["value"] and cx[6]["frac"] > cx[5]["frac"], (cx[5], cx[6])
    # the line checks itself: ms_per_step (median sample) x the steps every table made inside the timed samples = the timed
    # seconds (all samples), which fit inside this run's wall clock; `lib` names the kernel sources the figures were measured on
    from pokerl_amd import _lib as L
    assert c["timed_steps_per_table"] == c["steps"] * c["reps"] * c["samples"] == c["config"]["launch_stats"]["steps"]
    assert abs(c["ms_per_step"] * 1e-3 * c["timed_steps_per_table"] - c["timed_s"]) < 0.15 * c["timed_s"], (c["ms_per_step"], c["timed_steps_per_table"], c["timed_s"])
    assert c["lib"] == L.lib().pk_build_info().decode() and c["lib"].startswith("abi=6 src=")
    assert isinstance(crf["profile_stale"], bool)
    assert abs(cx[1]["hand_evals_per_s"] / cx[1]["value"] - 1.0) < 1e-3       # configs[4]: one in-game evaluation per env-step

    # ---- the detail file: what the line held up to round 4
    assert c["detail"] == "bench_detail.json"
    r = json.load(open(detail_path))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert abs(r["value"] - 65536 * 256 / (r["ms_per_step"] * 256 / 1e3)) / r["value"] < 1e-6 and abs(r["value"] - c["value"]) / r["value"] < 1e-4
    assert r["reps"] == 2048 and r["samples"] == 7 and len(r["sample_seconds"]) == 7   # 2 048 blocks of 256 steps = 524 288 per sample
    rf = r["roofline"]
    # the binding roofline leads: VALU issue, from the committed rocprofv3 PMC summary of this workload x this run's launch time
    assert rf["bound"] == "valu-issue" and rf["unit"] == "wave-instr/s" and rf["peak"] == 256 * 4 * 2.4e9 / 2
    assert 0.0 < rf["frac"] <= 0.5 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert rf["waves_per_simd"] == 1.0 and 0.3 < rf["lanes_active"] <= 1.0 and rf["source"].endswith("_summary.json")
    cm = rf["ceiling_mix"]
    assert cm["peak"] < rf["peak"] and 0.0 < cm["frac_of_ceiling"] <= 1.0 and abs(cm["frac_of_ceiling"] - rf["achieved"] / cm["peak"]) < 1e-9
    # self-consistent timing: HIP-event time per launch / steps per launch can never exceed the wall-clock time per step
    assert rf["kernel_ms"] / rf["steps_per_launch"] <= r["ms_per_step"] * 1.0005, (rf["kernel_ms"], rf["steps_per_launch"], r["ms_per_step"])
    st = r["config"]["launch_stats"]
    assert st["steps"] == 256 * 2048 * 7 and abs(rf["steps_per_launch"] - 256 * 2048 / rf["launches_timed"]) < 1e-9 and rf["launches_timed"] * 6 < st["launches"] < rf["launches_timed"] * 8
    # measured HBM traffic (one read + one write of the table state per launch) and SURVEY 8d's algorithmic figure, nested
    assert rf["traffic"] and 0.5 < rf["traffic"] / (65536 * 2 * 290) < 1.5 and rf["traffic_source"].endswith("_summary.json")
    hb = rf["hbm_algorithmic"]
    assert hb["unit"] == "GB/s" and hb["peak"] == 8000.0 and abs(hb["frac"] - hb["achieved"] / hb["peak"]) < 1e-9
    assert abs(hb["achieved"] - hb["algorithmic_bytes_per_launch"] / (rf["kernel_ms"] * 1e-3) / 1e9) / hb["achieved"] < 1e-6
    assert abs(hb["algorithmic_bytes_per_launch"] - 478 * 65536 * rf["steps_per_launch"]) < 1.0
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["unit"] == "env-steps/s" and cb["sample"]
    assert "all_cores" in cb and cb["all_cores"]["cores"] >= 1
    ev = r["evaluator"]
    assert ev["roofline"]["bytes_per_eval"] == 12 and ev["hand_evals_per_s"] > 0 and 0.0 < ev["roofline"]["frac"] <= 1.0
    assert "HYBRID" in rf["note"] and "COMMITTED" in rf["note"]      # says where VALU-per-wave-step comes from
    assert cm["half_rate_share_source"] and 0.4 < cm["half_rate_share"] < 0.8
    # the other single-GPU BASELINE configs, Game.step with the caller's actions and the PokerGameEnv path are driver-timed legs
    xs = r["extra_workloads"]
    assert len(xs) == 12 and [("configs[1]" in xs[0]["name"]), ("configs[4]" in xs[1]["name"])] == [True, True]
    assert "one launch per call" in xs[2]["name"] and xs[2]["launch_stats"]["max"] == 20 and xs[2]["value"] < r["value"]
    for x in xs:
        for k in ("name", "short", "value", "unit", "kernel", "kernel_ms", "launches", "roofline"):
            assert k in x, (x.get("name"), k)
        xr = x["roofline"]
        assert x["value"] > 0 and x["kernel_ms"] > 0 and x["launches"] > 0 and xr["bound"] in ("valu-issue", "hbm")
        assert xr["frac"] is not None and 0.0 < xr["frac"] <= 1.0 and xr["source"].endswith("_summary.json"), x["name"]
        assert abs(xr["frac"] - xr["achieved"] / xr["peak"]) < 1e-9
    # configs[4] is the showdown-heavy half of the metric: one in-game evaluation per env-step
    assert abs(xs[1]["hand_evals_per_s"] / xs[1]["value"] - 1.0) < 1e-3 and xs[0]["unit"] == xs[1]["unit"] == "env-steps/s"
    assert 0.99 < xs[7]["ready_fraction_per_launch"] < 1.0 and "k_step_async" in xs[7]["kernel"]
    for i, x in enumerate(xs[3:9]):                                # Game.step legs: the HBM roofline on SURVEY 8d's bytes (+ the observation row's) + measured traffic
        xr = x["roofline"]
        row = 168 if i in (2, 3) else 0                            # PK_OBS_PACKED_BYTES(6)
        assert x["unit"] == "env-steps/s" and "k_step" in x["kernel"] and xr["bound"] == "hbm" and x["tables_with_error_bits"] < 40
        assert xr["obs_row_bytes"] == row and abs(xr["achieved"] - (478 + row) * x["value"] / 1e9) / xr["achieved"] < 1e-6 and xr["traffic"] > 0.5 * 65536 * 478
    assert xs[5]["kernel"] == "k_step + k_obs_packed" and xs[6]["kernel"] == "k_step[+packed row]" and xs[5]["launches"] == 2 * xs[6]["launches"]
    assert xs[6]["roofline"]["traffic"] < xs[5]["roofline"]["traffic"]       # measured: the getter kernel re-reads the tables the step kernel has just stored
    for x in xs[9:]:                                               # PokerGameEnv legs: measured HBM traffic beside the VALU figure
        assert x["unit"] == "env.step/s" and x["kernel"].startswith("k_env_step") and 0.0 < x["roofline"]["hbm"]["frac"] <= 1.0
        assert x["roofline"]["traffic"] > 0 and 4.0 < x["game_steps_per_env_step"] < 8.0
    assert xs[9]["ready_fraction_per_launch"] == 1.0 and 0.3 < xs[10]["ready_fraction_per_launch"] < 1.0
    # every figure that rests on a committed counter summary says whether that summary was measured on THIS library's sources
    assert abs(sum(r["sample_seconds"]) - c["timed_s"]) < 1e-3 * c["timed_s"]
    src = json.load(open(os.path.join(root, "profiles", rf["source"])))
    assert rf["profile_stale"] == (src.get("source_hash") != L.source_hash()) == crf["profile_stale"]
    for short, full in zip(cx, xs):
        assert bool(short.get("profile_stale", False)) == bool(full["roofline"].get("profile_stale", False)), short["name"]


@pytest.mark.parametrize("N,policy", [(6, 0), (9, 1), (2, 0), (10, 0)])
def test_full_width_waves_lockstep_and_env(HB, O, monkeypatch, N, policy):
    """Batches below 32 769 tables spread over 1 024 waves with fewer than 64 tables each (Hot::tpb), so the small
    lockstep / env cases above never fill a wave.  PK_TPB=64 forces full waves: pk_step (k_step), pk_pick_actions,
    pk_reset with a mask, pk_env_reset / pk_env_step and the getters against the oracle with 64 tables per wave."""
    monkeypatch.setenv("PK_TPB", "64")
    T = 2048 + 37                                      # ragged last wave
    o = O.OracleGame(T, N, seed=4242)
    h = HB(T, N, seed=4242)
    o.reset(); h.reset()
    for s in range(160):
        a = o.pick_actions(policy)
        assert np.array_equal(a, h.pick_actions(policy))
        fo, eo = o.step(a)
        fh, eh = h.step(a)
        assert np.array_equal(fo, fh) and np.array_equal(eo, eh), s
        over = (fo & 1).astype(np.uint8)
        if over.any():
            o.reset(mask=over); h.reset(mask=over)
        if s % 20 == 0:
            assert_same(o.snapshot(), h.snapshot(), "step %d" % s)
    assert_same(o.snapshot(), h.snapshot(), "lockstep")
    o.env_reset(None, policy); h.env_reset(None, policy)
    assert_same(o.snapshot(), h.snapshot(), "env reset")
    for s in range(40):
        a = o.pick_actions(0)
        ro, do, ho, eo = o.env_step(a, policy)
        rh, dh, hh, eh = h.env_step(a, policy)
        assert GU.bits_equal(ro, rh) and np.array_equal(do, dh) and np.array_equal(ho, hh) and np.array_equal(eo, eh), s
        m = ((do != 0) | (eo != 0)).astype(np.uint8)
        if m.any():
            o.env_reset(m, policy); h.env_reset(m, policy)
    assert_same(o.snapshot(), h.snapshot(), "env")
