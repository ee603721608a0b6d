"""GPU tests of PokerGameEnv with ONE AGENT PER SEAT (reference pokerl/envs/game_env.py:13-18: a list of agent callables,
`self.agents[active_player](state)` at :25, :43, :51): in-kernel agents per seat, seats played by the caller
(pk_env_step_multi_d: tables yield where such a seat is to act), the call agent, and the reference-style constructor.
Everything bit-exact against the oracle (itself pinned to reference-generated fixtures with mixed agent lists)."""
import numpy as np
import pytest

import golden_util as GU

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import loader
    return loader


@pytest.fixture(scope="module")
def R():
    from oracle import rng_spec
    return rng_spec


def mask_bits(rows):
    return (rows[:, 3:10] > 0).astype(np.uint32) @ (1 << np.arange(7, dtype=np.uint32))


class SpecAgent:
    """A HOST agent that plays an rng_spec policy by the book: the action the in-kernel agent of that policy would take for
    the table's current step serial and valid mask.  Batched: gets the observation rows of all the tables that wait for it."""
    batched = True

    def __init__(self, R, env_ref, policy, seed, base=0):
        self.R, self.env_ref, self.policy, self.seed, self.base, self.calls = R, env_ref, policy, seed, base, 0

    def __call__(self, rows, tables):
        self.calls += 1
        serial = self.env_ref[0].game.step_serial          # readable while env calls are in flight (pk_get_serials)
        bits = mask_bits(rows)
        return np.array([self.R.pick_action(self.seed, self.base + int(t), int(serial[t]), int(b), self.policy)
                         for t, b in zip(tables, bits)], np.int32)


def test_reference_style_constructor_and_in_kernel_agent_lists(O):
    """VecPokerGameEnv(agents=[...]) takes the reference's list form: one agent per opponent seat.  In-kernel agents per
    seat (RandomAgent / CallAgent / AllInAgent markers or Policy values) against the oracle with the same list."""
    import pokerl_amd
    from pokerl_amd import AllInAgent, CallAgent, Policy, RandomAgent
    with pytest.raises(ValueError):
        pokerl_amd.VecPokerGameEnv([RandomAgent()] * 2, num_tables=4, num_players=4)       # three opponents expected
    with pytest.raises(TypeError):
        pokerl_amd.VecPokerGameEnv([RandomAgent(), "nope", RandomAgent()], num_tables=4, num_players=4)
    for T, N, agents, pols in [(2048, 4, [RandomAgent()] * 3, [0, 0, 0]),                          # the reference's examples
                               (1000, 4, [CallAgent(), RandomAgent(), AllInAgent()], [2, 0, 1]),
                               (700, 6, [Policy.CALL, Policy.RANDOM, CallAgent(), Policy.ALL_IN, RandomAgent()], [2, 0, 2, 1, 0]),
                               (512, 2, [CallAgent()], [2])]:
        env = pokerl_amd.VecPokerGameEnv(agents, num_tables=T, num_players=N, seed=2024)
        assert env.agents[0] is None and len(env.agents) == N and env.player_agent == 0   # game_env.py:17-18
        o = O.OracleGame(T, N, seed=2024)
        obs = env.reset(); o.env_reset(None, pols)
        assert GU.bits_equal(o.f64(0), env.game.credits) and (obs[:, 0] == 0).all()
        for s in range(40):
            a = o.pick_actions(0)
            ro, do, ho, eo = o.env_step(a, pols)
            obs, r, d, h, e = env.step(a, strict=False)
            assert GU.bits_equal(ro, r) and np.array_equal(do != 0, d) and np.array_equal(ho != 0, h) and np.array_equal(eo, e), (N, s)
            m = (do != 0).astype(np.uint8)
            if m.any():
                o.env_reset(m, pols); env.reset(m)
        snap_o = o.snapshot()
        g = env.game
        for k, got in (("credits", g.credits), ("payoffs", g.payoffs), ("cards", g.deck), ("states", g.player_states),
                       ("step_serial", g.step_serial), ("hand_serial", g.hand_serial)):
            assert GU.bits_equal(snap_o[k], got), (N, k)
        env.close()


def test_host_agents_reproduce_the_in_kernel_agents(O, R):
    """Seats played by the CALLER: whenever such a seat is to act the table yields, the host agent is called with that
    seat's StateView rows and its action goes back.  Host agents that apply an in-kernel policy's rule must reproduce that
    policy's trajectory bit for bit -- every opponent external; a mix of external and in-kernel seats; per-table
    (non-batched) reference-style callables that get a StateView."""
    import pokerl_amd
    from pokerl_amd import AllInAgent, CallAgent, RandomAgent
    seed = 777

    def call_rule(state):                 # a plain reference-style agent: agent(state: StateView) -> int (agents/agent.py:11-13)
        assert state.player != 0 and len(state.player_cards) == 2
        return 2 if state.valid_actions[2] else (1 if state.valid_actions[1] else 6)

    for T, N, pols, external in [(1024, 4, [0, 2, 1], [1, 2, 3]), (600, 6, [0, 2, 0, 1, 2], [1, 4]), (300, 3, [2, 2], [1]),
                                 (2048 + 5, 5, [0, 0, 0, 0], [2, 3])]:
        ref = []
        agents = []
        for seat, pol in enumerate(pols, start=1):
            if seat in external:
                agents.append(call_rule if (pol == 2 and N == 3) else SpecAgent(R, ref, pol, seed))
            else:
                agents.append([RandomAgent(), AllInAgent(), CallAgent()][pol])
        env = pokerl_amd.VecPokerGameEnv(agents, num_tables=T, num_players=N, seed=seed)
        ref.append(env)
        assert sorted(env._external) == external
        o = O.OracleGame(T, N, seed=seed)
        env.reset(); o.env_reset(None, pols)
        assert GU.bits_equal(o.f64(0), env.game.credits)
        for s in range(25):
            a = o.pick_actions(0)
            ro, do, ho, eo = o.env_step(a, pols)
            obs, r, d, h, e = env.step(a, strict=False)
            assert GU.bits_equal(ro, r) and np.array_equal(do != 0, d) and np.array_equal(ho != 0, h) and np.array_equal(eo, e), (N, s)
            m = (do != 0).astype(np.uint8)
            if m.any():
                o.env_reset(m, pols); env.reset(m)
        snap_o = o.snapshot()
        g = env.game
        for k, got in (("credits", g.credits), ("payoffs", g.payoffs), ("bets", g.bets), ("cards", g.deck),
                       ("states", g.player_states), ("step_serial", g.step_serial), ("hand_serial", g.hand_serial)):
            assert GU.bits_equal(snap_o[k], got), (N, k)
        assert all(a.calls > 0 for a in agents if isinstance(a, SpecAgent))
        env.close()


def test_host_agent_errors_and_busy_contract():
    """An invalid action from a host agent is refused like Game.step refuses it (game.py:649-651): ValueError, the table
    untouched and nothing left in flight; while tables wait for a host agent the other entry points are busy."""
    import pokerl_amd
    from pokerl_amd import _lib as L
    from pokerl_amd.hipmem import DeviceBuffer

    class Bad:
        batched = True

        def __call__(self, rows, tables):
            return np.full(len(tables), 1, np.int32)      # CHECK: invalid pre-flop (high_bet != 0)

    T, N = 256, 3
    env = pokerl_amd.VecPokerGameEnv([Bad(), Bad()], num_tables=T, num_players=N, seed=3)
    with pytest.raises(ValueError, match="invalid move"):
        env.reset()
    g = env.game
    assert (g.step_serial == 0).all() and (g.hand_serial == 1).all()      # Game.reset() ran, no opponent step did
    g.credits                                                              # nothing is in flight any more
    env.close()
    # raw API: yielded tables keep every other entry point busy until pk_env_end_multi_d
    T, N = 512, 4
    env = pokerl_amd.VecPokerGameEnv([Bad(), pokerl_amd.RandomAgent(), Bad()], num_tables=T, num_players=N, seed=3)   # seats 1 and 3 are the caller's
    g = env.game
    D = 17 + 3 * N
    rew, done, hand, terr, obs, who, ready, act, rst = (DeviceBuffer(T * 8), DeviceBuffer(T), DeviceBuffer(T), DeviceBuffer(T),
                                                        DeviceBuffer(T * D * 8), DeviceBuffer(T), DeviceBuffer(T), DeviceBuffer(T * 4), DeviceBuffer(T))

    def call_rule(rows):
        v = rows[:, 3:10] > 0
        return np.where(v[:, 2], 2, np.where(v[:, 1], 1, 6)).astype(np.int32)

    def launch(a, reset=False):
        act.upload(a)
        env.step_multi_d(act.ptr, rst.ptr if reset else None, rew.ptr, done.ptr, hand.ptr, terr.ptr, obs.ptr, who.ptr, ready.ptr,
                         max_passes=0, auto_reset=False)
        g.sync()
        return ready.download(np.uint8, T), who.download(np.uint8, T), obs.download(np.float64, T * D).reshape(T, D)

    rst.upload(np.ones(T, np.uint8))
    r, w, rows = launch(np.full(T, L.ACTION_SKIP, np.int32), reset=True)
    idle = np.zeros(T, bool)
    for _ in range(40):           # everybody calls until some tables wait for the caller's opponent seats while others have returned
        assert set(np.unique(r)) <= {1, 2, 3}
        idle |= r == 1
        assert np.array_equal(rows[r != 3, 0].astype(np.uint8), w[r != 3])          # the row is the view of the seat to act
        assert (w[r == 1] == 0).all() and np.isin(w[r == 2], [1, 3]).all()
        if (r == 2).any() and idle.sum() > T // 8:
            break
        a = np.full(T, L.ACTION_SKIP, np.int32)
        a[r == 2] = call_rule(rows[r == 2])
        if not (r == 2).any():    # every table has returned: seat 0 calls on all of them
            a = call_rule(rows); idle[:] = False
        r, w, rows = launch(a)
    waiting = r == 2
    assert waiting.any() and idle.any() and not (waiting & idle).any()
    with pytest.raises(L.PokerlHipError):
        g.credits
    with pytest.raises(L.PokerlHipError):                                 # the agents may not change while calls are in flight
        L.check(g._lib.pk_env_step_multi_d(g._h, act.ptr, None, env.seat_policies ^ (15 << 8), 0, 0, rew.ptr, done.ptr, hand.ptr,
                                           terr.ptr, obs.ptr, who.ptr, ready.ptr), g._h)
    # an idle table with PK_ACTION_SKIP is left alone (ready 3); a yielded one that gets no valid action keeps waiting
    serial = g.step_serial.copy()
    r2, w2, _ = launch(np.full(T, L.ACTION_SKIP, np.int32))
    assert (r2[idle] == 3).all() and (r2[waiting] == 2).all() and np.array_equal(w2[waiting], w[waiting])
    assert (terr.download(np.uint8, T)[waiting] == L.TERR_INVALID_ACTION).all()
    assert np.array_equal(serial, g.step_serial)
    env.end_multi()
    assert np.array_equal(serial, g.step_serial)
    g.credits
    env.close()


def test_multi_agent_bounded_launches_deliver_the_synchronous_sequences(O, R):
    """pk_env_step_multi_d with a pass budget and auto-reset, seat 0 and two opponent seats played by the caller with
    rules that are functions of the delivered row: per table the sequence of delivered (reward, done, hand) is the
    oracle's, whatever launch delivers it; yields, returns and budget-exhausted tables mix freely."""
    import pokerl_amd
    from pokerl_amd import CallAgent, RandomAgent, _lib as L
    from pokerl_amd.hipmem import DeviceBuffer

    def call_rule(rows):
        v = rows[:, 3:10] > 0
        return np.where(v[:, 2], 2, np.where(v[:, 1], 1, 6)).astype(np.int32)

    # (calling stations as the caller's seats would need one launch per Game.step of the endless games they play once
    # seat 0 is broke -- 8 192 launches up to PK_TERR_ENV_CAP -- so the caller's opponent seats shove instead)
    seen_all = {0: 0, 1: 0, 2: 0}
    for T, N, pols, passes in [(4096, 6, [1, 0, 1, 2, 0], 3), (1000, 4, [1, 1, 0], 1), (2048, 3, [1, 0], 9)]:
        K = 30
        ext_seats = [s for s, p in enumerate(pols, start=1) if p == 1]          # the all-in seats are played by the caller
        agents = [(lambda st: 6) if p == 1 else (RandomAgent() if p == 0 else CallAgent()) for p in pols]
        o = O.OracleGame(T, N, seed=99)
        o.env_reset(None, pols)
        want = dict(rew=np.zeros((T, K)), done=np.zeros((T, K), np.uint8), hand=np.zeros((T, K), np.uint8), err=np.zeros((T, K), np.uint8))
        for k in range(K):
            a = o.pick_actions(2)                                                # seat 0 plays the call rule as well
            ro, do, ho, eo = o.env_step(a, pols)
            # calling stations can play on for ever once seat 0 is broke: the reference's loops (game_env.py:41, :49) would
            # spin, PK_TERR_ENV_CAP ends the step and auto_reset starts a new episode, like a finished one
            assert not (eo & ~np.uint8(O.ERR_ENV_CAP | O.ERR_HAND_CAP)).any()
            m = ((do != 0) | (eo != 0)).astype(np.uint8)
            if m.any():
                o.env_reset(m, pols)
            want["rew"][:, k], want["done"][:, k], want["hand"][:, k], want["err"][:, k] = ro, do, ho, eo
        env = pokerl_amd.VecPokerGameEnv(agents, num_tables=T, num_players=N, seed=99)
        g = env.game
        D = 17 + 3 * N
        rew, done, hand, terr, obs, who, ready, act, rst = (DeviceBuffer(T * 8), DeviceBuffer(T), DeviceBuffer(T), DeviceBuffer(T),
                                                            DeviceBuffer(T * D * 8), DeviceBuffer(T), DeviceBuffer(T), DeviceBuffer(T * 4), DeviceBuffer(T))
        # the compact row beside the dense one (pk_set_env_obs_packed): the env kernel writes both from registers for every table it
        # delivers -- also the YIELDED seat's view (ready 2)
        pdt = pokerl_amd.packed_dtype(N)
        packed = DeviceBuffer(T * pdt.itemsize)
        L.check(g._lib.pk_set_env_obs_packed(g._h, packed.ptr), g._h)
        got = {k: np.zeros_like(v) for k, v in want.items()}
        count = np.full(T, -1, np.int64)                  # -1: the delivery of the initial reset is still to come
        rst.upload(np.ones(T, np.uint8))
        a = np.full(T, L.ACTION_SKIP, np.int32)
        seen = {0: 0, 1: 0, 2: 0}
        launches = 0
        first = True
        while count.min() < K:
            launches += 1
            assert launches < 400 * K, "no progress"
            act.upload(a)
            env.step_multi_d(act.ptr, rst.ptr if first else None, rew.ptr, done.ptr, hand.ptr, terr.ptr, obs.ptr, who.ptr, ready.ptr,
                             max_passes=passes, auto_reset=True)
            first = False
            g.sync()
            r, w = ready.download(np.uint8, T), who.download(np.uint8, T)
            for v in (0, 1, 2):
                seen[v] += int((r == v).sum())
            te = terr.download(np.uint8, T)
            assert not te[r == 2].any()
            rows = obs.download(np.float64, T * D).reshape(T, D)
            deliv = (r == 1) | (r == 2)
            prow = packed.download(np.uint8, T * pdt.itemsize).view(pdt)
            assert pokerl_amd.unpack_obs(prow[deliv], N).tobytes() == rows[deliv].tobytes()
            ret = r == 1
            idx = np.nonzero(ret & (count >= 0) & (count < K))[0]
            c = count[idx]
            got["err"][idx, c] = te[idx]
            got["rew"][idx, c] = rew.download(np.float64, T)[idx]
            got["done"][idx, c] = done.download(np.uint8, T)[idx]
            got["hand"][idx, c] = hand.download(np.uint8, T)[idx]
            count[ret] += 1
            assert (w[ret] == 0).all() and np.isin(w[r == 2], ext_seats).all()
            a = np.full(T, -1, np.int32)                  # garbage for tables in flight (ready 0): must be ignored
            a[r == 1] = call_rule(rows[r == 1])           # seat 0: the call rule on its delivered row
            a[r == 2] = 6                                 # the caller's opponent seats: all-in
        assert seen[2] > 0, seen                          # tables yielded to the caller's seats
        for v in seen:
            seen_all[v] += seen[v]
        for k in want:
            same = GU.bits_equal(want[k], got[k]) if want[k].dtype == np.float64 else np.array_equal(want[k], got[k])
            assert same, (T, N, k)
        env.end_multi()
        L.check(g._lib.pk_set_env_obs_packed(g._h, None), g._h)
        env.close()
    assert seen_all[0] > 0, seen_all                      # ... and some ran out of passes with their call still in flight


def test_env_pool_plays_the_single_handle_trajectories(O):
    """VecPokerGameEnvPool: several batches (own handle + stream each) behind one object; with global table ids the pool's
    tables play exactly what one handle holding all of them -- and the oracle -- play."""
    import pokerl_amd
    T, N = 3000, 5
    pool = pokerl_amd.VecPokerGameEnvPool(pokerl_amd.Policy.RANDOM, num_tables=T, num_batches=4, num_players=N, seed=11, table_id_base=500)
    assert len(pool) == 4 and sum(s.stop - s.start for s in pool.slices) == T
    o = O.OracleGame(T, N, seed=11, table_id_base=500)
    obs = pool.reset(); o.env_reset(None, 0)
    assert obs.shape == (T, 17 + 3 * N)
    for s in range(30):
        a = o.pick_actions(0)
        ro, do, ho, eo = o.env_step(a, 0)
        obs, r, d, h, e = pool.step(a, strict=False)
        assert GU.bits_equal(ro, r) and np.array_equal(do != 0, d) and np.array_equal(ho != 0, h) and np.array_equal(eo, e), s
        if do.any():
            o.env_reset(do, 0)
            for env, sl in zip(pool.envs, pool.slices):
                if do[sl].any():
                    env.reset(do[sl])
    assert GU.bits_equal(o.f64(0), np.concatenate([e.game.credits for e in pool.envs]))
    # strict: an invalid action anywhere raises before ANY batch of the pool is stepped
    before = np.concatenate([e.game.step_serial for e in pool.envs])
    bad = o.pick_actions(0); bad[T - 1] = 7
    with pytest.raises(ValueError, match="table %d" % (T - 1)):
        pool.step(bad)
    assert np.array_equal(before, np.concatenate([e.game.step_serial for e in pool.envs]))
    obs, r, d, h = pool.step(o.pick_actions(0))
    assert obs.shape[0] == T and len(r) == T
    pool.close()


GATHER_GPU = r'''
import os, sys
import torch                      # first: the process must settle on torch's HIP runtime
import torch.distributed as dist
sys.path.insert(0, %r)
import numpy as np
import pokerl_amd
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29561")
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
local = int(os.environ.get("LOCAL_RANK", str(rank)))
torch.cuda.set_device(local)
dist.init_process_group("nccl", device_id=torch.device("cuda", local), rank=rank, world_size=world)      # RCCL
TOTAL = 3000 * world + 1                                                 # uneven shards: the padding rows are exercised
n_local, base = pokerl_amd.shard_tables(TOTAL, rank, world)
g = pokerl_amd.VecGame(n_local, num_players=5, seed=9, device=local, table_id_base=base)
g.reset()
g.rollout(200, 0)
# the job's tables in ONE handle on this rank's GPU: what every rank must receive (global table ids: same trajectories)
whole = pokerl_amd.VecGame(TOTAL, num_players=5, seed=9, device=local)
whole.reset()
whole.rollout(200, 0)
# a busy default stream right before the gather: the export on the handle's own stream must not race torch's work on the
# send tensor (round 3: torch.zeros over the whole tensor could land after the export)
busy = torch.randn(4096, 4096, device="cuda")
for field, want in ((3, whole.payoffs), (0, whole.credits)):
    for _ in range(8):
        busy = busy @ busy * 1e-4
    got = pokerl_amd.gather_f64(g, field, dist)                          # pk_get_f64_d -> all_gather on the device
    assert got.shape == want.shape and got.tobytes() == want.tobytes(), (rank, field)
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print("GATHER-OK world=%%d" %% world)
'''


def test_optional_payoff_gather_over_rccl(tmp_path):
    """north_star's optional RCCL gather of payoffs: the local block is exported on the device (pk_get_f64_d) into the send
    tensor of an RCCL all-gather.  One rank per visible GPU, at most 8 (a one-GPU box: one rank -- the collective and the
    device path are the real ones; on a node the same test exercises N > 1), uneven shards, a busy default stream."""
    import os
    import subprocess
    import sys
    torch = pytest.importorskip("torch")
    import pokerl_amd
    world = max(1, min(pokerl_amd.device_count(), 8))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "gather.py"
    script.write_text(GATHER_GPU % root)
    if world == 1:
        cmd = [sys.executable, str(script)]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", "29561", str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0 and ("GATHER-OK world=%d" % world) in out.stdout, (out.stdout[-2000:], out.stderr[-3000:])


def test_pool_over_a_device_list_plays_the_single_handle_trajectories(O):
    """VecPokerGameEnvPool(devices=[...]): SURVEY 8e's single-process form -- one handle and one Python thread per entry of
    `devices` (here the one GPU twice; a node lists its eight), tables sharded like ranks.  The pool's tables play what one
    handle holding all of them plays, through reset / step (threads) and through the pinned send / recv pipeline."""
    import pokerl_amd
    T, N = 1500, 4
    cfg = dict(num_players=N, seed=5, table_id_base=100)
    pool = pokerl_amd.VecPokerGameEnvPool(pokerl_amd.Policy.RANDOM, num_tables=T, devices=[0, 0, 0], **cfg)
    one = pokerl_amd.VecPokerGameEnv(pokerl_amd.Policy.RANDOM, num_tables=T, **cfg)
    assert len(pool) == 3 and pool.devices == [0, 0, 0] and [s.stop - s.start for s in pool.slices] == [500, 500, 500]
    assert pool.reset().tobytes() == one.reset().tobytes()
    for it in range(6):
        acts = one.game.pick_actions(0)
        a = pool.step(acts, strict=False)          # (no resets in between: finished games report game.py:473's assertion as an error
        b = one.step(acts, strict=False)           #  bit, identically on both sides)
        assert len(a) == len(b) == 5
        for x, y in zip(a, b):
            assert np.asarray(x).tobytes() == np.asarray(y).tobytes(), it
    acts = one.game.pick_actions(0)
    outs = pool.step_pipelined(acts, obs='dense')
    obs, rew, done, hand, terr = one.step(acts, strict=False)
    assert np.concatenate([o[4] for o in outs]).tobytes() == terr.tobytes()
    assert np.concatenate([o[0] for o in outs]).tobytes() == obs.tobytes()
    assert np.concatenate([o[1] for o in outs]).tobytes() == rew.tobytes()
    assert np.array_equal(np.concatenate([o[2] for o in outs]), done)
    # an invalid action: by default (strict=False, nothing blocks before the launches) it is NOT raised -- that table is left unstepped and
    # its terr says so; strict=True checks every batch first (in parallel on the pool's threads) and raises before any table of any batch moved
    bad = one.game.pick_actions(0); bad[1203] = 9; bad[1499] = -3
    before = [e.game.step_serial for e in pool.envs]
    with pytest.raises(ValueError, match=r"table 1203\)"):
        pool.step_pipelined(bad, obs=None, strict=True)
    assert all(np.array_equal(b, e.game.step_serial) for b, e in zip(before, pool.envs))
    terr = np.concatenate([o[4] for o in pool.step_pipelined(bad, obs=None)])
    assert terr[1203] == 1 and terr[1499] == 1 and int((terr == 1).sum()) == 2
    after = np.concatenate([e.game.step_serial for e in pool.envs])
    moved = np.delete(after, [1203, 1499]) > np.delete(np.concatenate(before), [1203, 1499])    # (tables whose game is over do not step: no resets in this test)
    assert after[1203] == np.concatenate(before)[1203] and after[1499] == np.concatenate(before)[1499] and moved.mean() > 0.5
    pool.close(); one.close()


def test_single_table_drop_in_reads_like_the_reference():
    """pokerl_amd.Game / PokerGameEnv: the reference's own shapes (one table, scalars, Card lists, StateView objects).
    The loop of examples/random_game.py:8-12 against the reference-generated trajectory of that very configuration
    (tests/golden/game_n4_example_cfg: 1000 / 40 / 20, four seats), table by table, plus the env fixture with mixed agents."""
    import pokerl_amd
    from pokerl_amd import AllInAgent, CallAgent, Game, PokerGameEnv, PokerMoves, RandomAgent
    z, meta = GU.load_npz("game_n4_example_cfg")
    for t in range(2):
        game = Game(num_players=meta["n"], seed=meta["seed"], table_id_base=meta["table_id_base"] + t, **meta["cfg"])
        game.reset()
        assert game.num_players == 4 and not game.game_over and game.community_cards == []
        for s in range(meta["steps"]):
            onehot, valid = game.get_valid_actions()
            action = int(z["actions"][s, t])
            assert onehot[action] == 1.0 and action in list(valid)
            state = game.active_state
            assert state.player == game.active_player and len(state.player_cards) == 2
            over, hand, turn = game.step(action)
            assert (int(over) | int(hand) << 1 | int(turn) << 2) == int(z["flags"][s, t]), (t, s)
            assert GU.bits_equal(z["post_credits"][s, t], game.credits) and GU.bits_equal(z["post_payoffs"][s, t], game.payoffs)
            assert GU.bits_equal(z["post_pending"][s, t], game.pending_bets) and game.turn == int(z["post_turn"][s, t])
            assert [c.value for c in game.deck] == z["post_cards"][s, t].tolist()
            assert len(game.community_cards) == (0 if game.turn == 0 else game.turn + 2)
            assert game.pot == float(np.sum(game.bets)) and game.get_hand_for(1)[5:] == game.get_cards_of(1)
            if over:
                game.reset()
        with pytest.raises(NotImplementedError):
            game.step(1.5)                                                        # game.py:700
        bad = next(a for a in range(7) if game.get_valid_actions()[0][a] == 0) if (game.get_valid_actions()[0] == 0).any() else None
        if bad is not None:
            with pytest.raises(ValueError, match="invalid move"):
                game.step(bad)                                                    # game.py:649-651
        game.close()
    z, meta = GU.load_npz("env_n4_mixed_opponents")                               # agents = [call, random, all-in]
    env = PokerGameEnv([CallAgent(), RandomAgent(), AllInAgent()], num_players=4, seed=meta["seed"], table_id_base=meta["table_id_base"], **meta["cfg"])
    state = env.reset()
    assert state.player == 0 and env.game.active_player == 0
    for s in range(60):
        action = int(z["actions"][s, 0])
        state, reward, done, hand = env.step(action)
        assert np.float64(reward).tobytes() == z["reward"][s, 0].tobytes() and done == bool(z["done"][s, 0]) and hand == bool(z["hand_over"][s, 0]), s
        assert GU.bits_equal(z["post_credits"][s, 0], env.game.credits)
        if done:
            state = env.reset()
    assert isinstance(state.valid_actions, np.ndarray) and PokerMoves.FOLD == 0
    env.close()


def test_call_agent_rollout_and_pick_vs_oracle(O):
    """PK_POLICY_CALL in the rollout kernels (k_rollout_call, fused and one step per launch), pk_pick_actions and the lockstep
    path, against the oracle's call agent (rng_spec: CALL if valid, else CHECK if valid, else ALL_IN)."""
    from hip_backend import HipBackend as HB
    for T, N, K in [(4096, 6, 120), (1000, 3, 200), (65536, 2, 40)]:
        o = O.OracleGame(T, N, seed=321)
        h = HB(T, N, seed=321)
        o.reset(); h.reset()
        a = o.pick_actions(2)
        assert np.array_equal(a, h.pick_actions(2)) and set(np.unique(a)) <= {1, 2, 6}
        co, err = o.rollout(K, 2, True)
        ch = h.rollout(K, 2, True)
        assert ch.tolist() == co.tolist(), (T, N)
        co2, _ = o.rollout(7, 2, True)
        ch2 = h.rollout(7, 2, True, fused=False)
        assert ch2.tolist() == co2.tolist()
        for _ in range(5):                                   # asynchronous calls (deferred / merged) of the call agents
            h.g.rollout(9, 2, True, True, counters=False)
        o.rollout(45, 2, True)
        so, sh = o.snapshot(), h.snapshot()
        for k in GU.SNAP_FIELDS:
            assert GU.bits_equal(so[k], sh[k]), (T, N, k)
        h.g.close()


def test_env_sub_batches_inside_one_handle(O):
    """pk_set_env_batches: the handle's tables split into contiguous ranges on internal streams; a bounded
    pk_env_step_async_d call launches ONE range (reading actions only inside it) and delivers the range launched longest ago
    (pk_env_last_range).  Per table the delivered (reward, done, hand, obs row) sequence is the synchronous one's -- with
    seat 0's actions supplied by the caller as a function of the delivered row -- and a drain delivers everything."""
    import pokerl_amd
    from pokerl_amd import _lib as L
    from pokerl_amd.hipmem import DeviceBuffer
    lib = L.lib()

    def choose(row, k):   # seat 0's host-side policy: the (k mod #valid)-th valid action of the delivered row
        mask = row[:, 3:10] > 0
        nth = k % mask.sum(axis=1)
        return ((np.cumsum(mask, axis=1) - 1 == nth[:, None]) & mask).argmax(axis=1).astype(np.int32)

    # busy: the handle's stream has unfinished work when the call is made (it waits for an event of ANOTHER handle's long rollout), so
    # the launch is ordered after the caller's stream with an event pair; otherwise that stream is idle and nothing is recorded at all
    from pokerl_amd.hipmem import DeviceEvent
    other = pokerl_amd.VecGame(65536, num_players=6, seed=99)
    other.reset()
    other_ev = DeviceEvent()
    for T, N, opp, K, passes, B, busy in [(3 * 2048 + 100, 6, 0, 25, 5, 3, False), (4096, 3, 1, 20, 2, 4, True), (1000, 9, 0, 15, 7, 8, False),
                                          (2 * 2048 + 7, 6, 0, 12, 4, 3, True)]:
        D = 17 + 3 * N
        rew, done, hand, terr, obs, ready, act = (DeviceBuffer(T * 8), DeviceBuffer(T), DeviceBuffer(T), DeviceBuffer(T),
                                                  DeviceBuffer(T * D * 8), DeviceBuffer(T), DeviceBuffer(T * 4))
        outputs = lambda: (rew.download(np.float64, T), done.download(np.uint8, T), hand.download(np.uint8, T),
                           obs.download(np.float64, T * D).reshape(T, D))
        # ---- the synchronous sequences (checked against the oracle)
        env = pokerl_amd.VecPokerGameEnv(opp, num_tables=T, num_players=N, seed=4321)
        g = env.game
        o = O.OracleGame(T, N, seed=4321)
        row = env.reset(); o.env_reset(None, opp)
        want = dict(rew=np.zeros((T, K)), done=np.zeros((T, K), np.uint8), hand=np.zeros((T, K), np.uint8), obs=np.zeros((T, K, D)))
        for k in range(K):
            a = choose(row, np.full(T, k))
            ro, do, ho, eo = o.env_step(a, opp)
            m = ((do != 0) | ((eo & 12) != 0)).astype(np.uint8)
            if m.any():
                o.env_reset(m, opp)
            act.upload(a)
            L.check(lib.pk_env_step_fused_d(g._h, act.ptr, 0, opp, 1, rew.ptr, done.ptr, hand.ptr, terr.ptr, obs.ptr), g._h)
            g.sync()
            want["rew"][:, k], want["done"][:, k], want["hand"][:, k], want["obs"][:, k] = outputs()
            row = want["obs"][:, k]
            assert GU.bits_equal(ro, want["rew"][:, k]) and np.array_equal(do, want["done"][:, k]) and np.array_equal(ho, want["hand"][:, k])
        g.close()
        # ---- the same tables through one handle with B sub-batches
        env = pokerl_amd.VecPokerGameEnv(opp, num_tables=T, num_players=N, seed=4321)
        g = env.game
        row = env.reset().copy()
        nb = env.set_env_batches(B)
        assert 1 < nb <= B
        b0, e0, fresh = env.last_range()
        assert (b0, fresh) == (0, True) and 0 < e0 < T and e0 % 64 == 0
        got = {k: np.zeros_like(v) for k, v in want.items()}
        count = np.zeros(T, np.int64)
        waiting = np.ones(T, bool)                  # tables that are to be given an action (all of them after the reset)
        ranges, launches, in_flight_seen = set(), 0, 0
        while count.min() < K:
            launches += 1
            assert launches < 100 * K * nb, ("no progress", T, N, B)
            lb, le, _ = env.last_range()            # the range this call will launch = the one delivered last
            a = np.full(T, -1, np.int32)            # garbage outside the range and for tables in flight: must be ignored
            idx = np.nonzero(waiting[lb:le])[0] + lb
            a[idx] = choose(row[idx], count[idx])
            waiting[lb:le] = False
            act.upload(a)
            if busy:
                other.rollout(1500, 0, True, True, counters=False)      # ~3 ms of work on ANOTHER stream ...
                other.record_event(other_ev.handle)
                g.wait_event(other_ev.handle)                           # ... that this handle's stream now waits for
            env.step_async_d(act.ptr, rew.ptr, done.ptr, hand.ptr, terr.ptr, obs.ptr, ready.ptr, max_passes=passes)
            g.sync()
            db, de, fresh = env.last_range()
            ranges.add((db, de))
            if fresh:                               # not launched yet: untouched, every table awaits its first action
                assert waiting[db:de].all()
                continue
            r = ready.download(np.uint8, T)[db:de] != 0
            in_flight_seen += int((~r).sum())
            assert not terr.download(np.uint8, T)[db:de][r].any()
            out = outputs()
            t_idx = np.nonzero(r)[0] + db
            row[t_idx] = out[3][t_idx]
            use = t_idx[count[t_idx] < K]
            c = count[use]
            got["rew"][use, c], got["done"][use, c], got["hand"][use, c], got["obs"][use, c] = (x[use] for x in out)
            count[t_idx] += 1
            waiting[t_idx] = True
        assert len(ranges) == nb and in_flight_seen > 0, (ranges, in_flight_seen)
        for k in want:
            same = GU.bits_equal(want[k], got[k]) if want[k].dtype == np.float64 else np.array_equal(want[k], got[k])
            assert same, (T, N, B, k)
        with pytest.raises(L.PokerlHipError):
            g.credits                               # steps are in flight
        with pytest.raises(L.PokerlHipError):
            env.set_env_batches(2)
        # drain: every range runs to its end, everything is delivered, the handle is readable again
        a = np.full(T, -1, np.int32)
        a[waiting] = choose(row[waiting], count[waiting])
        act.upload(a)
        env.step_async_d(act.ptr, rew.ptr, done.ptr, hand.ptr, terr.ptr, obs.ptr, ready.ptr, max_passes=0)
        g.sync()
        assert env.last_range() == (0, T, False) and (ready.download(np.uint8, T) != 0).all()
        assert GU.bits_equal(obs.download(np.float64, T * D).reshape(T, D), g.observations)
        env.close()
    other.close()
