"""CPU-side tests (no GPU): the C-ABI library loads and exports every declared symbol, the host mirror's pure-host
logic, sharding, and the distributed aggregation of bench.py under gloo with world_size 2."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from pokerl_amd import _lib, build
    build.build_lib()
    header = open(os.path.join(ROOT, "include", "pokerl_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(pk_[a-z0-9_]+)\s*\(", header)))
    assert declared == sorted(_lib.SYMBOLS)
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert _lib.lib().pk_abi_version() == _lib.ABI_VERSION == 6


# The table kernels that DO use scratch memory, and why: the 168-register variants of the fused rollout (three waves per SIMD) spill at eight to
# ten seats -- measured to pay at eight seats from 524 288 tables on (pk_create picks it there; never at nine / ten): pk_kernels.hpp, PK_OCC_CAP.
SCRATCH_ALLOWED = {"k_rollout_occ3<8>", "k_rollout_occ3<9>", "k_rollout_occ3<10>", "k_rollout_occ3_allin<8>", "k_rollout_occ3_allin<9>", "k_rollout_occ3_allin<10>"}


def test_no_table_kernel_uses_scratch():
    """`.private_segment_fixed_size` of every kernel in the BUILT library's gfx950 code objects (tools/kernel_meta.py reads the metadata notes
    of the embedded offload bundles): 0 -- a dispatch of such a kernel sets up no scratch memory -- for every kernel but the allow-list above.
    (Round 5's k_step_async<6> / <14> carried a dead 36-byte stack object: VERDICT r05 weak #7.)"""
    from pokerl_amd import _lib, build
    build.build_lib()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_meta
    ks = kernel_meta.kernels(_lib.LIB_PATH)
    table_kernels = [k for k in ks if re.match(r"k_(reset|make_fresh|pick|rollout|step|env_)", k)]
    # 13 kernels x seats 2..16 + the two occ3 variants and k_rollout_allin_tab x seats 2..10 + k_rollout_tab x seats 2..6
    assert len(table_kernels) == 15 * 13 + 9 * 3 + 5, len(table_kernels)
    # the table-evaluator variants' one-wave workgroups must fit a CU four at a time (160 KB of LDS: 40 960 bytes each), or 65 536 tables run at half rate
    tab = {k: d["lds"] for k, d in ks.items() if k.startswith(("k_rollout_tab<", "k_rollout_allin_tab<"))}
    assert len(tab) == 5 + 9 and all(32768 < v <= 40960 for v in tab.values()), tab
    for base in ("k_step", "k_step_async", "k_rollout", "k_env_step_async"):
        assert all("%s<%d>" % (base, n) in ks for n in build.SEATS), base
    bad = {k: d["private_segment"] for k, d in ks.items() if d["private_segment"] != 0 and k not in SCRATCH_ALLOWED}
    assert not bad, bad
    assert all(ks[k]["private_segment"] > 0 for k in SCRATCH_ALLOWED)        # (the allow-list names real spills only: shrink it when one goes away)


def test_no_device_fails_loudly():
    import pokerl_amd
    if pokerl_amd.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(pokerl_amd.PokerlHipError, match="no CPU fallback"):
        pokerl_amd.VecGame(4, num_players=3)
    with pytest.raises(pokerl_amd.PokerlHipError):
        pokerl_amd.eval_hand(["AD", "KD"])


def test_create_refuses_bad_arguments_before_looking_for_a_device():
    """The argument checks of pk_create run on any machine: seat counts outside the ABI's range and NON-FINITE money (the
    library is built with -fno-honor-nans: an inf stack would become NaN at the first all-in) are PK_E_INVALID_ARG; a valid
    configuration gets as far as the device check."""
    import ctypes as C
    from pokerl_amd import _lib as L
    lib = L.lib()

    def create(n=6, credits=None, scalar=100.0, bb=2.0, sb=1.0):
        h = C.c_void_p()
        sc = None if credits is None else np.ascontiguousarray(credits, np.float64)
        rc = lib.pk_create(C.byref(h), 0, 64, n, L.ptr(sc), scalar, bb, sb, 0, 1, 0)
        msg = (lib.pk_last_error(None) or b"").decode()
        if rc == L.PK_OK:
            lib.pk_destroy(h)
        return rc, msg

    inf, nan = float("inf"), float("nan")
    for kw in (dict(scalar=inf), dict(scalar=nan), dict(bb=inf), dict(sb=-inf), dict(bb=nan),
               dict(credits=[100, 100, inf, 100, 100, 100]), dict(credits=[100, nan, 1, 1, 1, 1])):
        rc, msg = create(**kw)
        assert rc == L.PK_E_INVALID_ARG and "finite" in msg, (kw, rc, msg)
    for n in (1, L.MAX_PLAYERS + 1, 24):
        rc, msg = create(n=n)
        assert rc == L.PK_E_INVALID_ARG and "num_players" in msg, (n, rc, msg)
    rc, msg = create()
    assert rc in (L.PK_OK, L.PK_E_NO_DEVICE), (rc, msg)
    with pytest.raises(ValueError):
        import pokerl_amd
        pokerl_amd.VecGame(4, num_players=L.MAX_PLAYERS + 1)


def test_packed_observation_rows_unpack_to_the_dense_layout():
    """state_view.packed_dtype / unpack_obs: the host side of PK_OBS_PACKED_BYTES (the device side is a -m gpu test)."""
    from pokerl_amd import packed_dtype, unpack_obs
    header = open(os.path.join(ROOT, "include", "pokerl_hip.h")).read()
    assert "#define PK_OBS_PACKED_BYTES(n) (16 + 8 * (3 * (n) + 1))" in header
    for n in (2, 6, 9, 15):
        dt = packed_dtype(n)
        assert dt.itemsize == 16 + 8 * (3 * n + 1) and dt.fields['minimum_raise_value'][1] == 16
        rng = np.random.default_rng(n)
        rows = np.zeros(5, dt)
        rows['player'] = rng.integers(0, n, 5); rows['turn'] = [0, 1, 2, 3, 4]; rows['valid_bits'] = [0x43, 0x7f, 0x45, 0x41, 0x47]
        rows['player_cards'] = rng.integers(0, 0x3d, (5, 2))
        cc = rng.integers(0, 0x3d, (5, 5)).astype(np.uint8)
        for i, turn in enumerate(rows['turn']):
            cc[i, (0 if turn == 0 else turn + 2):] = 0xFF
        rows['community_cards'] = cc
        for f in ('credits', 'bets', 'pending_bets'):
            rows[f] = rng.random((5, n)) * 100 - 3
        rows['minimum_raise_value'] = rng.random(5)
        dense = unpack_obs(rows, n)
        assert dense.shape == (5, 17 + 3 * n)
        assert dense[:, 0].tolist() == rows['player'].tolist() and dense[:, 1].tolist() == [0, 1, 2, 3, 4]
        assert dense[1, 3:10].tolist() == [1] * 7 and dense[0, 3:10].tolist() == [1, 1, 0, 0, 0, 0, 1]
        assert dense[0, 12:17].tolist() == [-1] * 5 and (dense[1, 12:15] >= 0).all() and dense[1, 15:17].tolist() == [-1, -1]
        assert dense[:, 17:17 + n].tobytes() == rows['credits'].tobytes() and dense[:, 2].tobytes() == rows['minimum_raise_value'].tobytes()
        # raw bytes in, same rows out
        assert np.array_equal(unpack_obs(rows.view(np.uint8).reshape(5, -1), n), dense)


def test_pool_plan_shards_like_ranks():
    """VecPokerGameEnvPool(devices=[...]): one contiguous block per device, cut as shard_tables cuts them for ranks."""
    from pokerl_amd import VecPokerGameEnvPool, shard_tables
    slices, devs = VecPokerGameEnvPool.plan(524288, devices=list(range(8)))
    assert devs == list(range(8)) and [(s.stop - s.start, s.start) for s in slices] == [shard_tables(524288, r, 8) for r in range(8)]
    slices, devs = VecPokerGameEnvPool.plan(10, devices=[0, 0, 1])
    assert [(s.start, s.stop) for s in slices] == [(0, 4), (4, 7), (7, 10)] and devs == [0, 0, 1]
    slices, devs = VecPokerGameEnvPool.plan(2, num_batches=4, default_device=3)
    assert [(s.start, s.stop) for s in slices] == [(0, 1), (1, 2)] and devs == [3, 3]


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "pokerl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src, os.path.join(dirpath, f)  # not imported, linked, loaded or even named


def test_shard_tables():
    from pokerl_amd import shard_tables
    for total, world in [(524288, 8), (65536, 1), (10, 3), (7, 8)]:
        parts = [shard_tables(total, r, world) for r in range(world)]
        assert sum(n for n, _ in parts) == total
        pos = 0
        for n, base in parts:
            assert base == pos
            pos += n
    assert shard_tables(524288, 3, 8) == (65536, 196608)
    with pytest.raises(ValueError):
        shard_tables(10, 3, 3)


def test_card_helpers_match_reference_encoding():
    from pokerl_amd import cards
    # pokerl/cards.py: '1D' and 'AD' are both the ace of diamonds, value (2<<4)|0; rank property makes it 13
    assert cards.card_value("AD") == cards.card_value("1D") == 0x20
    assert cards.card_rank(0x20) == 13 and cards.card_suit(0x20) == 2 and cards.card_id(0x20) == 26
    assert cards.card_value("KC") == 0x3c and cards.card_value((5, 1)) == 0x15 and cards.card_value((13, 0)) == 0
    d = cards.default_deck_values()
    assert d[:5].tolist() == [0x00, 0x10, 0x20, 0x30, 0x01] and len(set(d.tolist())) == 52
    from oracle import rng_spec
    assert d.tolist() == rng_spec.canonical_deck_values()


def test_enums_match_golden_constants():
    from pokerl_amd import HandRanking, PlayerState, PokerMoves
    assert (HandRanking.STRAIGHT_FLUSH, HandRanking.HIGH, HandRanking.NONE) == (1, 9, 10)
    assert (PokerMoves.FOLD, PokerMoves.CHECK, PokerMoves.CALL, PokerMoves.RAISE_TEN, PokerMoves.ALL_IN) == (0, 1, 2, 3, 6)
    assert (PlayerState.FOLDED, PlayerState.ACTIVE, PlayerState.CALLED, PlayerState.ALL_IN, PlayerState.BROKEN) == (0, 1, 2, 3, 4)


WORKER = r'''
import os, sys, json
sys.path.insert(0, %r)
import bench
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
ctx = bench.DistContext(backend="gloo")
n, base = bench.shard(65536 * world, ctx)
# stand-in workload: rank r "runs" n*10 steps in (1 + r) seconds
total_steps, seconds = ctx.aggregate(n * 10, 1.0 + rank)
bases = ctx.sum_list([base if r == rank else 0 for r in range(world)])     # every rank's table_id_base, gathered by SUM
dist_block = ctx.describe(rank %% 4, n)                                       # the `dist` block of the bench line (collective)
ctx.barrier()
if rank == 0:
    print(json.dumps(dict(n=n, base=base, total=total_steps, seconds=seconds, world=world, bases=bases, dist=dist_block,
                          cfg=bench.baseline_config_index(65536, 6, "random", world))))
ctx.close()
'''


def test_bench_line_is_compact():
    """bench.py's stdout line must fit the ~8 KB tail of stdout the driver keeps (round 4's 20.9 KB line was never parsed): the full
    result of that very run (profiles/r04_bench_driver.json) goes through compact_line() and comes out under 4 KB with every contract
    key, `roofline` and `cpu_baseline`; what is dropped stays in the detail file."""
    import json
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_driver.json")))
    assert len(json.dumps(full)) > 20000
    full["dist"] = {"backend": "nccl", "world": 8, "devices": list(range(8)), "tables_per_rank": [65536] * 8, "collectives_on_step_path": 0}
    full["extra_workloads"].append({"name": "a leg that failed", "error": "RuntimeError: " + "x" * 500})
    text = bench.compact_line(full)
    assert len(text) < bench.LINE_LIMIT == 4096 and "\n" not in text
    c = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "evaluator", "extra_workloads", "dist", "detail"):
        assert k in c, k
    assert abs(c["value"] - full["value"]) / full["value"] < 1e-4 and c["config"]["workload"] == full["config"]["workload"]
    rf = c["roofline"]
    assert set(rf) == {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "steps_per_launch", "source", "hbm_algorithmic"}
    assert rf["bound"] == "valu-issue" and abs(rf["frac"] - full["roofline"]["frac"]) < 1e-4 and list(rf["hbm_algorithmic"]) == ["frac"]
    assert c["cpu_baseline"]["cores"] == 1 and c["cpu_baseline"]["all_cores"]["cores"] == 16 and c["cpu_baseline"]["kind"] == "port"
    assert len(c["extra_workloads"]) == 7 and all(len(x["name"]) <= 40 for x in c["extra_workloads"])
    assert c["extra_workloads"][-1]["error"].startswith("RuntimeError") and len(c["extra_workloads"][-1]["error"]) <= 80
    assert c["extra_workloads"][3]["bound"] == "valu-issue" and 0 < c["extra_workloads"][3]["hbm_frac"] < 1      # env leg: measured HBM beside VALU
    assert c["dist"]["world"] == 8
    # round 6: the line checks itself -- the timed seconds, the steps per table inside them and the library's source hash travel in it -- and a
    # roofline figure that rests on a committed counter summary of OTHER kernel sources says so
    full.update({"reps": 26215, "samples": 7, "timed_steps_per_table": 20 * 26215 * 7, "timed_s": 7.85, "lib": "abi=6 src=0123456789abcdef"})
    full["roofline"]["profile_stale"] = True
    full["extra_workloads"][3]["roofline"]["profile_stale"] = True
    full["extra_workloads"][0]["roofline"]["profile_stale"] = False
    c = json.loads(bench.compact_line(full))
    assert c["timed_steps_per_table"] == 3670100 and c["timed_s"] == 7.85 and c["reps"] == 26215 and c["samples"] == 7 and c["lib"].endswith("0123456789abcdef")
    assert c["roofline"]["profile_stale"] is True and c["extra_workloads"][3]["profile_stale"] is True and "profile_stale" not in c["extra_workloads"][0]
    assert len(bench.compact_line(full)) < bench.LINE_LIMIT


def test_source_hash_ignores_comments_and_is_embedded():
    """pokerl_amd/build.py source_hash(): over the kernel sources without comments / white space + the compiler flags; the built library
    reports the hash it was built from (pk_build_info) and the summaries under profiles/ that were made in round 6 or later carry one."""
    import glob
    import json
    from pokerl_amd import _lib, build
    build.build_lib()
    h = build.source_hash()
    assert re.fullmatch(r"[0-9a-f]{16}", h) and _lib.lib().pk_build_info().decode() == "abi=%d src=%s" % (_lib.ABI_VERSION, h) and _lib.source_hash() == h
    import bench
    assert bench.profile_stale({}) and bench.profile_stale({"source_hash": "0" * 16}) and not bench.profile_stale({"source_hash": h})
    for f in glob.glob(os.path.join(ROOT, "profiles", "r0[6-9]_*_summary.json")):
        assert re.fullmatch(r"[0-9a-f]{16}", json.load(open(f)).get("source_hash") or ""), f


def test_bench_aggregation_gloo_world2(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29531", str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["world"] == 2 and r["n"] == 65536 and r["base"] == 0
    assert r["total"] == 131072 * 10          # units of ALL ranks
    assert r["seconds"] == 2.0                # MAX over ranks
    assert r["dist"] == {"backend": "gloo", "world": 2, "devices": [0, 1], "tables_per_rank": [65536, 65536], "collectives_on_step_path": 0}


def test_bench_aggregation_gloo_world8(tmp_path):
    """BASELINE configs[3] end to end without a node: the `--gpus 8` code path of bench.py (DistContext, contiguous shards by
    global table id, barrier, SUM of units, MAX of seconds) with eight gloo ranks -- shards (65 536, r * 65 536)."""
    script = tmp_path / "w8.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                          "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["world"] == 8 and r["n"] == 65536 and r["base"] == 0 and r["cfg"] == 3
    assert r["bases"] == [k * 65536 for k in range(8)]          # rank k hosts tables [k * 65 536, (k + 1) * 65 536)
    assert r["total"] == 524288 * 10                            # units of ALL ranks
    assert r["seconds"] == 8.0                                  # MAX over ranks (rank 7: 1 + 7 s)
    assert r["dist"]["world"] == 8 and r["dist"]["devices"] == [0, 1, 2, 3, 0, 1, 2, 3] and r["dist"]["tables_per_rank"] == [65536] * 8


GATHER_WORKER = r'''
import os, sys, json
sys.path.insert(0, %r)
import numpy as np
import torch.distributed as dist
from pokerl_amd import sharding
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n_local, base = sharding.shard_tables(1001, rank, world)          # uneven shards: 334 / 334 / 333

class FakeGame:                                                   # the host side of gather_f64 needs only these
    num_tables, num_players, device = n_local, 3, 0
    def _f64(self, field):
        t = np.arange(base, base + n_local, dtype=np.float64)[:, None]
        return t * 10 + np.arange(3)[None, :] + field * 0.25

out = sharding.gather_f64(FakeGame(), 3, dist)
if rank == 0:
    t = np.arange(1001, dtype=np.float64)[:, None]
    print(json.dumps(dict(ok=bool(np.array_equal(out, t * 10 + np.arange(3)[None, :] + 0.75)), shape=list(out.shape))))
dist.destroy_process_group()
'''


def test_optional_payoff_gather_gloo_world3(tmp_path):
    """north_star's optional payoff gather (no collective on the step path): every rank receives the field of all tables in
    global table order; gloo ranks exchange host arrays (RCCL ranks exchange device buffers filled by pk_get_f64_d)."""
    script = tmp_path / "g.py"
    script.write_text(GATHER_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
                          "--master-addr", "127.0.0.1", "--master-port", "29547", str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["ok"] and r["shape"] == [1001, 3]


def test_agents_module_maps_in_kernel_agents_and_applies_their_rules_on_the_host():
    """pokerl_amd.agents (reference pokerl/agents): in-kernel markers map to policy nibbles, anything else is a host agent;
    called on the host the markers apply the same RULE to a StateView (agents/random.py:12-18 for the random one)."""
    from pokerl_amd import AllInAgent, CallAgent, PokerAgent, Policy, RandomAgent, StateView
    from pokerl_amd.agents import kernel_policy
    assert [kernel_policy(a) for a in (RandomAgent(), AllInAgent(), CallAgent(), Policy.CALL, 1, np.int64(0))] == [0, 1, 2, 2, 1, 0]
    assert kernel_policy(lambda s: 0) is None and kernel_policy(PokerAgent()) is None and kernel_policy(True) is None
    with pytest.raises(ValueError):
        kernel_policy(7)                                   # not a Policy
    with pytest.raises(NotImplementedError):
        PokerAgent()(None)                                 # agents/agent.py:11-13
    n = 3
    row = np.array([1, 0, 2.0,  1, 0, 1, 1, 0, 0, 1,  0x20, 0x3c,  -1, -1, -1, -1, -1,  90, 80, 70,  0, 0, 0,  1, 2, 0], np.float64)
    sv = StateView(row, n)                                 # valid: FOLD, CALL, RAISE_TEN, ALL_IN
    assert CallAgent()(sv) == 2 and AllInAgent()(sv) == 6
    np.random.seed(3)
    assert {RandomAgent()(sv) for _ in range(200)} == {0, 2, 3, 6}
    row[5] = 0                                             # CALL invalid, CHECK invalid -> the call agent shoves
    assert CallAgent()(StateView(row, n)) == 6
    row[4] = 1
    assert CallAgent()(StateView(row, n)) == 1             # ... or checks where it can


def test_state_view_mirror_fields_and_pickle():
    """StateView / Card host mirrors (reference game.py:39-240, cards.py:4-72): built from a dense observation row."""
    import pickle
    from pokerl_amd import Card, StateView
    n = 3
    row = np.array([1, 2, 4.5,  1, 0, 1, 1, 0, 0, 1,  0x20, 0x3c,  0x01, 0x15, 0x2b, 0x09, -1,
                    90, 80, 70,  5, 6, 7,  1, 2, 3], np.float64)
    sv = StateView(row, n)
    assert (sv.player, sv.turn, sv.num_players, sv.minimum_raise_value) == (1, 2, 3, 4.5)
    assert list(sv.valid_action_indices) == [0, 2, 3, 6]
    assert [c.value for c in sv.player_cards] == [0x20, 0x3c] and len(sv.community_cards) == 4   # turn 2: four cards
    assert sv.player_cards[0].rank == 13 and sv.player_cards[0].suit == 2 and sv.player_cards[0].id == 26
    assert sv.pot == 18.0 and sv.high_bet == 3.0 and sv.credit == 80.0
    assert len(sv.player_hand) == 6
    state = sv.__getstate__()
    assert len(state) == 10 and state[0] == 1 and state[3] == 2          # tuple order of game.py:211-223
    sv2 = pickle.loads(pickle.dumps(sv))
    assert sv2.credit == 80.0 and sv2.player_cards == sv.player_cards
    assert Card("AD") == Card(0x20) == Card((13, 2)) and repr(Card("KC"))
