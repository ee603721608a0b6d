#!/usr/bin/env python3
"""Fuzz the CPU oracle against the REAL reference on random odd configurations (build container only: imports
/root/reference through make_golden's injection harness; nothing is written).  A configuration on which the reference
itself never returns (Game.step spins, docs/history.md section 2) is skipped after a time-out.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/fuzz_oracle_vs_reference.py [rounds] [min_seats max_seats]
(seats default 2..10: the generator sequence of the recorded runs; `200 11 15` fuzzes the wide tables)
"""
import os
import random
import signal
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import json  # noqa: E402

import make_golden as MG  # noqa: E402  (installs the injections)
import golden_util as GU  # noqa: E402
from oracle import loader as O  # noqa: E402

STACKS = [0.5, 1, 2, 3, 5, 10, 37.5, 100, 1000, 1e6]
BLINDS = [0, 0.25, 0.5, 1, 2, 3, 7.5, 40, 250]


class Timeout(Exception):
    pass


def _alarm(*_):
    raise Timeout()


def make_oracle(meta):
    cfg = meta["cfg"]
    return O.OracleGame(meta["tables"], meta["n"], cfg["start_credits"], cfg["big_blind"], cfg["small_blind"],
                        seed=meta["seed"], table_id_base=meta["table_id_base"])


def env_cap_hit(out, meta):
    """True if the oracle stops one of this trajectory's env calls with ORC_ERR_ENV_CAP (8)."""
    import numpy as np
    b = make_oracle(meta)
    opp = meta["opp_policy"]
    b.env_reset(None, opp)
    for s in range(meta["steps"]):
        _, done, _, err = b.env_step(out["actions"][s].astype(np.int32), opp)
        if (err & 8).any():
            return True
        if err.any():
            return False
        if done.any():
            b.env_reset(done, opp)
    return False


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2, 10)
    rng = random.Random(2026)
    signal.signal(signal.SIGALRM, _alarm)
    ok = skipped = capped = 0
    for i in range(rounds):
        n = rng.randint(lo, hi)
        same = rng.random() < 0.5
        start = rng.choice(STACKS) if same else [rng.choice(STACKS) for _ in range(n)]
        if same and isinstance(start, float) and start != int(start):
            start = [start] * n              # a float scalar would take the list branch of game.py:408 and fail
        cfg = dict(start_credits=int(start) if (same and not isinstance(start, list)) else start,
                   big_blind=rng.choice(BLINDS), small_blind=rng.choice(BLINDS))
        policy = 1 if rng.random() < 0.25 else 0
        seed, base, dealer = rng.getrandbits(63), rng.getrandbits(32), rng.randrange(n)
        signal.alarm(20)
        try:
            out = MG.game_trajectory(n, policy, seed, 3, 120, base, cfg, dealer=dealer)
        except Timeout:
            skipped += 1
            print("skip (reference spins): n=%d cfg=%s policy=%d" % (n, cfg, policy), flush=True)
            continue
        finally:
            signal.alarm(0)
        meta = json.loads(str(out["meta"]))
        try:
            GU.replay_game(make_oracle, "fuzz%d" % i, loaded=(out, meta))
        except AssertionError as e:
            import numpy as np
            hs = out["post_hand_serial"].astype(np.int64)
            prev = np.concatenate([out["init_hand_serial"][None].astype(np.int64), hs[:-1]])
            if "err [" in str(e) and (hs - prev).max() > 4096:
                capped += 1      # the reference rolled > PK_HAND_CAP hands inside one step: the documented cap rule
                print("cap  (reference rolled %d hands in one step): n=%d cfg=%s" % ((hs - prev).max(), n, cfg), flush=True)
                continue
            print("MISMATCH n=%d cfg=%s policy=%d seed=%d base=%d dealer=%d\n%s" % (n, cfg, policy, seed, base, dealer, e))
            return 1
        ok += 1
    print("fuzz oracle vs reference: %d configurations identical, %d skipped (reference never returns within 20 s), "
          "%d stopped by the hand cap where the reference rolled > 4096 hands in one step" % (ok, skipped, capped))
    # ---- PokerGameEnv.reset / step (envs/game_env.py:20-53) on odd configurations: reward / done / hand and the whole
    #      table state after every env.step and after every reset of a finished episode
    env_ok = env_skipped = env_capped = 0
    for i in range(max(10, rounds // 3)):
        n = rng.randint(lo, hi)
        same = rng.random() < 0.5
        stacks = [x for x in STACKS if x >= 2]
        start = int(rng.choice([x for x in stacks if x == int(x)])) if same else [rng.choice(stacks) for _ in range(n)]
        cfg = dict(start_credits=start, big_blind=rng.choice([b for b in BLINDS if 0 < b <= 40]),
                   small_blind=rng.choice([b for b in BLINDS if 0 < b <= 40]))
        opp = 1 if rng.random() < 0.25 else 0
        if i % 2:   # round 3: one agent per opponent seat (random / all-in / call), as PokerGameEnv(agents=[...]) takes them
            opp = [rng.choice([0, 0, 1, 2]) for _ in range(n - 1)]
        seed, base = rng.getrandbits(63), rng.getrandbits(32) & 0xFFFFFF00
        signal.alarm(30)
        try:
            out = MG.env_trajectory(n, 0, opp, seed, 3, 40, base, cfg)
        except Timeout:
            env_skipped += 1
            print("skip (reference env spins): n=%d cfg=%s opp=%s" % (n, cfg, opp), flush=True)
            continue
        finally:
            signal.alarm(0)
        meta = json.loads(str(out["meta"]))
        try:
            GU.replay_env(make_oracle, "envfuzz%d" % i, loaded=(out, meta))
        except AssertionError as e:
            if env_cap_hit(out, meta):      # the documented divergence: the reference plays on past ORC_ENV_STEP_CAP opponent steps
                env_capped += 1
                print("env cap (reference auto-plays more than 8 192 opponent steps in one call): n=%d cfg=%s opp=%s" % (n, cfg, opp), flush=True)
                continue
            print("ENV MISMATCH n=%d cfg=%s opp=%s seed=%d base=%d\n%s" % (n, cfg, opp, seed, base, e))
            return 1
        env_ok += 1
    print("fuzz oracle vs reference, PokerGameEnv: %d configurations identical, %d skipped, %d stopped by the env step cap"
          % (env_ok, env_skipped, env_capped))
    return 0


if __name__ == "__main__":
    sys.exit(main())
