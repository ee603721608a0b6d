#!/usr/bin/env python3
"""Times the REFERENCE itself (pure Python, imported read-only from /root/reference) on this container's cores: the loop
of examples/random_game.py:8-12 (random agent over `state.valid_action_indices`, `Game.step(agent(game.active_state))`,
`game.reset()` when the game is over), logging at WARNING, for N = 2 / 6 / 9 seats with Game's default configuration
(start_credits=100, blinds 2/1, game.py:246-251) -- BASELINE configs[0] plumbing -- with 1 process and with one process
per core.  BUILD CONTAINER ONLY (the reference does not exist on the GPU box); output: profiles/r02_reference_timing.json.

    PYTHONDONTWRITEBYTECODE=1 python tools/time_reference.py [seconds_per_run]
"""
import json
import multiprocessing as mp
import os
import platform
import sys
import time

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(args):
    n, seconds, seed = args
    sys.path.insert(0, "/root/reference")
    import logging
    import numpy as np
    from pokerl import Game
    logging.getLogger().setLevel(logging.WARNING)
    np.random.seed(seed)
    agent = lambda state: np.random.choice(list(state.valid_action_indices))   # examples/random_game.py:8
    game = Game(num_players=n)                                                 # defaults: 100 / 2 / 1
    game.reset()
    steps = games = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(200):
            done, *_ = game.step(agent(game.active_state))                     # examples/random_game.py:12
            steps += 1
            if done:
                games += 1
                game.reset()
    return steps, games, time.perf_counter() - t0


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
    ncores = len(os.sched_getaffinity(0))
    cpu = ""
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    import numpy
    out = dict(what="reference pokerl Game.step loop of examples/random_game.py:8-12, random agents, default config, "
                    "auto-reset, logging WARNING", cpu=cpu, cores=ncores, python=platform.python_version(),
               numpy=numpy.__version__, seconds_per_run=seconds, runs=[])
    for n in (2, 6, 9):
        s, g, dt = worker((n, seconds, 1))
        row = dict(num_players=n, processes=1, steps_per_s=s / dt, games=g)
        out["runs"].append(row)
        print(row, flush=True)
        with mp.Pool(ncores) as pool:
            t0 = time.perf_counter()
            res = pool.map(worker, [(n, seconds, 100 + i) for i in range(ncores)])
            wall = time.perf_counter() - t0
        row = dict(num_players=n, processes=ncores, steps_per_s=sum(r[0] for r in res) / max(r[2] for r in res),
                   wall_s=wall, games=sum(r[1] for r in res))
        out["runs"].append(row)
        print(row, flush=True)
    dst = os.path.join(ROOT, "profiles", "r02_reference_timing.json")
    json.dump(out, open(dst, "w"), indent=1)
    print("wrote", dst)


if __name__ == "__main__":
    main()
