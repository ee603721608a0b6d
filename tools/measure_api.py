#!/usr/bin/env python3
"""Rates of the host-buffer entry points (PCIe/launch-inclusive; never the bench `value`). Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pokerl_amd

T, N = 65536, 6
g = pokerl_amd.VecGame(T, num_players=N); g.reset()
for name, fn in [("pk_pick_actions + pk_step (host numpy in/out, strict=False)", lambda: g.step(g.pick_actions(0), strict=False)),
                 ("pk_get_obs (dense StateView rows to host)", lambda: g.observations)]:
    fn(); t0 = time.perf_counter(); n = 50
    for _ in range(n):
        out = fn()
        if isinstance(out, tuple) and (out[0]).any(): g.reset(mask=out[0].astype(np.uint8))
    dt = (time.perf_counter() - t0) / n
    print("%-62s %8.1f us/call  %7.1f M table-ops/s" % (name, dt * 1e6, T / dt / 1e6))
env = pokerl_amd.VecPokerGameEnv(0, num_tables=T, num_players=N); env.reset()
a = np.full(T, 6, np.int32)
t0 = time.perf_counter(); n = 30
for _ in range(n):
    obs, r, d, h = env.step(env.game.pick_actions(0))
    if d.any(): env.reset(d.astype(np.uint8))
dt = (time.perf_counter() - t0) / n
print("%-62s %8.1f us/call  %7.1f M env.step/s" % ("VecPokerGameEnv.step + obs (opponents in-kernel)", dt * 1e6, T / dt / 1e6))
