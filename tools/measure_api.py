#!/usr/bin/env python3
"""Rates of the host-buffer entry points (PCIe/launch-inclusive; never the bench `value`). Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pokerl_amd
from pokerl_amd import packed_dtype, pinned_empty

T, N = 65536, 6
D = 17 + 3 * N


def rate(name, fn, n=50, unit="table-ops/s", post=None):
    out = fn()
    if post:
        post(out)
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
        if post:
            post(out)
    dt = (time.perf_counter() - t0) / n
    print("%-86s %8.1f us/call  %7.1f M %s" % (name, dt * 1e6, T / dt / 1e6, unit), flush=True)


g = pokerl_amd.VecGame(T, num_players=N); g.reset()
def reset_over(out):
    if isinstance(out, tuple) and out[0].any(): g.reset(mask=out[0].astype(np.uint8))
rate("pk_pick_actions + pk_step (host numpy in/out, strict=False)", lambda: g.step(g.pick_actions(0), strict=False), post=reset_over)
rate("pk_get_obs (dense StateView rows, 18.4 MB, fresh pageable array per call)", lambda: g.observations)
pin = pinned_empty((T, D), np.float64)
rate("pk_get_obs into a PINNED array (pk_host_alloc)", lambda: g.observations_of(None, out=pin.array))
pinp = pinned_empty(T, packed_dtype(N))
rate("pk_get_obs_packed (11.0 MB) into a pinned array", lambda: g.observations_packed_of(None, out=pinp.array))
rate("pk_get_obs_packed, fresh pageable array per call", lambda: g.observations_packed_of(None))

env = pokerl_amd.VecPokerGameEnv(0, num_tables=T, num_players=N); env.reset()
def reset_done(out):
    d = out[2]
    if d.any(): env.reset(d.astype(np.uint8))
rate("VecPokerGameEnv.step + obs (reference shape: strict check, fresh arrays, dense rows)",
     lambda: env.step(env.game.pick_actions(0)), n=30, unit="env.step/s", post=reset_done)
acts = np.zeros(T, np.int32)
def fast(obs):
    def f():
        env.send(env.game.pick_actions(0), obs=obs, auto_reset=True, strict=True)
        return env.recv()
    return f
rate("send + recv: pinned outputs, device-side check, auto-reset, PACKED rows", fast('packed'), n=30, unit="env.step/s")
rate("send + recv: pinned outputs, device-side check, auto-reset, DENSE rows", fast('dense'), n=30, unit="env.step/s")
rate("send + recv: pinned outputs, no observation rows", fast(None), n=30, unit="env.step/s")
env.close()
# two envs pipelined: env B's launch overlaps env A's device-to-host copies
pool = pokerl_amd.VecPokerGameEnvPool(0, num_tables=2 * T, num_batches=2, num_players=N); pool.reset()
a2 = np.concatenate([e.game.pick_actions(0) for e in pool.envs])
def piped():
    global a2
    outs = pool.step_pipelined(a2, obs='packed', auto_reset=True)
    a2 = np.concatenate([e.game.pick_actions(0) for e in pool.envs])
    return outs
fn = piped; fn(); t0 = time.perf_counter(); n = 30
for _ in range(n): fn()
dt = (time.perf_counter() - t0) / n
print("%-86s %8.1f us/call  %7.1f M env.step/s" % ("pool of two envs, step_pipelined (packed rows; incl. pick_actions of both)", dt * 1e6, 2 * T / dt / 1e6))
pool.close()
