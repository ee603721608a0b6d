#!/usr/bin/env python3
"""The loop of the reference's examples/random_game.py with the CALLER's policy and every buffer in HBM: observation rows and valid
masks are read on the device, a (toy) policy kernel would write actions_d, Game.step runs on the device buffers and finished games
restart inside the step's launch (pk_step_auto_d).  Here the "policy" is the library's own in-kernel pick written to actions_d
(pk_pick_actions_d) -- replace it with your network; a torch user passes tensor.data_ptr() instead of DeviceBuffer.

    python examples/device_game_loop.py [tables=65536] [steps=2000]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import pokerl_amd  # noqa: E402
from pokerl_amd import _lib as L  # noqa: E402
from pokerl_amd.hipmem import DeviceBuffer  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
N = 6
game = pokerl_amd.VecGame(T, num_players=N)
game.reset()
actions, flags, terr = DeviceBuffer(T * 4), DeviceBuffer(T), DeviceBuffer(T)
obs = DeviceBuffer(T * (17 + 3 * N) * 8)                       # dense StateView rows of the player to act: what the policy reads
L.check(L.lib().pk_get_obs_d(game._h, -1, obs.ptr), game._h)   # the rows of the freshly reset tables (one launch, once) ...
game.set_step_obs(obs, None)                                   # ... from now on every step_d / step_async_d writes `game.active_state` itself
                                                               #     (pk_set_step_obs: from the step kernel's registers, no launch of its own)
for _ in range(4000):                                          # warm-up (~0.1 s): the first launches of a process load code objects and ramp the clock
    game.pick_actions_d(actions, pokerl_amd.Policy.RANDOM)
    game.step_d(actions, flags, terr, auto_reset=True)
game.sync()
games = 0
t0 = time.perf_counter()
for s in range(steps):
    # `obs` holds the row of every table's player to act here (written by the previous step_d): what a network would read
    game.pick_actions_d(actions, pokerl_amd.Policy.RANDOM)     # <- your policy kernel goes here: int32[T], valid for each active player
    game.step_d(actions, flags, terr, auto_reset=True)         # Game.step on every table; a finished game is Game.reset() on the spot
    if s % 500 == 499:                                         # looking at the flags is a host round trip: do it rarely
        game.sync()
        games += int((flags.download(np.uint8, T) & L.FLAG_GAME_OVER).sum())
game.sync()
dt = time.perf_counter() - t0
assert not (terr.download(np.uint8, T) & L.TERR_INVALID_ACTION).any()
print("%d tables x %d device-resident Game.steps in %.3f s = %.2f G steps/s (%.1f us per step of the whole batch); "
      "%d games ended in the sampled steps; mean pot now %.2f" % (T, steps, dt, T * steps / dt / 1e9, dt / steps * 1e6, games, game.pot.mean()))

# The same loop as BOUNDED launches (pk_step_async_d): a synchronous step launch lasts as long as the one table whose step rolls through
# several hands (game.py:607-611 plays hands nobody can act in); here such a table stays in flight, ready[t] says whose step has returned,
# and the loop acts on those.  Drain (max_hands=0) before touching the game through any other method.
ready = DeviceBuffer(T)
delivered = 0
game.sync()
t0 = time.perf_counter()
for s in range(steps):
    game.pick_actions_d(actions, pokerl_amd.Policy.RANDOM)     # a device reader: allowed while steps are in flight (rows of tables in flight: ignore)
    game.step_async_d(actions, flags, terr, ready, max_hands=1, auto_reset=True)
    if s % 500 == 499:
        game.sync()
        delivered += int(ready.download(np.uint8, T).sum())    # (sampled: 99.98 % of the tables are ready after every launch)
game.sync()
te = terr.download(np.uint8, T)
assert not (te[ready.download(np.uint8, T) != 0] & L.TERR_INVALID_ACTION).any()      # the picks were valid for every table that took one
# A drain with an actions buffer is a full step call: it would step the ready tables AGAIN with whatever the buffer holds.  actions = None (or -1 for
# a table) means "no step": those tables come back untouched with TERR_INVALID_ACTION (by design, not an error); the tables in flight finish.
game.step_async_d(None, flags, terr, ready, max_hands=0, auto_reset=True)      # drain only (actions None): every table ready, nobody stepped
game.sync()
assert (ready.download(np.uint8, T) != 0).all() and not (terr.download(np.uint8, T) & ~np.uint8(L.TERR_INVALID_ACTION | L.TERR_HAND_CAP)).any()
dt = time.perf_counter() - t0
print("bounded launches: %.1f us per step of the whole batch (%.2f G steps/s if every table were ready; %.4f of them were in the sampled launches)"
      % (dt / steps * 1e6, T * steps / dt / 1e9, delivered / float(T * max(1, steps // 500))))
game.set_step_obs(None, None)                                  # the handle keeps raw pointers: unset them before freeing the buffer
for b in (actions, flags, terr, obs, ready):
    b.free()
game.close()
