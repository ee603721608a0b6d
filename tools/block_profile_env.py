#!/usr/bin/env python3
"""Diagnostic: where a wave's cycles go inside the asynchronous PokerGameEnv.step kernel (s_memtime stamps of the -DPK_PROFILE
build, pokerl_amd/libpokerl_hip_prof.so).  Run on the GPU box:  python tools/block_profile_env.py [N] [passes] [tables]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("POKERL_HIP_LIB", os.path.join(ROOT, "pokerl_amd", "libpokerl_hip_prof.so"))
import numpy as np  # noqa: E402
import pokerl_amd  # noqa: E402
from pokerl_amd import _lib as L  # noqa: E402
from pokerl_amd.hipmem import DeviceBuffer  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 8
T = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
D = 17 + 3 * N
env = pokerl_amd.VecPokerGameEnv(0, num_tables=T, num_players=N)
g = env.game
env.reset()
rew, done, hand, terr, obs, ready = (DeviceBuffer(T * 8), DeviceBuffer(T), DeviceBuffer(T), DeviceBuffer(T), DeviceBuffer(T * D * 8), DeviceBuffer(T))
step = lambda: env.step_async_d(None, rew.ptr, done.ptr, hand.ptr, terr.ptr, obs.ptr, ready.ptr, max_passes=passes)
for _ in range(100):
    step()
g.sync()
lib = L.lib()
lib.pk_prof_read.argtypes = [C.c_void_p, C.c_void_p]
buf = np.zeros(16, np.uint64)
lib.pk_prof_read(g._h, L.ptr(buf))           # reads and clears
launches, delivered = 400, 0
for _ in range(launches):
    step()
    g.sync()
    delivered += int((ready.download(np.uint8, T) != 0).sum())
lib.pk_prof_read(g._h, L.ptr(buf))
names = ["passes (pick + begin_step + cursor + retire)", "load + census between the rounds", "end_pre", "eval", "sidepot", "setup", "deal", "other (loop exit)",
         None, None, None, None, None, None, "episode reset (reset_state + deal)", "action draws (Philox refills)"]
waves = (T + 63) // 64
tot = float(buf[:8].sum() + buf[14] + buf[15])
print("N=%d T=%d passes=%d: %.3f of the tables ready per launch (diagnostic build; read shares, not time)" % (N, T, passes, delivered / float(T * launches)))
for i, n in enumerate(names):
    if n is None:
        continue
    print("  %-48s %6.1f %%   %8.0f cycles per wave and launch" % (n, 100 * buf[i] / tot, buf[i] / waves / launches))
print("  betting passes per launch %.2f  end_blocks per launch %.2f  eval passes %.2f   total cycles per wave and launch %.0f" % (
    buf[8] / waves / launches, buf[9] / waves / launches, buf[10] / waves / launches, tot / waves / launches))
env.step_async_d(None, rew.ptr, done.ptr, hand.ptr, terr.ptr, obs.ptr, ready.ptr, max_passes=0)
g.sync()
