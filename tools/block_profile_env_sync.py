#!/usr/bin/env python3
"""Diagnostic: where a wave's cycles go inside the SYNCHRONOUS fused PokerGameEnv.step kernel (-DPK_PROFILE build).
Run on the GPU box:  python tools/block_profile_env_sync.py [N] [tables]"""
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
T = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
D = 17 + 3 * N
env = pokerl_amd.VecPokerGameEnv(0, num_tables=T, num_players=N)
g = env.game
env.reset()
lib = L.lib()
rew, done, hand, terr, obs = (DeviceBuffer(T * 8), DeviceBuffer(T), DeviceBuffer(T), DeviceBuffer(T), DeviceBuffer(T * D * 8))
step = lambda: L.check(lib.pk_env_step_fused_d(g._h, None, 0, 0, 1, rew.ptr, done.ptr, hand.ptr, terr.ptr, obs.ptr), g._h)
for _ in range(30):
    step()
g.sync()
lib.pk_prof_read.argtypes = [C.c_void_p, C.c_void_p]
buf = np.zeros(16, np.uint64)
lib.pk_prof_read(g._h, L.ptr(buf))
launches = 100
for _ in range(launches):
    step()
g.sync()
lib.pk_prof_read(g._h, L.ptr(buf))
names = ["passes (pick + begin_step + cursor + retire)", "load + census between the rounds", "end_pre", "eval", "sidepot", "setup", "deal", "other (loop exit)",
         None, None, None, None, None, None, "episode reset (reset_state + deal)", "action draws (Philox refills)"]
waves = (T + 63) // 64
tot = float(buf[:8].sum() + buf[14] + buf[15])
print("N=%d T=%d synchronous fused env.step (diagnostic build; read shares, not time); cycles are SUMS over the waves, the launch lasts as long as its slowest wave" % (N, T))
for i, n in enumerate(names):
    if n is not None:
        print("  %-48s %6.1f %%   %8.0f cycles per wave and launch" % (n, 100 * buf[i] / tot, buf[i] / waves / launches))
print("  betting passes per launch %.2f  end_blocks per launch %.2f  eval passes %.2f   total cycles per wave and launch %.0f" % (
    buf[8] / waves / launches, buf[9] / waves / launches, buf[10] / waves / launches, tot / waves / launches))
