#!/usr/bin/env python3
"""Diagnostic: where a wave's cycles go inside the fused rollout kernel (s_memtime stamps, -DPK_PROFILE build).
Run on the GPU box:  python tools/block_profile.py [N] [policy]   (uses pokerl_amd/libpokerl_hip_prof.so)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("POKERL_HIP_LIB", os.path.join(ROOT, "pokerl_amd", "libpokerl_hip_counts.so" if os.environ.get("PK_COUNTS") else "libpokerl_hip_prof.so"))
import numpy as np  # noqa: E402
import pokerl_amd  # noqa: E402
from pokerl_amd import _lib as L  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
policy = int(sys.argv[2]) if len(sys.argv) > 2 else 0
T, K = int(os.environ.get("PK_BP_T", "65536")), int(os.environ.get("PK_BP_K", "512"))
FUSED = os.environ.get("PK_BP_FUSED", "1") != "0"   # 0: K single-step launches (k_rollout_single: the state round-trips HBM every step)
g = pokerl_amd.VecGame(T, num_players=N)
g.reset()
g.rollout(K, policy, True, FUSED)
lib = L.lib()
lib.pk_prof_read.argtypes = [C.c_void_p, C.c_void_p]
buf = np.zeros(16, np.uint64)
lib.pk_prof_read(g._h, L.ptr(buf))
launches = 4
ms, c = g.time_rollout(K, policy, True, FUSED, launches)
if not FUSED:
    ms *= K     # time_rollout reports per LAUNCH; below everything is per block of K steps
lib.pk_prof_read(g._h, L.ptr(buf))
names = ["action+valid", "cursor", "end_pre", "eval", "sidepot", "setup", "deal", "other"]
waves = max(1, (T + 63) // 64) if T >= 65536 else 1024
tot = float(buf[:8].sum())
print("N=%d T=%d %s policy=%d  %.3f ms/launch  %.2f G env-steps/s (diagnostic build; read shares, not time)" % (N, T, "fused" if FUSED else "single-step launches", policy, ms, T * K / ms / 1e6))
for i, n in enumerate(names):
    print("  %-14s %6.1f %%   %8.0f cycles/wave-step" % (n, 100 * buf[i] / tot, buf[i] / waves / launches / K))
print("  cursor passes/step %.2f  end_blocks/step %.2f  eval passes/step %.2f  sidepot iters/step %.2f" % tuple(
    buf[8 + i] / waves / launches / K for i in range(4)))
print("  total cycles/wave-step %.0f" % (tot / waves / launches / K))
if buf[11]:
    ends = float(buf[9])
    print("  per end_block: %.1f lanes served; side-pot loop %.2f wave-iterations with %.1f lanes active each" % (
        buf[13] / ends, buf[11] / ends, buf[12] / max(1.0, float(buf[11]))))
