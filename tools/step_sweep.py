#!/usr/bin/env python3
"""Diagnostic (GPU box): Game.step with caller-supplied actions on device buffers (pk_step_d, reference pokerl/game.py:621-700) as a
function of the batch size -- where the single-step path is bound by the serial chain of its slowest table (a launch lasts as long as
the wave that rolls most hands inside the step) and where by the bytes it moves.

    python tools/step_sweep.py [N] [T ...]

Per batch size: (a) the device-resident loop pick + step + reset(game_over)  -- three launches per step;
                (b) pk_step_d ALONE, replaying the actions (a) recorded       -- one launch per step (+ the masked reset);
both HIP-event timed on the handle's stream; algorithmic bytes = (2*(35N+21)+16) per env-step."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import pokerl_amd  # noqa: E402
from pokerl_amd import _lib as L  # noqa: E402
from pokerl_amd.hipmem import DeviceBuffer, DeviceEvent  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
SIZES = [int(x) for x in sys.argv[2:]] or [4096, 16384, 65536, 131072, 262144, 524288, 1048576]
STEPS = int(os.environ.get("PK_SWEEP_STEPS", "300"))
WARM = 100
B_STEP = 2 * (35 * N + 21) + 16


def run(T):
    g = pokerl_amd.VecGame(T, num_players=N)
    flags, terr = DeviceBuffer(T), DeviceBuffer(T)
    rec = DeviceBuffer(T * 4 * (WARM + STEPS))            # every step's actions, for the replay
    ev0, ev1 = DeviceEvent(), DeviceEvent()

    def act(s):
        return C.c_void_p(rec.ptr.value + s * T * 4)

    def loop(lo, hi, pick):
        for s in range(lo, hi):
            if pick:
                g.pick_actions_d(act(s), 0)
            g.step_d(act(s), flags, terr)
            g.reset_d(flags, L.FLAG_GAME_OVER)

    out = {}
    for name, pick in (("pick+step+reset", True), ("step+reset (replayed actions)", False)):
        g.set_serials(0, 0)                                # the replay deals the same hands (serials first: pk_reset deals)
        g.reset()
        loop(0, WARM, pick)
        g.sync()
        g.record_event(ev0.handle)
        loop(WARM, WARM + STEPS, pick)
        g.record_event(ev1.handle)
        g.sync()
        ms = DeviceEvent.elapsed_ms(ev0, ev1) / STEPS
        bad = int((terr.download(np.uint8, T) != 0).sum())
        out[name] = (ms, bad)
    steps_done = int(g.step_serial.sum())
    g.close()
    for b in (flags, terr, rec):
        b.free()
    return out, steps_done


print("N=%d  %d timed steps per size; B_step = %d B" % (N, STEPS, B_STEP))
for T in SIZES:
    res, done = run(T)
    for name, (ms, bad) in res.items():
        rate = T / (ms * 1e-3)
        print("T=%8d  %-32s %9.2f us/step  %7.3f G env-steps/s  alg HBM %6.1f GB/s = %.3f of 8 TB/s  (tables with error bits at the end: %d)"
              % (T, name, ms * 1e3, rate / 1e9, rate * B_STEP / 1e9, rate * B_STEP / 8e12, bad), flush=True)
