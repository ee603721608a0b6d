#!/usr/bin/env python3
"""Why do HIP events and rocprofv3 disagree on k_eval7_tab_stream (0.907 ms per pass by events = 0.44 of HBM, 1.012 ms
average kernel duration by rocprofv3 = 0.40)?  One process times the SAME launches three ways: (a) `reps` passes back to back inside one event
pair (what bench.py's evaluator leg and pk_time_eval7_d report: device time / passes), (b) every pass inside an event pair of its own with a
device synchronise in between (an ISOLATED launch: ramp + body + tail), (c) -- when run under `rocprofv3 --kernel-trace` -- the profiler's
start / end timestamps of those very dispatches (tools/eval7_reconcile.sh reads them back and reports, for the back-to-back group, the
overlap between consecutive dispatches).
usage: python3 tools/eval7_reconcile.py [log2_hands] [reps]        -> one JSON line"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pokerl_amd import judger  # noqa: E402
from pokerl_amd.hipmem import DeviceBuffer  # noqa: E402

log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 28
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
m = 1 << log2
hands, out = DeviceBuffer(m * 8), DeviceBuffer(m * 4)
judger.make_hands(hands.ptr, m)
import ctypes as C  # noqa: E402
from pokerl_amd import _lib as L  # noqa: E402
from pokerl_amd.hipmem import DeviceEvent  # noqa: E402
hip = C.CDLL("libamdhip64.so")
lib = L.lib()


def isolated():
    """ONE pass on an idle device inside its own event pair (legacy default stream, where pk_eval7_d launches; it synchronises the device)."""
    e0, e1 = DeviceEvent(), DeviceEvent()
    assert hip.hipDeviceSynchronize() == 0 and hip.hipEventRecord(e0.handle, None) == 0
    L.check(lib.pk_eval7_d(0, hands.ptr, C.c_size_t(m), out.ptr, 1))
    assert hip.hipEventRecord(e1.handle, None) == 0
    return DeviceEvent.elapsed_ms(e0, e1)


judger.time_eval7_stream(hands.ptr, m, out.ptr, True, 3)                                   # warm-up (table build, clocks): 1 + 3 dispatches
b2b = [judger.time_eval7_stream(hands.ptr, m, out.ptr, True, reps) for _ in range(3)]     # (a) ms per pass, back to back (each call: 1 untimed + reps timed dispatches)
iso = [isolated() for _ in range(reps)]                                                    # (b) one pass per event pair, device idle before it
b2b2 = [judger.time_eval7_stream(hands.ptr, m, out.ptr, True, reps) for _ in range(2)]    # (a) again: the order of the two does not matter
iso_sorted = sorted(iso)
res = {"hands": m, "reps": reps, "back_to_back_ms_per_pass": b2b + b2b2,
       "isolated_ms": {"min": iso_sorted[0], "median": iso_sorted[len(iso) // 2], "max": iso_sorted[-1], "mean": sum(iso) / len(iso)},
       "launch_groups": [4] + [1 + reps] * 3 + [1] * reps + [1 + reps] * 2,
       "gbps": {"back_to_back": 12.0 * m / (min(b2b + b2b2) * 1e-3) / 1e9, "isolated_median": 12.0 * m / (iso_sorted[len(iso) // 2] * 1e-3) / 1e9}}
print(json.dumps(res))
hands.free(); out.free()
