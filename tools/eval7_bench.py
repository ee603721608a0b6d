#!/usr/bin/env python3
"""Diagnostic / profiling driver (GPU box) of the stand-alone streaming evaluator: pk_eval7_d over 2^LOG2 device-resident
7-card hands, REPS timed passes.  usage: tools/eval7_bench.py [LOG2=28] [REPS=5] [distinct=1]   -> one JSON line"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pokerl_amd import judger  # noqa: E402
from pokerl_amd.hipmem import DeviceBuffer  # noqa: E402

log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 28
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
distinct = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
m = 1 << log2
hands, out = DeviceBuffer(m * 8), DeviceBuffer(m * 4)
judger.make_hands(hands.ptr, m)
# The first ~10 dispatches of a process run while the clock still ramps (1.0 ms per pass falling to the steady 0.83: profiles/r06_eval7_reconcile.txt --
# round 5's "events say 0.907, rocprofv3 says 1.012 ms" was a FIVE-pass profiling run against a bench leg that comes after seconds of GPU work).  Warm up
# first, then time; a profiler that wants the steady state skips the first WARM + 1 dispatches of the trace (tools/summarize_eval7.py).
WARM = 24
judger.time_eval7_stream(hands.ptr, m, out.ptr, distinct, WARM - 1)     # (1 untimed + WARM - 1 timed dispatches)
ms = judger.time_eval7_stream(hands.ptr, m, out.ptr, distinct, reps)
print(json.dumps(dict(hands=m, reps=reps, warm_dispatches=WARM, distinct=distinct, kernel_ms=ms, hand_evals_per_s=m / (ms * 1e-3),
                      algorithmic_GBps=12.0 * m / (ms * 1e-3) / 1e9)))
