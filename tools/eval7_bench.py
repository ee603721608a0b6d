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
ms = judger.time_eval7_stream(hands.ptr, m, out.ptr, distinct, reps)
print(json.dumps(dict(hands=m, reps=reps, distinct=distinct, kernel_ms=ms, hand_evals_per_s=m / (ms * 1e-3),
                      algorithmic_GBps=12.0 * m / (ms * 1e-3) / 1e9)))
