#!/usr/bin/env python3
"""Diagnostic (GPU box): throughput of pk_eval_hands_d -- judger.eval_hand on device-resident hands of 0..7 cards (the
partial-hand feature of examples/q_learning.py:29-33).  usage: tools/eval_hands_bench.py [LOG2=24]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pokerl_amd import _lib as L
from pokerl_amd.hipmem import DeviceBuffer

log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 24
m = 1 << log2
rng = np.random.default_rng(3)
lib = L.lib()
from pokerl_amd import judger
hands = DeviceBuffer(m * 8)
judger.make_hands(hands.ptr, m)                                       # 7 DISTINCT cards per hand (what a game produces)
distinct = hands.download(np.uint64, m).view(np.uint8).reshape(m, 8)[:, :7].copy()
hands.free()
repeated = distinct.copy()
sel = rng.random(m) < 0.35
repeated[sel, 1] = repeated[sel, 0]                                   # 35 % of the hands repeat a card (the reference's tests do)
CASES = (("7 distinct cards", distinct, None),
         ("2/5/6/7 distinct cards, mixed in the batch", distinct, rng.choice(np.array([2, 5, 6, 7], np.uint8), m)),
         ("7 cards, 35 % of the hands with a repeated card", repeated, None))
if os.environ.get("PK_EHB_CASE"):          # one case only (tools/profile_eval_hands.sh profiles them one by one)
    CASES = (CASES[int(os.environ["PK_EHB_CASE"])],)
for name, cards, ncards in CASES:
    d_c, d_n, d_r, d_k, d_nk = DeviceBuffer(m * 7), DeviceBuffer(m), DeviceBuffer(m), DeviceBuffer(m * 4), DeviceBuffer(m)
    d_c.upload(cards)
    if ncards is not None:
        d_n.upload(ncards)
    hip = C.CDLL("libamdhip64.so")
    def run():
        L.check(lib.pk_eval_hands_d(0, d_c.ptr, d_n.ptr if ncards is not None else None, C.c_size_t(m), d_r.ptr, d_k.ptr, d_nk.ptr, None))
    run(); hip.hipDeviceSynchronize()
    t0 = time.perf_counter(); reps = 10
    for _ in range(reps):
        run()
    hip.hipDeviceSynchronize()
    dt = (time.perf_counter() - t0) / reps
    print("pk_eval_hands_d, %s: %.1f G evals/s (%.3f ms per 2^%d hands; 13 B/eval algorithmic = %.0f GB/s)" % (name, m / dt / 1e9, dt * 1e3, log2, 13 * m / dt / 1e9))
    for b in (d_c, d_n, d_r, d_k, d_nk):
        b.free()
