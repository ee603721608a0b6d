#!/usr/bin/env python3
"""Exhaustive pin of the 7-card evaluator: runs the REAL reference pokerl.judger.eval_hand
(imported read-only from /root/reference; build container only) on all C(52,7) = 133 784 560
distinct hands and writes a position-sensitive digest to eval7_digest.json.

Hand i (lexicographic rank of the ascending 7-subset of canonical deck indices 0..51, i.e.
itertools.combinations order; card value of index c = ((c%4)<<4)|(c//4), cards.py:77) yields
v_i = rank<<20 | get_kickers_value(kickers).  digest = sum_i mix64(v_i ^ (i * 0x9E3779B97F4A7C15)) mod 2^64
(mix64 = splitmix64 finaliser); also per-category counts and per-first-card partial digests.
"""
import itertools
import json
import math
import multiprocessing as mp
import os
import sys
import time

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402

GOLD = np.uint64(0x9E3779B97F4A7C15)


def mix64(z):
    z = z.copy()
    z ^= z >> np.uint64(30)
    z *= np.uint64(0xBF58476D1CE4E5B9)
    z ^= z >> np.uint64(27)
    z *= np.uint64(0x94D049BB133111EB)
    z ^= z >> np.uint64(31)
    return z


def prefix_offset(a, b):
    off = sum(math.comb(51 - x, 6) for x in range(a))
    off += sum(math.comb(51 - y, 5) for y in range(a + 1, b))
    return off


def work(ab):
    from pokerl.cards import Card
    from pokerl.judger import eval_hand, get_kickers_value
    a, b = ab
    cards = [Card(((c % 4) << 4) | (c // 4)) for c in range(52)]
    head = [cards[a], cards[b]]
    vals = []
    for rest in itertools.combinations(cards[b + 1:], 5):
        rank, kick = eval_hand(head + list(rest))
        vals.append((rank << 20) | get_kickers_value(kick))
    if not vals:
        return a, 0, [0] * 11, 0
    v = np.array(vals, np.uint64)
    idx = np.arange(len(v), dtype=np.uint64) + np.uint64(prefix_offset(a, b))
    with np.errstate(over="ignore"):
        h = int(np.sum(mix64(v ^ (idx * GOLD)), dtype=np.uint64))
    counts = np.bincount((v >> np.uint64(20)).astype(np.int64), minlength=11).tolist()
    return a, h, counts, len(v)


def main():
    t0 = time.time()
    tasks = [(a, b) for a in range(52) for b in range(a + 1, 52)]
    per_first = [0] * 52
    counts = [0] * 11
    total = 0
    procs = int(os.environ.get("PROCS", "7"))
    with mp.Pool(procs) as pool:
        for a, h, c, n in pool.imap_unordered(work, tasks, chunksize=4):
            per_first[a] = (per_first[a] + h) % (1 << 64)
            counts = [x + y for x, y in zip(counts, c)]
            total += n
    assert total == math.comb(52, 7)
    digest = sum(per_first) % (1 << 64)
    out = dict(hands=total, digest="%016x" % digest, per_first_card=["%016x" % x for x in per_first],
               category_counts=counts, seconds=round(time.time() - t0, 1),
               note="reference pokerl.judger.eval_hand over all 7-subsets; see make_eval_digest.py")
    with open(os.path.join(HERE, "eval7_digest.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(out["digest"], counts, out["seconds"])


if __name__ == "__main__":
    main()
