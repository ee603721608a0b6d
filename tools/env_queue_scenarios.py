#!/usr/bin/env python3
"""Diagnostic (GPU box): does the sub-batched PokerGameEnv.step (pk_set_env_batches, bench.py's extra leg) run at the same rate
whatever the PROCESS did before?  Round 4 found it did not -- 3.6 G env.step/s in a fresh process, 2.6 G after one hipMemcpy, 1.5 G
after another PokerGameEnv handle had been used: barrier packets of the cross-stream waits stalling hardware queues that share a
pipe (docs/history.md section 5) -- and this is the script that pinned it down.  Each scenario runs in a process of its own.
usage: python tools/env_queue_scenarios.py [scenario ...]     (no argument: all of them, one child process each)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SCENARIOS = ["fresh", "twice", "after_memcpy", "after_rollout", "after_env_sync", "after_env_async", "after_evaluator"]


def main(which):
    import numpy as np
    import bench
    import pokerl_amd
    from pokerl_amd.hipmem import DeviceBuffer
    ctx = bench.DistContext()
    kw = dict(tables=524288, players=6, async_passes=8, inner=3, steps=1500, warmup=300)

    def run(tag):
        r = bench.env_workload(ctx, 0, **kw)
        print("%-18s %.3f G env.step/s  (%.1f us per call)" % (tag, r["env_steps"] / r["seconds"] / 1e9, r["seconds"] / r["steps"] * 1e6), flush=True)

    if which == "twice":
        run("fresh")
    elif which == "after_memcpy":
        b = DeviceBuffer(1 << 20); b.upload(np.zeros(1 << 18, np.uint8)); b.download(np.uint8, 16); b.free()
    elif which == "after_rollout":
        bench.rollout_workload(ctx, 0, 65536, 6, "random", 20, 5, min_steps=131072, samples=2)
    elif which == "after_env_sync":
        bench.env_workload(ctx, 0, tables=65536, players=6, steps=200, warmup=20)
    elif which == "after_env_async":
        bench.env_workload(ctx, 0, tables=65536, players=6, async_passes=8, steps=500, warmup=50)
    elif which == "after_evaluator":
        bench.evaluator_leg(0, 24, 3)
    run(which)


if __name__ == "__main__":
    if len(sys.argv) == 2 and sys.argv[1] in SCENARIOS and os.environ.get("PK_SCENARIO_CHILD"):
        main(sys.argv[1])
    else:
        for w in (sys.argv[1:] or SCENARIOS):
            subprocess.run([sys.executable, os.path.abspath(__file__), w], env=dict(os.environ, PK_SCENARIO_CHILD="1"))
