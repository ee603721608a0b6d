import sys, os, json, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bench
import pokerl_amd
from pokerl_amd.hipmem import DeviceBuffer
ctx = bench.DistContext()
kw = dict(tables=524288, players=6, async_passes=8, inner=3, steps=1500, warmup=300)
def run(tag):
    r = bench.env_workload(ctx, 0, **kw)
    print(tag, "%.3f G  (%.1f us per call)" % (r["env_steps"] / r["seconds"] / 1e9, r["seconds"] / r["steps"] * 1e6), flush=True)
which = sys.argv[1]
if which == "e":
    b = DeviceBuffer(1 << 20); b.upload(np.zeros(1 << 18, np.uint8)); x = b.download(np.uint8, 16); b.free()
    run("after DeviceBuffer alloc/upload/download/free")
elif which == "f":
    p = pokerl_amd.VecPokerGameEnvPool(0, num_tables=65536, num_batches=1, num_players=6); p.reset(); p.close()
    run("after pool create/reset/close")
elif which == "g":
    p = pokerl_amd.VecGame(65536, num_players=6); p.reset(); o = p.observations; p.close()
    run("after VecGame create/reset/observations/close")
elif which == "h":
    p = pokerl_amd.VecGame(65536, num_players=6); p.reset(); p.rollout(100, 0); p.close()
    run("after VecGame create/reset/rollout(counters)/close")
elif which == "i":
    b = DeviceBuffer(1 << 20); b.free()
    run("after DeviceBuffer alloc/free only")
if which == "c":
    bench.env_workload(ctx, 0, tables=65536, players=6, steps=200, warmup=20)
    run("after a sync env leg")
elif which == "d":
    bench.evaluator_leg(0, 24, 3)
    run("after evaluator leg")
elif which == "a":
    run("fresh process"); run("second time same process")
elif which == "j":
    bench.env_workload(ctx, 0, tables=65536, players=6, async_passes=8, steps=500, warmup=50)
    run("after an async B=1 env leg")
elif which == "k":
    bench.rollout_workload(ctx, 0, 65536, 6, "random", 20, 5, min_steps=131072, samples=3, coalesce=0)
    run("after a coalesce-0 rollout leg")
elif which == "l":
    bench.rollout_workload(ctx, 0, 65536, 6, "random", 20, 5, min_steps=131072, samples=2)
    run("after a coalesced rollout leg")
if which == "o":
    import gc
    bench.env_workload(ctx, 0, tables=65536, players=6, steps=200, warmup=20)
    gc.collect()
    run("after a sync env leg + gc.collect()")
elif which == "p":
    import gc
    gc.disable()
    bench.env_workload(ctx, 0, tables=65536, players=6, steps=200, warmup=20)
    run("after a sync env leg, gc disabled")
elif which == "q":
    bench.env_workload(ctx, 0, tables=65536, players=6, steps=2, warmup=0)
    run("after a 2-step sync env leg")
elif which == "r":
    p = pokerl_amd.VecPokerGameEnvPool(0, num_tables=65536, num_batches=1, num_players=6)
    from pokerl_amd import _lib as L
    g = p.envs[0].game
    L.check(L.lib().pk_env_reset_d(g._h, None, 0), g._h); g.sync(); x = g.step_serial; p.close()
    run("after pool + pk_env_reset_d + step_serial + close")
if which in ("s", "t", "u", "v"):
    from pokerl_amd import _lib as L
    from pokerl_amd.hipmem import DeviceEvent
    T = 65536
    p = pokerl_amd.VecPokerGameEnvPool(0, num_tables=T, num_batches=1, num_players=6)
    g = p.envs[0].game
    lib = L.lib()
    L.check(lib.pk_env_reset_d(g._h, None, 0), g._h); g.sync()
    if which == "s":
        e0, e1 = DeviceEvent(), DeviceEvent()
        g.record_event(e0.handle); g.record_event(e1.handle); g.sync(); print(DeviceEvent.elapsed_ms(e0, e1))
    bufs = []
    if which in ("t", "u", "v"):
        bufs = [DeviceBuffer(T * 4), DeviceBuffer(T * 8), DeviceBuffer(T), DeviceBuffer(T), DeviceBuffer(T), DeviceBuffer(T * 35 * 8)]
    if which == "t":
        L.check(lib.pk_env_step_fused_d(g._h, None, 0, 0, 1, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, bufs[4].ptr, bufs[5].ptr), g._h); g.sync()
    if which == "v":
        L.check(lib.pk_env_step_fused_d(g._h, None, 0, 0, 1, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, bufs[4].ptr, None), g._h); g.sync()
    for b in bufs: b.free()
    p.close()
    run("variant " + which)
