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
