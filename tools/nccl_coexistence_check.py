#!/usr/bin/env python3
"""GPU-box check: torch.distributed's RCCL backend ("nccl") and libpokerl_hip.so in ONE process (what bench.py does at
N > 1), here with world_size 1: init, collectives on the rank's device, kernels on the handle's stream, same device."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import pokerl_amd

g = pokerl_amd.VecGame(65536, num_players=6, device=0)
g.reset()
t = torch.ones(4, device="cuda")
dist.all_reduce(t)
c = g.rollout(256, 0)
x = torch.tensor([float(c["steps"])], dtype=torch.float64, device="cuda")
dist.all_reduce(x, op=dist.ReduceOp.SUM)
dist.barrier()
torch.cuda.synchronize()
assert x.item() == 65536 * 256 and t.sum().item() == 4
# a torch tensor's memory handed to the device-pointer entry points
obs = torch.empty((65536, 17 + 18), dtype=torch.float64, device="cuda")
from pokerl_amd import _lib as L
L.check(L.lib().pk_get_obs_d(g._h, -1, obs.data_ptr()), g._h)
g.sync()
assert torch.equal(obs.cpu(), torch.from_numpy(g.observations))
dist.destroy_process_group()
print("nccl + libpokerl_hip coexistence ok: %d env-steps, obs via torch tensor ok" % c["steps"])
