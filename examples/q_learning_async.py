#!/usr/bin/env python3
"""examples/q_learning_device.py with the ASYNCHRONOUS environment call (pk_env_step_async_d): the learner acts on the
tables whose PokerGameEnv.step has returned (`ready`), the others carry on in flight on the device.  One synchronous
env.step of 65 536 tables lasts as long as its slowest table (a seat 0 that goes broke during an opponent's step waits for
the end of the game, pokerl/envs/game_env.py:49-52); here a launch lasts `PASSES` betting passes, whatever the stragglers
do.  Per table the sequence of steps, rewards and observations is exactly the synchronous one.

    python examples/q_learning_async.py [num_tables] [launches] [passes]

torch is only the learner (the user's side of the boundary); the environment is plain HIP behind the C ABI.
"""
import ctypes as C
import os
import sys
import time

import torch  # first: torch ships its own HIP runtime and the process must settle on one

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pokerl_amd  # noqa: E402
from pokerl_amd import _lib as L  # noqa: E402
from pokerl_amd import judger  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
LAUNCHES = int(sys.argv[2]) if len(sys.argv) > 2 else 600
PASSES = int(sys.argv[3]) if len(sys.argv) > 3 else 8
N = 4                                                   # q_learning.py:18-19: three random opponents
D = 17 + 3 * N
dev = torch.device("cuda", 0)
env = pokerl_amd.VecPokerGameEnv(pokerl_amd.Policy.RANDOM, num_tables=T, num_players=N)
g, lib = env.game, L.lib()
g.set_stream(torch.cuda.current_stream(dev).cuda_stream)   # the handle runs in program order with torch's kernels
stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

obs = torch.zeros((T, D), dtype=torch.float64, device=dev)
reward = torch.zeros(T, dtype=torch.float64, device=dev)
done = torch.zeros(T, dtype=torch.uint8, device=dev)
hand = torch.zeros(T, dtype=torch.uint8, device=dev)
terr = torch.zeros(T, dtype=torch.uint8, device=dev)
ready = torch.ones(T, dtype=torch.uint8, device=dev)           # every table is ready for its first action
cards = torch.zeros((T, 7), dtype=torch.uint8, device=dev)
ncards = torch.zeros(T, dtype=torch.uint8, device=dev)
rank = torch.zeros(T, dtype=torch.uint8, device=dev)
kick = torch.zeros(T, dtype=torch.int32, device=dev)
Q = torch.zeros((10, 7), dtype=torch.float64, device=dev)      # q_learning.py:24-26: HandRanking.NONE x NUM_MOVES
p = lambda t: C.c_void_p(t.data_ptr())


def features():
    """get_state (q_learning.py:29-33) of every table's LAST DELIVERED observation row."""
    hole, comm = obs[:, 10:12], obs[:, 12:17]
    vis = comm >= 0
    cards[:, :2] = hole.to(torch.uint8)
    cards[:, 2:] = torch.where(vis, comm, torch.zeros_like(comm)).to(torch.uint8)
    ncards.copy_((2 + vis.sum(dim=1)).to(torch.uint8))
    judger.eval_hands_d(p(cards), p(ncards), T, p(rank), p(kick), None, device=0, stream=stream)
    return rank.long() - 1, obs[:, 3:10] > 0


def act(s, valid, eps):
    q = Q[s].masked_fill(~valid, -1e30)
    soft = torch.distributions.Categorical(logits=q).sample()
    uni = torch.distributions.Categorical(probs=valid.double()).sample()
    return torch.where(torch.rand(T, device=dev) < eps, uni, soft).to(torch.int32)


L.check(lib.pk_env_reset_d(g._h, None, 0), g._h)
L.check(lib.pk_get_obs_d(g._h, -1, p(obs)), g._h)
s, valid = features()
u = act(s, valid, 1.0)                                   # the action each table is executing (or will execute next)
alpha, gamma = 0.01, 1.0
delivered = torch.zeros((), dtype=torch.int64, device=dev)
total = torch.zeros((), dtype=torch.float64, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(LAUNCHES):
    # tables in flight ignore their entry of `u`; ready tables start the action chosen for them last time round
    env.step_async_d(p(u), p(reward), p(done), p(hand), p(terr), p(obs), p(ready), max_passes=PASSES)
    r = ready.bool()                                     # rows of reward / done / hand / obs are valid where r
    ns, nvalid = features()
    nxt = Q[ns].masked_fill(~nvalid, -1e30).max(dim=1).values
    target = reward + (1 - hand.double()) * gamma * nxt                         # q_learning.py:85
    idx = (s * 7 + u.long())[r]
    td = alpha * (target[r] - Q.view(-1)[idx])
    Q.view(-1).index_add_(0, idx, td / torch.bincount(idx, minlength=70)[idx].double())   # batch-mean of q_learning.py:86
    total += reward[r].sum()
    delivered += r.sum()
    nu = act(ns, nvalid, 1 - (it / LAUNCHES) ** 2)
    s, u = torch.where(r, ns, s), torch.where(r, nu, u)  # tables in flight keep the state / action they started from
torch.cuda.synchronize()
dt = time.perf_counter() - t0
n = int(delivered.item())
print("%d tables, %d bounded launches of %d passes: %d env.steps delivered (%.0f %% ready per launch), %.1f M env.step/s "
      "incl. the learner, mean reward %.4f" % (T, LAUNCHES, PASSES, n, 100.0 * n / (T * LAUNCHES), n / dt / 1e6, total.item() / max(1, n)))
# drain before touching the tables through any other entry point
env.step_async_d(p(u), p(reward), p(done), p(hand), p(terr), p(obs), p(ready), max_passes=0)
torch.cuda.synchronize()
names = pokerl_amd.HandRanking.as_string
for r_ in (9, 7, 3):
    print("Q[%-10s] = %s" % (names[r_], [round(x, 3) for x in Q[r_ - 1].tolist()]))
g.use_own_stream()
g.close()
