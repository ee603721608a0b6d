#!/usr/bin/env python3
"""The reference's examples/q_learning.py (tabular Q-learning on the hand-ranking feature against random agents) for
65 536 tables at once with NOTHING leaving the GPU: observations, the partial-hand rank feature
(`eval_hand(state.player_hand)` on 2 / 5 / 6 / 7-card hands, q_learning.py:29-33 -> pk_eval_hands_d), the Q table, the
softmax policy (torch) and the environment (pk_env_step_fused_d: PokerGameEnv.step + reset of finished episodes +
next observation in one launch) all live in HBM and run on ONE stream (pk_set_stream).

    python examples/q_learning_device.py [num_tables] [steps]

torch is only the learner here (the user's side of the boundary); the environment is plain HIP behind the C ABI.
"""
import ctypes as C
import os
import sys

import torch  # first: torch ships its own HIP runtime and the process must settle on one

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pokerl_amd  # noqa: E402
from pokerl_amd import _lib as L  # noqa: E402
from pokerl_amd import judger  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 400
N = 4                                                   # q_learning.py:18-19: three random opponents
D = 17 + 3 * N
dev = torch.device("cuda", 0)
env = pokerl_amd.VecPokerGameEnv(pokerl_amd.Policy.RANDOM, num_tables=T, num_players=N)
g, lib = env.game, L.lib()
g.set_stream(torch.cuda.current_stream(dev).cuda_stream)   # the handle now runs in program order with torch's kernels
stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

obs = torch.zeros((T, D), dtype=torch.float64, device=dev)
reward = torch.zeros(T, dtype=torch.float64, device=dev)
done = torch.zeros(T, dtype=torch.uint8, device=dev)
hand = torch.zeros(T, dtype=torch.uint8, device=dev)
terr = torch.zeros(T, dtype=torch.uint8, device=dev)
cards = torch.zeros((T, 7), dtype=torch.uint8, device=dev)
ncards = torch.zeros(T, dtype=torch.uint8, device=dev)
rank = torch.zeros(T, dtype=torch.uint8, device=dev)
kick = torch.zeros(T, dtype=torch.int32, device=dev)
Q = torch.zeros((10, 7), dtype=torch.float64, device=dev)      # q_learning.py:24-26: HandRanking.NONE x NUM_MOVES
p = lambda t: C.c_void_p(t.data_ptr())


def features():
    """get_state (q_learning.py:29-33): HandRanking of player_hand = hole cards + visible community cards; valid mask."""
    hole, comm = obs[:, 10:12], obs[:, 12:17]
    vis = comm >= 0
    n = 2 + vis.sum(dim=1)
    # visible community cards are a prefix (game.py:278), so hole + community[:k] packs left-aligned
    cards[:, :2] = hole.to(torch.uint8)
    cards[:, 2:] = torch.where(vis, comm, torch.zeros_like(comm)).to(torch.uint8)
    ncards.copy_(n.to(torch.uint8))
    judger.eval_hands_d(p(cards), p(ncards), T, p(rank), p(kick), None, device=0, stream=stream)
    return rank.long() - 1, obs[:, 3:10] > 0            # state index 0..8 (rank 1..9), valid-action mask [T,7]


def act(s, valid, eps):
    """predict (q_learning.py:35-42) with the eps-greedy exploration of :79-80, per table."""
    q = Q[s].masked_fill(~valid, -1e30)
    soft = torch.distributions.Categorical(logits=q).sample()
    uni = torch.distributions.Categorical(probs=valid.double()).sample()
    return torch.where(torch.rand(T, device=dev) < eps, uni, soft).to(torch.int32)


L.check(lib.pk_env_reset_d(g._h, None, 0), g._h)
L.check(lib.pk_get_obs_d(g._h, -1, p(obs)), g._h)
s, valid = features()
alpha, gamma = 0.01, 1.0
total = torch.zeros((), dtype=torch.float64, device=dev)
for step in range(STEPS):
    u = act(s, valid, 1 - (step / STEPS) ** 2)
    L.check(lib.pk_env_step_fused_d(g._h, p(u), 0, 0, 1, p(reward), p(done), p(hand), p(terr), p(obs)), g._h)
    ns, nvalid = features()                             # for tables whose episode ended this is the NEW episode's state
    nxt = Q[ns].masked_fill(~nvalid, -1e30).max(dim=1).values
    target = reward + (1 - hand.double()) * gamma * nxt                       # q_learning.py:85
    idx = s * 7 + u.long()
    td = alpha * (target - Q.view(-1)[idx])
    Q.view(-1).index_add_(0, idx, td / torch.bincount(idx, minlength=70)[idx].double())   # batch-mean of q_learning.py:86
    total += reward.sum()
    s, valid = ns, nvalid
torch.cuda.synchronize()
print("%d tables x %d env steps on the device; mean reward per env step %.4f" % (T, STEPS, total.item() / (T * STEPS)))
names = pokerl_amd.HandRanking.as_string
for r in (9, 7, 3):
    print("Q[%-10s] = %s" % (names[r], [round(x, 3) for x in Q[r - 1].tolist()]))
g.use_own_stream()
g.close()
