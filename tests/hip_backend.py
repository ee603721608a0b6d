"""Adapter: the pokerl_amd product (HIP kernels behind the C ABI) seen through the backend interface of golden_util."""
import numpy as np

import pokerl_amd
from pokerl_amd import _lib as L


class HipBackend:
    def __init__(self, tables, n, start_credits=100, big_blind=2, small_blind=1, seed=0x706F6B65726C, table_id_base=0):
        self.env = pokerl_amd.VecPokerGameEnv(0, num_tables=tables, num_players=n, start_credits=start_credits,
                                              big_blind=big_blind, small_blind=small_blind, seed=seed,
                                              table_id_base=table_id_base)
        self.g = self.env.game
        self.T, self.N = tables, n

    @classmethod
    def from_meta(cls, meta):
        cfg = meta["cfg"]
        return cls(meta["tables"], meta["n"], cfg["start_credits"], cfg["big_blind"], cfg["small_blind"],
                   seed=meta["seed"], table_id_base=meta["table_id_base"])

    def set_serials(self, hand_serial, step_serial):
        self.g.set_serials(hand_serial, step_serial)

    def reset(self, mask=None, dealer=0):
        self.g.reset(mask=mask, dealer=dealer)

    def step(self, actions):
        out = self.g.step(np.asarray(actions, np.int32), strict=False)
        over, hand, turn, terr = out
        flags = over.astype(np.uint8) | (hand.astype(np.uint8) << 1) | (turn.astype(np.uint8) << 2)
        return flags, terr

    def pick_actions(self, policy):
        return self.g.pick_actions(policy)

    def env_reset(self, mask=None, opp_policy=0):
        self.env.opp_policy = opp_policy
        self.env.reset(mask)

    def env_step(self, actions, opp_policy=0):
        self.env.opp_policy = opp_policy
        g = self.g
        a = np.ascontiguousarray(actions, np.int32)
        reward = np.zeros(self.T, np.float64)
        done = np.zeros(self.T, np.uint8)
        hand = np.zeros(self.T, np.uint8)
        terr = np.zeros(self.T, np.uint8)
        if isinstance(opp_policy, (list, tuple)) or self.env.opp_policy is None:   # one agent per seat: the multi-agent entry point
            out = self.env.step(a, strict=False)
            return out[1], out[2].astype(np.uint8), out[3].astype(np.uint8), out[4]
        rc = g._lib.pk_env_step(g._h, L.ptr(a), opp_policy, L.ptr(reward), L.ptr(done), L.ptr(hand), L.ptr(terr))
        L.check(rc, g._h, allow_table_errors=True)
        return reward, done, hand, terr

    def rollout(self, K, policy, auto_reset=True, fused=True):
        c = self.g.rollout(K, policy, auto_reset, fused)
        return np.array([c["steps"], c["hands"], c["evals"], c["games"]], np.uint64)

    def snapshot(self):
        g = self.g
        onehot, _ = g.get_valid_actions()
        valid = (onehot.astype(np.uint8) << np.arange(7, dtype=np.uint8)).sum(axis=1).astype(np.uint8)
        srank, skick = g.hand_rankings
        return dict(active=g.active_player.astype(np.uint8), turn=g.turn.astype(np.uint8),
                    dealer=g.dealer_idx.astype(np.uint8), sb=g.small_blind_idx.astype(np.uint8),
                    bb=g.big_blind_idx.astype(np.uint8), hand=g.hand, states=g.player_states,
                    credits=g.credits, bets=g.bets, pending=g.pending_bets, payoffs=g.payoffs,
                    min_raise=g.minimum_raise_value, cards=g.deck, srank=srank, skick=skick, valid=valid,
                    hand_serial=g.hand_serial, step_serial=g.step_serial)
