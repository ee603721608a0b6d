"""pokerl_amd -- MI355X-native vectorised No-Limit Hold'em hot path (drop-in for sneppy/pokerl's
Game.step/reset, judger and PokerGameEnv.step over thousands of tables).  HIP kernels behind a ctypes C ABI."""
from .enums import CardRank, CardSuit, HandRanking, PlayerState, PokerMoves, Policy
from .game import VecGame
from .envs import VecPokerGameEnv, VecPokerGameEnvPool
from .agents import AllInAgent, CallAgent, PokerAgent, RandomAgent
from .judger import compare_hands, compare_rankings, eval_hand, eval_hands
from .sharding import gather_f64, shard_tables
from .single import Game, PokerGameEnv
from .state_view import Card, StateView, packed_dtype, unpack_obs
from .hipmem import pinned_empty
from ._lib import PokerlHipError, device_count

__all__ = ['Game', 'PokerGameEnv', 'VecGame', 'VecPokerGameEnv', 'VecPokerGameEnvPool', 'eval_hand', 'eval_hands', 'compare_rankings', 'compare_hands',
           'shard_tables', 'gather_f64', 'Card', 'StateView', 'HandRanking', 'PokerMoves', 'PlayerState', 'CardRank', 'CardSuit', 'Policy',
           'PokerlHipError', 'device_count', 'packed_dtype', 'unpack_obs', 'pinned_empty', 'PokerAgent', 'RandomAgent', 'AllInAgent', 'CallAgent']
