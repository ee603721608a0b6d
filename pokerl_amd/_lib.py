"""ctypes binding of libpokerl_hip.so (C ABI: include/pokerl_hip.h).  There is NO CPU fallback: if the HIP library is
missing or no MI355X is visible, calls raise."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("POKERL_HIP_LIB") or os.path.join(HERE, "libpokerl_hip.so")  # override: diagnostic builds

PK_OK, PK_E_INVALID_ARG, PK_E_NO_DEVICE, PK_E_HIP, PK_E_OOM, PK_E_TABLE, PK_E_BUSY = 0, -1, -2, -3, -4, -5, -6
TERR_INVALID_ACTION, TERR_NO_WINNER, TERR_HAND_CAP, TERR_ENV_CAP = 1, 2, 4, 8
FLAG_GAME_OVER, FLAG_HAND_OVER, FLAG_TURN_OVER = 1, 2, 4
F_CREDITS, F_BETS, F_PENDING_BETS, F_PAYOFFS = 0, 1, 2, 3
I_ACTIVE_PLAYER, I_TURN, I_DEALER_IDX, I_SMALL_BLIND_IDX, I_BIG_BLIND_IDX, I_HAND = range(6)
TF_POT, TF_HIGH_BET, TF_MIN_RAISE = 0, 1, 2
ACTION_SKIP = -2   # pk_env_step_multi_d: leave an idle table alone (PK_ACTION_SKIP)
POLICY_EXTERNAL = 15
ABI_VERSION = 6
NUM_COUNTERS = 4
MIN_PLAYERS, MAX_PLAYERS = 2, 16

# every symbol include/pokerl_hip.h declares (tests check the library exports each one)
SYMBOLS = ["pk_abi_version", "pk_device_count", "pk_last_error", "pk_create", "pk_destroy", "pk_num_tables",
           "pk_num_players", "pk_reset", "pk_step", "pk_step_d", "pk_get_valid_actions", "pk_get_f64",
           "pk_get_min_raise", "pk_get_player_states", "pk_get_i32", "pk_get_cards", "pk_get_hand_ranks",
           "pk_eval_hands", "pk_compare_rankings", "pk_eval7_prefix", "pk_pick_actions", "pk_rollout",
           "pk_env_reset", "pk_env_step", "pk_get_obs", "pk_sync", "pk_time_rollout", "pk_get_obs_d",
           "pk_get_valid_actions_d", "pk_env_step_d", "pk_env_reset_d", "pk_eval7_d", "pk_make_hands_d", "pk_time_eval7_d",
           "pk_get_serials", "pk_set_serials", "pk_get_table_f64", "pk_get_game_over", "pk_eval_hands_d",
           "pk_pick_actions_d", "pk_flush", "pk_get_owed", "pk_env_step_fused_d", "pk_env_step_async_d", "pk_set_tuning", "pk_get_stream", "pk_set_stream", "pk_wait_event",
           "pk_record_event", "pk_use_own_stream", "pk_set_coalesce", "pk_get_launch_stats", "pk_env_step_multi_d", "pk_env_end_multi_d", "pk_get_f64_d", "pk_set_env_batches", "pk_env_last_range",
           "pk_get_obs_packed", "pk_get_obs_packed_d", "pk_set_env_obs_packed", "pk_host_alloc", "pk_host_free", "pk_check_actions",
           "pk_env_step_begin", "pk_env_step_end", "pk_reset_d", "pk_step_auto_d", "pk_stream_pool_drain", "pk_step_async_d", "pk_set_step_obs", "pk_build_info"]


class PokerlHipError(RuntimeError):
    pass


_lib = None
_vp = C.c_void_p


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PokerlHipError(
            "%s not found: build it with `python -m pokerl_amd.build` (hipcc, gfx950). "
            "pokerl_amd has no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.pk_abi_version.restype = C.c_int
    L.pk_device_count.restype = C.c_int
    L.pk_last_error.restype = C.c_char_p
    L.pk_last_error.argtypes = [_vp]
    L.pk_build_info.restype = C.c_char_p
    L.pk_create.argtypes = [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, _vp, C.c_double, C.c_double, C.c_double,
                            C.c_int, C.c_uint64, C.c_uint32]
    L.pk_destroy.argtypes = [_vp]
    L.pk_num_tables.argtypes = [_vp]
    L.pk_num_players.argtypes = [_vp]
    L.pk_reset.argtypes = [_vp, _vp, C.c_int]
    L.pk_reset_d.argtypes = [_vp, _vp, C.c_int, C.c_int]
    L.pk_step.argtypes = [_vp, _vp, _vp, _vp]
    L.pk_step_d.argtypes = [_vp, _vp, _vp, _vp]
    L.pk_step_auto_d.argtypes = [_vp, _vp, _vp, _vp]
    L.pk_step_async_d.argtypes = [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int]
    L.pk_get_valid_actions.argtypes = [_vp, C.c_int, _vp]
    L.pk_get_f64.argtypes = [_vp, C.c_int, _vp]
    L.pk_get_f64_d.argtypes = [_vp, C.c_int, _vp]
    L.pk_get_min_raise.argtypes = [_vp, _vp]
    L.pk_get_player_states.argtypes = [_vp, _vp]
    L.pk_get_i32.argtypes = [_vp, C.c_int, _vp]
    L.pk_get_cards.argtypes = [_vp, _vp]
    L.pk_get_hand_ranks.argtypes = [_vp, _vp, _vp]
    L.pk_eval_hands.argtypes = [C.c_int, _vp, _vp, C.c_size_t, _vp, _vp, _vp]
    L.pk_compare_rankings.argtypes = [C.c_int, _vp, _vp, C.c_int, C.c_size_t, _vp]
    L.pk_eval7_prefix.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.POINTER(C.c_size_t)]
    L.pk_pick_actions.argtypes = [_vp, C.c_int, _vp]
    L.pk_rollout.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]
    L.pk_env_reset.argtypes = [_vp, _vp, C.c_int]
    L.pk_env_step.argtypes = [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp]
    L.pk_get_obs.argtypes = [_vp, C.c_int, _vp]
    L.pk_get_obs_d.argtypes = [_vp, C.c_int, _vp]
    L.pk_get_valid_actions_d.argtypes = [_vp, C.c_int, _vp]
    L.pk_env_step_d.argtypes = [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp]
    L.pk_env_reset_d.argtypes = [_vp, _vp, C.c_int]
    L.pk_eval7_d.argtypes = [C.c_int, _vp, C.c_size_t, _vp, C.c_int]
    L.pk_make_hands_d.argtypes = [C.c_int, C.c_uint64, C.c_size_t, _vp]
    L.pk_time_eval7_d.argtypes = [C.c_int, _vp, C.c_size_t, _vp, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.pk_sync.argtypes = [_vp]
    L.pk_get_serials.argtypes = [_vp, _vp, _vp]
    L.pk_set_serials.argtypes = [_vp, _vp, _vp]
    L.pk_get_table_f64.argtypes = [_vp, C.c_int, _vp]
    L.pk_get_game_over.argtypes = [_vp, _vp]
    L.pk_eval_hands_d.argtypes = [C.c_int, _vp, _vp, C.c_size_t, _vp, _vp, _vp, _vp]
    L.pk_pick_actions_d.argtypes = [_vp, C.c_int, _vp]
    L.pk_flush.argtypes = [_vp]
    L.pk_env_step_fused_d.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]
    L.pk_env_step_async_d.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]
    L.pk_env_step_multi_d.argtypes = [_vp, _vp, _vp, C.c_uint64, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
    L.pk_env_end_multi_d.argtypes = [_vp]
    L.pk_set_env_batches.argtypes = [_vp, C.c_int]
    L.pk_env_last_range.argtypes = [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.pk_get_obs_packed.argtypes = [_vp, C.c_int, _vp]
    L.pk_get_obs_packed_d.argtypes = [_vp, C.c_int, _vp]
    L.pk_set_env_obs_packed.argtypes = [_vp, _vp]
    L.pk_set_step_obs.argtypes = [_vp, _vp, _vp]
    L.pk_host_alloc.argtypes = [C.POINTER(_vp), C.c_size_t]
    L.pk_host_free.argtypes = [_vp]
    L.pk_check_actions.argtypes = [_vp, _vp, C.POINTER(C.c_int32)]
    L.pk_env_step_begin.argtypes = [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]
    L.pk_env_step_end.argtypes = [_vp]
    L.pk_stream_pool_drain.argtypes = [C.c_int]
    L.pk_get_owed.argtypes = [_vp, _vp]
    L.pk_set_tuning.argtypes = [_vp, C.c_int, C.c_int]
    L.pk_get_stream.argtypes = [_vp, C.POINTER(_vp)]
    L.pk_set_stream.argtypes = [_vp, _vp]
    L.pk_use_own_stream.argtypes = [_vp]
    L.pk_set_coalesce.argtypes = [_vp, C.c_int]
    L.pk_get_launch_stats.argtypes = [_vp, _vp, C.c_int]
    L.pk_wait_event.argtypes = [_vp, _vp]
    L.pk_record_event.argtypes = [_vp, _vp]
    L.pk_time_rollout.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), _vp]
    for name in SYMBOLS:
        if name not in ("pk_last_error", "pk_build_info"):
            getattr(L, name).restype = C.c_int
    if L.pk_abi_version() != ABI_VERSION:
        raise PokerlHipError("libpokerl_hip.so ABI version mismatch")
    _lib = L
    return L


def ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


def check(rc, handle=None, allow_table_errors=False):
    if rc == PK_OK or (allow_table_errors and rc == PK_E_TABLE):
        return rc
    msg = lib().pk_last_error(handle)
    raise PokerlHipError("libpokerl_hip: error %d: %s" % (rc, msg.decode() if msg else "?"))


def source_hash():
    """The hash of the kernel sources the loaded library was built from (pk_build_info; pokerl_amd/build.py source_hash)."""
    info = lib().pk_build_info().decode()
    return info.split("src=")[-1]


def device_count():
    return lib().pk_device_count()
