"""ctypes loader for libazamd.so (the HIP engine). There is no CPU fallback: importing the
compute classes without the built library, or without a GPU, raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AZAMD_LIB selects another build of the same library (e.g. a diagnostic build with in-kernel stamps)
LIB_PATH = os.environ.get("AZAMD_LIB") or os.path.join(_HERE, "libazamd.so")

AZ_OK, AZ_ERR_ARG, AZ_ERR_HIP, AZ_ERR_STATE, AZ_ERR_ILLEGAL, AZ_ERR_NOCHILD, AZ_ERR_CAPACITY = 0, -1, -2, -3, -4, -5, -6
AZ_FEAT_F32_NCHW, AZ_FEAT_BF16_NHWC = 0, 1
AZ_NET_BF16, AZ_NET_F32 = 0, 1


class AzConfig(C.Structure):
    _fields_ = [("board_size", C.c_int32), ("history_len", C.c_int32), ("games", C.c_int32),
                ("simulations", C.c_int32), ("inference_batch", C.c_int32), ("topk_expand", C.c_int32),
                ("dedup_leaves", C.c_int32), ("device", C.c_int32), ("komi", C.c_double),
                ("c_base", C.c_double), ("c_init", C.c_double), ("dirichlet_eps", C.c_double),
                ("virtual_loss", C.c_double)]


class AzNetConfig(C.Structure):
    _fields_ = [("board_size", C.c_int32), ("in_planes", C.c_int32), ("channels", C.c_int32),
                ("blocks", C.c_int32), ("use_se", C.c_int32), ("se_hidden", C.c_int32),
                ("value_hidden", C.c_int32), ("precision", C.c_int32), ("device", C.c_int32)]


class AzLaneConfig(C.Structure):
    _fields_ = [("temp_moves", C.c_int32), ("temp_mid_moves", C.c_int32), ("temp_start", C.c_double), ("temp_mid", C.c_double),
                ("temp_end", C.c_double), ("resign_threshold", C.c_double), ("disable_resign", C.c_int32),
                ("resign_min_move", C.c_int32), ("dirichlet_alpha", C.c_double), ("collect_samples", C.c_int32),
                ("rng_threads", C.c_int32), ("use_graph", C.c_int32)]


class AzLaneResult(C.Structure):
    _fields_ = [("counters", C.c_int64 * 8), ("host_seconds_rng", C.c_double), ("n_over", C.c_int32), ("over_slot", C.c_void_p),
                ("over_winner", C.c_void_p), ("bits", C.c_void_p), ("caps", C.c_void_p), ("pi32", C.c_void_p),
                ("to_play", C.c_void_p), ("difficulty", C.c_void_p), ("actions", C.c_void_p), ("root_value", C.c_void_p)]


# name -> (restype, argtypes); every symbol declared in include/azamd.h
_VP, _I32, _I64 = C.c_void_p, C.c_int32, C.c_int64
SIGNATURES = {
    "az_last_error": (C.c_char_p, []),
    "az_zobrist_key": (C.c_uint64, [_I32]),
    "az_engine_create": (_I32, [C.POINTER(AzConfig), C.POINTER(_VP)]),
    "az_engine_destroy": (_I32, [_VP]),
    "az_engine_action_size": (_I32, [_VP]),
    "az_engine_feature_planes": (_I32, [_VP]),
    "az_engine_seen_capacity": (_I32, [_VP]),
    "az_engine_max_rows": (_I32, [_VP]),
    "az_rules_reset": (_I32, [_VP, _VP, _VP]),
    "az_rules_play": (_I32, [_VP, _VP, _VP, _VP]),
    "az_rules_legal_mask": (_I32, [_VP, _VP, _VP]),
    "az_rules_features": (_I32, [_VP, _VP, _I32, _VP]),
    "az_rules_status": (_I32, [_VP, _VP, _VP, _VP]),
    "az_rules_export": (_I32, [_VP, _VP, _VP, _VP, _VP, _VP]),
    "az_rules_import": (_I32, [_VP, _VP, _VP, _VP, _VP, _VP]),
    "az_mcts_begin": (_I32, [_VP, _VP, _VP]),
    "az_mcts_select": (_I32, [_VP, _VP]),
    "az_mcts_pending": (_I32, [_VP, C.POINTER(_I32), _VP]),
    "az_mcts_set_root_reuse": (_I32, [_VP, _I32]),
    "az_mcts_gather_rows": (_I32, [_VP, C.POINTER(_I32), _VP]),
    "az_mcts_set_async": (_I32, [_VP, _I32]),
    "az_mcts_run_rounds": (_I32, [_VP, _VP, _VP, _I32, _VP, _VP, _I32, _VP]),
    "az_net_capturable": (_I32, [_VP, _I32]),
    "az_mcts_rows_device": (_I32, [_VP, C.POINTER(_VP)]),
    "az_mcts_gather": (_I32, [_VP, _VP, _I32, _VP]),
    "az_mcts_row_games": (_I32, [_VP, _VP, _VP]),
    "az_mcts_expand": (_I32, [_VP, _VP, _VP, _VP, _VP]),
    "az_mcts_root_legal_counts": (_I32, [_VP, _VP, _VP]),
    "az_mcts_result": (_I32, [_VP, _VP, _VP, _VP]),
    "az_mcts_last_priors": (_I32, [_VP, _VP, _VP]),
    "az_mcts_stats": (_I32, [_VP, _VP, _VP]),
    "az_net_create": (_I32, [C.POINTER(AzNetConfig), C.POINTER(_VP)]),
    "az_net_destroy": (_I32, [_VP]),
    "az_net_param_count": (_I64, [_VP]),
    "az_net_load": (_I32, [_VP, _VP, _I64]),
    "az_net_forward": (_I32, [_VP, _VP, _I32, _VP, _VP, _VP]),
    "az_net_reserve": (_I32, [_VP, _I32]),
    "az_net_forward_static": (_I32, [_VP, _VP, _I32, _VP, _VP, _VP]),
    "az_net_static_max_rows": (_I64, [_VP]),
    "az_net_indirect_passes": (_I32, [_VP, _I32]),
    "az_net_is_fused": (_I32, [_VP, C.POINTER(_I32)]),
    "az_net_forward_indirect": (_I32, [_VP, _VP, _VP, _I32, _VP, _VP, _VP]),
    "az_net_profile_spans": (_I32, [_VP, _VP, C.c_int64, C.POINTER(C.c_int64)]),
    "az_net_set_tower_workgroups": (_I32, [_VP, _I32]),
    "az_net_set_device_choice": (_I32, [_VP, _I32]),
    "az_net_profile": (_I32, [_VP, _I32, C.POINTER(C.c_double), C.POINTER(_I64), C.POINTER(_I64)]),
    "az_replay_gather": (_I32, [_VP, _VP, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _VP, _VP, _VP]),
    "az_pack_planes": (_I32, [_VP, _I32, _I32, _I32, _I32, _VP, _VP, _VP]),
    "az_lane_create": (_I32, [_VP, _VP, _VP, C.POINTER(AzLaneConfig), _VP, _I32, _VP, _I32, _VP, _VP, C.POINTER(_VP)]),
    "az_lane_destroy": (_I32, [_VP]),
    "az_lane_slots": (_I32, [_VP, C.POINTER(_VP), C.POINTER(_VP)]),
    "az_lane_prepare": (_I32, [_VP, _VP]),
    "az_lane_launch": (_I32, [_VP, _I32, _VP]),
    "az_lane_finish": (_I32, [_VP, C.POINTER(AzLaneResult)]),
    "az_streams_overlap": (_I32, [_VP, _VP, _I32, _I32, C.POINTER(_I32)]),
    "az_rng_create": (_I32, [_I32, C.POINTER(_VP)]),
    "az_rng_destroy": (_I32, [_VP]),
    "az_rng_seed": (_I32, [_VP, _I32, C.c_uint32]),
    "az_rng_dirichlet": (_I32, [_VP, _VP, C.c_double, _VP, _I32, _I32]),
    "az_rng_random_sample": (_I32, [_VP, _VP, _VP]),
}

_lib = None


class NativeError(RuntimeError):
    pass


def lib():
    """Load libazamd.so, binding every declared entry point. Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc: int, what: str = "") -> int:
    """Translate a negative az_status into the exception the upstream Python code raises."""
    if rc >= 0:
        return rc
    msg = lib().az_last_error().decode() or what
    if rc == AZ_ERR_ILLEGAL:
        raise ValueError(msg or "Illegal move")
    if rc == AZ_ERR_NOCHILD:
        raise RuntimeError("No child selected in MCTS")
    raise NativeError(f"{what}: {msg} (status {rc})")
