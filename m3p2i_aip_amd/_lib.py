"""ctypes binding of libm3p2i_hip.so (C-ABI declared in include/m3p2i_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or a call fails this
module raises.  Build with ``python -m m3p2i_aip_amd.build`` (hipcc, gfx950).
"""
from __future__ import annotations

import ctypes as C
import os

MAX_NU = 9
TOPK = 20
ABI_VERSION = 4

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("M3P2I_HIP_LIB") or os.path.join(_HERE, "lib", "libm3p2i_hip.so")

ENV_POINT, ENV_PANDA = 0, 1
HALTON_PLAIN, HALTON_FAURE = 0, 1
IPC_HANDLE_BYTES = 64
TASKS = {"navigation": 0, "push": 1, "pull": 2, "push_pull": 3, "reach": 4, "pick": 5,
         "place": 6, "idle": 7}

(BUF_STATES, BUF_ACTIONS, BUF_COST_HORIZON, BUF_TRAJ_COST, BUF_TRAJ_COST_ALL, BUF_WEIGHTS,
 BUF_WEIGHTS_1, BUF_WEIGHTS_2, BUF_MEAN, BUF_MEAN_1, BUF_MEAN_2, BUF_BEST, BUF_BEST_1,
 BUF_BEST_2, BUF_ACTION_OUT, BUF_TOP_IDX, BUF_TOP_TRAJS, BUF_REDUCE, BUF_NOISE,
 BUF_PENDING_FORCE, BUF_INFO, BUF_SIM_WORLD, BUF_RECORD, BUF_RECORDS_ALL, BUF_NOISE_ALL, BUF_COV,
 BUF_RECORD_B, BUF_RECORDS_B_ALL, BUF_COUNT) = range(29)


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int), ("device", C.c_int), ("K_global", C.c_int),
                ("K_local", C.c_int), ("k_offset", C.c_int), ("T", C.c_int), ("nu", C.c_int),
                ("env_type", C.c_int), ("multi_modal", C.c_int), ("mode_simple", C.c_int),
                ("sampling_random", C.c_int), ("sample_null_action", C.c_int),
                ("filter_u", C.c_int), ("u_per_command", C.c_int),
                ("u_min", C.c_float * MAX_NU), ("u_max", C.c_float * MAX_NU),
                ("noise_sigma_diag", C.c_float * MAX_NU), ("u_scale", C.c_float),
                ("gamma", C.c_float), ("lambda_", C.c_float), ("step_size_mean", C.c_float),
                ("kp_suction", C.c_float), ("pre_height_diff", C.c_float), ("dt", C.c_float),
                ("substeps", C.c_int), ("solver_iters", C.c_int), ("cube_on_shelf", C.c_int),
                ("sim_only", C.c_int), ("shard_mix", C.c_int), ("noise_abs_cost", C.c_int),
                ("update_cov", C.c_int), ("full_sigma", C.c_int), ("noise_mu", C.c_float * MAX_NU),
                ("noise_sigma_full", C.c_float * (MAX_NU * MAX_NU)), ("seed", C.c_ulonglong)]


class PointWorld(C.Structure):
    _fields_ = [("robot", C.c_float * 4), ("box", C.c_float * 7), ("dyn_obs", C.c_float * 7)]


class Info(C.Structure):
    _fields_ = [("eta", C.c_float), ("eta_1", C.c_float), ("eta_2", C.c_float),
                ("beta", C.c_float), ("beta_1", C.c_float), ("beta_2", C.c_float),
                ("iters", C.c_int), ("iters_1", C.c_int), ("iters_2", C.c_int),
                ("best_idx", C.c_int), ("best_idx_1", C.c_int), ("best_idx_2", C.c_int),
                ("wsum_push", C.c_float), ("wsum_pull", C.c_float),
                ("pull_preference", C.c_int), ("calls", C.c_int)]


INFO_WORDS = C.sizeof(Info) // 4
INFO_PULL_PREFERENCE = Info.pull_preference.offset // 4   # index of the field in an int32 view of M3_BUF_INFO


class Timing(C.Structure):
    _fields_ = [("rollout_ms", C.c_float), ("update_ms", C.c_float),
                ("finalize_ms", C.c_float), ("total_ms", C.c_float)]


# every symbol include/m3p2i_hip.h declares: (name, restype, argtypes)
_H = C.c_void_p
_FP = C.c_void_p  # float* (host or device), passed as integer addresses
SYMBOLS = [
    ("m3_abi_version", C.c_int, []),
    ("m3_build_id", C.c_char_p, []),
    ("m3_last_error", C.c_char_p, [_H]),
    ("m3_default_config", None, [C.POINTER(Config), C.c_int]),
    ("m3_create", C.c_int, [C.POINTER(Config), C.POINTER(_H)]),
    ("m3_destroy", None, [_H]),
    ("m3_set_stream", C.c_int, [_H, C.c_void_p]),
    ("m3_enable_timing", C.c_int, [_H, C.c_int]),
    ("m3_set_rollout_lanes", C.c_int, [_H, C.c_int]),
    ("m3_set_panda_lanes_per_sample", C.c_int, [_H, C.c_int]),
    ("m3_panda_lanes_per_sample_used", C.c_int, [_H]),
    ("m3_panda_near_share", C.c_int, [_H]),
    ("m3_set_panda_reach_cost_kernel", C.c_int, [_H, C.c_int]),
    ("m3_set_update_launches", C.c_int, [_H, C.c_int]),
    ("m3_set_ladder_spins", C.c_int, [_H, C.c_int]),
    ("m3_set_wave_order", C.c_int, [_H, C.c_int]),
    ("m3_relabel_samples", C.c_int, [_H]),
    ("m3_set_noise", C.c_int, [_H, _FP, C.c_int]),
    ("m3_set_noise_knots", C.c_int, [_H, _FP, C.c_int, C.c_int, C.c_float, C.c_int]),
    ("m3_set_noise_global", C.c_int, [_H, _FP, C.c_int]),
    ("m3_set_noise_knots_global", C.c_int, [_H, _FP, C.c_int, C.c_int, C.c_float, C.c_int]),
    ("m3_set_noise_halton", C.c_int, [_H, C.c_int, C.c_int, C.c_float]),
    ("m3_set_noise_halton_scrambled", C.c_int, [_H, C.c_int, C.c_int, C.c_float, C.c_int]),
    ("m3_sample_noise", C.c_int, [_H]),
    ("m3_set_call_count", C.c_int, [_H, C.c_uint]),
    ("m3_set_objective", C.c_int, [_H, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_int]),
    ("m3_set_avoid_dyn_obs", C.c_int, [_H, C.c_int]),
    ("m3_set_multi_modal", C.c_int, [_H, C.c_int]),
    ("m3_set_plan", C.c_int, [_H, C.c_int, _FP]),
    ("m3_set_action_out", C.c_int, [_H, C.c_void_p]),
    ("m3_set_beta", C.c_int, [_H, C.c_float]),
    ("m3_reset", C.c_int, [_H]),
    ("m3_set_world_point", C.c_int, [_H, C.POINTER(PointWorld)]),
    ("m3_set_world_point_raw", C.c_int, [_H, C.POINTER(C.c_float)]),
    ("m3_bind_sim_point", C.c_int, [_H, _FP, _FP, C.c_int, C.c_int, C.c_int]),
    ("m3_set_world_panda_raw", C.c_int, [_H, C.POINTER(C.c_float)]),
    ("m3_bind_sim_panda", C.c_int, [_H, _FP, _FP, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("m3_command", C.c_int, [_H, _FP]),
    ("m3_rollout", C.c_int, [_H]),
    ("m3_update", C.c_int, [_H]),
    ("m3_finalize", C.c_int, [_H]),
    ("m3_update_finalize", C.c_int, [_H]),
    ("m3_p2p_export", C.c_int, [_H, C.c_void_p]),
    ("m3_p2p_connect", C.c_int, [_H, C.c_void_p, C.c_int]),
    ("m3_p2p_connect_local", C.c_int, [_H, C.POINTER(_H), C.c_int]),
    ("m3_p2p_put", C.c_int, [_H]),
    ("m3_p2p_wait", C.c_int, [_H]),
    ("m3_p2p_exchange", C.c_int, [_H]),
    ("m3_p2p_put_ch", C.c_int, [_H, C.c_int]),
    ("m3_p2p_wait_ch", C.c_int, [_H, C.c_int]),
    ("m3_p2p_exchange_b", C.c_int, [_H]),
    ("m3_update_b", C.c_int, [_H]),
    ("m3_record_b_len", C.c_int, [_H]),
    ("m3_p2p_status", C.c_int, [_H, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("m3_p2p_set_timeout_ms", C.c_int, [_H, C.c_int, C.c_int]),
    ("m3_p2p_clear_error", C.c_int, [_H]),
    ("m3_p2p_detach", C.c_int, [_H]),
    ("m3_p2p_set_memory_kind", C.c_int, [_H, C.c_int]),
    ("m3_get_buffer", C.c_int, [_H, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_longlong)]),
    ("m3_reduce_len", C.c_int, [_H]),
    ("m3_record_len", C.c_int, [_H]),
    ("m3_get_info", C.c_int, [_H, C.POINTER(Info)]),
    ("m3_get_timing", C.c_int, [_H, C.POINTER(Timing)]),
    ("m3_sim_bind_views", C.c_int, [_H, _FP, _FP, _FP, _FP, C.c_int, C.c_int]),
    ("m3_sim_pull_state", C.c_int, [_H]),
    ("m3_sim_shift_actor", C.c_int, [_H, C.c_int, C.c_float, C.c_float, C.c_float]),
    ("m3_sim_step_with_target", C.c_int, [_H, C.c_void_p]),
    ("m3_sim_push_state", C.c_int, [_H]),
    ("m3_sim_set_velocity_target", C.c_int, [_H, _FP]),
    ("m3_sim_apply_body_forces", C.c_int, [_H, _FP]),
    ("m3_sim_step", C.c_int, [_H]),
    ("m3_cost", C.c_int, [_H, _FP]),
    ("m3_sim_suction_forces", C.c_int, [_H, C.c_float, _FP]),
    ("m3_sim_check_and_apply_suction", C.c_int, [_H, _FP, C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
]

_lib = None


class M3Error(RuntimeError):
    pass


def load():
    """Load the HIP library.  Raises if it has not been built -- there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise M3Error(f"{LIB_PATH} not found: build it with `python -m m3p2i_aip_amd.build` "
                      "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # torch must be imported FIRST: it maps its bundled libamdhip64.so.7, and our NEEDED entry
    # then resolves to that same runtime (one HIP runtime per process), so device pointers
    # and streams are interchangeable between torch and the library.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    if lib.m3_abi_version() != ABI_VERSION:
        raise M3Error("libm3p2i_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc, handle=None):
    if rc != 0:
        lib = load()
        msg = lib.m3_last_error(handle)
        raise M3Error(f"m3p2i_hip error {rc}: {msg.decode() if msg else ''}")
