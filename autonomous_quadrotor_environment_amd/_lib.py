"""ctypes binding of libqenv.so (the C-ABI declared in include/qenv.h).

The HIP library is the product: there is no CPU fallback.  If the shared object is
missing or a symbol cannot be bound, importing this module raises immediately.
"""
import ctypes as C
import os

# torch first: its wheel ships its own libamdhip64.so; libqenv.so must bind to THAT runtime instance (the one holding
# the device context and the streams we are handed), which happens only if it is already loaded when CDLL resolves
# libqenv.so's dependency.  Loaded the other way round, the process ends up with two HIP runtimes and every call
# into libqenv.so fails with "no ROCm-capable device is detected".
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QENV_LIB") or os.path.join(_HERE, "libqenv.so")  # QENV_LIB: developer A/B builds

QENV_ABI_VERSION = 10
F32, F64 = 0, 1
MAIN, OPT = 0, 1
RK45, DP5, RK4 = 0, 1, 2
INTEGRATORS = {"rk45": RK45, "dp5": DP5, "rk4": RK4}
VARIANTS = {"main": MAIN, "opt": OPT}
CTRL_CONST, CTRL_LQR, CTRL_PID, CTRL_PIDPOS = 0, 1, 2, 3
STEP_FULL, STEP_DYNAMICS_ONLY, STEP_NO_AUX = 0, 1, 2
STEP_PUSH_WINDOW = 4
SENSOR_ROWS = 31
FLAG_DONE, FLAG_SOLVED, FLAG_HAS_SHAPING = 1, 2, 4

(FIELD_STATE, FIELD_ANG, FIELD_PREV_ANG, FIELD_ANG_VEL, FIELD_VQ, FIELD_W, FIELD_STEP_EFFORT,
 FIELD_CLIPPED_ACTION, FIELD_ACCEL, FIELD_PREV_SHAPING, FIELD_REWARD, FIELD_ABS_SUM, FIELD_EP_RETURN,
 FIELD_PID, FIELD_I, FIELD_FLAGS, FIELD_WINDOW, FIELD_LAST_RETURN, FIELD_EPISODES, FIELD_ROBUST_PARAMS,
 FIELD_GUST, FIELD_GUST_EVALS, FIELD_OUTER) = range(23)
OUTER_NONE, OUTER_VELPID, OUTER_POSPI = 0, 1, 2


class QenvConfig(C.Structure):
    _fields_ = [("n_envs", C.c_int64), ("t_step", C.c_double), ("n_max", C.c_int32), ("T", C.c_int32),
                ("training", C.c_int32), ("direct_control", C.c_int32), ("clipped", C.c_int32),
                ("variant", C.c_int32), ("integrator", C.c_int32), ("dtype", C.c_int32), ("compute", C.c_int32)]


class QenvError(RuntimeError):
    pass


class PpoStepArgs(C.Structure):  # qenv_ppo_step_args (include/qenv.h)
    _fields_ = [("k_in", C.c_int32), ("hidden", C.c_int32)] + [(n, C.c_void_p) for n in ("W1", "b1", "W2", "b2", "W3", "b3")] + \
               [("std", C.c_double), ("noise", C.c_void_p), ("seed", C.c_uint64), ("step", C.c_uint64), ("env_offset", C.c_int64)] + \
               [(n, C.c_void_p) for n in ("out_mean", "out_action", "out_logprob", "obs_store", "out_reward", "out_done", "out_mask",
                                          "out_valid")] + [("inband", C.c_int32), ("key_by_episode", C.c_int32), ("obs_rows_from", C.c_int32),
                                                           ("obs_slices", C.c_int32)]


# every symbol include/qenv.h declares: (name, restype, argtypes)
_vp, _i, _i64 = C.c_void_p, C.c_int, C.c_int64
_dptr = C.POINTER(C.c_double)
SYMBOLS = {
    "qenv_abi_version": (_i, []),
    "qenv_last_error": (C.c_char_p, []),
    "qenv_workspace_bytes": (C.c_size_t, [C.POINTER(QenvConfig)]),
    "qenv_create": (_i, [C.POINTER(QenvConfig), _vp, C.c_size_t, _vp, C.POINTER(_vp)]),
    "qenv_destroy": (_i, [_vp]),
    "qenv_set_stream": (_i, [_vp, _vp]),
    "qenv_field": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(C.c_int32)]),
    "qenv_permute_envs": (_i, [_vp, _vp, _vp, C.c_size_t]),
    "qenv_reset": (_i, [_vp, _vp, _vp, _vp, _vp, _i]),
    "qenv_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i]),
    "qenv_rollout": (_i, [_vp, _i, _i, _dptr, _i, _i]),
    "qenv_reward_function": (_i, [_vp, _i64] + [_vp] * 10),
    "qenv_mix": (_i, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "qenv_rollout_track": (_i, [_vp, _i, _vp, _i]),
    "qenv_lqr_action": (_i, [_vp, _vp]),
    "qenv_pid_action": (_i, [_vp, _dptr, _vp]),
    "qenv_pid_reset": (_i, [_vp, _vp]),
    "qenv_pidpos_action": (_i, [_vp, _dptr, _vp]),
    "qenv_window_push": (_i, [_vp, _vp, _vp, _i]),
    "qenv_mlp_forward": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "qenv_mlp_image_bytes": (C.c_size_t, [_i, _i]),
    "qenv_mlp_prepare": (_i, [_vp, _i, _i, _i, _i] + [_vp] * 7),
    "qenv_sensor_reset": (_i, [_vp, _vp, _vp, C.c_uint64, C.c_uint64, _i64, _i, _dptr]),
    "qenv_sensor_read": (_i, [_vp, _vp, _vp, C.c_uint64, C.c_uint64, _i64, _i, _dptr, _vp]),
    "qenv_sensor_ops": (_i, [_vp, _vp, C.POINTER(C.c_int32), _i, _vp, _i, _dptr, _vp]),
    "qenv_sensor_bias_reset": (_i, [_vp, _vp, _vp, _i, _dptr]),
    "qenv_attach_sensor": (_i, [_vp, _vp, _vp, C.c_uint64, _i64, _i, _dptr, _vp, C.c_uint64]),
    "qenv_dense_tail": (_i, [_i64, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "qenv_dense_tail_rows": (_i, [_i64, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "qenv_dense_tail_chunks": (_i, [_i64, C.c_int32, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "qenv_dense_tail_chunks_strided": (_i, [_i64, C.c_int32, _i64, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "qenv_sample_action": (_i, [_vp, _vp, C.c_double, _vp, C.c_uint64, C.c_uint64, _i64, _vp, _vp]),
    "qenv_policy_act": (_i, [_vp, _i, _i] + [_vp] * 6 + [C.c_double, _vp, C.c_uint64, C.c_uint64, _i64, _vp, _vp, _vp, _vp]),
    "qenv_sample_init_states": (_i, [_vp, C.c_uint64, C.c_uint64, _i64, _vp, _vp]),
    "qenv_reset_sampled": (_i, [_vp, C.c_uint64, C.c_uint64, _i64, _vp, _i]),
    "qenv_ppo_step": (_i, [_vp, _vp]),
    "qenv_ppo_flush": (_i, [_vp]),
    "qenv_ppo_rollout": (_i, [_vp, _vp, _i]),
    "qenv_policy_rollout": (_i, [_vp, _i, _i] + [_vp] * 6 + [_i, _vp, _vp, _vp, _vp]),
    "qenv_outer_velpid": (_i, [_vp, _vp, C.c_int32]),
    "qenv_outer_pospi": (_i, [_vp, _vp, C.c_double, C.c_double, C.c_double]),
    "qenv_outer_reset": (_i, [_vp, _vp]),
    "qenv_set_robust": (_i, [_vp, _i]),
    "qenv_set_ppo_fused": (_i, [_vp, _i]),
    "qenv_set_ppo_fused_block": (_i, [_vp, _i]),
    "qenv_get_ppo_fused": (_i, [_vp]),
    "qenv_robust_params": (_i, [_vp, _vp, C.c_uint64, _i64, _vp]),
    "qenv_robust_gusts": (_i, [_vp, _vp, C.c_int32, C.c_int32, C.c_uint64, _i64]),
    "qenv_gae": (_i, [_i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "qenv_adv_normalize": (_i, [_i, _i64, _vp, _vp, _i, _vp]),
    "qenv_gae_masked": (_i, [_i, _i64, _i] + [_vp] * 9),
    "qenv_adv_normalize_masked": (_i, [_i, _i64, _vp, _vp, _vp, _i, _vp]),
    "qenv_visual_reward": (_i, [_i, _i64, _i] + [_vp] * 13),
    "qenv_ppo_loss": (_i, [_i, _i64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_double, _vp, _vp, _vp, _vp]),
    "qenv_mlp_backward": (_i, [_i64, _i, _i, _i, _i, _i] + [_vp] * 14),
    "qenv_mlp_backward_ppo": (_i, [_i64, _i, _i, _i, _i] + [_vp] * 5 + [C.c_double] * 3 + [_i64] + [_vp] * 13),
    "qenv_mlp_backward_ppo_masked": (_i, [_i64, _i, _i, _i, _i] + [_vp] * 6 + [C.c_double] * 3 + [_i64] + [_vp] * 13),
    "qenv_weight_grad": (_i, [_i64, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "qenv_mlp_grad_ppo_workspace_bytes": (C.c_size_t, []),
    "qenv_mlp_grad_ppo": (_i, [_i64, _i, _i, _i, _i, _vp, _i64] + [_vp] * 5 + [C.c_double] * 3 + [_i64] + [_vp] * 15),
    "qenv_weight_grad_workspace_bytes": (C.c_size_t, [_i, _i]),
    "qenv_weight_grad_deterministic": (_i, [_i64, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "qenv_gae_workspace_bytes": (C.c_size_t, [_i64]),
    "qenv_gae_deterministic": (_i, [_i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "qenv_adam_step": (_i, [_i64, _vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_double, C.c_double, _i, _vp]),
    "qenv_adam_step_counted": (_i, [_i64, _vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_double, C.c_double, _vp, _vp]),
    "qenv_debug_copy_rows": (_i, [_i, _vp, _vp, _i, _i64, _vp]),
}

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
        "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc); there is no CPU fallback.")

lib = C.CDLL(LIB_PATH)
for _name, (_res, _args) in SYMBOLS.items():
    _fn = getattr(lib, _name)  # AttributeError here = header and library out of sync
    _fn.restype = _res
    _fn.argtypes = _args
if lib.qenv_abi_version() != QENV_ABI_VERSION:
    raise ImportError(f"libqenv.so ABI {lib.qenv_abi_version()} != binding {QENV_ABI_VERSION}")


def check(rc, what=""):
    if rc != 0:
        msg = lib.qenv_last_error()
        raise QenvError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())
