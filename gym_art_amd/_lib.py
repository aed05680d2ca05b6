"""ctypes binding of libgaq.so (include/gaq.h).

There is deliberately NO fallback: if the HIP library is missing or no GPU is
visible, constructing an env raises.  (`load()` itself only dlopens, so the
CPU-only container can still check that every declared symbol is exported.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GAQ_LIB") or os.path.join(_HERE, "libgaq.so")      # GAQ_LIB: measurement builds (tools/aux_variants.sh)
ABI_VERSION = 5
STATE_PLANES = 42
AUX_WORDS = 17

CTRL_RAW_ZERO_MIDDLE, CTRL_RAW, CTRL_MELLINGER = 0, 1, 2
NOISE_OFF, NOISE_PHILOX, NOISE_INPUT = 0, 1, 2
REW_QUADROTOR, REW_MULTI_LOG = 0, 1
OBS_BODY_FRAME, OBS_APPEND_H, OBS_APPEND_ACC, OBS_APPEND_ACT = 1, 2, 4, 8
OBS_QUAT, OBS_APPEND_T2W, OBS_APPEND_T2T = 16, 32, 64


class GaqModel(C.Structure):
    """gaq_model: the constants QuadrotorDynamics.update_model derives (quadrotor.py:142-208)."""
    _fields_ = [("mass", C.c_double), ("inertia", C.c_double * 3), ("thrust_max", C.c_double * 4),
                ("torque_max", C.c_double * 4), ("prop_pos", C.c_double * 12), ("damp_time_up", C.c_double),
                ("damp_time_down", C.c_double), ("linearity", C.c_double), ("arm", C.c_double),
                ("ou_sigma", C.c_double), ("vel_damp", C.c_double), ("damp_omega_quadratic", C.c_double),
                ("c_drag", C.c_double), ("c_roll", C.c_double)]


MODEL_DOUBLES = C.sizeof(GaqModel) // 8
MODEL_FIELDS = [("mass", 1), ("inertia", 3), ("thrust_max", 4), ("torque_max", 4), ("prop_pos", 12),
                ("damp_time_up", 1), ("damp_time_down", 1), ("linearity", 1), ("arm", 1), ("ou_sigma", 1),
                ("vel_damp", 1), ("damp_omega_quadratic", 1), ("c_drag", 1), ("c_roll", 1)]


class GaqRewCoeff(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("pos", "effort", "crash", "orient", "yaw", "rot", "attitude", "spin",
                                         "action_change", "vel", "pos_offset", "pos_log_weight", "pos_linear_weight")]


class GaqSenseNoise(C.Structure):
    _fields_ = [("enabled", C.c_int32)] + [(k, C.c_float) for k in (
        "pos_norm_std", "pos_unif_range", "vel_norm_std", "vel_unif_range", "quat_norm_std", "quat_unif_range",
        "gyro_noise_density", "acc_static_noise_std", "acc_dynamic_noise_ratio", "gyro_norm_std", "gyro_random_walk",
        "gyro_bias_correlation_time")]


class GaqSwarm(C.Structure):
    _fields_ = [("agents", C.c_int32)] + [(k, C.c_float) for k in (
        "goal_radius", "collision_dist", "prox_dist", "w_collision", "w_prox")] + [("response", C.c_int32)]


class GaqQuadParams(C.Structure):
    """gaq_quad_params: one parameter tree, flat (quad_params.TREE_LEAVES order)."""
    _fields_ = [("body", C.c_double * 4), ("payload", C.c_double * 4), ("arms", C.c_double * 4), ("motors", C.c_double * 3),
                ("propellers", C.c_double * 3), ("motor_pos", C.c_double * 3), ("arms_pos", C.c_double * 2),
                ("payload_pos", C.c_double * 3), ("damp", C.c_double * 2), ("noise", C.c_double * 1), ("motor", C.c_double * 11)]


TREE_DOUBLES = C.sizeof(GaqQuadParams) // 8


class GaqCounters(C.Structure):          # include/gaq.h: gaq_counters (checkpoint / resume)
    _fields_ = [("step_index", C.c_uint64), ("reset_calls", C.c_uint64)]


class GaqRandomizer(C.Structure):
    _fields_ = [("sampler", C.c_int32), ("every", C.c_int32), ("ratio", C.c_double * TREE_DOUBLES), ("base", GaqQuadParams)]


class GaqConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("abi_version", C.c_uint32), ("num_envs", C.c_int64),
                ("env_id_offset", C.c_int64), ("device", C.c_int32), ("seed", C.c_uint64), ("sim_freq", C.c_double),
                ("sim_steps", C.c_int32), ("ep_len", C.c_int32), ("room_size", C.c_double), ("gravity", C.c_double),
                ("t2w_std", C.c_double), ("t2t_std", C.c_double),
                ("control", C.c_int32), ("noise", C.c_int32), ("reward_mode", C.c_int32), ("obs_flags", C.c_int32),
                ("auto_reset", C.c_int32), ("init_random_state", C.c_int32), ("resample_goal", C.c_int32),
                ("per_env_params", C.c_int32), ("compact_done", C.c_int32), ("obs_state_alias", C.c_int32), ("fp32_state", C.c_int32), ("excite", C.c_int32),
                ("aux_outputs", C.c_int32), ("action_f32", C.c_int32), ("sense_input", C.c_int32), ("swarm", GaqSwarm),
                ("rew", GaqRewCoeff), ("sense", GaqSenseNoise),
                ("model", GaqModel)]


class GaqPlanInfo(C.Structure):          # include/gaq.h: gaq_plan_info (kernel selection without a device)
    _fields_ = [(k, C.c_int32) for k in ("obs_dim", "state_layout", "fp32", "step_variant", "step_instantiated", "launchable",
                                         "rollout_variant", "rollout_instantiated", "lds_per_wave", "rows_variant", "ctr_variant",
                                         "ctr_waves", "ctr_shift", "ctr_inc0")]


# every symbol include/gaq.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("gaq_num_devices", C.c_int, []),
    ("gaq_last_error", C.c_char_p, []),
    ("gaq_abi_version", C.c_int, []),
    ("gaq_is_diag_build", C.c_int, []),
    ("gaq_plan", C.c_int, [C.POINTER(GaqConfig), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(GaqPlanInfo)]),
    ("gaq_create", C.c_int, [C.POINTER(GaqConfig), C.POINTER(_P)]),
    ("gaq_destroy", C.c_int, [_P]),
    ("gaq_obs_dim", C.c_int, [_P]),
    ("gaq_obs_is_state", C.c_int, [_P]),
    ("gaq_state_layout", C.c_int, [_P]),
    ("gaq_num_envs", C.c_int64, [_P]),
    ("gaq_kernel_variant", C.c_int, [_P]),
    ("gaq_launch_variant", C.c_int, [_P]),
    ("gaq_launched_variants", C.c_int, [C.c_int, C.POINTER(C.c_uint32), C.c_int]),
    ("gaq_set_params", C.c_int, [_P, _P, C.c_int64, C.c_int64]),
    ("gaq_set_params_indexed", C.c_int, [_P, _P, _P, C.c_int64]),
    ("gaq_set_randomizer", C.c_int, [_P, C.POINTER(GaqRandomizer)]),
    ("gaq_randomize_dev", C.c_int, [_P, _P, _P]),
    ("gaq_set_param_trees", C.c_int, [_P, _P, C.c_int32, C.c_int64, C.c_int64]),
    ("gaq_get_params", C.c_int, [_P, _P, C.c_int64, C.c_int64]),
    ("gaq_get_param_trees", C.c_int, [_P, _P, C.c_int64, C.c_int64]),
    ("gaq_get_counters", C.c_int, [_P, _P, _P, _P]),
    ("gaq_set_counters", C.c_int, [_P, _P, _P, _P]),
    ("gaq_reset", C.c_int, [_P, _P, _P]),
    ("gaq_reset_dev", C.c_int, [_P, _P, _P, _P]),
    ("gaq_step", C.c_int, [_P, _P, _P, _P, _P]),
    ("gaq_step_dev", C.c_int, [_P, _P, _P, _P, _P, _P]),
    ("gaq_step_many_dev", C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P]),
    ("gaq_set_noise_input_dev", C.c_int, [_P, _P]),
    ("gaq_set_sense_input_dev", C.c_int, [_P, _P]),
    ("gaq_set_action_dtype", C.c_int, [_P, C.c_int32]),
    ("gaq_get_aux", C.c_int, [_P, _P]),
    ("gaq_get_state", C.c_int, [_P, _P]),
    ("gaq_set_state", C.c_int, [_P, _P]),
    ("gaq_observe", C.c_int, [_P, _P]),
    ("gaq_done_list", C.c_int, [_P, _P, C.c_int64, C.POINTER(C.c_int64)]),
    ("gaq_set_terminal_obs_dev", C.c_int, [_P, _P]),
    ("gaq_track_episodes", C.c_int, [_P, C.c_int32]),
    ("gaq_episode_stats", C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    C.POINTER(C.c_double), C.c_int32]),
    ("gaq_pack_rows_dev", C.c_int, [_P, _P, _P, _P, _P, _P]),
    ("gaq_set_packed_rows_dev", C.c_int, [_P, _P]),
    ("gaq_nan_count", C.c_int, [_P, C.POINTER(C.c_int64)]),
    ("gaq_last_kernel_ms", C.c_int, [_P, C.POINTER(C.c_float)]),
    ("gaq_set_graph_safe", C.c_int, [_P, C.c_int32]),
    ("gaq_set_timing", C.c_int, [_P, C.c_int32]),
    ("gaq_hbm_copy_dev", C.c_int, [_P, _P, C.c_size_t, _P]),
    ("gaq_synchronize", C.c_int, [_P]),
    ("gaq_stream", C.c_void_p, [_P]),
    # one batch over several devices, one process (include/gaq.h: gaq_sharded)
    ("gaq_create_sharded", C.c_int, [C.POINTER(GaqConfig), C.POINTER(C.c_int32), C.c_int32, C.POINTER(_P)]),
    ("gaq_sharded_from_handles", C.c_int, [C.POINTER(_P), C.c_int32, C.POINTER(_P)]),
    ("gaq_destroy_sharded", C.c_int, [_P]),
    ("gaq_sharded_num_shards", C.c_int, [_P]),
    ("gaq_sharded_shard", C.c_void_p, [_P, C.c_int32]),
    ("gaq_sharded_range", C.c_int, [_P, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    ("gaq_sharded_num_envs", C.c_int64, [_P]),
    ("gaq_shard_range", C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("gaq_reset_sharded_dev", C.c_int, [_P, _P, _P, _P]),
    ("gaq_step_sharded_dev", C.c_int, [_P, _P, _P, _P, _P, _P]),
    ("gaq_reset_sharded", C.c_int, [_P, _P, _P]),
    ("gaq_step_sharded", C.c_int, [_P, _P, _P, _P, _P]),
    ("gaq_synchronize_sharded", C.c_int, [_P]),
]

_lib = None


class GaqError(RuntimeError):
    pass


def load():
    """dlopen libgaq.so and bind every declared entry point.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # torch wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.so.1 (same SONAMEs as /opt/rocm's).
        # Whichever copy is loaded first serves the whole process, and torch stops seeing the GPU when it is
        # not its own -- so when torch is installed it has to be imported before libgaq.so is dlopened.
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise GaqError("libgaq.so not found at %s -- build it with `python __graft_entry__.py` or "
                       "`make -C gym_art_amd/csrc` (hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)       # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.gaq_abi_version() != ABI_VERSION:
        raise GaqError("libgaq ABI %d != binding ABI %d" % (lib.gaq_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc):
    """Translate a gaq_status into the exception the reference would raise."""
    if rc == 0:
        return
    msg = load().gaq_last_error().decode("utf-8", "replace")
    if rc == -1:
        raise ValueError("gaq: " + msg)
    if rc == -3:
        raise ValueError(msg)            # 'QuadEnv: reward is Nan' (quadrotor.py:636)
    raise GaqError("gaq (status %d): %s" % (rc, msg))


def ptr(a):
    """Raw pointer of a NumPy array / torch tensor / int / None."""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return C.c_void_p(a.ctypes.data)
    if hasattr(a, "data_ptr"):     # torch tensor
        assert a.is_contiguous()
        return C.c_void_p(a.data_ptr())
    raise TypeError("cannot take a pointer of %r" % type(a))


def models_to_rows(models):
    """dict of arrays ([N] or [N,k], keys = gaq_model fields) -> contiguous [N, MODEL_DOUBLES] float64."""
    n = int(np.asarray(models["mass"]).reshape(-1).shape[0])
    rows = np.zeros((n, MODEL_DOUBLES), dtype=np.float64)
    col = 0
    for name, width in MODEL_FIELDS:
        v = np.asarray(models[name], dtype=np.float64).reshape(n, width)
        rows[:, col:col + width] = v
        col += width
    assert col == MODEL_DOUBLES
    return rows


def rows_to_models(rows):
    """[N, MODEL_DOUBLES] float64 (gaq_get_params) -> dict of arrays keyed like gaq_model's fields."""
    rows = np.asarray(rows, dtype=np.float64)
    out, col = {}, 0
    for name, width in MODEL_FIELDS:
        out[name] = rows[:, col].copy() if width == 1 else rows[:, col:col + width].copy()
        col += width
    return out


def row_to_model(row):
    m = GaqModel()
    C.memmove(C.byref(m), np.ascontiguousarray(row, dtype=np.float64).ctypes.data, C.sizeof(GaqModel))
    return m
