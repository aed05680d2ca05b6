"""Helpers for the -m gpu parity tests: drive libgaq.so through its C ABI (ctypes) with model
constants taken verbatim from the golden fixtures."""
import ctypes as C

import numpy as np

from gym_art_amd import _lib
from tests import hh

REW_Q = {"pos": 1., "effort": 0.05, "action_change": 0., "crash": 1., "orient": 1., "yaw": 0., "rot": 0.,
         "attitude": 0., "spin": 0.1, "vel": 0., "pos_offset": 0.1, "pos_log_weight": 1., "pos_linear_weight": 0.1}


def model_row(const):
    """golden `const_*` block -> gaq_model as a [33] float64 row."""
    return np.concatenate([
        [float(const["mass"])], np.asarray(const["inertia"], float), np.asarray(const["thrust_max"], float),
        np.asarray(const["torque_max"], float), np.asarray(const["prop_pos"], float).reshape(12),
        [float(const["damp_time_up"]), float(const["damp_time_down"]), float(const["motor_linearity"]),
         float(const["arm"]), float(const["thrust_noise_sigma"]), float(const["vel_damp"]),
         float(const["damp_omega_quadratic"]), float(const["C_rot_drag"]), float(const["C_rot_roll"])]])


class Handle(object):
    """Thin RAII wrapper over gaq_create / gaq_destroy + the host-pointer entry points."""

    def __init__(self, n, dt, sim_steps, ep_len, const=None, rows=None, control=0, noise=0, reward_mode=0,
                 obs_flags=0, rew=None, auto_reset=0, seed=0, env_id_offset=0, compact_done=0, init_random_state=0,
                 resample_goal=0, device=0, alias=0, fp32=0, sense=None, room_size=10.0, force_generic=False,
                 action_f32=0, sense_input=0, aux=0, per_env=None):
        self.lib = _lib.load()
        cfg = _lib.GaqConfig()
        cfg.struct_size = C.sizeof(cfg)
        cfg.abi_version = _lib.ABI_VERSION
        cfg.num_envs, cfg.env_id_offset, cfg.device, cfg.seed = n, env_id_offset, device, seed
        cfg.sim_freq, cfg.sim_steps, cfg.ep_len = 1.0 / dt, sim_steps, ep_len
        cfg.room_size, cfg.gravity = float(room_size), 9.81
        cfg.t2w_std, cfg.t2t_std = 0.005, 0.0005            # QuadrotorEnv's constructor defaults (quadrotor.py:658)
        cfg.control, cfg.noise, cfg.reward_mode, cfg.obs_flags = control, noise, reward_mode, obs_flags
        cfg.auto_reset, cfg.init_random_state, cfg.resample_goal = auto_reset, init_random_state, resample_goal
        cfg.per_env_params = (1 if rows is not None else 0) if per_env is None else int(per_env)
        cfg.compact_done = compact_done
        cfg.obs_state_alias = alias
        cfg.fp32_state = fp32
        cfg.action_f32, cfg.sense_input, cfg.aux_outputs = action_f32, sense_input, aux
        rc = dict(REW_Q)
        if reward_mode == 1:
            rc.update({"effort": 0.01, "spin": 0.})
        if rew:
            rc.update(rew)
        for k, v in rc.items():
            setattr(cfg.rew, k, float(v))
        if sense is not None:        # SensorNoise() defaults (sensor_noise.py:58-63) overridden by `sense`
            prm = dict(pos_norm_std=0.005, pos_unif_range=0., vel_norm_std=0.01, vel_unif_range=0., quat_norm_std=0.,
                       quat_unif_range=0., gyro_noise_density=0.000175, acc_static_noise_std=0.002,
                       acc_dynamic_noise_ratio=0.005, gyro_norm_std=0., gyro_random_walk=0.0105, gyro_bias_correlation_time=1000.)
            prm.update(sense)
            cfg.sense.enabled = 1
            for k, v in prm.items():
                setattr(cfg.sense, k, float(v))
        if const is not None:
            cfg.model = _lib.row_to_model(model_row(const))
        self.h = C.c_void_p()
        import os
        if force_generic:       # diagnostic switch of the library: run the generic instantiation whatever the options
            os.environ["GAQ_FORCE_GENERIC"] = "1"
        try:
            _lib.check(self.lib.gaq_create(C.byref(cfg), C.byref(self.h)))
        finally:
            os.environ.pop("GAQ_FORCE_GENERIC", None)
        self.n = n
        self.D = self.lib.gaq_obs_dim(self.h)
        self.alias = bool(self.lib.gaq_obs_is_state(self.h))
        if rows is not None:
            rows = np.ascontiguousarray(rows, dtype=np.float64)
            _lib.check(self.lib.gaq_set_params(self.h, _lib.ptr(rows), 0, n))

    def close(self):
        if self.h:
            self.lib.gaq_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_state(self, planes):
        planes = np.ascontiguousarray(planes, dtype=np.float64)
        assert planes.shape == (42, self.n)
        _lib.check(self.lib.gaq_set_state(self.h, _lib.ptr(planes)))

    def get_state(self):
        st = np.empty((42, self.n))
        _lib.check(self.lib.gaq_get_state(self.h, _lib.ptr(st)))
        return st

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.n, 4)
        obs = np.empty((self.n, self.D), np.float32)
        rew = np.empty(self.n, np.float32)
        done = np.empty(self.n, np.uint8)
        _lib.check(self.lib.gaq_step(self.h, _lib.ptr(a), _lib.ptr(obs), _lib.ptr(rew), _lib.ptr(done)))
        return obs, rew, done.astype(bool)

    def reset(self, mask=None):
        obs = np.empty((self.n, self.D), np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        _lib.check(self.lib.gaq_reset(self.h, _lib.ptr(m), _lib.ptr(obs)))
        return obs

    def observe(self):
        obs = np.empty((self.n, self.D), np.float32)
        _lib.check(self.lib.gaq_observe(self.h, _lib.ptr(obs)))
        return obs

    def get_aux(self):
        aux = np.empty((self.n, _lib.AUX_WORDS), np.float32)
        _lib.check(self.lib.gaq_get_aux(self.h, _lib.ptr(aux)))
        return aux

    def done_list(self):
        idx = np.empty(self.n, np.uint32)
        cnt = C.c_int64(0)
        _lib.check(self.lib.gaq_done_list(self.h, _lib.ptr(idx), self.n, C.byref(cnt)))
        return np.sort(idx[:cnt.value])


def planes_from_blocks(blocks, n):
    """[42, n] state planes: env i starts from the initial state of blocks[i % len(blocks)]."""
    st = np.zeros((42, n))
    for i in range(n):
        b = blocks[i % len(blocks)]
        dt = float(b["dt"])
        st[:39, i] = hh.pack_state(b["init_pos"], b["init_vel"], b["init_rot"], b["init_omega"], b["goal"],
                                 svd_ctr=int(round(float(b["init_svd"]) / dt)))
    return st


def run_blocks(handle, blocks, n, normals_fn=None):
    """Step `handle` through the (zero-padded) action sequences of the blocks; returns per-block outputs
    for the first replica of each block plus the max spread between replicas (lane independence)."""
    nb = len(blocks)
    T = max(b["obs"].shape[0] for b in blocks)
    handle.set_state(planes_from_blocks(blocks, n))
    obs_all = np.zeros((T, n, handle.D), np.float32)
    rew_all = np.zeros((T, n), np.float32)
    done_all = np.zeros((T, n), bool)
    for t in range(T):
        a = np.zeros((n, 4), np.float32)
        for i in range(n):
            b = blocks[i % nb]
            if "actions" in b and t < b["actions"].shape[0]:
                a[i] = b["actions"][t]
        if normals_fn is not None:
            normals_fn(t)
        obs_all[t], rew_all[t], done_all[t] = handle.step(a)
    outs = []
    spread = 0.0
    for k, b in enumerate(blocks):
        Tb = b["obs"].shape[0]
        outs.append(dict(obs=obs_all[:Tb, k], reward=rew_all[:Tb, k], done=done_all[:Tb, k]))
        for r in range(k + nb, n, nb):
            spread = max(spread, float(np.max(np.abs(obs_all[:Tb, r] - obs_all[:Tb, k]))))
    return outs, spread
