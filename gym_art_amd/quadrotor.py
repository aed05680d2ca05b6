"""QuadrotorEnv: the reference's Gym surface over the fused HIP kernel.

Mirrors gym_art/quadrotor/quadrotor.py:647-1156 (`class QuadrotorEnv`): same class name, the same
constructor keywords with the same defaults, `reset()`, `step(a)`, `observation_space`,
`action_space`, `spec`, `resample_dynamics()`, pickling by constructor arguments -- so the loops
written against the reference (`reset(); while not done: step()`, quadrotor.py:1278-1305,
:1424-1428) run unchanged with `num_envs=1`.  Extension keywords select the batched mode:

    num_envs      N environments stepped by one kernel launch (default 1)
    device        HIP device ordinal (default: LOCAL_RANK or 0)
    device_ids    [d0, d1, ...]: ONE env object, ONE process, the batch cut into contiguous shards (whole 64-env tiles) over these GPUs;
                  step() / reset() fan out over the shards' own streams and return the stacked arrays / tensors on d0, bit-equal to one
                  handle holding all N envs (gym_art_amd/multi_device.py; include/gaq.h gaq_sharded).  A single id is just `device`
    backend       "hip" (the only one: there is deliberately no CPU path in the product; "cpu" raises)
    host_seed     seed of the HOST-side parameter sampler alone (default: `seed`; the shards of a multi-device env share `seed` but not this)
    seed          seed of the on-device counter-based RNG (reset states, thrust noise)
    auto_reset    re-initialise finished envs inside the step launch (default: num_envs > 1)
    env_id_offset global index of env 0 (multi-GPU shards; RNG streams follow the global index)
    thrust_noise  "philox" (default; on-device OU noise), "off", or "input" (caller-supplied normals)
    reward        "quadrotor" (default) or "multi" (log-distance reward of quadrotor_multi.py:550-650)
    info          build the reference's per-step info dict ('rewards', 'obs_comp', 'dyn_params'; quadrotor.py:
                  993-1028) on the host from the device state.  Default: on for num_envs == 1 (drop-in loops
                  read it), off for batches (it costs a device->host state copy per step).
    alias_obs     how the integrator state is stored (gaq_config.obs_state_alias).  None (default): split state with
                  library-owned heads, the returned observation is a copy you may edit freely.  True: the returned
                  observation tensor IS the state's fp32 head (least HBM traffic, what bench.py times) -- device-tensor
                  callers must then leave the tensor returned by step k untouched and alive until step k+1 has run
                  (GAQ_CHECK_ALIAS=1 turns violations into errors); NumPy callers always get copies.  False: fp64
                  state planes and a write-only observation tensor.
    sense_noise_input  take the sensor-noise draws from set_sense_input() instead of the device RNG (parity tests)
    terminal_observation  batched auto-reset mode: info["terminal_observation"] holds the LAST observation of every episode that ended in the
                  step (rows valid where done; the vector-env convention of Gym / Garage samplers -- the row returned by step() belongs
                  to the new episode); written by the step launch, fetched only on steps that end episodes
    out_ring      NumPy callers of big batches: hand out the step's obs / reward / done arrays from a ring of `out_ring` preallocated sets
                  instead of fresh arrays (0, the default: fresh arrays per call, like the reference).  A fresh 75-MB observation array
                  costs more in page faults than the whole PCIe round trip of the step (12 vs 2.3 ms per step at N = 2^20): with a ring an
                  array stays valid for `out_ring - 1` further steps (gym.vector's `copy=False` is the same trade with a ring of one)
    randomize_on_device  per-env parameter sampling (dyn_sampler_1 = RelativeSampler around a shipped model) and the
                  QuadLink / update_model derivation inside the library, on the GPU (gaq_set_randomizer): per-episode
                  re-randomisation (dynamics_randomize_every) then costs microseconds per step instead of a host round trip.
                  None (default): on for batches when the configuration allows it; True: required; False: host pipeline.

Everything numeric happens in libgaq.so on the GPU; there is no CPU path here.
"""
import copy
import ctypes as C
import math
import os

import numpy as np

from . import _lib
from . import quad_params as qp
from . import quadrotor_randomization as quad_rand
from .spaces import Box, EnvBase, EnvSpec

GRAV = 9.81
_CCW = np.array([-1., 1., -1., 1.])          # propeller directions (quadrotor.py:166)

OBS_FLAGS = {
    # get_state.py:5,134,147,219,236,249 -- the six observation packers that work in the reference
    "xyz_vxyz_R_omega": 0,
    "xyz_vxyz_R_omega_h": _lib.OBS_APPEND_H,
    "xyzr_vxyzr_R_omega": _lib.OBS_BODY_FRAME,
    "xyzr_vxyzr_R_omega_h": _lib.OBS_BODY_FRAME | _lib.OBS_APPEND_H,
    "xyz_vxyz_R_omega_acc_act": _lib.OBS_APPEND_ACC | _lib.OBS_APPEND_ACT,
    "xyz_vxyz_R_omega_act": _lib.OBS_APPEND_ACT,
    # get_state.py:276-384 -- complete in the reference but NameError as shipped (the module never imports `normal` / `R2quat`);
    # pinned by the patched-import fixture G15 (tests/golden/make_golden.py)
    "xyz_vxyz_R_omega_t2w": _lib.OBS_APPEND_T2W,
    "xyzr_vxyzr_R_omega_t2w": _lib.OBS_BODY_FRAME | _lib.OBS_APPEND_T2W,
    "xyz_vxyz_R_omega_t2w_t2t": _lib.OBS_APPEND_T2W | _lib.OBS_APPEND_T2T,
    "xyz_vxyz_quat_omega": _lib.OBS_QUAT,
    "xyzr_vxyzr_quat_omega": _lib.OBS_QUAT | _lib.OBS_BODY_FRAME,
    "xyzr_vxyzr_quat_omega_h": _lib.OBS_QUAT | _lib.OBS_BODY_FRAME | _lib.OBS_APPEND_H,
}

REW_DEFAULT = {   # quadrotor.py:799-806
    "pos": 1., "effort": 0.05, "action_change": 0., "crash": 1., "orient": 1., "yaw": 0., "rot": 0.,
    "attitude": 0., "spin": 0.1, "vel": 0.}
REW_DEFAULT_MULTI = {   # quadrotor_multi/quadrotor_multi.py:811-818
    "pos": 1., "pos_offset": 0.1, "pos_log_weight": 1., "pos_linear_weight": 0.1, "effort": 0.01,
    "action_change": 0., "crash": 1., "orient": 1., "yaw": 0., "rot": 0., "attitude": 0., "spin": 0., "vel": 0.}


def _is_torch(x):
    return hasattr(x, "data_ptr") and hasattr(x, "device")


class DynamicsView(object):
    """`env.dynamics`: read access to the batched device state with the attribute names of
    QuadrotorDynamics (quadrotor.py:60-540).  Arrays carry a leading env axis unless num_envs == 1."""

    def __init__(self, env):
        self._env = env
        self.omega_max = 40.   # quadrotor.py:91-94
        self.vxyz_max = 3.
        self.acc_max = 3. * GRAV
        self.prop_ccw = np.array([-1., 1., -1., 1.])

    def _planes(self):
        return self._env.get_state()

    def _sq(self, a):
        return a[0] if self._env.num_envs == 1 else a

    @property
    def pos(self):
        return self._sq(self._planes()[0:3].T.copy())

    @property
    def vel(self):
        return self._sq(self._planes()[3:6].T.copy())

    @property
    def rot(self):
        return self._sq(self._planes()[6:15].T.reshape(-1, 3, 3).copy())

    @property
    def omega(self):
        return self._sq(self._planes()[15:18].T.copy())

    @property
    def thrust_rot_damp(self):
        return self._sq(self._planes()[18:22].T.copy())

    @property
    def thrust_cmds_damp(self):
        return self._sq(self._planes()[22:26].T.copy())

    def __getattr__(self, name):
        models = self.__dict__["_env"].models
        alias = {"motor_linearity": "linearity", "motor_damp_time_up": "damp_time_up",
                 "motor_damp_time_down": "damp_time_down", "damp_omega_quadratic": "damp_omega_quadratic",
                 "C_rot_drag": "c_drag", "C_rot_roll": "c_roll"}
        key = alias.get(name, name)
        if key in models:
            v = models[key]
            if key == "prop_pos":
                v = v.reshape(-1, 4, 3)
            return self._sq(v)
        extra = self.__dict__["_env"].models_extra
        if key in extra:
            return self._sq(extra[key])
        raise AttributeError(name)


class QuadrotorEnv(EnvBase):
    metadata = {'render.modes': ['human', 'rgb_array'], 'video.frames_per_second': 50}

    def __init__(self, dynamics_params="DefaultQuad", dynamics_change=None,
                 dynamics_randomize_every=None, dyn_sampler_1=None, dyn_sampler_2=None,
                 raw_control=True, raw_control_zero_middle=True, dim_mode='3D', tf_control=False, sim_freq=200.,
                 sim_steps=2, obs_repr="xyz_vxyz_R_omega", ep_time=7, obstacles_num=0, room_size=10,
                 init_random_state=False, rew_coeff=None, sense_noise=None, verbose=False, gravity=GRAV,
                 resample_goal=False, t2w_std=0.005, t2t_std=0.0005, excite=False, dynamics_simplification=False,
                 num_envs=1, device=None, seed=None, auto_reset=None, env_id_offset=0, thrust_noise="philox",
                 reward="quadrotor", compact_done=False, alias_obs=None, info=None, swarm=None, precision="fp64",
                 sense_noise_input=False, randomize_on_device=None, terminal_observation=False, out_ring=0,
                 device_ids=None, host_seed=None, backend="hip"):
        kwargs = dict(locals())
        kwargs.pop("self")
        self._ctor_kwargs = copy.deepcopy(kwargs)      # pickling by constructor args (quadrotor.py:688)
        # device_ids=[d0, d1, ...]: ONE env object, ONE process, the batch cut into contiguous shards over those GPUs;
        # step() / reset() fan out over the shards' own streams and hand back the stacked arrays / tensors on d0 (gym_art_amd/
        # multi_device.py, include/gaq.h gaq_sharded).  A single id is just `device`.
        self.device_ids = None if device_ids is None else [int(d) for d in device_ids]
        if self.device_ids is not None:
            if len(self.device_ids) == 0:
                raise ValueError("device_ids must name at least one device")
            if device is not None and int(device) != self.device_ids[0]:
                raise ValueError("device and device_ids[0] disagree")
            device = self.device_ids[0]
            if len(self.device_ids) > 1:
                from .multi_device import multi_device_class
                self.__class__ = multi_device_class(type(self))      # the handle-facing methods now go to the shards
        # ---- options of the reference that this path does not implement: fail loudly --------------------
        if backend != "hip":
            # SURVEY 8b sketches backend="hip"|"cpu"; the tier's rule is that a product path with ANY CPU fallback voids the parity claims,
            # so there is exactly one backend and asking for another fails loudly (the CPU restatements live under oracle/: test infrastructure)
            raise NotImplementedError("backend=%r: gym_art_amd has no CPU backend -- the only product path is the HIP library (libgaq.so)" % (backend,))
        if dim_mode != '3D':
            raise ValueError('QuadEnv: Unknown dimensionality mode %s (only 3D is built; the 1D/2D controllers '
                             'of the reference crash in _step, quadrotor.py:1002)' % dim_mode)
        if tf_control:
            raise NotImplementedError("tf_control (TF1 graph, quadrotor_control.py:388-511) is out of scope")
        if obstacles_num:
            raise NotImplementedError("obstacles are broken in the reference (quadrotor.py:870) and out of scope")
        self._sense = self._parse_sense_noise(sense_noise)
        self.excite = bool(excite)
        self._swarm = self._parse_swarm(swarm, int(num_envs), int(env_id_offset))
        if obs_repr not in OBS_FLAGS:
            raise AttributeError("module 'get_state' has no attribute 'state_%s'" % obs_repr)
        if reward not in ("quadrotor", "multi"):
            raise ValueError("reward must be 'quadrotor' or 'multi'")
        if precision not in ("fp64", "fp32"):
            raise ValueError("precision must be 'fp64' (the parity path) or 'fp32' (throughput-first, ~1e-4 drift)")
        self.precision = precision
        if thrust_noise not in ("philox", "off", "input"):
            raise ValueError("thrust_noise must be 'philox', 'off' or 'input'")

        self.num_envs = int(num_envs)
        self.init_random_state = init_random_state
        self.room_size = room_size
        self.obs_repr = obs_repr
        self.sim_steps = sim_steps
        self.dim_mode = dim_mode
        self.raw_control = raw_control
        self.raw_control_zero_middle = raw_control_zero_middle
        self.dynamics_randomize_every = dynamics_randomize_every
        self.verbose = verbose
        self.gravity = gravity
        self.resample_goal = resample_goal
        self.t2w_std, self.t2w_min, self.t2w_max = t2w_std, 1.5, 10.0       # the t2w / t2t observation components (:706-712)
        self.t2t_std, self.t2t_min, self.t2t_max = t2t_std, 0.005, 1.0
        self.dynamics_simplification = dynamics_simplification
        self.room_box = np.array([[-room_size, -room_size, 0], [room_size, room_size, room_size]], dtype=np.float64)
        self.box = 2.0
        self.traj_count = 0
        self._auto_reset = bool(self.num_envs > 1) if auto_reset is None else bool(auto_reset)
        self._seed_value = int(seed) if seed is not None else int.from_bytes(os.urandom(4), "little")
        # (host_seed: the seed of the HOST-side parameter sampler alone -- the shards of a multi-device env share `seed`, which keys the
        #  device's random streams by global env index, but must not draw the same parameter sets on the host)
        self._rng = np.random.RandomState((self._seed_value if host_seed is None else int(host_seed)) & 0x7FFFFFFF)
        self._thrust_noise = thrust_noise
        self._reward = reward
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = int(device)
        self.env_id_offset = int(env_id_offset)
        self._compact_done = bool(compact_done)
        # alias_obs (the obs tensor doubles as state): the default is ON only where the library owns the observation
        # buffer -- NumPy in / NumPy out callers get copies -- and opt-in for device-tensor loops (step_dev / torch
        # actions), where an in-place edit of the returned tensor would corrupt the physics.  None = that rule.
        self._alias_request = alias_obs
        self._info = bool(self.num_envs == 1) if info is None else bool(info)
        self._sense_input = bool(sense_noise_input)
        self._terminal_observation = bool(terminal_observation)
        self._term_buf = None
        self._out_ring, self._ring, self._ring_pos = int(out_ring), None, 0
        self._one = None       # num_envs == 1: the persistent buffers of the drop-in loop and their pointers (step)
        self._action_f32 = True      # arithmetic of RawControl on the caller's dtype: float32 arrays unless told otherwise
        self._actions = [np.zeros((self.num_envs, 4)), np.zeros((self.num_envs, 4))]
        self._per_env_traj = np.zeros(self.num_envs, dtype=np.int64)
        self._obs_ref = None          # keeps the previous observation tensor alive (alias mode: it is state)

        # ---- episode parameters (quadrotor.py:789-795) ---------------------------------------------------
        self.ep_time = ep_time
        self.dt = 1.0 / sim_freq
        self.sim_freq = sim_freq
        self.metadata = dict(self.metadata)
        self.metadata["video.frames_per_second"] = sim_freq / self.sim_steps
        self.ep_len = int(self.ep_time / (self.dt * self.sim_steps))
        self.control_freq = sim_freq / sim_steps
        self.tick = 0
        self.crashed = False

        # ---- reward weights (quadrotor.py:799-818) -------------------------------------------------------
        self.rew_coeff = dict(REW_DEFAULT if reward == "quadrotor" else REW_DEFAULT_MULTI)
        if rew_coeff is not None:
            assert isinstance(rew_coeff, dict)
            assert set(rew_coeff.keys()).issubset(set(self.rew_coeff.keys()))
            self.rew_coeff.update(rew_coeff)
        for key in self.rew_coeff.keys():
            self.rew_coeff[key] = float(self.rew_coeff[key])

        # ---- dynamics parameters and their randomisation (quadrotor.py:738-764) -------------------------
        self.dyn_base_sampler = getattr(quad_rand, dynamics_params)()
        self.dynamics_change = copy.deepcopy(dynamics_change)
        self._sampler_1 = self._make_sampler(dyn_sampler_1)
        self._sampler_2 = self._make_sampler(dyn_sampler_2)
        self._per_env = (dynamics_params == "RandomQuad" or dyn_sampler_1 is not None or dyn_sampler_2 is not None)
        self._handle = None
        self._models_host, self._models_extra_host, self._models_cache, self._extra_cache = None, None, None, None
        self._dyn_params_cache = None
        self._dev_rand = self._decide_device_randomizer(randomize_on_device, dynamics_params, dyn_sampler_1, dyn_sampler_2)
        self._lib = _lib.load()
        self.dynamics = DynamicsView(self)
        self.resample_dynamics()              # also (re)creates the device handle

        self.observation_space = self.make_observation_space()
        self.action_space = self._make_action_space()
        # (what vector-env samplers look for; the spaces above describe ONE env whatever num_envs is, like the reference's)
        self.single_observation_space, self.single_action_space = self.observation_space, self.action_space
        self.spec = EnvSpec(id='Quadrotor-v0', max_episode_steps=self.ep_len)
        self._last_obs = None
        if self._terminal_observation and self._auto_reset:
            # the vector-env convention: with auto-reset the observation returned with done = 1 is the first one of the NEW episode; the last
            # one of the finished episode -- what the reference returns with done=True, needed to bootstrap a value at this time-limit
            # truncation -- goes to info["terminal_observation"] (rows valid where done); written by the step launch itself
            import torch
            self._term_buf = torch.zeros((self.num_envs, self.obs_dim), dtype=torch.float32, device=torch.device("cuda", self.device))
            self.set_terminal_obs(self._term_buf)
        self.reset()

    SWARM_DEFAULTS = dict(agents=8, goal_radius=0.5, collision_dist=None, prox_dist=None, w_collision=1.0, w_prox=0.5,
                          collision_response=True)

    @classmethod
    def _parse_swarm(cls, swarm, num_envs, env_id_offset):
        """Swarm layer options (include/gaq.h gaq_swarm; this build's own specification, DESIGN.md "Swarm layer").
        collision_dist defaults to 2 x arm (rotor discs touching), prox_dist to 4 x collision_dist."""
        if swarm is None:
            return None
        unknown = set(swarm) - set(cls.SWARM_DEFAULTS)
        if unknown:
            raise TypeError("unknown swarm option '%s'" % sorted(unknown)[0])
        prm = dict(cls.SWARM_DEFAULTS, **swarm)
        a = int(prm["agents"])
        if a < 2 or a > 16 or (a & (a - 1)):
            raise ValueError("swarm agents must be a power of two in [2, 16]")
        if num_envs % a or env_id_offset % a:
            raise ValueError("num_envs and env_id_offset must be multiples of the number of agents per world")
        return prm

    @property
    def actions(self):
        """env.actions (quadrotor.py:943-944): [current, previous] action arrays, float64."""
        self._actions = [np.asarray(x, dtype=np.float64) for x in self._actions]
        return self._actions

    @actions.setter
    def actions(self, v):
        self._actions = list(v)

    @property
    def goal(self):
        """env.goal (quadrotor.py:1078-1081, :957-963): (0, 0, 2) unless resample_goal / excite / a swarm formation
        move it per env."""
        if not (self.resample_goal or self.excite or self._swarm) or getattr(self, "_handle", None) is None:
            return np.array([0., 0., 2.])
        g = self.get_state()[34:37].T.copy()
        return g[0] if self.num_envs == 1 else g

    # ------------------------------------------------------------------------------------------------
    SENSE_DEFAULTS = dict(pos_norm_std=0.005, pos_unif_range=0., vel_norm_std=0.01, vel_unif_range=0., quat_norm_std=0.,
                          quat_unif_range=0., gyro_norm_std=0., gyro_noise_density=0.000175, gyro_random_walk=0.0105,
                          gyro_bias_correlation_time=1000., bypass=False, acc_static_noise_std=0.002,
                          acc_dynamic_noise_ratio=0.005)     # SensorNoise.__init__ (sensor_noise.py:58-63)

    def _parse_sense_noise(self, sense_noise):
        """update_sense_noise (quadrotor.py:838-849): None -> bypass, "default" -> SensorNoise(), dict -> SensorNoise(**dict)."""
        if sense_noise is None:
            return None
        if isinstance(sense_noise, str):
            if sense_noise != "default":
                raise ValueError("ERROR: QuadEnv: sense_noise parameter is of unknown type: " + str(sense_noise))
            prm = dict(self.SENSE_DEFAULTS)
        elif isinstance(sense_noise, dict):
            unknown = set(sense_noise) - set(self.SENSE_DEFAULTS)
            if unknown:
                raise TypeError("__init__() got an unexpected keyword argument '%s'" % sorted(unknown)[0])
            prm = dict(self.SENSE_DEFAULTS, **sense_noise)
        else:
            raise ValueError("ERROR: QuadEnv: sense_noise parameter is of unknown type: " + str(sense_noise))
        if prm["bypass"]:
            return None
        return prm

    def _make_sampler(self, spec):
        if spec is None:
            return None
        spec = copy.deepcopy(spec)
        cls = spec.pop("class")
        return getattr(quad_rand, cls)(params=None, **spec)

    # ---- per-env parameters: host arrays, or read back from the device when the library samples them ---------------
    def _decide_device_randomizer(self, want, dynamics_params, s1, s2):
        why = None
        if not self._per_env:
            why = "the model is not randomised per env"
        elif dynamics_params not in ("Crazyflie", "DefaultQuad", "MediumQuad", "CrazyflieLowInertia", "RandomQuad"):
            why = "unknown base sampler %s" % dynamics_params
        elif dynamics_params == "RandomQuad" and (s1 is not None or s2 is not None or self.dynamics_change is not None):
            why = "RandomQuad is sampled on the device only without dynamics_change / further samplers"
        elif dynamics_params != "RandomQuad" and (s2 is not None or not isinstance(s1, dict) or s1.get("class") != "RelativeSampler"):
            why = "needs dyn_sampler_1 = RelativeSampler and no dyn_sampler_2"
        elif dynamics_params != "RandomQuad" and s1.get("sampler", "normal") not in ("normal", "uniform"):
            why = "unknown sampler %r" % (s1.get("sampler"),)
        elif self.dynamics_simplification:
            why = "dynamics_simplification needs the host pipeline"
        else:
            chg = (self.dynamics_change or {}).get("motor", {})
            if chg.get("C_drag", 0.) != 0. or chg.get("C_roll", 0.) != 0.:
                why = "rotor drag needs the generic kernel and the host pipeline"
        if want is True and why is not None:
            raise ValueError("randomize_on_device=True is not possible here: " + why)
        if want is None:
            want = self.num_envs > 1
        return bool(want) and why is None

    @property
    def models(self):
        """Derived constants per env (dict of [N] / [N,k] arrays, gaq_model's fields).  With the device randomizer they are
        read back from the GPU on demand (and cached until the parameters can have changed)."""
        if self._dev_rand and self._handle is not None:
            if self._models_cache is None:
                rows = np.empty((self.num_envs, _lib.MODEL_DOUBLES), dtype=np.float64)
                _lib.check(self._lib.gaq_get_params(self._handle, _lib.ptr(rows), 0, self.num_envs))
                self._models_cache = _lib.rows_to_models(rows)
            return self._models_cache
        return self._models_host

    @models.setter
    def models(self, v):
        self._models_host = v

    @property
    def models_extra(self):
        if self._dev_rand and self._handle is not None:
            if self._extra_cache is None:
                _, self._extra_cache = qp.derive_models(self.sampled_trees(), self.dynamics_simplification)
            return self._extra_cache
        return self._models_extra_host

    @models_extra.setter
    def models_extra(self, v):
        self._models_extra_host = v

    def sampled_trees(self):
        """Device randomizer: the parameter trees the envs currently fly with (batched tree, read back)."""
        rows = np.empty((self.num_envs, qp.TREE_DOUBLES), dtype=np.float64)
        _lib.check(self._lib.gaq_get_param_trees(self._handle, _lib.ptr(rows), 0, self.num_envs))
        return qp.unflatten_tree(rows, by_density=self._ctor_kwargs["dynamics_params"] == "RandomQuad")

    def _base_tree(self):
        tree = self.dyn_base_sampler.sample(1, rng=self._rng)
        if self.dynamics_change is not None:
            qp.update_tree(tree, qp.broadcast_tree(self.dynamics_change, 1))
        return tree

    def _resample_on_device(self, env_ids):
        """resample_dynamics with the sampler, QuadLink and update_model on the GPU (gaq_set_randomizer / gaq_randomize_dev)."""
        n = self.num_envs
        if self._handle is None:
            base = self._base_tree()
            if not qp.tree_is_flat_compatible(base):
                raise ValueError("randomize_on_device: the parameter tree is not of the shipped models' shape")
            models, extra = qp.derive_models(base, False)       # nominal model: noise / swarm settings of gaq_config
            self._models_host = {k: np.repeat(v, n, axis=0) for k, v in models.items()}
            self._models_extra_host = {k: np.repeat(v, n, axis=0) for k, v in extra.items()}
            self.dynamics_params_batched = base
            self.dynamics_params = qp.unbatch_tree(base, 0)
            dr, self._dev_rand = self._dev_rand, False          # (models property: host arrays while the handle is built)
            try:
                self._create_handle()
            finally:
                self._dev_rand = dr
            rz = _lib.GaqRandomizer()
            rz.every = int(self.dynamics_randomize_every or 0) if self._auto_reset else 0
            if self._ctor_kwargs["dynamics_params"] == "RandomQuad":
                rz.sampler = 2                          # randomquad_parameters per draw (quadrotor_randomization.py:142-243)
            else:
                spec = self._ctor_kwargs["dyn_sampler_1"]
                rz.sampler = 0 if spec.get("sampler", "normal") == "normal" else 1
                rz.ratio[:] = list(qp.ratio_rows(base, float(spec.get("noise_ratio", 0.)), spec.get("noise_ratio_custom"))[0])
            C.memmove(C.byref(rz.base), qp.flatten_tree(base)[0].ctypes.data, C.sizeof(rz.base))
            _lib.check(self._lib.gaq_set_randomizer(self._handle, C.byref(rz)))
            env_ids = None
        mask = None
        if env_ids is not None:
            m = np.zeros(n, dtype=np.uint8)
            m[np.asarray(env_ids, dtype=np.int64)] = 1
            import torch
            mask = torch.as_tensor(m, device=torch.device("cuda", self.device))
        _lib.check(self._lib.gaq_randomize_dev(self._handle, _lib.ptr(mask), None))
        _lib.check(self._lib.gaq_synchronize(self._handle))
        if mask is not None:
            import torch
            torch.cuda.synchronize(self.device)
        self._models_cache, self._extra_cache = None, None

    def _sample_params(self, n):
        """base sampler -> dynamics_change -> sampler 1 -> sampler 2 -> limits (quadrotor.py:1030-1053)."""
        tree = self.dyn_base_sampler.sample(n, rng=self._rng)
        if self.dynamics_change is not None:
            qp.update_tree(tree, qp.broadcast_tree(self.dynamics_change, n))
        if self._sampler_1 is not None:
            tree = self._sampler_1.sample(tree, rng=self._rng)
        if self._sampler_2 is not None:
            tree = self._sampler_2.sample(tree, rng=self._rng)
        return quad_rand.check_quad_param_limits(tree)

    def resample_dynamics(self, env_ids=None):
        """quadrotor.py:1030-1056.  MUST be followed by reset() (as in the reference).  With per-env
        randomisation `env_ids` restricts the resampling to those envs."""
        n = self.num_envs
        if self._dev_rand:
            return self._resample_on_device(env_ids)
        if self._handle is None or not self._per_env or env_ids is None:
            tree = self._sample_params(n if self._per_env else 1)
            models, extra = qp.derive_models(tree, self.dynamics_simplification)
            if not self._per_env:
                models = {k: np.repeat(v, n, axis=0) for k, v in models.items()}
                extra = {k: np.repeat(v, n, axis=0) for k, v in extra.items()}
                tree = qp.broadcast_tree(qp.unbatch_tree(tree, 0), n)
            self.dynamics_params_batched, self.models, self.models_extra = tree, models, extra
            first, rows = 0, _lib.models_to_rows(models)
        else:
            env_ids = np.asarray(env_ids, dtype=np.int64)
            tree = self._sample_params(len(env_ids))
            models, extra = qp.derive_models(tree, self.dynamics_simplification)
            for k, v in models.items():
                self.models[k][env_ids] = v
            for k, v in extra.items():
                self.models_extra[k][env_ids] = v
            self._dyn_params_cache = None               # (arrays edited in place: the info dict's constants are stale)
            first, rows = None, _lib.models_to_rows(models)
        self.dynamics_params = qp.unbatch_tree(self.dynamics_params_batched, 0)
        if self._handle is None:
            self._create_handle()
        if self._per_env:
            if first is not None:
                _lib.check(self._lib.gaq_set_params(self._handle, _lib.ptr(rows), 0, n))
            else:
                idx = np.ascontiguousarray(env_ids, dtype=np.int64)
                rows = np.ascontiguousarray(rows)        # (a named array: _lib.ptr holds no reference, a temporary would be gone before the call)
                _lib.check(self._lib.gaq_set_params_indexed(self._handle, _lib.ptr(rows), _lib.ptr(idx), len(idx)))

    def _create_handle(self):
        cfg = _lib.GaqConfig()
        cfg.struct_size = C.sizeof(_lib.GaqConfig)
        cfg.abi_version = _lib.ABI_VERSION
        cfg.num_envs = self.num_envs
        cfg.env_id_offset = self.env_id_offset
        cfg.device = self.device
        cfg.seed = self._seed_value
        cfg.sim_freq = float(self.sim_freq)
        cfg.sim_steps = int(self.sim_steps)
        cfg.ep_len = int(self.ep_len)
        cfg.room_size = float(self.room_size)
        cfg.gravity = float(self.gravity)
        cfg.t2w_std, cfg.t2t_std = float(self.t2w_std), float(self.t2t_std)
        if self.raw_control:
            cfg.control = _lib.CTRL_RAW_ZERO_MIDDLE if self.raw_control_zero_middle else _lib.CTRL_RAW
        else:
            cfg.control = _lib.CTRL_MELLINGER
        sigma_on = bool(np.any(self.models["ou_sigma"] != 0))
        noise = {"off": _lib.NOISE_OFF, "philox": _lib.NOISE_PHILOX, "input": _lib.NOISE_INPUT}[self._thrust_noise]
        if noise == _lib.NOISE_PHILOX and not sigma_on and not self._per_env:
            noise = _lib.NOISE_OFF           # thrust_noise_ratio == 0: the OU process is identically zero
        cfg.noise = noise
        cfg.reward_mode = _lib.REW_QUADROTOR if self._reward == "quadrotor" else _lib.REW_MULTI_LOG
        cfg.obs_flags = OBS_FLAGS[self.obs_repr]
        cfg.auto_reset = int(self._auto_reset)
        cfg.init_random_state = int(bool(self.init_random_state))
        cfg.resample_goal = int(bool(self.resample_goal))
        cfg.excite = int(self.excite)
        if self._swarm is not None:
            sw = self._swarm
            arm = float(np.max(self.models["arm"]))
            col = 2.0 * arm if sw["collision_dist"] is None else float(sw["collision_dist"])
            prox = 4.0 * col if sw["prox_dist"] is None else float(sw["prox_dist"])
            cfg.swarm.agents = int(sw["agents"])
            cfg.swarm.goal_radius, cfg.swarm.collision_dist, cfg.swarm.prox_dist = float(sw["goal_radius"]), col, prox
            cfg.swarm.w_collision, cfg.swarm.w_prox = float(sw["w_collision"]), float(sw["w_prox"])
            cfg.swarm.response = int(bool(sw["collision_response"]))
            self.swarm = dict(agents=int(sw["agents"]), goal_radius=float(sw["goal_radius"]), collision_dist=col,
                              prox_dist=prox, w_collision=float(sw["w_collision"]), w_prox=float(sw["w_prox"]),
                              collision_response=bool(sw["collision_response"]))
        cfg.per_env_params = int(self._per_env)
        cfg.compact_done = int(self._compact_done)
        # rotor drag / rolling moment need the generic kernel, which keeps plain fp64 state planes: with per-env parameters
        # the library only learns that when the parameters arrive, so the layout is decided here
        drag = bool(np.any(self.models["c_drag"] != 0) or np.any(self.models["c_roll"] != 0))
        cfg.obs_state_alias = 0 if (drag or self._alias_request is False) else 1 if self._alias_request is True else 2
        cfg.aux_outputs = int(self._info)
        cfg.action_f32 = int(self._action_f32)
        cfg.sense_input = int(self._sense_input)
        cfg.fp32_state = int(self.precision == "fp32")
        for k in ("pos", "effort", "crash", "orient", "yaw", "rot", "attitude", "spin", "action_change", "vel"):
            setattr(cfg.rew, k, self.rew_coeff[k])
        for k in ("pos_offset", "pos_log_weight", "pos_linear_weight"):
            setattr(cfg.rew, k, self.rew_coeff.get(k, 0.0))
        if self._sense is not None:
            cfg.sense.enabled = 1
            for k in ("pos_norm_std", "pos_unif_range", "vel_norm_std", "vel_unif_range", "quat_norm_std",
                      "quat_unif_range", "gyro_noise_density", "acc_static_noise_std", "acc_dynamic_noise_ratio",
                      "gyro_norm_std", "gyro_random_walk", "gyro_bias_correlation_time"):
                setattr(cfg.sense, k, float(self._sense[k]))
        cfg.model = _lib.row_to_model(_lib.models_to_rows(self.models)[0])
        h = C.c_void_p()
        _lib.check(self._lib.gaq_create(C.byref(cfg), C.byref(h)))
        self._handle = h
        self.obs_dim = self._lib.gaq_obs_dim(h)
        self.obs_is_state = bool(self._lib.gaq_obs_is_state(h))
        self.state_layout = int(self._lib.gaq_state_layout(h))     # 0 fp64 planes, 1 heads in the obs tensor, 2 library-owned heads
        self._noise_mode = noise

    # ------------------------------------------------------------------------------------------------
    def make_observation_space(self):
        """quadrotor.py:898-936: bounds by component name."""
        rb = self.room_box
        lim = {
            "xyz": [-(rb[1] - rb[0]), rb[1] - rb[0]], "xyzr": [-(rb[1] - rb[0]), rb[1] - rb[0]],
            "vxyz": [-3. * np.ones(3), 3. * np.ones(3)], "vxyzr": [-3. * np.ones(3), 3. * np.ones(3)],
            "acc": [-3. * GRAV * np.ones(3), 3. * GRAV * np.ones(3)], "R": [-np.ones(9), np.ones(9)],
            "omega": [-40. * np.ones(3), 40. * np.ones(3)], "h": [0. * np.ones(1), rb[1][2] * np.ones(1)],
            "act": [np.zeros(4), np.ones(4)],
            "t2w": [0. * np.ones(1), 5. * np.ones(1)], "t2t": [0. * np.ones(1), 1. * np.ones(1)],
            "quat": [-np.ones(4), np.ones(4)],
        }
        comps = self.obs_repr.split("_")
        low = np.concatenate([lim[c][0] for c in comps])
        high = np.concatenate([lim[c][1] for c in comps])
        if self._swarm is not None:   # (pos_j - pos_i, vel_j - vel_i) per neighbour
            k = int(self._swarm["agents"]) - 1
            nlo = np.concatenate([lim["xyz"][0], 2 * lim["vxyz"][0]])
            nhi = np.concatenate([lim["xyz"][1], 2 * lim["vxyz"][1]])
            low, high = np.concatenate([low] + [nlo] * k), np.concatenate([high] + [nhi] * k)
        self.obs_space_low_high = lim
        return Box(low, high, dtype=np.float32)

    def _make_action_space(self):
        if self.raw_control:   # RawControl.action_space (quadrotor_control.py:72-84)
            low = -np.ones(4) if self.raw_control_zero_middle else np.zeros(4)
            return Box(low, np.ones(4), dtype=np.float32)
        # NonlinearPositionController.action_space (quadrotor_control.py:514-522)
        t2w = float(np.asarray(self.models_extra["thrust_to_weight"]).reshape(-1)[0])
        c = 2 * np.pi
        return Box(np.array([-1.0, -5 * c, -5 * c, -c]), np.array([t2w - 1.0, 5 * c, 5 * c, c]), dtype=np.float32)

    def seed(self, seed=None):
        return self._seed(seed)

    def _seed(self, seed=None):
        """Reference: seeds only the reset-position stream (quadrotor.py:938-940).  Here: reseeds the host
        RNG used for parameter sampling; the device RNG seed is fixed at construction."""
        if seed is not None:
            self._rng = np.random.RandomState(int(seed) & 0x7FFFFFFF)
        return [seed]

    # ------------------------------------------------------------------------------------------------
    def _stream(self, like=None):
        if like is not None and _is_torch(like):
            import torch
            return C.c_void_p(torch.cuda.current_stream(like.device).cuda_stream)
        return None

    def reset(self, mask=None):
        """quadrotor.py:1149 -> _reset (:1059-1144).  `mask` ([N] bool/uint8) restricts the reset in batched
        mode.  Returns obs [obs_dim] (num_envs == 1) or [N, obs_dim]."""
        if self.dynamics_randomize_every is not None and self.num_envs == 1 and \
                (self.traj_count + 1) % self.dynamics_randomize_every == 0:
            self.resample_dynamics()                      # quadrotor.py:1063-1066
        obs = np.empty((self.num_envs, self.obs_dim), dtype=np.float32)
        m = None if mask is None else np.ascontiguousarray(np.asarray(mask).astype(np.uint8))
        self._c_reset(m, obs)
        self.tick = 0
        self.crashed = False
        self._actions = [np.zeros((self.num_envs, 4)), np.zeros((self.num_envs, 4))]
        return obs[0].astype(np.float64) if self.num_envs == 1 else obs

    # the two host-pointer calls of the Gym surface (one handle here; gym_art_amd/multi_device.py fans them out over its shards)
    def _c_reset(self, mask, obs):
        _lib.check(self._lib.gaq_reset(self._handle, _lib.ptr(mask), _lib.ptr(obs)))

    def _c_step(self, a, obs, rew, done):
        _lib.check(self._lib.gaq_step(self._handle, _lib.ptr(a), _lib.ptr(obs), _lib.ptr(rew), _lib.ptr(done)))

    def reset_dev(self, obs_out, mask=None):
        """Batched reset writing into a device tensor (torch, float32 [N, obs_dim]); asynchronous."""
        _lib.check(self._lib.gaq_reset_dev(self._handle, _lib.ptr(mask), _lib.ptr(obs_out), self._stream(obs_out)))
        self._obs_ref = obs_out
        return obs_out

    def step(self, action, out=None):
        """quadrotor.py:1155 -> _step (:942-1028).

        num_envs == 1: action [4] -> (obs [obs_dim] float64, reward float, done bool, info dict).
        num_envs  > 1: action [N,4] float32 (NumPy, or a torch tensor on this env's GPU) ->
                       (obs [N,obs_dim], reward [N], done [N], info); torch in -> torch out, zero copy;
                       `out=(obs, rew, done)` (torch path): write into these tensors instead of allocating three per call.
        """
        n = self.num_envs
        if _is_torch(action):
            import torch
            assert action.is_cuda and action.dtype == torch.float32 and tuple(action.shape) == (n, 4)
            a = action.contiguous()
            self._set_action_f32(True)
            if out is None:   # fresh tensors per call, like a Gym env hands out fresh arrays; `out=(obs, rew, done)` reuses the caller's
                obs = torch.empty((n, self.obs_dim), dtype=torch.float32, device=a.device)
                rew = torch.empty((n,), dtype=torch.float32, device=a.device)
                done = torch.empty((n,), dtype=torch.uint8, device=a.device)
            else:
                obs, rew, done = out
                assert tuple(obs.shape) == (n, self.obs_dim) and obs.dtype == torch.float32 and obs.is_contiguous()
                assert tuple(rew.shape) == (n,) and rew.dtype == torch.float32 and tuple(done.shape) == (n,) and done.dtype == torch.uint8
            self.step_dev(a, obs, rew, done)
            # per-episode re-randomisation: the device randomizer promoted the due envs inside the launch (step_dev dropped the
            # parameter caches) -- nothing to do here, and above all no read-back of `done`: that is a D2H copy and a stream
            # synchronisation per step.  Only the host-managed parameter path has to look at which envs finished.
            if self.dynamics_randomize_every is not None and self._per_env and not self._dev_rand and self._auto_reset:
                self._rerandomize_finished(self.done_indices() if self._compact_done else
                                           np.nonzero(done.cpu().numpy())[0])
            # (device tensor: rows valid where done == 1; no synchronisation)
            return obs, rew, done, ({"terminal_observation": self._term_buf} if self._term_buf is not None else {})
        # RawControl's arithmetic follows the dtype of the CALLER's array like the reference's does (quadrotor_control.py:
        # 88-92): a float32 array -> 0.5*(a+1) in float32; float64 (or a list) -> in float64.  The values travel as
        # float32 either way (the ABI's only action dtype).
        arr = np.asarray(action)
        self._set_action_f32(arr.dtype == np.float32)
        if n == 1 and self._out_ring == 0 and type(self)._c_step is QuadrotorEnv._c_step:
            # the single-env drop-in loop (BASELINE config 1): nothing of the call's arrays is handed out (the caller gets float64 copies /
            # scalars), so buffers and their pointers are made once -- four `ndarray.ctypes` objects per call cost 6 of a 43-us step
            one = self._one
            if one is None:
                bufs = (np.zeros((1, 4), np.float32), np.empty((1, self.obs_dim), np.float32), np.empty(1, np.float32), np.empty(1, np.uint8))
                one = self._one = bufs + tuple(_lib.ptr(b) for b in bufs)
            a, obs, rew, done = one[:4]
            a[0] = arr.reshape(4)
            _lib.check(self._lib.gaq_step(self._handle, *one[4:]))
            self.tick += 1
            a = a.copy()
            self._actions = [a.astype(np.float64), self._actions[0]]
            info = self._make_info_single(a, rew) if self._info else {}
            d = bool(done[0])
            self.traj_count += int(d)
            return obs[0].astype(np.float64), float(rew[0]), d, info
        a = np.ascontiguousarray(arr.astype(np.float32, copy=False).reshape(n, 4))
        if self._out_ring > 0:
            if self._ring is None:           # allocated AND touched once: the pages exist from here on
                self._ring = [self._host_arrays(n) for _ in range(self._out_ring)]
            obs, rew, done = self._ring[self._ring_pos]
            self._ring_pos = (self._ring_pos + 1) % self._out_ring
        else:
            obs = np.empty((n, self.obs_dim), dtype=np.float32)
            rew = np.empty((n,), dtype=np.float32)
            done = np.empty((n,), dtype=np.uint8)
        self._c_step(a, obs, rew, done)           # raises on NaN reward
        self.tick += 1
        # env.actions (quadrotor.py:943-944), float64 like the reference's; big batches that build no info dict keep the float32 array itself
        # (a 32-MB conversion per step at N = 2^20 otherwise) -- it is converted when somebody reads it (the `actions` property)
        # (`a` may BE the caller's array -- a contiguous float32 [n, 4] passes through the conversion above untouched -- and a sampler that
        #  refills one action buffer in place must not see env.actions[1], the PREVIOUS action, change under it: keep a copy (16 MB at 2^20))
        self._actions = [a.astype(np.float64) if (self._info or n <= 4096) else (a.copy() if a is arr or np.shares_memory(a, arr) else a),
                         self._actions[0]]
        info = (self._make_info_single(a, rew) if n == 1 else self._make_info(a, rew)) if self._info else {}
        if n == 1:
            d = bool(done[0])
            self.traj_count += int(d)
            return obs[0].astype(np.float64), float(rew[0]), d, info
        self.traj_count += int(done.sum())
        if self.dynamics_randomize_every is not None and self._auto_reset:
            self._rerandomize_finished(np.nonzero(done)[0])
        if self._term_buf is not None:      # fetched only on the steps that end episodes
            if done.any():
                info = dict(info)
                info["terminal_observation"] = self._term_buf.cpu().numpy()
        return obs, rew, done.view(np.bool_), info          # (0 / 1 bytes: a view, no copy)

    def _host_arrays(self, n):
        """One (obs, reward, done) set of the output ring.  Page-locked when torch can provide it (the arrays are NumPy views of pinned
        tensors, kept alive beside them): the device-to-host copies then go straight to the arrays at the link rate instead of through the
        runtime's staging buffers."""
        try:
            import torch
            ts = (torch.zeros((n, self.obs_dim), dtype=torch.float32, pin_memory=True), torch.zeros((n,), dtype=torch.float32, pin_memory=True),
                  torch.zeros((n,), dtype=torch.uint8, pin_memory=True))
            self._ring_pins = getattr(self, "_ring_pins", []) + [ts]
            return tuple(t.numpy() for t in ts)
        except Exception:       # no torch / no pinned allocator: pageable arrays
            return (np.zeros((n, self.obs_dim), dtype=np.float32), np.zeros((n,), dtype=np.float32), np.zeros((n,), dtype=np.uint8))

    def _set_action_f32(self, f32):
        if bool(f32) != self._action_f32:
            self._action_f32 = bool(f32)
            _lib.check(self._lib.gaq_set_action_dtype(self._handle, int(self._action_f32)))

    def _rerandomize_finished(self, finished):
        """dynamics_randomize_every in batched auto-reset mode (quadrotor.py:1063-1066 per env): an env whose
        NEXT episode index is a multiple of `dynamics_randomize_every` gets new parameters.  The in-kernel reset has
        already drawn its initial state, which does not depend on the model; like resample_dynamics the library
        clears that env's SVD counter and OU state."""
        if len(finished) == 0 or not self._per_env:
            return
        if self._dev_rand:          # the library re-randomises due envs right after the step launch
            self._models_cache, self._extra_cache = None, None
            return
        self._per_env_traj[finished] += 1
        due = finished[(self._per_env_traj[finished] + 1) % self.dynamics_randomize_every == 0]
        if len(due):
            self.resample_dynamics(env_ids=due)

    def _info_dyn_params(self):
        """info["dyn_params"] (quadrotor.py:1009-1025): constants of the current model(s); rebuilt only when the parameters change."""
        if self._dyn_params_cache is not None and self._dyn_params_cache[0] is self.models:
            return self._dyn_params_cache[1]
        n = self.num_envs
        sq = (lambda x: x[0]) if n == 1 else (lambda x: x)
        m = self.models
        dyn_params = {"mass": [sq(m["mass"])], "motor_linearity": [sq(m["linearity"])],
                      "motor_time_up": [sq(m["damp_time_up"])], "motor_time_down": [sq(m["damp_time_down"])],
                      "motor_assymetry": [sq(self.models_extra["motor_assymetry"])], "motor_pos": [sq(m["prop_pos"])],
                      "motor_ccw": [np.array([-1., 1., -1., 1.])], "t2w": [sq(self.models_extra["thrust_to_weight"])],
                      "t2t": [sq(self.models_extra["torque_to_thrust"])], "t2i": [sq(self.models_extra["torque_to_inertia"])],
                      "inertia": [sq(m["inertia"])],
                      "thrust_max": [sq(np.mean(m["thrust_max"], axis=1))], "torque_max": [sq(np.mean(m["torque_max"], axis=1))],
                      "arm": [sq(m["arm"])], "grav": [GRAV], "dt": [self.dt * self.sim_steps]}
        self._dyn_params_cache = (m, dyn_params)
        return dyn_params

    def _make_info_single(self, action, rew):
        """_make_info for num_envs == 1 (the drop-in loop, BASELINE config 1): the same entries from the same inputs, with scalar
        arithmetic instead of ~60 NumPy calls on 1-element arrays (the dict costs 25 us instead of 80 of a 116-us step)."""
        one = self.__dict__.get("_one_info")
        if one is None:        # persistent buffers + their pointers (the dict below hands out copies)
            bufs = (np.empty((_lib.STATE_PLANES, 1), dtype=np.float64), np.empty((1, _lib.AUX_WORDS), dtype=np.float32))
            one = self._one_info = bufs + tuple(_lib.ptr(b) for b in bufs)
        _lib.check(self._lib.gaq_get_state(self._handle, one[2]))
        _lib.check(self._lib.gaq_get_aux(self._handle, one[3]))
        st = one[0][:, 0].copy()
        aux = one[1][0].astype(np.float64)
        v = st.tolist()
        px, py, pz, vx, vy, vz = v[0:6]
        r00, r11, r22 = v[6], v[10], v[14]
        wx, wy, wz = v[15:18]
        gx, gy, gz = v[34:37]
        w = self.rew_coeff
        sqrt, acos = math.sqrt, math.acos
        dist = sqrt((gx - px) ** 2 + (gy - py) ** 2 + (gz - pz) ** 2)
        if self._reward == "quadrotor":
            cost_pos = w["pos"] * dist
        else:
            cost_pos = w["pos"] * (w["pos_log_weight"] * math.log(dist + w["pos_offset"]) + w["pos_linear_weight"] * dist)
        act = action[0].astype(np.float64)
        a, ap = act.tolist(), self.actions[1][0].tolist()
        clip1 = lambda x: -1. if x < -1. else (1. if x > 1. else x)
        raw = (("pos", dist, None), ("action", sqrt(a[0] ** 2 + a[1] ** 2 + a[2] ** 2 + a[3] ** 2), "effort"),
               ("crash", 1.0 if pz <= float(self.models["arm"][0]) else 0.0, "crash"), ("orient", -r22, "orient"), ("yaw", -r00, "yaw"),
               ("rot", acos(clip1(((r00 + r11 + r22) - 1.) / 2.)), "rot"), ("attitude", acos(clip1(r22)), "attitude"),
               ("spin", sqrt(wx * wx + wy * wy + wz * wz), "spin"),
               ("act_change", sqrt(sum((x - y) ** 2 for x, y in zip(a, ap))), "action_change"),
               ("vel", sqrt(vx * vx + vy * vy + vz * vz), "vel"))
        rewards = {"rew_main": -cost_pos, "rewraw_main": -dist}
        for k, val, wk in raw:
            rewards["rew_" + k] = -(cost_pos if wk is None else w[wk] * val)
            rewards["rewraw_" + k] = -val
        self.crashed = raw[2][1] > 0
        cmds = aux[13:17]
        obs_comp = {"xyz": [st[0:3]], "vxyz": [st[3:6]], "acc": [aux[0:3]], "omega": [st[15:18]], "omega_dot": [aux[3:6]],
                    "R": [st[6:15]], "act": [act], "act_clipped": [np.clip(aux[9:13], 0., 1.)], "act_filtered": [cmds],
                    "act_torque": [_CCW * cmds], "torque": [aux[6:9]]}
        return {"rewards": rewards, "obs_comp": obs_comp, "dyn_params": self._info_dyn_params()}

    def _make_info(self, action, rew):
        """The reference's info dict (quadrotor.py:993-1028) from the device state and the kernel's aux row (the last
        sub-step's accelerometer / omega_dot / torque, the controller output and thrust_cmds_damp).  Values follow
        compute_reward_weighted (:544-638 / quadrotor_multi.py:550-650)."""
        st = self.get_state()
        n = self.num_envs
        pos, vel, omega = st[0:3].T, st[3:6].T, st[15:18].T
        rot = st[6:15].T.reshape(n, 3, 3)
        goal = st[34:37].T
        w = self.rew_coeff
        dist = np.linalg.norm(goal - pos, axis=1)
        if self._reward == "quadrotor":
            cost_pos = w["pos"] * dist
        else:
            cost_pos = w["pos"] * (w["pos_log_weight"] * np.log(dist + w["pos_offset"]) + w["pos_linear_weight"] * dist)
        act, act_prev = action.astype(np.float64), self.actions[1]
        raw = {
            "pos": dist, "action": np.linalg.norm(act, axis=1),
            "crash": (pos[:, 2] <= self.models["arm"]).astype(np.float64), "orient": -rot[:, 2, 2], "yaw": -rot[:, 0, 0],
            "rot": np.arccos(np.clip(((rot[:, 0, 0] + rot[:, 1, 1] + rot[:, 2, 2]) - 1.) / 2., -1., 1.)),
            "attitude": np.arccos(np.clip(rot[:, 2, 2], -1., 1.)), "spin": np.linalg.norm(omega, axis=1),
            "act_change": np.linalg.norm(act - act_prev, axis=1), "vel": np.linalg.norm(vel, axis=1)}
        wkey = {"pos": None, "action": "effort", "crash": "crash", "orient": "orient", "yaw": "yaw", "rot": "rot",
                "attitude": "attitude", "spin": "spin", "act_change": "action_change", "vel": "vel"}
        sq = (lambda x: x[0]) if n == 1 else (lambda x: x)
        rewards = {"rew_main": sq(-cost_pos), "rewraw_main": sq(-dist)}
        for k, v in raw.items():
            cost = cost_pos if k == "pos" else w[wkey[k]] * v
            rewards["rew_" + k] = sq(-cost)
            rewards["rewraw_" + k] = sq(-v)
        self.crashed = sq(raw["crash"] > 0)
        aux = np.empty((n, _lib.AUX_WORDS), dtype=np.float32)          # last sub-step's acc / omega_dot / torque, controller
        _lib.check(self._lib.gaq_get_aux(self._handle, _lib.ptr(aux)))   # output, thrust_cmds_damp (gaq_config.aux_outputs)
        aux = aux.astype(np.float64)
        cmds = aux[:, 13:17]
        obs_comp = {"xyz": [sq(pos)], "vxyz": [sq(vel)], "acc": [sq(aux[:, 0:3])], "omega": [sq(omega)],
                    "omega_dot": [sq(aux[:, 3:6])], "R": [sq(rot.reshape(n, 9))], "act": [sq(act)],
                    "act_clipped": [sq(np.clip(aux[:, 9:13], 0., 1.))], "act_filtered": [sq(cmds)],
                    "act_torque": [sq(np.array([-1., 1., -1., 1.])[None] * cmds)], "torque": [sq(aux[:, 6:9])]}
        dyn_params = self._info_dyn_params()
        return {"rewards": rewards, "obs_comp": obs_comp, "dyn_params": dyn_params}

    def step_dev(self, actions, obs, rew, done, stream=None):
        """Asynchronous device-pointer step (gaq_step_dev) on the tensors' current torch stream.  With alias_obs=True
        `obs` is also the next step's INPUT (the state's fp32 head): keep it alive and unmodified until then."""
        st = self._stream(actions) if stream is None else C.c_void_p(stream)
        _lib.check(self._lib.gaq_step_dev(self._handle, _lib.ptr(actions), _lib.ptr(obs), _lib.ptr(rew), _lib.ptr(done), st))
        self._obs_ref = obs
        if self._dev_rand and self.dynamics_randomize_every:
            self._models_cache, self._extra_cache = None, None

    def bind_step(self, actions, obs, rew, done, stream=None):
        """step_dev with the pointer and stream look-ups done ONCE: returns a zero-argument callable that launches the step on
        these (persistent) device tensors.  At small batches a step is a ~9 us kernel and the per-call Python work of
        step_dev (torch stream query, four data_ptr conversions: ~6 us) is what bounds an eager loop; the bound call is one
        ctypes call.  The tensors must stay alive and on the same stream; results are those of step_dev."""
        st = self._stream(actions) if stream is None else C.c_void_p(stream)
        fn, h = self._lib.gaq_step_dev, self._handle
        pa, po, pr, pd = _lib.ptr(actions), _lib.ptr(obs), _lib.ptr(rew), _lib.ptr(done)
        keep = (actions, obs, rew, done)
        rand = self._dev_rand and bool(self.dynamics_randomize_every)

        def bound():
            rc = fn(h, pa, po, pr, pd, st)
            if rc:
                _lib.check(rc)
            self._obs_ref = keep[1]
            if rand:
                self._models_cache, self._extra_cache = None, None
        return bound

    def step_many_dev(self, actions, obs, rew, done, stream=None):
        """T fused-API steps: actions [T,N,4] -> obs [T,N,D], rew [T,N], done [T,N] (device tensors)."""
        T = int(actions.shape[0])
        st = self._stream(actions) if stream is None else C.c_void_p(stream)
        _lib.check(self._lib.gaq_step_many_dev(self._handle, T, _lib.ptr(actions), _lib.ptr(obs), _lib.ptr(rew),
                                               _lib.ptr(done), st))
        self._obs_ref = obs

    def pack_rows_dev(self, obs, rew, done, rows, stream=None):
        """rows[i] = [obs[i], reward[i], float(done[i])] ([N, obs_dim + 2] float32 device tensor): the multi-GPU return
        path's single-collective row (gaq_pack_rows_dev)."""
        st = self._stream(obs) if stream is None else C.c_void_p(stream)
        _lib.check(self._lib.gaq_pack_rows_dev(self._handle, _lib.ptr(obs), _lib.ptr(rew), _lib.ptr(done), _lib.ptr(rows), st))

    def set_packed_rows(self, rows):
        """Register a device tensor [N, obs_dim + 2] (float32): every following step_dev launch ALSO writes the packed rows
        [obs | reward | float(done)] of its outputs into it, from inside the step kernel (gaq_set_packed_rows_dev) -- the multi-GPU
        return path's single-collective row without the pack launch between the step and the gather.  None unregisters."""
        if rows is not None:
            assert tuple(rows.shape) == (self.num_envs, self.obs_dim + 2) and rows.is_contiguous()
        self._rows_ref = rows
        _lib.check(self._lib.gaq_set_packed_rows_dev(self._handle, _lib.ptr(rows)))

    @property
    def kernel_variant(self):
        """Feature mask of the step kernel this env launches (csrc/quad_core.hpp: enum Feature; gaq_kernel_variant)."""
        return int(self._lib.gaq_kernel_variant(self._handle))

    @property
    def launch_variant(self):
        """... and of the instantiation the next step launches: that kernel or its F_ROWS / F_CTR twin (gaq_launch_variant)."""
        return int(self._lib.gaq_launch_variant(self._handle))

    def set_sense_input(self, draws_dev):
        """sense_noise_input=True: the standard draws of the next step's three add_noise calls, device float32
        [3, 10, 3, N] (include/gaq.h gaq_set_sense_input_dev)."""
        self._sense_ref = draws_dev
        _lib.check(self._lib.gaq_set_sense_input_dev(self._handle, _lib.ptr(draws_dev)))

    def set_noise_input(self, normals_dev):
        """thrust_noise='input': normals for the next step, device float32 [sim_steps, 4, N]."""
        _lib.check(self._lib.gaq_set_noise_input_dev(self._handle, _lib.ptr(normals_dev)))

    def _raise_on_nan(self):
        cnt = C.c_int64(0)
        _lib.check(self._lib.gaq_nan_count(self._handle, C.byref(cnt)))
        if cnt.value:
            raise ValueError('QuadEnv: reward is Nan')      # quadrotor.py:633-636

    def check_finite(self):
        """Batched device mode: raise like the reference if any reward since the last check was non-finite."""
        self._raise_on_nan()

    # ------------------------------------------------------------------------------------------------
    def get_state(self):
        """Device state as [42, N] float64 planes (layout: include/gaq.h GAQ_STATE_PLANES)."""
        st = np.empty((_lib.STATE_PLANES, self.num_envs), dtype=np.float64)
        _lib.check(self._lib.gaq_get_state(self._handle, _lib.ptr(st)))
        return st

    def set_state(self, planes):
        st = np.ascontiguousarray(planes, dtype=np.float64)
        assert st.shape == (_lib.STATE_PLANES, self.num_envs)
        _lib.check(self._lib.gaq_set_state(self._handle, _lib.ptr(st)))

    def observe(self):
        obs = np.empty((self.num_envs, self.obs_dim), dtype=np.float32)
        _lib.check(self._lib.gaq_observe(self._handle, _lib.ptr(obs)))
        return obs[0].astype(np.float64) if self.num_envs == 1 else obs

    def done_indices(self):
        cap = self.num_envs
        idx = np.empty(cap, dtype=np.uint32)
        cnt = C.c_int64(0)
        _lib.check(self._lib.gaq_done_list(self._handle, _lib.ptr(idx), cap, C.byref(cnt)))
        return np.sort(idx[:cnt.value])

    def set_terminal_obs(self, term_obs):
        """Register a device tensor [N, obs_dim] that receives the last observation of every episode that ends
        (and is auto-reset) in a step; None unregisters."""
        self._term_ref = term_obs
        _lib.check(self._lib.gaq_set_terminal_obs_dev(self._handle, _lib.ptr(term_obs)))

    def track_episodes(self, enabled=True):
        _lib.check(self._lib.gaq_track_episodes(self._handle, int(enabled)))

    def episode_stats(self, clear=True):
        """Episodes finished since the last clear: dict(episodes, mean_return, std_return, mean_length)."""
        n, sr, sl, sq = C.c_int64(0), C.c_double(0), C.c_double(0), C.c_double(0)
        _lib.check(self._lib.gaq_episode_stats(self._handle, C.byref(n), C.byref(sr), C.byref(sl), C.byref(sq), int(clear)))
        k = max(n.value, 1)
        mean = sr.value / k
        return dict(episodes=n.value, mean_return=mean, std_return=max(sq.value / k - mean * mean, 0.0) ** 0.5,
                    mean_length=sl.value / k)

    def set_graph_safe(self, enabled=True):
        """Keep the step index (RNG key of noise and resets) in device memory so that step_dev / step_many_dev /
        reset_dev can be captured in a HIP graph (torch.cuda.graph) and draw fresh randomness on every replay.
        In the alias layout capture with one observation tensor used in place."""
        _lib.check(self._lib.gaq_set_graph_safe(self._handle, int(enabled)))

    def set_timing(self, enabled=True):
        _lib.check(self._lib.gaq_set_timing(self._handle, int(enabled)))

    def last_kernel_ms(self):
        ms = C.c_float(0)
        _lib.check(self._lib.gaq_last_kernel_ms(self._handle, C.byref(ms)))
        return ms.value

    def synchronize(self):
        _lib.check(self._lib.gaq_synchronize(self._handle))

    def render(self, mode='human', **kwargs):
        raise NotImplementedError("rendering (pyglet scene, quadrotor_visualization.py) is out of scope")

    def close(self):
        if getattr(self, "_handle", None) is not None:
            self._lib.gaq_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # pickling by constructor arguments, like gym.utils.EzPickle (quadrotor.py:647,688)
    # ---- checkpoint / resume (no reference counterpart: the reference pickles its constructor arguments only) ----------------
    def state_dict(self):
        """Everything a bit-exact continuation needs, as NumPy arrays / plain values: the state planes, the counters behind the
        RNG keys, the per-env episode / resample counts, the per-env models when they are managed on the host, and the Python-side
        bookkeeping.  `QuadrotorEnv(**same_kwargs).load_state_dict(d)` continues as if nothing had happened -- same thrust noise,
        same in-kernel resets, same re-randomised parameters (tests: test_checkpoint_resume_is_bit_exact)."""
        cnt = _lib.GaqCounters()
        traj = rc = None
        if self._per_env:
            traj = np.empty(self.num_envs, dtype=np.uint32)
            rc = np.empty(self.num_envs, dtype=np.uint32)
        _lib.check(self._lib.gaq_get_counters(self._handle, C.byref(cnt), _lib.ptr(traj), _lib.ptr(rc)))
        d = {"format": 1, "num_envs": self.num_envs, "state": self.get_state(), "step_index": int(cnt.step_index),
             "reset_calls": int(cnt.reset_calls), "episodes": traj, "resamples": rc, "tick": self.tick,
             "traj_count": self.traj_count, "actions": [a.copy() for a in self.actions], "crashed": copy.deepcopy(self.crashed),
             "per_env_traj": self._per_env_traj.copy(), "host_rng": self._rng.get_state(), "action_f32": self._action_f32}
        if self._per_env and not self._dev_rand:
            d["models"] = {k: v.copy() for k, v in self.models.items()}
            d["models_extra"] = {k: v.copy() for k, v in self.models_extra.items()}
            d["dynamics_params_batched"] = copy.deepcopy(self.dynamics_params_batched)
        return d

    def load_state_dict(self, d):
        """Inverse of state_dict() on an env built with the same constructor arguments."""
        if int(d.get("format", 0)) != 1 or int(d["num_envs"]) != self.num_envs:
            raise ValueError("state_dict of another format / batch size")
        if "models" in d:
            if not self._per_env or self._dev_rand:
                raise ValueError("state_dict holds host-managed per-env models; this env does not")
            self.models = {k: np.array(v) for k, v in d["models"].items()}
            self.models_extra = {k: np.array(v) for k, v in d["models_extra"].items()}
            self.dynamics_params_batched = copy.deepcopy(d["dynamics_params_batched"])
            self.dynamics_params = qp.unbatch_tree(self.dynamics_params_batched, 0)
            rows = _lib.models_to_rows(self.models)
            _lib.check(self._lib.gaq_set_params(self._handle, _lib.ptr(rows), 0, self.num_envs))
            self._dyn_params_cache = None
        cnt = _lib.GaqCounters()
        cnt.step_index, cnt.reset_calls = int(d["step_index"]), int(d["reset_calls"])
        traj = None if d["episodes"] is None else np.ascontiguousarray(d["episodes"], dtype=np.uint32)
        rc = None if d["resamples"] is None else np.ascontiguousarray(d["resamples"], dtype=np.uint32)
        _lib.check(self._lib.gaq_set_counters(self._handle, C.byref(cnt), _lib.ptr(traj), _lib.ptr(rc)))
        self.set_state(d["state"])          # after the parameters: gaq_set_params clears SVD counter / OU state, the planes restore them
        self._models_cache, self._extra_cache = None, None
        self.tick, self.traj_count = d["tick"], d["traj_count"]
        self.actions = [np.array(a) for a in d["actions"]]
        self.crashed = copy.deepcopy(d["crashed"])
        self._per_env_traj = np.array(d["per_env_traj"])
        self._rng.set_state(d["host_rng"])
        self._set_action_f32(bool(d["action_f32"]))
        return self

    def __getstate__(self):
        return {"_ctor_kwargs": self._ctor_kwargs}

    def __setstate__(self, d):
        self.__init__(**d["_ctor_kwargs"])
