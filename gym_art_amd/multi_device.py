"""One QuadrotorEnv over several GPUs in ONE process: `QuadrotorEnv(num_envs=N, device_ids=[0, 1, ..., 7])`.

SURVEY 8b / 8e ("single process, one stream per device"; BASELINE config 4): the reference's loops `reset(); while not done: step()`
(quadrotor.py:1278-1305, :1424-1428) run unchanged -- only the constructor call names the devices.  The batch is cut into contiguous
shards of whole 64-env tiles (whole swarm worlds), shard k is an ordinary single-device env on device_ids[k] with `env_id_offset` = its
first global index (the random streams are keyed by the global env index: results equal those of one handle holding all N envs, bit for
bit), and every step()/reset()/step_dev()/reset_dev() is ONE call into libgaq's sharded handle (include/gaq.h `gaq_sharded`), which fans
out over the shards' own HIP streams and lands the stacked obs / reward / done in the caller's arrays / tensors on device_ids[0].

How it is put together: QuadrotorEnv.__init__ swaps the instance's class for `multi_device_class(type(self))` = (this mixin, the user's
class) before anything touches a device, so whatever the user's class is (QuadrotorEnv, the fork's QuadrotorEnv, QuadrotorEnvMulti) keeps
its own behaviour and only the handle-facing methods below are replaced.  torch.distributed is not involved; for one process PER GPU see
gym_art_amd/sharding.py.
"""
import copy
import ctypes as C

import numpy as np

from . import _lib

_CLASSES = {}


def multi_device_class(cls):
    """The (mixin, cls) class of a multi-device instance of `cls` (cached; pickles as `cls` + constructor arguments)."""
    if issubclass(cls, _MultiDeviceMixin):
        return cls
    if cls not in _CLASSES:
        _CLASSES[cls] = type(cls.__name__ + "OnDevices", (_MultiDeviceMixin, cls), {"_user_class": cls, "__doc__": cls.__doc__})
    return _CLASSES[cls]


def _rebuild(cls, kwargs):
    return cls(**kwargs)


def shard_ranges(n, num_shards, align=1):
    """[(first, count)] of gaq_shard_range for every shard (pure host arithmetic; empty tail shards included)."""
    lib = _lib.load()
    f, c = C.c_int64(0), C.c_int64(0)
    out = []
    for k in range(int(num_shards)):
        _lib.check(lib.gaq_shard_range(int(n), int(num_shards), k, int(align), C.byref(f), C.byref(c)))
        out.append((f.value, c.value))
    return out


class _MultiDeviceMixin(object):
    """Handle-facing methods of QuadrotorEnv for an env whose batch lives on several devices (see the module docstring)."""

    _NOT_HERE = ("%s is a per-device facility: use it on the shards (env.shards[k]), each an ordinary single-device QuadrotorEnv over "
                 "the global envs [env.shard_ranges[k][0], +env.shard_ranges[k][1])")

    # ---- construction: called by QuadrotorEnv.__init__ where a single-device env creates its handle ----------------------------------
    def resample_dynamics(self, env_ids=None):
        if getattr(self, "_sharded", None) is None:
            return self._build_shards()
        if env_ids is None:
            for sh in self.shards:
                sh.resample_dynamics()
        else:
            ids = np.asarray(env_ids, dtype=np.int64)
            for sh, (f, c) in zip(self.shards, self.shard_ranges):
                mine = ids[(ids >= f) & (ids < f + c)] - f
                if len(mine):
                    sh.resample_dynamics(env_ids=mine)
        self._dyn_params_cache = None

    def _build_shards(self):
        if self._info:
            raise NotImplementedError("info=True (the per-step info dict) is not built for device_ids with several devices: read the "
                                      "shards' state instead (env.get_state(), env.shards[k])")
        if self._terminal_observation:
            raise NotImplementedError("terminal_observation=True is not built for device_ids with several devices")
        # the shards are instances of the BASE class: its constructor signature is what self._ctor_kwargs holds at this point (a
        # subclass -- the fork's env, QuadrotorEnvMulti -- translates its own arguments before it calls the base constructor, and puts
        # its own arguments into _ctor_kwargs when that returns)
        base = _base_class()
        kw = copy.deepcopy(self._ctor_kwargs)
        align = int(self._swarm["agents"]) if self._swarm is not None else 1
        spans = shard_ranges(self.num_envs, len(self.device_ids), align)
        self.shards, self.shard_ranges, self.shard_devices = [], [], []
        for k, ((f, c), d) in enumerate(zip(spans, self.device_ids)):
            if c == 0:
                continue             # fewer tiles than devices: the tail devices stay idle
            kw_k = dict(kw, num_envs=c, env_id_offset=self.env_id_offset + f, device=d, device_ids=None, seed=self._seed_value,
                        host_seed=(self._seed_value + 7919 * (k + 1)) & 0x7FFFFFFF, auto_reset=self._auto_reset, info=False,
                        randomize_on_device=self._dev_rand, terminal_observation=False, out_ring=0)
            sh = base(**kw_k)
            # (a shard's own constructor ended with a reset(), like every env's; this env's constructor is about to reset them all: wind
            #  the shard's reset-call counter -- a key of the reset streams -- back, so that the batch's draws are those of ONE env's)
            cnt = _lib.GaqCounters()
            cnt.step_index, cnt.reset_calls = 0, 0
            _lib.check(self._lib.gaq_set_counters(sh._handle, C.byref(cnt), None, None))
            if self._sense is not None and float(self._sense.get("gyro_norm_std", 0.0)) != 0.0:
                st = sh.get_state()          # ... and the gyro-bias walk, the one piece of state that survives resets (sensor_noise.py:98)
                st[39:42] = 0.0
                sh.set_state(st)
            self.shards.append(sh)
            self.shard_ranges.append((f, c))
            self.shard_devices.append(d)
        hs = (C.c_void_p * len(self.shards))(*[sh._handle for sh in self.shards])
        h = C.c_void_p()
        _lib.check(self._lib.gaq_sharded_from_handles(hs, len(self.shards), C.byref(h)))
        self._sharded = h
        s0 = self.shards[0]
        self.obs_dim, self.obs_is_state, self.state_layout = s0.obs_dim, s0.obs_is_state, s0.state_layout
        self.dynamics_params, self.dynamics_params_batched = s0.dynamics_params, None
        if self._swarm is not None:
            self.swarm = s0.swarm
        self._handle = None

    # ---- the Gym surface's two host-pointer calls and their device-tensor forms ------------------------------------------------------
    def _c_reset(self, mask, obs):
        _lib.check(self._lib.gaq_reset_sharded(self._sharded, _lib.ptr(mask), _lib.ptr(obs)))

    def _c_step(self, a, obs, rew, done):
        _lib.check(self._lib.gaq_step_sharded(self._sharded, _lib.ptr(a), _lib.ptr(obs), _lib.ptr(rew), _lib.ptr(done)))

    def reset_dev(self, obs_out, mask=None):
        """Batched reset into a device tensor [N, obs_dim] on device_ids[0]; asynchronous on its current stream."""
        _lib.check(self._lib.gaq_reset_sharded_dev(self._sharded, _lib.ptr(mask), _lib.ptr(obs_out), self._stream(obs_out)))
        self._obs_ref = obs_out
        return obs_out

    def step_dev(self, actions, obs, rew, done, stream=None):
        """Asynchronous step of the whole batch: tensors on device_ids[0], [N, 4] -> [N, obs_dim], [N], [N] (gaq_step_sharded_dev)."""
        st = self._stream(actions) if stream is None else C.c_void_p(stream)
        _lib.check(self._lib.gaq_step_sharded_dev(self._sharded, _lib.ptr(actions), _lib.ptr(obs), _lib.ptr(rew), _lib.ptr(done), st))
        self._obs_ref = obs
        if self._dev_rand and self.dynamics_randomize_every:
            for sh in self.shards:
                sh._models_cache, sh._extra_cache = None, None

    def bind_step(self, actions, obs, rew, done, stream=None):
        st = self._stream(actions) if stream is None else C.c_void_p(stream)
        fn, h = self._lib.gaq_step_sharded_dev, self._sharded
        pa, po, pr, pd = _lib.ptr(actions), _lib.ptr(obs), _lib.ptr(rew), _lib.ptr(done)
        keep = (actions, obs, rew, done)
        rand = self._dev_rand and bool(self.dynamics_randomize_every)

        def bound():
            rc = fn(h, pa, po, pr, pd, st)
            if rc:
                _lib.check(rc)
            self._obs_ref = keep[1]
            if rand:
                for sh in self.shards:
                    sh._models_cache, sh._extra_cache = None, None
        return bound

    def _set_action_f32(self, f32):
        self._action_f32 = bool(f32)
        for sh in self.shards:
            sh._set_action_f32(f32)

    # ---- state exchange, parameters, bookkeeping: stacked over the shards --------------------------------------------------------------
    def _cat(self, parts, axis=0):
        return np.concatenate(parts, axis=axis)

    @property
    def models(self):
        ms = [sh.models for sh in self.shards]
        return {k: self._cat([m[k] for m in ms]) for k in ms[0]}

    @models.setter
    def models(self, v):
        raise AttributeError(self._NOT_HERE % "assigning models")

    @property
    def models_extra(self):
        ms = [sh.models_extra for sh in self.shards]
        return {k: self._cat([m[k] for m in ms]) for k in ms[0]}

    @models_extra.setter
    def models_extra(self, v):
        raise AttributeError(self._NOT_HERE % "assigning models_extra")

    def sampled_trees(self):
        from . import quad_params as qp
        rows = self._cat([qp.flatten_tree(sh.sampled_trees()) for sh in self.shards])
        return qp.unflatten_tree(rows, by_density=self._ctor_kwargs["dynamics_params"] == "RandomQuad")

    @property
    def goal(self):
        if not (self.resample_goal or self.excite or self._swarm) or getattr(self, "_sharded", None) is None:
            return np.array([0., 0., 2.])
        return self.get_state()[34:37].T.copy()

    def get_state(self):
        return self._cat([sh.get_state() for sh in self.shards], axis=1)

    def set_state(self, planes):
        st = np.ascontiguousarray(planes, dtype=np.float64)
        assert st.shape == (_lib.STATE_PLANES, self.num_envs)
        for sh, (f, c) in zip(self.shards, self.shard_ranges):
            sh.set_state(st[:, f:f + c])

    def observe(self):
        return self._cat([sh.observe().reshape(c, self.obs_dim) for sh, (_, c) in zip(self.shards, self.shard_ranges)])

    def done_indices(self):
        return self._cat([sh.done_indices().astype(np.int64) + f for sh, (f, _) in zip(self.shards, self.shard_ranges)])

    def _raise_on_nan(self):
        err = None
        for sh in self.shards:             # every shard's counter is read (and cleared) before anything is raised
            try:
                sh._raise_on_nan()
            except ValueError as e:
                err = e
        if err is not None:
            raise err

    def track_episodes(self, enabled=True):
        for sh in self.shards:
            sh.track_episodes(enabled)

    def episode_stats(self, clear=True):
        n = sr = sl = sq = 0.0
        for sh in self.shards:
            a, b, c, d = C.c_int64(0), C.c_double(0), C.c_double(0), C.c_double(0)
            _lib.check(self._lib.gaq_episode_stats(sh._handle, C.byref(a), C.byref(b), C.byref(c), C.byref(d), int(clear)))
            n, sr, sl, sq = n + a.value, sr + b.value, sl + c.value, sq + d.value
        k = max(n, 1)
        mean = sr / k
        return dict(episodes=int(n), mean_return=mean, std_return=max(sq / k - mean * mean, 0.0) ** 0.5, mean_length=sl / k)

    def set_timing(self, enabled=True):
        for sh in self.shards:
            sh.set_timing(enabled)

    def last_kernel_ms(self):
        return max(sh.last_kernel_ms() for sh in self.shards)      # the shards run side by side

    def synchronize(self):
        _lib.check(self._lib.gaq_synchronize_sharded(self._sharded))

    @property
    def kernel_variant(self):
        return self.shards[0].kernel_variant

    @property
    def launch_variant(self):
        return self.shards[0].launch_variant

    # ---- per-device facilities ----------------------------------------------------------------------------------------------------------
    def _not_here(self, what):
        raise NotImplementedError(self._NOT_HERE % what)

    def step_many_dev(self, *a, **k):
        self._not_here("step_many_dev (fused rollouts)")

    def pack_rows_dev(self, *a, **k):
        self._not_here("pack_rows_dev")

    def set_packed_rows(self, *a, **k):
        self._not_here("set_packed_rows")

    def set_sense_input(self, *a, **k):
        self._not_here("set_sense_input")

    def set_noise_input(self, *a, **k):
        self._not_here("set_noise_input")

    def set_terminal_obs(self, *a, **k):
        self._not_here("set_terminal_obs")

    def set_graph_safe(self, *a, **k):
        self._not_here("set_graph_safe (a HIP graph is captured on one device)")

    # ---- lifetime, checkpoints, pickling ------------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_sharded", None) is not None:
            self._lib.gaq_destroy_sharded(self._sharded)        # before the shard handles it borrows
            self._sharded = None
        for sh in getattr(self, "shards", []):
            sh.close()

    def state_dict(self):
        return {"format": 1, "num_envs": self.num_envs, "multi_device": True, "shard_ranges": list(self.shard_ranges),
                "shards": [sh.state_dict() for sh in self.shards], "tick": self.tick, "traj_count": self.traj_count,
                "actions": [a.copy() for a in self.actions], "crashed": copy.deepcopy(self.crashed),
                "per_env_traj": self._per_env_traj.copy(), "host_rng": self._rng.get_state(), "action_f32": self._action_f32}

    def load_state_dict(self, d):
        if not d.get("multi_device") or int(d["num_envs"]) != self.num_envs or [tuple(x) for x in d["shard_ranges"]] != list(self.shard_ranges):
            raise ValueError("state_dict of another batch size / another split over devices")
        for sh, sd in zip(self.shards, d["shards"]):
            sh.load_state_dict(sd)
        self.tick, self.traj_count = d["tick"], d["traj_count"]
        self.actions = [np.array(a) for a in d["actions"]]
        self.crashed = copy.deepcopy(d["crashed"])
        self._per_env_traj = np.array(d["per_env_traj"])
        self._rng.set_state(d["host_rng"])
        self._set_action_f32(bool(d["action_f32"]))
        return self

    def __reduce__(self):
        return (_rebuild, (self._user_class, copy.deepcopy(self._ctor_kwargs)))


def _base_class():
    from .quadrotor import QuadrotorEnv
    return QuadrotorEnv
