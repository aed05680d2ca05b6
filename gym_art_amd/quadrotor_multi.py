"""Swarm environment for BASELINE config 5 ("8-agent swarm x 131072 worlds with neighbour-distance reward").

The reference snapshot has NO multi-agent code: gym_art/quadrotor_multi/quadrotor_multi.py is a single-agent fork of
quadrotor.py whose only change on this path is the log-distance position cost (:554).  Everything that makes a swarm --
worlds of several agents, the neighbour terms of reward and observation, formation goals -- is this build's OWN
specification (DESIGN.md "Swarm layer"); it is parity-unpinned and labelled so wherever it is reported.  What IS pinned:
per-agent dynamics (the same fused kernel, fixtures G2-G6) and the per-agent quadrotor_multi reward (fixture G7).

Layout: the batch holds num_worlds * num_agents envs; agent a of world w is env w * num_agents + a.  num_agents is a
power of two <= 16 (what gaq_create accepts: 16 agents' observation rows need 110 KB of the CU's 160 KB LDS), so a world never straddles a 64-env wave tile and neighbour exchange is a wave shuffle inside the
step kernel.  step()/reset() keep QuadrotorEnv's flat batched signature; `worlds(x)` reshapes [N, ...] -> [W, A, ...].
"""
import copy

from .quadrotor import GRAV
from .quadrotor import QuadrotorEnv as _QuadrotorEnv

# the keys of the fork's info["rewards"] (quadrotor_multi.py:627-640): no `rewraw_*` entries and no `rew_act_change`
_FORK_REWARD_KEYS = ("rew_main", "rew_pos", "rew_action", "rew_crash", "rew_orient", "rew_yaw", "rew_rot", "rew_attitude", "rew_spin",
                     "rew_vel")


class QuadrotorEnv(_QuadrotorEnv):
    """Drop-in for the fork's own `QuadrotorEnv` (gym_art/quadrotor_multi/quadrotor_multi.py:659-843): a script that does
    `from gym_art.quadrotor_multi.quadrotor_multi import QuadrotorEnv` switches to `from gym_art_amd.quadrotor_multi import
    QuadrotorEnv` with nothing else changed.  What the fork changes against gym_art/quadrotor/quadrotor.py, and therefore what this
    class changes against gym_art_amd.QuadrotorEnv:

      * constructor defaults: `dynamics_params="defaultquad"` (:665 -- which `quad_rand` does not have: the fork's own default raises
        AttributeError, and so does this one; callers pass "DefaultQuad" / "Crazyflie" / ...) and `ep_time=4` (:668);
      * sampler dicts are keyed "type", not "class" (:759-768); a dict without it is a KeyError there and here;
      * the reward: log-distance position cost (:554) with the fork's default weights (:811-818: effort 0.01, spin 0), i.e.
        `reward="multi"` of the base class -- pinned by fixtures G7 and G17 (the `multi` blocks);
      * info["rewards"] holds the fork's keys only (:627-640).

    Everything else -- dynamics, controllers, observations, reset, the batched / device extensions (`num_envs`, `device`, ...) -- is
    the base class."""

    def __init__(self, dynamics_params="defaultquad", dynamics_change=None,
                 dynamics_randomize_every=None, dyn_sampler_1=None, dyn_sampler_2=None,
                 raw_control=True, raw_control_zero_middle=True, dim_mode='3D', tf_control=False, sim_freq=200., sim_steps=2,
                 obs_repr="xyz_vxyz_R_omega", ep_time=4, obstacles_num=0, room_size=10, init_random_state=False,
                 rew_coeff=None, sense_noise=None, verbose=False, gravity=GRAV, resample_goal=False,
                 t2w_std=0.005, t2t_std=0.0005, excite=False, dynamics_simplification=False, **extensions):
        if "reward" in extensions:
            raise TypeError("the fork's QuadrotorEnv always uses its own (log-distance) reward")
        fork_kwargs = dict(locals())
        for k in ("self", "extensions", "__class__"):
            fork_kwargs.pop(k, None)

        def to_class_key(spec):      # "type" -> the base class's "class" (quadrotor_multi.py:759-762 vs quadrotor.py:748-751)
            if spec is None:
                return None
            spec = copy.deepcopy(spec)
            spec["class"] = spec.pop("type")          # KeyError without "type", like the reference
            return spec
        super().__init__(dynamics_params=dynamics_params, dynamics_change=dynamics_change,
                         dynamics_randomize_every=dynamics_randomize_every, dyn_sampler_1=to_class_key(dyn_sampler_1),
                         dyn_sampler_2=to_class_key(dyn_sampler_2), raw_control=raw_control,
                         raw_control_zero_middle=raw_control_zero_middle, dim_mode=dim_mode, tf_control=tf_control, sim_freq=sim_freq,
                         sim_steps=sim_steps, obs_repr=obs_repr, ep_time=ep_time, obstacles_num=obstacles_num, room_size=room_size,
                         init_random_state=init_random_state, rew_coeff=rew_coeff, sense_noise=sense_noise, verbose=verbose,
                         gravity=gravity, resample_goal=resample_goal, t2w_std=t2w_std, t2t_std=t2t_std, excite=excite,
                         dynamics_simplification=dynamics_simplification, reward="multi", **extensions)
        self._ctor_kwargs = dict(copy.deepcopy(fork_kwargs), **copy.deepcopy(extensions))      # pickling by constructor arguments

    def step(self, action, out=None):
        res = super().step(action, out=out)
        info = res[3]
        if isinstance(info, dict) and "rewards" in info:
            info["rewards"] = {k: info["rewards"][k] for k in _FORK_REWARD_KEYS}
        return res


class QuadrotorEnvMulti(_QuadrotorEnv):
    def __init__(self, num_agents=8, num_worlds=1, goal_radius=0.5, collision_dist=None, prox_dist=None,
                 quadcol_coeff=1.0, quadprox_coeff=0.5, dynamics_params="DefaultQuad", reward="multi", collision_response=True, **kw):
        if "num_envs" in kw or "swarm" in kw:
            raise TypeError("QuadrotorEnvMulti takes num_agents / num_worlds, not num_envs / swarm")
        self.num_agents, self.num_worlds = int(num_agents), int(num_worlds)
        kw.setdefault("auto_reset", True)
        super().__init__(dynamics_params=dynamics_params, reward=reward, num_envs=self.num_agents * self.num_worlds,
                         swarm=dict(agents=self.num_agents, goal_radius=goal_radius, collision_dist=collision_dist,
                                    prox_dist=prox_dist, w_collision=quadcol_coeff, w_prox=quadprox_coeff,
                                    collision_response=collision_response), **kw)
        kwargs = dict(num_agents=num_agents, num_worlds=num_worlds, goal_radius=goal_radius, collision_dist=collision_dist,
                      prox_dist=prox_dist, quadcol_coeff=quadcol_coeff, quadprox_coeff=quadprox_coeff,
                      dynamics_params=dynamics_params, reward=reward, collision_response=collision_response, **kw)
        self._ctor_kwargs = kwargs

    def worlds(self, x):
        """[N, ...] -> [num_worlds, num_agents, ...] (a view for NumPy arrays and torch tensors alike)."""
        return x.reshape((self.num_worlds, self.num_agents) + tuple(x.shape[1:]))
