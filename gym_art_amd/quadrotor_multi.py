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
import numpy as np

from .quadrotor import QuadrotorEnv


class QuadrotorEnvMulti(QuadrotorEnv):
    def __init__(self, num_agents=8, num_worlds=1, goal_radius=0.5, collision_dist=None, prox_dist=None,
                 quadcol_coeff=1.0, quadprox_coeff=0.5, dynamics_params="DefaultQuad", reward="multi", **kw):
        if "num_envs" in kw or "swarm" in kw:
            raise TypeError("QuadrotorEnvMulti takes num_agents / num_worlds, not num_envs / swarm")
        self.num_agents, self.num_worlds = int(num_agents), int(num_worlds)
        kw.setdefault("auto_reset", True)
        super().__init__(dynamics_params=dynamics_params, reward=reward, num_envs=self.num_agents * self.num_worlds,
                         swarm=dict(agents=self.num_agents, goal_radius=goal_radius, collision_dist=collision_dist,
                                    prox_dist=prox_dist, w_collision=quadcol_coeff, w_prox=quadprox_coeff), **kw)
        kwargs = dict(num_agents=num_agents, num_worlds=num_worlds, goal_radius=goal_radius, collision_dist=collision_dist,
                      prox_dist=prox_dist, quadcol_coeff=quadcol_coeff, quadprox_coeff=quadprox_coeff,
                      dynamics_params=dynamics_params, reward=reward, **kw)
        self._ctor_kwargs = kwargs

    def worlds(self, x):
        """[N, ...] -> [num_worlds, num_agents, ...] (a view for NumPy arrays and torch tensors alike)."""
        return x.reshape((self.num_worlds, self.num_agents) + tuple(x.shape[1:]))
