"""Dynamics-parameter samplers, vectorised over N envs.

Same names and sampler-dict format as the reference's gym_art/quadrotor/quadrotor_randomization.py
(`{"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"}`), but every sampler
works on a *batched* parameter tree (gym_art_amd.quad_params) and draws all N envs at once from a
numpy Generator/RandomState -- the reference instantiates Python objects per env and draws from
the global numpy RNG, so only the distributions agree, not the streams.

  check_quad_param_limits   :16-46      RelativeSampler / perturb_dyn_parameters  :345-358, :70-104
  AbsoluteSampler           :360-371    ConstValueSampler                        :373-377
  RandomQuad / randomquad_parameters :142-243   Crazyflie / DefaultQuad / MediumQuad :329-339
"""
import copy

import numpy as np

from . import quad_models
from . import quad_params as qp


def _walk(node, fn, path=()):
    for key, item in node.items():
        if isinstance(item, dict):
            _walk(item, fn, path + (key,))
        else:
            node[key] = fn(path + (key,), item)


def check_quad_param_limits(params, params_init=None):
    """Clip a batched tree to the reference's physical limits (:16-46).  With `params_init` the propeller
    radius is rescaled as r0 * (t2w_init / t2w_new)**0.5 (:41-44; the reference's variable names are swapped,
    the arithmetic here is what it executes)."""
    g = params["geom"]
    for key in ("body", "payload", "arms", "motors", "propellers"):
        for k in g[key]:
            g[key][k] = np.clip(g[key][k], 0.0, None)
    xyz = np.array(g["motor_pos"]["xyz"], dtype=np.float64)
    xyz[:, :2] = np.clip(xyz[:, :2], 0.005, None)
    g["motor_pos"]["xyz"] = xyz
    body_w = np.asarray(g["body"]["w"])
    g["payload_pos"]["xy"] = np.clip(g["payload_pos"]["xy"], (-body_w / 4.0)[:, None], (body_w / 4.0)[:, None])
    g["arms_pos"]["angle"] = np.clip(g["arms_pos"]["angle"], 0.0, 90.0)
    params["damp"]["vel"] = np.clip(params["damp"]["vel"], 0.0, 1.0)
    params["damp"]["omega_quadratic"] = np.clip(params["damp"]["omega_quadratic"], 0.0, 1.0)
    m = params["motor"]
    m["thrust_to_weight"] = np.clip(m["thrust_to_weight"], 1.2, None)
    m["torque_to_thrust"] = np.clip(m["torque_to_thrust"], 0.001, 1.0)
    m["linearity"] = np.clip(m["linearity"], 0.0, 1.0)
    m["assymetry"] = np.clip(m["assymetry"], 0.9, 1.1)
    for k in ("C_drag", "C_roll", "damp_time_up", "damp_time_down"):
        m[k] = np.clip(m[k], 0.0, None)
    if params_init is not None:
        r0 = np.asarray(params_init["geom"]["propellers"]["r"])
        t2w, t2w0 = np.asarray(params_init["motor"]["thrust_to_weight"]), np.asarray(m["thrust_to_weight"])
        g["propellers"]["r"] = r0 * (t2w / t2w0) ** 0.5
    return params


def _rng(rng):
    return np.random if rng is None else rng


class _ModelSampler(object):
    model = None

    def sample(self, n=1, rng=None):
        return qp.broadcast_tree(quad_models.model_params(self.model), n)


class Crazyflie(_ModelSampler):
    model = "crazyflie"


class DefaultQuad(_ModelSampler):
    model = "defaultquad"


class MediumQuad(_ModelSampler):
    model = "mediumquad"


class CrazyflieLowInertia(_ModelSampler):
    model = "crazyflie_lowinertia"


class RandomQuad(object):
    """randomquad_parameters (:142-243): a random quadrotor per env, in the reference's draw order."""

    def sample(self, n=1, rng=None):
        r = _rng(rng)
        U = lambda lo, hi, size=n: r.uniform(lo, hi, size=size)
        Nrm = lambda loc, scale, size=n: r.normal(loc, scale, size=size)
        dens = r.uniform(low=[500., 200., 500., 500., 200.], high=[2000., 2000., 2000., 4500., 300.], size=(n, 5))
        geom = {k: {"density": dens[:, i]} for i, k in enumerate(("body", "payload", "arms", "motors", "propellers"))}
        total_w = U(0.05, 0.2)
        total_l = np.clip(Nrm(1., 0.1), 1.0, None) * total_w
        motor_z = Nrm(0., total_w / 8.)
        geom["motor_pos"] = {"xyz": np.stack([total_w / 2., total_l / 2., motor_z], axis=1)}
        geom["motors"]["r"] = total_w * Nrm(0.1, 0.01)
        geom["motors"]["h"] = geom["motors"]["r"] * Nrm(1.0, 0.05)
        w_low, w_high = 0.25, 0.5
        w_coeff = U(w_low, w_high)
        geom["body"]["w"] = w_coeff * total_w
        l_scale = 1. - (w_coeff - w_low) / (w_high - w_low)
        geom["body"]["l"] = np.clip(Nrm(1., l_scale), 1.0, None) * geom["body"]["w"]
        geom["body"]["h"] = U(0.1, 1.5) * geom["body"]["w"]
        pl = r.uniform(0.25, 1.0, size=(n, 3))
        geom["payload"]["w"] = pl[:, 0] * geom["body"]["w"]
        geom["payload"]["l"] = pl[:, 1] * geom["body"]["l"]
        geom["payload"]["h"] = pl[:, 2] * geom["body"]["h"]
        geom["payload_pos"] = {"xy": r.normal(0., (geom["body"]["w"] / 10.)[:, None], size=(n, 2)),
                               "z_sign": np.sign(U(-1, 1))}
        geom["arms"]["w"] = total_w * Nrm(0.05, 0.005)
        geom["arms"]["h"] = total_w * Nrm(0.05, 0.005)
        geom["arms_pos"] = {"angle": Nrm(45., 10.), "z": motor_z - geom["motors"]["h"] / 2.}
        t2w = U(1.5, 3.5)
        geom["propellers"]["h"] = np.full(n, 0.01)
        geom["propellers"]["r"] = 0.3 * total_w * (t2w / 2.0) ** 0.5
        noise_ratio = U(0.01, 0.05)                     # the reference's order of draws (:216-224): noise ratio, motor time
        damp_up = U(0.15, 0.2)                          # constant up, its down scale, torque-to-thrust, the four asymmetries --
        down_scale = U(1.0, 1.0)                        # with n = 1 and the same seed the tree equals the reference's (fixture G21)
        t2t = U(0.005, 0.025)
        asym = r.uniform(0.9, 1.1, size=(n, 4))
        params = {
            "geom": geom,
            "damp": {"vel": np.zeros(n), "omega_quadratic": np.zeros(n)},
            "noise": {"thrust_noise_ratio": noise_ratio},
            "motor": {"thrust_to_weight": t2w, "torque_to_thrust": t2t,
                      "assymetry": asym, "linearity": np.ones(n), "C_drag": np.zeros(n),
                      "C_roll": np.zeros(n), "damp_time_up": damp_up, "damp_time_down": down_scale * damp_up},
        }
        return check_quad_param_limits(params)


def get_dyn_randomization_params(quad_params, noise_ratio=0., noise_ratio_params=None):
    """Tree of per-leaf noise ratios (:48-68)."""
    noise = copy.deepcopy(quad_params)
    _walk(noise, lambda path, item: noise_ratio)
    if noise_ratio_params is not None:
        qp.update_tree(noise, noise_ratio_params)
    return noise


def perturb_dyn_parameters(params, noise_params, sampler="normal", rng=None):
    """Sample every numeric leaf around its nominal value (:70-104): normal(loc=v, scale=|ratio/2 * v|) or
    uniform(v - v*ratio, v + v*ratio), then re-apply the limits."""
    r = _rng(rng)
    new = copy.deepcopy(params)

    def draw(path, val):
        ratio = noise_params
        for k in path:
            ratio = ratio[k]
        val = np.asarray(val, dtype=np.float64)
        ratio = np.asarray(ratio, dtype=np.float64)
        if ratio.ndim == 1 and val.ndim == 2:
            ratio = ratio[:, None]
        if sampler == "normal":
            return r.normal(loc=val, scale=np.abs((ratio / 2) * val))
        if sampler == "uniform":
            # (for a negative leaf low > high: numpy then returns low + (high - low) u like the reference's call does -- same interval)
            return r.uniform(low=val - val * ratio, high=val + val * ratio)
        raise KeyError("sample_" + sampler)

    _walk(new, draw)
    return check_quad_param_limits(new, params)


class RelativeSampler(object):
    def __init__(self, params, noise_ratio=0., noise_ratio_custom=None, sampler="normal"):
        self.noise_ratio = noise_ratio
        self.noise_ratio_custom = noise_ratio_custom
        self.sampler = sampler

    def sample(self, params, rng=None):
        noise = get_dyn_randomization_params(params, noise_ratio=self.noise_ratio,
                                             noise_ratio_params=self.noise_ratio_custom)
        return perturb_dyn_parameters(params, noise, sampler=self.sampler, rng=rng)


class AbsoluteSampler(object):
    """resample_dyn_parameters (:106-139): leaves of `noise_params` are objects with .min / .max."""

    def __init__(self, params, noise_params, sampler="uniform"):
        self.noise_params = copy.deepcopy(noise_params)
        self.sampler = sampler

    def sample(self, params, rng=None):
        r = _rng(rng)
        new = copy.deepcopy(params)

        def draw(path, val):
            mm = self.noise_params
            for k in path:
                mm = mm[k]
            val = np.asarray(val, dtype=np.float64)
            if self.sampler == "uniform":
                return r.uniform(low=mm.min * np.ones_like(val), high=mm.max * np.ones_like(val))
            mean, std = (mm.min + mm.max) / 2, (mm.max - mm.min) / 4
            return r.normal(loc=mean * np.ones_like(val), scale=std)

        _walk(new, draw)
        return check_quad_param_limits(new, params)


class ConstValueSampler(object):
    def __init__(self, params, params_change):
        self.params_change = copy.deepcopy(params_change)

    def sample(self, params, rng=None):
        n = qp.tree_size(params)
        qp.update_tree(params, qp.broadcast_tree(self.params_change, n))
        return params
