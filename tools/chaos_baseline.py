#!/usr/bin/env python3
"""How far apart do two runs of the SAME fp64 arithmetic end up when one starts one ulp away?  The NumPy oracle against
itself over 4096 random 500-step episodes of full-scale random actions (CPU only).  This is the floor under any
"max relative error vs the reference" figure for these episodes: no implementation that rounds differently from the
reference anywhere can do better than the tail printed here."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_art_amd import quad_params as qp, quadrotor_randomization as qr  # noqa: E402  (host-side parameter tables only)
from oracle import quad_oracle as qo  # noqa: E402

n, T = 4096, 500
out = {}
for per_env in (False, True):
    rng = np.random.RandomState(11)
    base = (qr.Crazyflie() if per_env else qr.DefaultQuad()).sample(n)
    base["noise"]["thrust_noise_ratio"] = np.zeros(n)
    tree = qr.RelativeSampler(base, noise_ratio=0.2).sample(base, rng=rng) if per_env else base
    models, _ = qp.derive_models(tree)
    p = qo.Params(n, mass=models["mass"], inertia=models["inertia"], thrust_max=models["thrust_max"],
                  torque_max=models["torque_max"], prop_pos=models["prop_pos"].reshape(n, 4, 3),
                  damp_time_up=models["damp_time_up"], damp_time_down=models["damp_time_down"],
                  linearity=models["linearity"], arm=models["arm"], ou_sigma=0 * models["ou_sigma"],
                  vel_damp=models["vel_damp"], damp_omega_quadratic=models["damp_omega_quadratic"],
                  C_drag=models["c_drag"], C_roll=models["c_roll"])
    cfg = qo.Config(sim_freq=200., sim_steps=2, ep_time=5)
    pos = rng.uniform(-2, 2, (n, 3)) + [0, 0, 2]
    pos[:, 2] = np.maximum(pos[:, 2], 0.25)
    vel = rng.uniform(-1, 1, (n, 3))
    q, r = np.linalg.qr(rng.normal(size=(n, 3, 3)))
    q = q * np.sign(np.einsum("nii->ni", r))[:, None, :]
    q[np.linalg.det(q) < 0, :, 0] *= -1
    om = rng.uniform(-3, 3, (n, 3))
    a, b = qo.State(n), qo.State(n)
    a.set_state(pos, vel, q, om)
    b.set_state(pos, np.nextafter(vel, np.inf), q, om)          # one ulp in the initial velocity ...
    b.omega = np.nextafter(b.omega, np.inf)                    # ... and in the angular velocity (translation never feeds back into rotation)
    worst = np.zeros(n)
    for t in range(T):
        act = rng.uniform(-1, 1, (n, 4)).astype(np.float32).astype(np.float64)
        oa, _, _ = qo.env_step(a, p, cfg, act)
        ob, _, _ = qo.env_step(b, p, cfg, act)
        worst = np.maximum(worst, np.max(np.abs(oa - ob) / np.maximum(np.abs(oa), 1.0), axis=1))
    qq = np.quantile(worst, [0.5, 0.99, 0.999, 1.0])
    out["crazyflie_randomized" if per_env else "hummingbird"] = {
        "median": qq[0], "p99": qq[1], "p99.9": qq[2], "max": qq[3],
        "frac_above_1e-7": float(np.mean(worst > 1e-7)), "frac_above_1e-6": float(np.mean(worst > 1e-6)),
        "frac_above_1e-5": float(np.mean(worst > 1e-5))}
print(json.dumps(out, indent=1))
