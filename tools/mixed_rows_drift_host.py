#!/usr/bin/env python3
"""Host-side check of the alias layout's MIXED residual rows (pos / vel / R 39 bits, omega exact; gaq_kernels.hpp kLoMix): the kernel
arithmetic header compiled for the host (tests/host_harness, storage emulation 2) against the NumPy oracle over whole
500-step episodes of full-scale random actions on random initial states with per-env randomised CrazyFlie parameters --
the population on which 16-bit residuals everywhere left 2.6 % of the episodes > 1e-6 away.  No GPU needed;
tools/oracle_drift.py is the same experiment on the device."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_art_amd import quad_params as qp, quadrotor_randomization as qr  # noqa: E402
from oracle import quad_oracle as qo  # noqa: E402
from tests import hh  # noqa: E402

n, T = int(os.environ.get("N", 2048)), 500
rng = np.random.RandomState(11)
base = qr.Crazyflie().sample(n)
base["noise"]["thrust_noise_ratio"] = np.zeros(n)
tree = qr.RelativeSampler(base, noise_ratio=0.2).sample(base, rng=rng)
models, _ = qp.derive_models(tree)
pos = (rng.uniform(-2, 2, (n, 3)) + [0, 0, 2]).astype(np.float32).astype(np.float64)
pos[:, 2] = np.maximum(pos[:, 2], 0.25)
vel = rng.uniform(-1, 1, (n, 3)).astype(np.float32).astype(np.float64)
q, r = np.linalg.qr(rng.normal(size=(n, 3, 3)))
q = q * np.sign(np.einsum("nii->ni", r))[:, None, :]
q[np.linalg.det(q) < 0, :, 0] *= -1
rot = q.astype(np.float32).astype(np.float64)
omega = rng.uniform(-3, 3, (n, 3)).astype(np.float32).astype(np.float64)
acts = rng.uniform(-1, 1, (T, n, 4)).astype(np.float32)
p = qo.Params(n, mass=models["mass"], inertia=models["inertia"], thrust_max=models["thrust_max"],
              torque_max=models["torque_max"], prop_pos=models["prop_pos"].reshape(n, 4, 3),
              damp_time_up=models["damp_time_up"], damp_time_down=models["damp_time_down"],
              linearity=models["linearity"], arm=models["arm"], ou_sigma=0 * models["ou_sigma"],
              vel_damp=models["vel_damp"], damp_omega_quadratic=models["damp_omega_quadratic"],
              C_drag=models["c_drag"], C_roll=models["c_roll"])
cfg = qo.Config(sim_freq=200., sim_steps=2, ep_time=5)
s = qo.State(n)
s.set_state(pos, vel, rot, omega)
ref = np.zeros((T, n, 18))
for t in range(T):
    ref[t], _, _ = qo.env_step(s, p, cfg, acts[t].astype(np.float64))
out = {}
for store, label in ((2, "mixed_rows(pos/vel/R 39 bits, omega exact)"), (0, "fp64_planes")):
    worst = np.zeros(n)
    for i in range(n):
        const = dict(mass=models["mass"][i], inertia=models["inertia"][i], thrust_max=models["thrust_max"][i],
                     torque_max=models["torque_max"][i], prop_pos=models["prop_pos"][i], damp_time_up=models["damp_time_up"][i],
                     damp_time_down=models["damp_time_down"][i], motor_linearity=models["linearity"][i], arm=models["arm"][i],
                     thrust_noise_sigma=0., vel_damp=models["vel_damp"][i], damp_omega_quadratic=models["damp_omega_quadratic"][i],
                     C_rot_drag=0., C_rot_roll=0.)
        m = hh.make_model(const)
        c = hh.make_cfg(0.005, 2, 500, m)
        st = hh.pack_state(pos[i], vel[i], rot[i], omega[i], [0., 0., 2.])
        o = hh.rollout(c, m, st, acts[:, i], variant=2, store_f32=store, want_traj=False)["obs"]
        worst[i] = np.max(np.abs(o.astype(np.float64) - ref[:, i]) / np.maximum(np.abs(ref[:, i]), 1.0))
    qq = np.quantile(worst, [0.5, 0.99, 0.999, 1.0])
    out[label] = {"episodes": n, "median": qq[0], "p99": qq[1], "p99.9": qq[2], "max": qq[3],
                  "frac_above_1e-6": float(np.mean(worst > 1e-6)), "frac_above_1e-5": float(np.mean(worst > 1e-5))}
print(json.dumps(out, indent=1))
