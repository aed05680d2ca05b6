#!/usr/bin/env python3
"""How many residual bits does the split state need?  The kernel arithmetic header compiled for the host (tests/host_harness, storage
emulation 100 + bits: truncated fp32 head + `bits` residual mantissa bits for every integrator word) against the NumPy oracle over whole
500-step Hummingbird episodes (random initial states, action scales 1 / 0.3 / 0.05).  16 bits is what the alias layouts store
(DESIGN.md 3); fewer would save 2 B per word and bit of traffic but leave the 1e-5 parity bar.  No GPU needed."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import quad_oracle as qo
from tests import hh, golden_util as gu
n, T = int(os.environ.get("N", 512)), 500
rng = np.random.RandomState(11)
const = dict(gu.sub(gu.load("g2_hummingbird_raw"), "const_"))
pos = (rng.uniform(-2, 2, (n, 3)) + [0, 0, 2]).astype(np.float32).astype(np.float64)
pos[:, 2] = np.maximum(pos[:, 2], 0.25)
vel = rng.uniform(-1, 1, (n, 3)).astype(np.float32).astype(np.float64)
q, r = np.linalg.qr(rng.normal(size=(n, 3, 3)))
q = q * np.sign(np.einsum("nii->ni", r))[:, None, :]
q[np.linalg.det(q) < 0, :, 0] *= -1
rot = q.astype(np.float32).astype(np.float64)
omega = rng.uniform(-3, 3, (n, 3)).astype(np.float32).astype(np.float64)
scale = rng.choice([1.0, 0.3, 0.05], size=n)
acts = (rng.uniform(-1, 1, (T, n, 4)) * scale[None, :, None]).astype(np.float32)
p = qo.Params(n, mass=const["mass"], inertia=const["inertia"], thrust_max=const["thrust_max"], torque_max=const["torque_max"],
              prop_pos=np.asarray(const["prop_pos"]).reshape(4, 3), damp_time_up=const["damp_time_up"], damp_time_down=const["damp_time_down"],
              linearity=const["motor_linearity"], arm=const["arm"], ou_sigma=0., vel_damp=const["vel_damp"],
              damp_omega_quadratic=const["damp_omega_quadratic"], C_drag=0., C_roll=0.)
cfg = qo.Config(sim_freq=200., sim_steps=2, ep_time=5)
s = qo.State(n); s.set_state(pos, vel, rot, omega)
ref = np.zeros((T, n, 18))
for t in range(T):
    ref[t], _, _ = qo.env_step(s, p, cfg, acts[t].astype(np.float64))
m = hh.make_model(const); c = hh.make_cfg(0.005, 2, 500, m)
out = {}
for store, label in ((0, "fp64 state"), (116, "fp32 head + 16 bits (shipped)"), (112, "fp32 head + 12 bits"), (110, "fp32 head + 10 bits"),
                     (108, "fp32 head + 8 bits"), (104, "fp32 head + 4 bits"), (1, "fp32 state, rounded to nearest")):
    worst = np.zeros(n)
    for i in range(n):
        st = hh.pack_state(pos[i], vel[i], rot[i], omega[i], [0., 0., 2.])
        o = hh.rollout(c, m, st, acts[:, i], variant=0, store_f32=store, want_traj=False)["obs"]
        worst[i] = np.max(np.abs(o.astype(np.float64) - ref[:, i]) / np.maximum(np.abs(ref[:, i]), 1.0))
    qq = np.quantile(worst, [0.5, 0.99, 1.0])
    out[label] = {"episodes": n, "median": qq[0], "p99": qq[1], "max": qq[2], "frac_above_1e-6": float(np.mean(worst > 1e-6)),
                  "frac_above_1e-5": float(np.mean(worst > 1e-5))}
print(json.dumps(out, indent=1))
