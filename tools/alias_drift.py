#!/usr/bin/env python3
"""How far does the split-state (alias) layout drift from the plain fp64 layout over one 500-step episode of full-scale
random actions?  The plain layout tracks the reference to 6e-8 on every fixture, so it stands in for it here on
thousands of trajectories (the fixtures hold a dozen).  Prints the distribution of the per-trajectory max rel. error."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_art_amd import QuadrotorEnv  # noqa: E402

n, steps = 4096, 500
out = {}
for model in ("DefaultQuad", "Crazyflie"):
    for scale in (1.0, 0.5):
        kw = dict(dynamics_params=model, num_envs=n, ep_time=5, seed=5, thrust_noise="off", auto_reset=False)
        a_env, p_env = QuadrotorEnv(alias_obs=True, **kw), QuadrotorEnv(alias_obs=False, **kw)
        assert a_env.obs_is_state and not p_env.obs_is_state
        p_env.set_state(a_env.get_state())
        rng = np.random.RandomState(7)
        worst = np.zeros(n)
        for t in range(steps):
            act = (scale * rng.uniform(-1, 1, (n, 4))).astype(np.float32)
            oa, _, _, _ = a_env.step(act)
            op, _, _, _ = p_env.step(act)
            err = np.max(np.abs(oa.astype(np.float64) - op) / np.maximum(np.abs(op), 1.0), axis=1)
            worst = np.maximum(worst, err)
        q = np.quantile(worst, [0.5, 0.9, 0.99, 0.999, 1.0])
        out["%s scale %.1f" % (model, scale)] = {"median": q[0], "p90": q[1], "p99": q[2], "p99.9": q[3], "max": q[4],
                                                 "frac_above_1e-5": float(np.mean(worst > 1e-5)),
                                                 "frac_above_1e-6": float(np.mean(worst > 1e-6))}
        a_env.close(); p_env.close()
print(json.dumps(out, indent=1))
