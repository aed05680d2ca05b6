"""One CPU worker of bench.py's `cpu_baseline` leg: the NumPy oracle (quad_oracle.py) stepping N Hummingbird envs,
thrust noise on, for a time budget; prints one JSON line.  TEST / MEASUREMENT INFRASTRUCTURE ONLY (see
quad_oracle.py's header): run as a child process so that several can use several cores, and so that nothing here
shares a process with the GPU runtime.

    python -m oracle.cpu_worker --envs 16384 --seconds 10 [--one-env-loop]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import quad_oracle as qo  # noqa: E402

# Hummingbird ("DefaultQuad") derived constants, as QuadrotorDynamics.update_model computes them (the
# values are checked against the reference in tests/test_quad_params.py::test_hummingbird_constants_quoted_in_survey)
HUMMINGBIRD = dict(mass=0.816, inertia=(3.746575e-3, 3.746575e-3, 6.149342e-3), thrust_max=5.603472, torque_max=0.2801736,
                   prop_pos=((0.12, -0.12, 7.174e-3), (-0.12, -0.12, 7.174e-3), (-0.12, 0.12, 7.174e-3), (0.12, 0.12, 7.174e-3)),
                   damp_time_up=0.0, damp_time_down=0.0, linearity=1.0, arm=0.169706, ou_sigma=0.01, vel_damp=0.0,
                   damp_omega_quadratic=0.0, C_drag=0.0, C_roll=0.0)


def run(n, seconds, seed=0):
    p = qo.Params(n, **HUMMINGBIRD)
    cfg = qo.Config(sim_freq=200., sim_steps=2, ep_time=5)
    s = qo.State(n)
    rng = np.random.RandomState(seed)
    qo.reset(s, p, cfg, rng)
    steps = 0
    t0 = time.perf_counter()
    while True:
        a = rng.uniform(-1, 1, size=(n, 4)).astype(np.float32).astype(np.float64)
        nz = rng.randn(cfg.sim_steps, n, 4)
        _, _, done = qo.env_step(s, p, cfg, a, nz)
        if done.any():
            qo.reset(s, p, cfg, rng, idx=np.where(done)[0])
        steps += 1
        el = time.perf_counter() - t0
        if el > seconds and steps >= 3:
            return steps, el


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=16384)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    steps, el = run(args.envs, args.seconds, args.seed)
    print(json.dumps({"envs": args.envs, "steps": steps, "seconds": el, "env_steps_per_s": args.envs * steps / el}))
