#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point (gaq_step: NumPy actions in, NumPy obs/reward/done out).
Never the benchmarked number (bench.py keeps everything resident in HBM); quoted in DESIGN.md section 6."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_art_amd import QuadrotorEnv  # noqa: E402

out = []
for n, ring in ((1 << 20, 0), (1 << 20, 4), (65536, 0), (65536, 4), (1, 0)):
    env = QuadrotorEnv(num_envs=n, ep_time=5, seed=0, auto_reset=True, out_ring=ring)
    env.reset()
    a = np.random.RandomState(0).uniform(-1, 1, (n, 4)).astype(np.float32)
    steps = 20 if n > 1000 else 2000
    for _ in range(3):
        env.step(a if n > 1 else a[0])
    t0 = time.perf_counter()
    for _ in range(steps):
        env.step(a if n > 1 else a[0])
    dt = time.perf_counter() - t0
    out.append({"num_envs": n, "out_ring": ring, "steps": steps, "ms_per_step": dt / steps * 1e3, "env_steps_per_s": n * steps / dt,
                "path": ("QuadrotorEnv.step(numpy) -> gaq_step (small batch: the step launch reads the actions from and writes obs/reward/done into "
                         "mapped host memory itself, no copies; + the info dict's export launch)" if n <= 8192 else
                         "QuadrotorEnv.step(numpy) -> gaq_step (H2D actions, kernel, D2H obs/reward/done, pageable host memory)") +
                        ("; outputs from a ring of %d preallocated array sets (out_ring)" % ring if ring else "; fresh output arrays per call")})
    env.close()
print(json.dumps(out))
