#!/usr/bin/env python3
"""Where a single-env step() goes on the host: cProfile of 20 000 calls of the drop-in loop (num_envs=1, Mellinger, complete info dict)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_art_amd import QuadrotorEnv  # noqa: E402

for kw in (dict(raw_control=False), dict()):
    env = QuadrotorEnv(num_envs=1, ep_time=5, seed=0, **kw)
    env.reset()
    a = np.zeros(4, np.float32)
    for _ in range(500):
        env.step(a)
    t0 = time.perf_counter()
    for _ in range(5000):
        env.step(a)
    print(kw, "%.2f us per step()" % ((time.perf_counter() - t0) / 5000 * 1e6))
    t0 = time.perf_counter()
    o = np.empty((1, env.obs_dim), np.float32); r = np.empty(1, np.float32); d = np.empty(1, np.uint8); aa = a.reshape(1, 4)
    for _ in range(5000):
        env._c_step(aa, o, r, d)
    print(kw, "%.2f us per _c_step (the library call alone)" % ((time.perf_counter() - t0) / 5000 * 1e6))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20000):
        env.step(a)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
    env.close()
