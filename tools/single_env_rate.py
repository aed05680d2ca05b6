#!/usr/bin/env python3
"""BASELINE config 1 through the drop-in class: ONE Hummingbird env stepped like the reference's own loops (quadrotor.py:1278-1305):
`obs, rew, done, info = env.step(action)` per call, Mellinger controller or RawControl, info dict on / off.  The reference does
2.7e3 (Mellinger) / 3.2e3 (RawControl) of these per second on one CPU core (tests/golden/reference_timing.json); here every call is a
kernel launch plus a device-to-host copy, so the figure is a latency, not a throughput.  Prints one JSON object."""
import cProfile
import io
import json
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_art_amd import QuadrotorEnv  # noqa: E402

out = {}
for label, kw in (("mellinger_info", dict(raw_control=False)), ("mellinger_noinfo", dict(raw_control=False, info=False)),
                  ("raw_info", dict(raw_control=True)), ("raw_noinfo", dict(raw_control=True, info=False))):
    env = QuadrotorEnv(seed=1, **kw)
    env.reset()
    rng = np.random.RandomState(0)
    acts = rng.uniform(-1, 1, (4096, 4))
    for t in range(300):
        env.step(acts[t])
    n = 3000
    t0 = time.perf_counter()
    for t in range(n):
        o, r, d, info = env.step(acts[t % 4096])
        if d:
            env.reset()
    dt = time.perf_counter() - t0
    out[label] = {"steps_per_s": n / dt, "us_per_step": dt / n * 1e6}
    if label == "mellinger_info" and os.environ.get("PROFILE"):
        pr = cProfile.Profile(); pr.enable()
        for t in range(1000):
            env.step(acts[t])
        pr.disable()
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18)
        sys.stderr.write(s.getvalue())
    env.close()
print(json.dumps(out, indent=1))
