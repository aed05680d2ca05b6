#!/usr/bin/env python3
"""Microseconds per step of the default configuration (alias layout) at small batches, eager (bind_step) and as 32-step graph replays.
python3 tools/small_batch_rate.py [N ...]   (environment knobs under test are set by the caller: GAQ_LIB, HIP_FORCE_DEV_KERNARG ...)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from gym_art_amd import QuadrotorEnv  # noqa: E402

dev = torch.device("cuda", 0)
K = 32
out = {}
for n in [int(a) for a in sys.argv[1:]] or [16384, 65536, 131072]:
    kw = dict(num_envs=n, ep_time=5, seed=0, alias_obs=True)
    if os.environ.get("SB_MODEL"):
        kw["dynamics_params"] = os.environ["SB_MODEL"]
    env = QuadrotorEnv(**kw)
    obs = torch.empty((n, env.obs_dim), device=dev); rew = torch.empty(n, device=dev); done = torch.empty(n, dtype=torch.uint8, device=dev)
    act = torch.rand((n, 4), device=dev) * 2 - 1
    env.reset_dev(obs)
    step = env.bind_step(act, obs, rew, done)

    def timed(fn, iters):
        for _ in range(100):
            fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(7):
            t0 = time.perf_counter()
            for _ in range(iters):
                fn()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / iters)
        return best * 1e6
    eager = timed(step, 3000)
    ev = env.launch_variant
    env.set_graph_safe(True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        env.step_dev(act, obs, rew, done)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(K):
            env.step_dev(act, obs, rew, done)
    graph = timed(g.replay, 300) / K
    out[n] = {"eager_us": round(eager, 3), "graph_us": round(graph, 3), "variants": [ev, env.launch_variant],
              "frac_352B_eager": round(n * 352 / (eager * 1e-6) / 8e12, 3)}
    env.close()
print(json.dumps(out))
