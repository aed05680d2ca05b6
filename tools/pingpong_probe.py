#!/usr/bin/env python3
"""Does updating the state heads OUT of place help beyond the Infinity Cache?  alias layout: the caller's obs tensor holds the heads; stepping
into the same tensor updates them in place, alternating two tensors reads one and writes the other (72 of the 277 B per env-step)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from gym_art_amd import QuadrotorEnv  # noqa: E402

dev = torch.device("cuda")
for n in (1 << 20, 1 << 22, 1 << 23):
    for mode in ("in place", "two obs tensors", "in place", "two obs tensors"):
        env = QuadrotorEnv(num_envs=n, ep_time=5, seed=0, alias_obs=True)
        obs = [torch.empty((n, 18), device=dev) for _ in range(2)]
        rew = torch.empty(n, device=dev); done = torch.empty(n, dtype=torch.uint8, device=dev)
        acts = [torch.rand((n, 4), device=dev) * 2 - 1 for _ in range(2)]
        env.reset_dev(obs[0])
        k = 600 if n <= (1 << 20) else 200
        sel = (lambda t: 0) if mode == "in place" else (lambda t: t & 1)
        for t in range(k):
            env.step_dev(acts[t & 1], obs[sel(t + 1)], rew, done)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(k):
            env.step_dev(acts[t & 1], obs[sel(t + 1)], rew, done)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / k
        print("N=%8d  %-16s %8.2f us per step  frac(352 B) %.3f" % (n, mode, dt * 1e6, 352.0 * n / dt / 8e12))
        env.close()
        del obs, rew, done, acts
