#!/usr/bin/env python3
"""Where a step goes at small batches: microseconds per step of the default Hummingbird configuration
(alias layout) at N = 16 384 ... 2^20, launched eagerly and as 32-step HIP-graph replays, for
  empty    a 1-element torch kernel in place of the step (the launch / dependency-chain floor of that launch mode)
  move     the real step kernel with the arithmetic skipped (GAQ_ABLATE=1, which only a MEASUREMENT build of the library honours: run
           with GAQ_LIB pointing at one -- tools/latency_breakdown.sh builds it with -DGAQ_DIAG_BUILD): HBM -> LDS -> registers -> LDS -> HBM only
  no_noise the real kernel without thrust noise (no Philox / Box-Muller, no OU plane)
  full     the real kernel
so that  launch = empty,  data path = move - empty,  arithmetic = full - move,  noise = full - no_noise  (read the GRAPH columns: the
eager `empty` is a torch op and measures Python, not the GPU).
Eager launches go through QuadrotorEnv.bind_step (one ctypes call per step) with the host-side step counter; the graph replays run in
graph-safe mode (round 3: one kernel node per step, the F_CTR twin advances the device-resident counter itself).
GPU needed.  bash tools/latency_breakdown.sh > profiles/rNN_latency_breakdown.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

dev = torch.device("cuda", 0)
K = 32


def timed(fn, iters):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / iters)
    return best * 1e6


def graph_of(step):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(K):
            step()
    return g


def measure(n, what):
    from gym_art_amd import QuadrotorEnv
    if what == "empty":
        x = torch.zeros(1, device=dev)
        step = lambda: x.add_(1.0)
    else:
        os.environ.pop("GAQ_ABLATE", None)
        if what == "move":
            os.environ["GAQ_ABLATE"] = "1"
        env = QuadrotorEnv(num_envs=n, ep_time=5, seed=0, alias_obs=True, thrust_noise="off" if what == "no_noise" else "philox")
        os.environ.pop("GAQ_ABLATE", None)
        obs = torch.empty((n, 18), device=dev); rew = torch.empty(n, device=dev); done = torch.empty(n, dtype=torch.uint8, device=dev)
        act = torch.rand((n, 4), device=dev) * 2 - 1
        env.reset_dev(obs)
        step = env.bind_step(act, obs, rew, done)          # eager: one ctypes call per step, stream looked up once
    eager = timed(step, 2000)
    variants = None
    if what != "empty":
        variants = {"eager": env.launch_variant}
        env.set_graph_safe(True)
        variants["graph"] = env.launch_variant
        step = lambda: env.step_dev(act, obs, rew, done)   # capture: the stream is torch's CAPTURING stream, looked up per call
    g = graph_of(step)
    graph = timed(g.replay, 200) / K
    return {"eager_us": eager, "graph_us": graph, **({"kernel_variants": variants} if variants else {})}


from gym_art_amd import _lib  # noqa: E402
assert _lib.load().gaq_is_diag_build() == 1, "run with GAQ_LIB=<a -DGAQ_DIAG_BUILD library> (tools/latency_breakdown.sh)"
out = {"what": __doc__.split("GPU needed")[0].strip(), "K_steps_per_graph": K, "library": _lib.LIB_PATH, "rows": []}
for n in (16384, 65536, 131072, 262144, 1 << 20):
    row = {"N": n}
    for what in ("empty", "move", "no_noise", "full"):
        row[what] = measure(n, what)
    for mode in ("eager_us", "graph_us"):
        row["breakdown_" + mode] = {"launch": row["empty"][mode], "data_path": row["move"][mode] - row["empty"][mode],
                                    "arithmetic": row["full"][mode] - row["move"][mode],
                                    "of_which_noise": row["full"][mode] - row["no_noise"][mode]}
    row["frac_352B_graph"] = n * 352 / (row["full"]["graph_us"] * 1e-6) / 8e12
    row["frac_352B_eager"] = n * 352 / (row["full"]["eager_us"] * 1e-6) / 8e12
    out["rows"].append(row)
    sys.stderr.write(json.dumps(row) + "\n")
print(json.dumps(out, indent=1))
