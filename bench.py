#!/usr/bin/env python
"""bench.py -- env-steps/s of the fused quadrotor step kernel on MI355X (driver contract: one JSON line).

Workload (BASELINE.json metric: "env-steps/sec at N=2^20 Hummingbird; achieved HBM GB/s"):
  N = 2^20 Hummingbird (`DefaultQuad`) envs PER GPU, RawControl zero-middle, sim_freq 200, sim_steps 2,
  ep_time 5 (ep_len 500 -> 501 steps/episode, in-kernel auto-reset), obs `xyz_vxyz_R_omega` (18 floats),
  OU thrust noise ON (on-device Philox), default reward weights; actions i.i.d. U(-1,1) float32 resident in HBM
  (a ring of pre-generated [N,4] tensors).  One "step" = one launch of the fused kernel over the whole batch
  (= sim_steps sub-steps + reward + obs + done + reset per env).
  N > 1 GPUs: one process per GPU, contiguous env-index shards (weak scaling: 2^20 envs per GPU), and the
  north_star's single RCCL gather of the stacked observation tensor to rank 0 after every step.

Extra objects in the JSON line: `roofline` (algorithmic bytes / measured kernel time vs 8 TB/s HBM) and
`cpu_baseline` (the NumPy oracle timed on this box's host cores on a bounded sample; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_ALG = 352           # algorithmic bytes per env-step (DESIGN.md "Byte accounting")
HBM_PEAK_GBPS = 8000.0


def pmc_traffic_per_env_step(alias, variant):
    """HBM bytes per env-step of the step kernel as measured with rocprofv3 PMC counters (separate --pmc passes,
    gfx950 FETCH_SIZE correction, tools/pmc_summary.py) and committed under profiles/; None when no profile matches
    this kernel variant.  `variant`: "default" (Hummingbird, noise on), "c3" (randomised CrazyFlie) or None."""
    if variant == "default":
        name = "r01_v8_pmc.json" if alias else "r01_v2_pmc.json"
    elif variant == "c3" and alias:
        name = "r01_v9_pmc_c3.json"
    else:
        return None, None
    path = os.path.join(ROOT, "profiles", name)
    try:
        with open(path) as f:
            return float(json.load(f)["_derived"]["traffic_bytes_per_env_step"]), "profiles/" + name
    except Exception:
        return None, None


def cpu_baseline(seconds_budget=12.0):
    """The CPU restatement (oracle/quad_oracle.py, vectorised NumPy fp64) on a bounded sample of the same workload:
    N = 16 384 Hummingbird envs per worker, thrust noise on.  Workers are child processes (oracle/cpu_worker.py):
    first one alone (the 1-core rate), then one per available core (the all-cores rate = `value`)."""
    import subprocess
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))                   # a 1-GPU box's CPU share is 16
    n = 16384
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    worker = [sys.executable, "-m", "oracle.cpu_worker", "--envs", str(n)]

    def launch(k, seconds, envs=None):
        cmd = list(worker) + ["--seconds", "%.2f" % seconds, "--seed", str(k)]
        if envs is not None:
            cmd[cmd.index("--envs") + 1] = str(envs)
        return subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)

    def collect(procs):
        return [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in procs]

    t_each = max(seconds_budget / 3.0, 1.0)
    one = collect([launch(0, t_each)])[0]
    loop = collect([launch(0, min(2.0, t_each), envs=1)])[0]
    many = collect([launch(k, t_each) for k in range(cores)])
    total = sum(r["envs"] * r["steps"] for r in many) / max(r["seconds"] for r in many)
    ref_rate = None
    try:
        with open(os.path.join(ROOT, "tests", "golden", "reference_timing.json")) as f:
            ref_rate = float(json.load(f)["env_steps_per_s"]["raw"])
    except Exception:
        pass
    return {"value": total, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "oracle/quad_oracle.py (NumPy fp64, vectorised) in %d worker processes x N=%d Hummingbird envs, noise on, "
                      "%.1f s each (%d env steps per worker); host reports %d cpus, %d usable"
                      % (cores, n, t_each, many[0]["steps"], os.cpu_count() or 0, avail),
            "one_core": {"value": one["env_steps_per_s"], "unit": "env-steps/s", "cores": 1,
                         "what": "one worker alone, N=%d x %d steps" % (n, one["steps"])},
            "one_env_per_call": {"value": loop["env_steps_per_s"], "unit": "env-steps/s", "cores": 1,
                                 "what": "the same oracle stepped like the reference: N=1 per call in a Python loop, on this host"},
            "reference_in_build_container": {"value": ref_rate, "unit": "env-steps/s", "cores": 1,
                                             "what": "unmodified reference QuadrotorEnv.step, RawControl, measured where "
                                                     "/root/reference exists (tests/golden/reference_timing.json)"}}


def main():
    # stdout carries exactly ONE JSON line: everything else written to fd 1 by native libraries (RCCL prints
    # its banner there) goes to stderr
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1100, help="timed steps (default spans two 501-step episodes)")
    ap.add_argument("--warmup", type=int, default=2000,
                    help="untimed steps first (default 0.1 s of them: the first ~50 ms after idle run ~5 %% slow while the "
                         "clocks ramp, see `repeats` in the output)")
    ap.add_argument("--envs", type=int, default=1 << 20, help="envs per GPU")
    ap.add_argument("--model", default="DefaultQuad")
    ap.add_argument("--randomize", action="store_true", help="config 3: per-env RelativeSampler(0.2) parameters")
    ap.add_argument("--no-noise", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="multi-GPU: skip the observation gather")
    ap.add_argument("--no-alias", action="store_true", help="keep obs and state separate (gaq_config.obs_state_alias=0)")
    ap.add_argument("--rollout", type=int, default=0, metavar="T",
                    help="time gaq_step_many_dev with T open-loop steps per call (fused rollout kernel) instead of "
                         "one launch per step; each of --steps timed iterations is then one T-step call")
    ap.add_argument("--fp32", action="store_true",
                    help="fp32 arithmetic and state (gaq_config.fp32_state): throughput-first, OUTSIDE the 1e-5 parity bar; "
                         "never the headline number")
    ap.add_argument("--reward", default="quadrotor", choices=["quadrotor", "multi"],
                    help="'multi' = the log-distance reward of the quadrotor_multi fork (quadrotor_multi.py:554)")
    ap.add_argument("--graph", type=int, default=0, metavar="K",
                    help="capture K consecutive single-step launches in one HIP graph (gaq_set_graph_safe) and time "
                         "replays; each of --steps timed iterations is then one K-step replay")
    ap.add_argument("--swarm", type=int, default=0, metavar="A",
                    help="config 5: worlds of A agents with the neighbour reward / observation terms (this build's own "
                         "specification, parity-unpinned); --envs stays the number of agents per GPU")
    ap.add_argument("--prime-ms", type=float, default=150.0,
                    help="milliseconds of scratch GPU work before the warm-up steps (clock ramp after idle); 0 = none")
    ap.add_argument("--repeats", type=int, default=5, help="timed regions in all (the first one is the reported value)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from gym_art_amd import QuadrotorEnv
    from gym_art_amd.sharding import ShardedQuadrotorEnv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for --gpus N"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_dist = os.environ.get("GAQ_BENCH_FORCE_DIST") == "1"     # exercise the RCCL path on a single rank
    if world > 1 or force_dist:
        dist.init_process_group("nccl", device_id=dev)

    n = args.envs
    kw = dict(dynamics_params=args.model, ep_time=5, sim_freq=200., sim_steps=2, seed=0, auto_reset=True,
              thrust_noise="off" if args.no_noise else "philox", alias_obs=not args.no_alias)
    if args.randomize:
        kw["dyn_sampler_1"] = {"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"}
    if args.reward != "quadrotor":
        kw.update(reward=args.reward)
    if args.fp32:
        kw.update(precision="fp32")
    if args.swarm:
        kw.update(reward="multi", swarm=dict(agents=args.swarm))
    sharded = ShardedQuadrotorEnv(n * world, **kw)      # contiguous global index range per rank
    assert (sharded.first, sharded.count) == (rank * n, n)
    env = sharded.env
    D = env.obs_dim
    ring = 8
    gen = torch.Generator(device=dev)
    gen.manual_seed(rank)
    actions = [torch.rand((n, 4), device=dev, generator=gen) * 2 - 1 for _ in range(ring)]
    do_gather = (world > 1 or force_dist) and not args.no_gather
    sharded.reset()

    roll = args.rollout
    if roll:
        assert world == 1, "--rollout is a single-GPU measurement"
        acts_T = (torch.rand((roll, n, 4), device=dev, generator=gen) * 2 - 1)
        obs_T = torch.empty((roll, n, D), device=dev)
        rew_T = torch.empty((roll, n), device=dev)
        done_T = torch.empty((roll, n), dtype=torch.uint8, device=dev)

    graph = None
    if args.graph:
        assert world == 1 and not roll, "--graph is a single-GPU, single-step-launch measurement"
        env.set_graph_safe(True)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            env.step_dev(actions[0], sharded.obs, sharded.reward, sharded.done)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for k in range(args.graph):
                env.step_dev(actions[k % ring], sharded.obs, sharded.reward, sharded.done)
        roll = args.graph          # accounting: K env steps per timed iteration

    def one_step(t):
        if graph is not None:
            graph.replay()
        elif roll:
            env.step_many_dev(acts_T, obs_T, rew_T, done_T)
        else:
            sharded.step(actions[t % ring], gather=do_gather)

    # Bring the GPU out of idle before anything is counted: the first ~50 ms of work after idle run ~5 % slow while the
    # clocks ramp (see `repeats`).  This is a scratch fill loop, not steps of the benchmark; the W warm-up steps and the K
    # timed steps below are untouched.  `--prime-ms 0` switches it off; the line reports what was done.
    if args.prime_ms > 0:
        scratch = torch.empty(64 << 20, dtype=torch.float32, device=dev)
        p0 = time.perf_counter()
        while (time.perf_counter() - p0) * 1e3 < args.prime_ms:
            for _ in range(8):
                scratch.add_(1.0)
            torch.cuda.synchronize()
        del scratch
    for t in range(args.warmup):
        one_step(t)
    # HIP events on the launch stream (torch's current stream = the stream step_dev launches on).  Single GPU:
    # one pair brackets the whole timed region (per-launch pairs put barrier packets between the kernels).
    # Multi GPU: the region also holds the gathers, so the kernel is timed with a pair per launch.
    per_launch = do_gather
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps if per_launch else 1)]
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if not per_launch:
        ev[0][0].record()
    for t in range(args.steps):
        if per_launch:
            ev[t][0].record()
        if graph is not None:
            graph.replay()
        elif roll:
            env.step_many_dev(acts_T, obs_T, rew_T, done_T)
        else:
            env.step_dev(actions[t % ring], sharded.obs, sharded.reward, sharded.done)
        if per_launch:
            ev[t][1].record()
        if do_gather:
            sharded.gather_obs()
    if not per_launch:
        ev[0][1].record()
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    elapsed = time.perf_counter() - t0
    # SURVEY 8(d) asks for the median of five runs: the line's `value` stays the contract's single K-step region above;
    # four more identical regions follow (single GPU only) and all five rates go into `repeats`
    extra = []
    if world == 1 and not force_dist and args.repeats > 1:
        for _ in range(args.repeats - 1):
            torch.cuda.synchronize()
            r0 = time.perf_counter()
            for t in range(args.steps):
                one_step(t)
            torch.cuda.synchronize()
            extra.append(time.perf_counter() - r0)
    env.check_finite()
    kern_ms = float(np.sum([a.elapsed_time(b) for a, b in ev])) / args.steps
    if dist.is_initialized():
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        total_envs = n * world
        env_steps_per_iter = total_envs * (roll if roll else 1)
        value = env_steps_per_iter * args.steps / elapsed
        b_alg = B_ALG + (128 if args.randomize else 0) + (24 * (args.swarm - 1) if args.swarm else 0)   # + neighbour obs words
        plain_run = not (args.swarm or args.no_noise or args.fp32 or args.reward != "quadrotor")
        variant = "default" if plain_run and not args.randomize and args.model == "DefaultQuad" else \
                  "c3" if plain_run and args.randomize and args.model == "Crazyflie" else None
        per_env, src = pmc_traffic_per_env_step(env.obs_is_state, variant)   # profiles exist for these two kernels
        kernel_name = "step_kernel"
        if roll and not args.graph:
            # a fused T-step launch reads state (+ parameters) once and writes it once; per step only the action
            # comes in (16 B) and obs + reward + done go out (72 + 4 + 1 B): SURVEY 8(d)'s words, amortised over T
            b_alg = 93.0 + (b_alg - 93.0) / roll
            per_env, src, kernel_name = None, None, "rollout_kernel"
        achieved = n * (roll if roll else 1) * b_alg / (kern_ms * 1e-3) / 1e9
        line = {
            "metric": "env-steps/sec (whole node) at N=2^20 Hummingbird; achieved HBM GB/s",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.fp32 else "f64", "data": "synthetic", "primed_ms": args.prime_ms,
            "config": {"workload": "N=%d %s envs per GPU (%d total), RawControl, sim_freq=200 sim_steps=2 ep_time=5, "
                                   "obs xyz_vxyz_R_omega, thrust noise %s, auto-reset, %s%s%s"
                                   % (n, args.model, total_envs, "off" if args.no_noise else "on (Philox OU)",
                                      "fp32 arithmetic, the fp32 obs tensor is the whole state (reduced precision: outside the parity bar)" if args.fp32
                                      else "fp64-grade split state with its fp32 head aliased to the obs tensor" if env.obs_is_state
                                      else "fp64 state planes + separate obs tensor",
                                      (", per-env randomized params" if args.randomize else "") +
                                      (", quadrotor_multi log-distance reward" if args.reward == "multi" and not args.swarm else "") +
                                      (", swarm worlds of %d agents: neighbour reward + observation terms, quadrotor_multi "
                                       "log-distance reward (own specification, parity-unpinned)" % args.swarm if args.swarm else ""),
                                      (", RCCL obs gather to rank 0" if do_gather else "") +
                                      (", HIP graph of %d single-step launches per replay" % args.graph if args.graph else
                                       ", fused open-loop rollouts of T=%d steps per launch" % roll if roll else "")),
                       "envs_per_gpu": n, "total_envs": total_envs, "obs_dim": D,
                       "parallelism": "env-shard x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": None if per_env is None else per_env * n,
                         "traffic_source": src, "kernel": kernel_name, "kernel_ms": kern_ms,
                         "alg_bytes_per_launch": n * (roll if roll else 1) * b_alg, "alg_bytes_per_env_step": b_alg},
        }
        if extra:
            rates = [env_steps_per_iter * args.steps / e for e in [elapsed] + extra]
            line["repeats"] = {"values": rates, "median": float(np.median(rates)), "min": min(rates), "max": max(rates)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        json_out.write(json.dumps(line) + "\n")
        json_out.flush()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
