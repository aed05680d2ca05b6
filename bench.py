#!/usr/bin/env python
"""bench.py -- env-steps/s of the fused quadrotor step kernel on MI355X (driver contract: one JSON line).

Workload (BASELINE.json metric: "env-steps/sec (whole node) at N=2^20 Hummingbird; achieved HBM GB/s"):
  2^20 Hummingbird (`DefaultQuad`) envs IN ALL, RawControl zero-middle, sim_freq 200, sim_steps 2, ep_time 5
  (ep_len 500 -> 501 steps/episode, in-kernel auto-reset), obs `xyz_vxyz_R_omega` (18 floats), OU thrust noise ON
  (on-device Philox), default reward weights; actions i.i.d. U(-1,1) float32 resident in HBM (a ring of pre-generated
  [N,4] tensors).  One "step" = one launch of the fused kernel over the whole batch (= sim_steps sub-steps + reward +
  obs + done + reset per env).
  --gpus 1: the whole batch on one GPU (it fits: 0.13 GB).
  --gpus N > 1: BASELINE config 4 -- the same 2^20 envs sharded by contiguous index range, 2^20 / N per GPU (131 072
  at N = 8), one process per GPU, and after every step ONE RCCL gather of the packed [obs | reward | done] rows to rank 0
  (`"scaling": "strong"`: the total is fixed).  `--envs-per-gpu M` is the weak-scaling variant (M envs on every GPU).

Launching: `python bench.py --gpus N` starts its own N ranks (fresh child processes, one per GPU, from a parent that
never touches the GPU) unless it is already running under torch.distributed.run (WORLD_SIZE set), in which case
WORLD_SIZE must equal --gpus.  Either way a rank whose GPU does not exist fails the whole run, loudly.

Timing: W untimed warm-up steps, then `--repeats` (5) regions of exactly K steps, each bracketed by barrier +
synchronize, max over ranks per region; `value` is the MEDIAN region's rate and `repeats` lists all.

Extra objects in the JSON line: `roofline` (algorithmic bytes / measured kernel time vs 8 TB/s HBM, plus the PMC traffic
of the same kernel when a profile of the same sources is committed, plus `peak_measured` / `frac_of_measured`: a 1-GiB copy with the
step kernels' access shape timed in the same process right after priming -- SURVEY 8d's "fraction against both nominal and measured-copy
bandwidth"), `cpu_baseline` (the CPU port timed on this box's
host cores on a bounded sample; rank 0, N=1 only), `layouts` (N=1: the median of three timed regions for each other state layout -- `value` is
timed on the opt-in alias layout, the Python class's own default is `shadow`), `staggered_episodes` (N=1: the timed configuration
with desynchronised episodes, i.e. in-kernel resets in every launch; the median of its regions like `value`), `sustained` (N=1: the timed
configuration for about four seconds of consecutive steps in one region: the rate the clocks settle at), `beyond_infinity_cache` (N=1: the
timed configuration at 2^22 envs, where the 256-MB Infinity Cache cannot hold what one step writes for the next to read) and, whenever a collective runs (N > 1, or
GAQ_BENCH_FORCE_DIST=1 on one rank), `phases` (kernel / pack / gather time per step from HIP events on rank 0) and
`variants` (one extra region each without a gather, with the obs-only gather, and with the pack as a separate launch), so
that an N > 1 number can be attributed.  `config.overrides` lists every GAQ_* environment override in effect; a measurement
build running with GAQ_ABLATE prints no `value`.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_ALG = 352           # algorithmic bytes per env-step (DESIGN.md "Byte accounting")
HBM_PEAK_GBPS = 8000.0
TOTAL_ENVS = 1 << 20


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1100, help="timed steps per region (default spans two 501-step episodes)")
    ap.add_argument("--warmup", type=int, default=2000,
                    help="untimed steps first (default 0.1 s of them: the first ~50 ms after idle run ~5 %% slow while the "
                         "clocks ramp)")
    ap.add_argument("--envs", type=int, default=TOTAL_ENVS,
                    help="envs IN ALL (default 2^20, the metric's size); each of N GPUs gets envs / N")
    ap.add_argument("--envs-per-gpu", type=int, default=0,
                    help="weak-scaling variant: this many envs on every GPU (overrides --envs)")
    ap.add_argument("--model", default="DefaultQuad")
    ap.add_argument("--randomize", action="store_true", help="config 3: per-env RelativeSampler(0.2) parameters")
    ap.add_argument("--no-noise", action="store_true")
    ap.add_argument("--gather", default="packed", choices=["packed", "obs", "none"],
                    help="multi-GPU return path per step: ONE gather of the packed [obs|reward|done] rows (default), ONE "
                         "gather of obs alone, or none (policy sharded with the envs)")
    ap.add_argument("--no-gather", action="store_true", help="same as --gather none")
    ap.add_argument("--layout", default="alias", choices=["alias", "shadow", "plain"],
                    help="state layout (gaq_config.obs_state_alias): 'alias' = the fp32 head of the split state lives in the obs "
                         "tensor (1; least traffic; bench default), 'shadow' = library-owned heads + a copy to the obs tensor (2; "
                         "the Python class's default), 'plain' = fp64 planes + write-only obs (0)")
    ap.add_argument("--no-alias", dest="layout", action="store_const", const="plain", help="same as --layout plain")
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
                         "specification, parity-unpinned); --envs stays the number of agents")
    ap.add_argument("--randomize-every", type=int, default=0, metavar="E",
                    help="config 3 with per-episode re-randomisation on the device (dynamics_randomize_every=E)")
    ap.add_argument("--stagger", action="store_true",
                    help="start the envs at uniformly staggered episode phases (resets spread over all steps)")
    ap.add_argument("--prime-ms", type=float, default=150.0,
                    help="milliseconds of scratch GPU work before the warm-up steps (clock ramp after idle); 0 = none")
    ap.add_argument("--repeats", type=int, default=5, help="timed K-step regions; `value` is their median rate")
    ap.add_argument("--sustain-s", type=float, default=4.0,
                    help="N = 1: seconds of consecutive steps in the extra `sustained` region (0 = none); not part of `value`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-layouts", action="store_true", help="skip the extra regions that time the other two state layouts (N = 1)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args(argv)


# ---- parent mode: start one fresh process per GPU -------------------------------------------------------------
def spawn_ranks(n, script=None, argv=None):
    """`python bench.py --gpus N` without torch.distributed.run: this process (which has imported nothing that touches the
    GPU) starts N children -- RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one per GPU -- relays rank 0's JSON line and
    fails if any rank fails.  (`script` / `argv`: what each rank runs; the CPU tests substitute a stand-in rank.)"""
    script = os.path.abspath(__file__) if script is None else script
    argv = sys.argv[1:] if argv is None else list(argv)
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        # GAQ_BENCH_REHEARSAL=1: every rank on GPU 0 over gloo (RCCL cannot put two ranks on one device) -- the whole N > 1 code path
        # of this file on a 1-GPU box; the number it prints means nothing and the line says so
        rehearsal = os.environ.get("GAQ_BENCH_REHEARSAL") == "1"
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0" if rehearsal else str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GAQ_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, script] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        time.sleep(0.2)
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = (r, p.returncode)
                break
    if failed is None:
        for r, p in enumerate(procs):
            if p.returncode != 0:
                failed = (r, p.returncode)
                break
    if failed is not None:
        for p in procs:                      # the others sit in a rendezvous or a collective that will never complete
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=15)
            except Exception:
                p.kill()
        sys.stderr.write("bench.py: rank %d of %d exited with code %s -- no result\n" % (failed[0], n, failed[1]))
        sys.exit(failed[1] if isinstance(failed[1], int) and failed[1] > 0 else 1)
    out = procs[0].stdout.read().decode("utf-8", "replace").strip().splitlines()
    lines = [ln for ln in out if ln.startswith("{")]
    if not lines:
        sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
        sys.exit(1)
    sys.stdout.write(lines[-1] + "\n")
    sys.stdout.flush()


# ---- helpers --------------------------------------------------------------------------------------------------
def source_sha256():
    """hash of the kernel sources: PMC profiles record it, and a profile of other sources is not reported as this
    kernel's traffic"""
    h = hashlib.sha256()
    for rel in ("gym_art_amd/csrc/gaq_kernels.hpp", "gym_art_amd/csrc/quad_core.hpp"):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def pmc_traffic_per_env_step(kernel_key):
    """HBM bytes per env-step of the step kernel as measured with rocprofv3 PMC counters (separate --pmc passes,
    gfx950 FETCH_SIZE correction, tools/pmc_summary.py) and committed under profiles/pmc_index.json:
    {kernel_key: {"file": ..., "source_sha256": ...}}.  Returns (bytes or None, source or None, stale flag)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_index.json")) as f:
            ent = json.load(f).get(kernel_key)
        if not ent:
            return None, None, False
        with open(os.path.join(ROOT, "profiles", ent["file"])) as f:
            per_env = float(json.load(f)["_derived"]["traffic_bytes_per_env_step"])
    except Exception:
        return None, None, False
    if ent.get("source_sha256") != source_sha256():
        return None, "profiles/%s (STALE: taken from other kernel sources, not reported)" % ent["file"], True
    return per_env, "profiles/" + ent["file"], False


def cpu_baseline(seconds_budget=12.0):
    """The CPU port on a bounded sample of the same workload (Hummingbird, thrust noise on, auto-reset):
    `value` = oracle/cpu_native.cpp -- the kernel arithmetic header compiled by g++, OpenMP over the batch -- on all
    usable cores; beside it the NumPy oracle (one core, and one env per call like the reference)."""
    import subprocess
    env = dict(os.environ, OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    env.pop("OMP_NUM_THREADS", None)            # the compiled port sets its own thread count per run

    def run(mod, *a):
        p = subprocess.run([sys.executable, "-m", mod] + [str(x) for x in a], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.DEVNULL, text=True)
        return json.loads(p.stdout.strip().splitlines()[-1])

    from oracle.cpu_native import usable_cores
    host, aff, quota = usable_cores()
    t_each = max(seconds_budget / 5.0, 1.0)
    n_native = 1 << 18
    tries = sorted({aff, max(1, min(aff, 16))} | ({max(1, int(round(quota)))} if quota else set()))
    native = [run("oracle.cpu_native", "--envs", n_native, "--seconds", "%.2f" % t_each, "--threads", t) for t in tries]
    best = max(native, key=lambda r: r["env_steps_per_s"])
    one = run("oracle.cpu_native", "--envs", 1 << 16, "--seconds", "%.2f" % min(t_each, 2.0), "--threads", 1)
    env["OMP_NUM_THREADS"] = "1"
    npy = run("oracle.cpu_worker", "--envs", 16384, "--seconds", "%.2f" % min(t_each, 2.5))
    loop = run("oracle.cpu_worker", "--envs", 1, "--seconds", "%.2f" % min(t_each, 2.0))
    ref_rate = None
    try:
        with open(os.path.join(ROOT, "tests", "golden", "reference_timing.json")) as f:
            ref_rate = float(json.load(f)["env_steps_per_s"]["raw"])
    except Exception:
        pass
    return {"value": best["env_steps_per_s"], "unit": "env-steps/s", "cores": best["threads"], "kind": "port",
            "reference_equivalent": "one_env_per_call",
            "reading": "`value` is this repo's own arithmetic header compiled for the CPU and batched (a port: the fastest CPU form of the "
                       "same work); the entry closest to what the REFERENCE does on this host is `one_env_per_call` (NumPy fp64, one env "
                       "per Python call); `reference_in_build_container` is the unmodified reference itself but timed on ANOTHER host "
                       "(the build container: it cannot travel to this box)",
            "sample": "oracle/cpu_native.cpp (the kernel's arithmetic header compiled by g++ -O3, fp64, OpenMP over the batch): "
                      "N=%d Hummingbird envs, noise on, auto-reset, %d batch steps in %.1f s on %d threads; host reports %d "
                      "cpus, %d in the affinity mask, cgroup quota %s; thread counts tried: %s"
                      % (n_native, best["steps"], best["seconds"], best["threads"], host, aff,
                         "none" if quota is None else "%.1f" % quota,
                         ", ".join("%d -> %.3g" % (r["threads"], r["env_steps_per_s"]) for r in native)),
            "compiled_one_core": {"value": one["env_steps_per_s"], "unit": "env-steps/s", "cores": 1,
                                  "what": "the same compiled port on one thread, N=65536"},
            "numpy_one_core": {"value": npy["env_steps_per_s"], "unit": "env-steps/s", "cores": 1,
                               "what": "oracle/quad_oracle.py (vectorised NumPy fp64), N=16384, one process"},
            "one_env_per_call": {"value": loop["env_steps_per_s"], "unit": "env-steps/s", "cores": 1,
                                 "what": "the NumPy oracle stepped like the reference: N=1 per call in a Python loop, on this host"},
            "reference_in_build_container": {"value": ref_rate, "unit": "env-steps/s", "cores": 1,
                                             "what": "unmodified reference QuadrotorEnv.step, RawControl, measured on ANOTHER host: the "
                                                     "build container (8 vCPU), where /root/reference exists "
                                                     "(tests/golden/reference_timing.json)"}}


# ---- one rank -------------------------------------------------------------------------------------------------
def worker(args):
    import numpy as np
    # stdout carries exactly ONE JSON line: everything else written to fd 1 by native libraries (RCCL prints
    # its banner there) goes to stderr
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: launch `python bench.py --gpus N` (it starts its own ranks) or "
                         "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`\n" % (args.gpus, world))
        sys.exit(2)

    if os.environ.get("GAQ_BENCH_REHEARSAL") == "1":      # every rank on GPU 0 (also under torch.distributed.run, which numbers them)
        local = 0
        os.environ["LOCAL_RANK"] = "0"
    import torch
    import torch.distributed as dist
    ndev = torch.cuda.device_count()
    if local >= ndev:
        sys.stderr.write("bench.py: rank %d needs GPU %d but this box has %d GPU(s): cannot run --gpus %d here\n"
                         % (rank, local, ndev, args.gpus))
        sys.exit(3)
    from gym_art_amd.sharding import ShardedQuadrotorEnv

    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_dist = os.environ.get("GAQ_BENCH_FORCE_DIST") == "1"     # exercise the RCCL path on a single rank
    rehearsal = os.environ.get("GAQ_BENCH_REHEARSAL") == "1" and world > 1     # gloo ranks sharing GPU 0 (see spawn_ranks)
    if world > 1 or force_dist:
        if force_dist and world == 1:
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29511")):
                os.environ.setdefault(k, v)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    rccl_ranks = dist.get_world_size() if (dist.is_initialized() and not rehearsal) else 0

    if args.envs_per_gpu:
        n, scaling = args.envs_per_gpu, "weak"
    else:
        if args.envs % world:
            sys.stderr.write("bench.py: --envs %d is not divisible by %d GPUs\n" % (args.envs, world))
            sys.exit(2)
        n, scaling = args.envs // world, "strong"
    total_envs = n * world
    kw = dict(dynamics_params=args.model, ep_time=5, sim_freq=200., sim_steps=2, seed=0, auto_reset=True,
              thrust_noise="off" if args.no_noise else "philox",
              alias_obs={"alias": True, "shadow": None, "plain": False}[args.layout])
    if args.randomize:
        kw["dyn_sampler_1"] = {"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"}
    if args.randomize_every:
        kw["dynamics_randomize_every"] = args.randomize_every
    if args.reward != "quadrotor":
        kw.update(reward=args.reward)
    if args.fp32:
        kw.update(precision="fp32")
    if args.swarm:
        kw.update(reward="multi", swarm=dict(agents=args.swarm))
    sharded = ShardedQuadrotorEnv(total_envs, always_collective=force_dist, **kw)      # contiguous global index range per rank
    assert (sharded.first, sharded.count) == (rank * n, n)
    env = sharded.env
    D = env.obs_dim
    ring = 8
    gen = torch.Generator(device=dev)
    gen.manual_seed(rank)
    actions = [torch.rand((n, 4), device=dev, generator=gen) * 2 - 1 for _ in range(ring)]
    gather = "none" if args.no_gather else args.gather
    if world == 1 and not force_dist:
        gather = "none"
    sharded.reset()
    if args.stagger:
        st = env.get_state()                       # tick plane: episode phases spread uniformly over an episode
        st[37] = np.arange(n) % (env.ep_len + 1)
        env.set_state(st)

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

    # persistent buffers: the pointer / stream look-ups of step_dev are done once per action tensor (QuadrotorEnv.bind_step)
    bound = [env.bind_step(a, sharded.obs, sharded.reward, sharded.done) for a in actions]

    def one_step(t, mode=None):
        mode = gather if mode is None else mode
        if graph is not None:
            graph.replay()
        elif roll:
            env.step_many_dev(acts_T, obs_T, rew_T, done_T)
        else:
            bound[t % ring]()
            if mode == "packed":
                sharded.gather_packed(done_as_float=True)     # the gathered rows ARE the result: views, no extra pass on rank 0
            elif mode == "obs":
                sharded.gather_obs()

    # Bring the GPU out of idle before anything is counted: the first ~50 ms of work after idle run ~5 % slow while the
    # clocks ramp.  This is a scratch fill loop, not steps of the benchmark; the W warm-up steps and the K timed steps
    # below are untouched.  `--prime-ms 0` switches it off; the line reports what was done.
    scratch = torch.empty(64 << 20, dtype=torch.float32, device=dev) if args.prime_ms > 0 else None

    def prime_and_warm_up(ms=None):
        """Before EVERY timed region (the repeats are regions of their own: each gets what the first one gets): the scratch loop, then the W
        untimed warm-up steps.  With the driver's K = 20 a region is 1 ms of work between two synchronisations, and the clocks sag over a
        train of such bursts: the later regions of a run read 5-10 % slower than the first without this."""
        if scratch is not None:
            p0 = time.perf_counter()
            while (time.perf_counter() - p0) * 1e3 < (args.prime_ms if ms is None else ms):
                for _ in range(8):
                    scratch.add_(1.0)
                torch.cuda.synchronize()
        for t in range(args.warmup):
            one_step(t)
    prime_and_warm_up()

    def measured_copy_bandwidth():
        """"also measure an on-box copy kernel and quote the fraction against both nominal and measured-copy bandwidth".  A 1-GiB
        copy with the step kernels' access shape (one 16-byte load + one 16-byte store per lane: libgaq's gaq_hbm_copy_dev), timed with HIP
        events on the launch stream right after priming: 2 x bytes / time, best of five launches after two untimed ones."""
        from gym_art_amd import _lib
        nbytes = 1 << 30
        try:
            src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        except Exception:
            return None
        src.zero_()
        st = torch.cuda.current_stream(dev).cuda_stream
        lib = _lib.load()
        best = None
        for k in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(lib.gaq_hbm_copy_dev(_lib.ptr(dst), _lib.ptr(src), nbytes, st))
            e1.record()
            torch.cuda.synchronize()
            if k >= 2:
                ms = e0.elapsed_time(e1)
                best = ms if best is None else min(best, ms)
        del src, dst
        return 2.0 * nbytes / (best * 1e-3) / 1e9
    copy_gbps = measured_copy_bandwidth() if (world == 1 and not force_dist) else None

    def timed_region(step_fn, steps, bracket=True):
        """exactly `steps` calls of step_fn(t) between barrier + synchronize on both sides; the max over ranks of the wall time, and
        (bracket) the device time between one pair of HIP events on the launch stream around them"""
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if bracket:
            ev[0].record()
        for t in range(steps):
            step_fn(t)
        if bracket:
            ev[1].record()
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        el = time.perf_counter() - t0
        if dist.is_initialized():
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el, (ev[0].elapsed_time(ev[1]) / steps if bracket else None)

    # HIP events on the launch stream (torch's current stream = the stream step_dev launches on).  Without a gather one
    # pair brackets a whole timed region (per-launch pairs put barrier packets between the kernels); with a gather the
    # region also holds the collectives, so the phases are timed with events per launch (first region only).
    per_launch = gather != "none"
    regions, kern_ms_all = [], []
    phases = None
    for rep in range(max(1, args.repeats)):
        if rep > 0:
            prime_and_warm_up(args.prime_ms / 5)      # (the GPU is not idle here, only bursty: a fifth of the first dose keeps the clocks up)
        if per_launch and rep == 0:
            E = lambda: torch.cuda.Event(enable_timing=True)
            evs = [(E(), E(), E(), E()) for _ in range(args.steps)]

            def phased_step(t):
                e0, e1, e2, e3 = evs[t]
                e0.record()
                bound[t % ring]()
                e1.record()
                if gather == "packed" and not sharded.fused_rows:
                    env.pack_rows_dev(sharded.obs, sharded.reward, sharded.done, sharded._rows[:sharded.count])
                e2.record()
                if gather == "packed":
                    sharded.gather_packed(done_as_float=True, packed=True)
                else:
                    sharded.gather_obs()
                e3.record()
            el, _ = timed_region(phased_step, args.steps, bracket=False)
            regions.append(el)
            k_ms = float(np.sum([a.elapsed_time(b) for a, b, _, _ in evs])) / args.steps
            kern_ms_all.append(k_ms)
            phases = {"kernel_ms": k_ms,
                      "pack_ms": float(np.sum([b.elapsed_time(c) for _, b, c, _ in evs])) / args.steps if not sharded.fused_rows else 0.0,
                      "gather_ms": float(np.sum([c.elapsed_time(d) for _, _, c, d in evs])) / args.steps,
                      "what": "HIP events on rank 0's launch stream around each step's launches, first timed region; pack_ms = 0: the "
                              "packed rows are written by the step launch itself (gaq_set_packed_rows_dev)"
                              if sharded.fused_rows else "HIP events on rank 0's launch stream around each step's launches, first timed region"}
        else:
            el, k_ms = timed_region(one_step, args.steps, bracket=not per_launch)
            regions.append(el)
            if k_ms is not None:
                kern_ms_all.append(k_ms)
    env.check_finite()
    elapsed = float(np.median(regions))
    kern_ms = float(np.median(kern_ms_all))

    # what the collective costs, by leaving parts of it out: one more region each (every rank takes part)
    variants = None
    if per_launch:
        variants = {}

        def region_of(mode, label, what):
            el, _ = timed_region(lambda t: one_step(t, mode), args.steps, bracket=False)
            variants[label] = {"value": total_envs * args.steps / el, "ms_per_step": el / args.steps * 1e3, "what": what}
        region_of("none", "gather_none", "no collective: the policy is sharded with the envs (pure data parallelism)")
        region_of("obs", "gather_obs", "ONE gather of the obs tensor alone ([count,%d] fp32)" % D)
        if gather == "packed" and sharded.fused_rows:
            sharded.set_fused_rows(False)
            region_of("packed", "packed_unfused", "the packed gather with the rows made by a separate pack launch between the step and the "
                                                  "collective (round 2's path)")
            sharded.set_fused_rows(True)

    # N = 1: the other state layouts, the better of two regions each (the Python class's own default is `shadow`; `value` is timed on --layout)
    def median_region(step_fn, reps=3):
        """the MEDIAN (by wall time) of `reps` timed regions, like `value` (not "the better of two")"""
        runs = sorted(timed_region(step_fn, args.steps) for _ in range(reps))
        return runs[len(runs) // 2]

    layouts = staggered = big = sustained = None
    plain_run = not (args.swarm or args.no_noise or args.fp32 or args.reward != "quadrotor" or args.randomize_every or args.randomize or
                     args.model != "DefaultQuad" or roll or args.graph or args.stagger)
    if world == 1 and not force_dist and plain_run and not args.no_layouts:
        layouts = {}
        for name in ("alias", "shadow", "plain"):
            if name == args.layout:          # the configuration `value` is timed on: its own (median) numbers, not one more region
                e2, el, k_ms = env, elapsed, kern_ms
            else:
                e2 = ShardedQuadrotorEnv(total_envs, **dict(kw, alias_obs={"alias": True, "shadow": None, "plain": False}[name]))
                e2.reset()
                b2 = [e2.env.bind_step(a, e2.obs, e2.reward, e2.done) for a in actions]
                for t in range(max(min(args.warmup, 1000), 500)):     # (creating the env left the GPU idle: bring it back to steady state)
                    b2[t % ring]()
                el, k_ms = median_region(lambda t: b2[t % ring]())
            key = "default_" + name
            per_env, src, stale = pmc_traffic_per_env_step(key) if n == TOTAL_ENVS else (None, None, False)
            ent = {"us_per_step": el / args.steps * 1e6, "kernel_us": k_ms * 1e3, "value": total_envs * args.steps / el,
                   "frac": n * B_ALG / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                   "traffic_bytes_per_env_step": per_env, "traffic_stale": stale,
                   "what": {"alias": "split state, fp32 heads IN the caller's obs tensor (opt-in: alias_obs=True; what `value` is timed on)",
                            "shadow": "split state, library-owned heads + a copy to the caller's obs tensor: the class default of "
                                      "gym_art_amd.QuadrotorEnv (no contract on the obs tensor)",
                            "plain": "fp64 state planes + write-only obs tensor (alias_obs=False)"}[name]}
            if per_env is not None:
                ent["measured_frac"] = per_env * n / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
            layouts[name] = ent
            ent["regions"] = len(regions) if e2 is env else 3      # (the timed layout: the median of the line's regions; the others: the median of three)
            if e2 is not env:
                e2.env.close()
        # ... and the timed layout once more with the episodes DESYNCHRONISED (phases spread uniformly over an episode, as in a sampler that
        # has been running for a while): every launch then resets n / (ep_len + 1) envs in-kernel, whatever K is -- the default workload
        # resets all envs together at steps 501, 1002, ..., which a short timed region never contains
        e3 = ShardedQuadrotorEnv(total_envs, **kw)
        e3.reset()
        st = e3.env.get_state()
        st[37] = np.arange(n) % (e3.env.ep_len + 1)
        e3.env.set_state(st)
        b3 = [e3.env.bind_step(a, e3.obs, e3.reward, e3.done) for a in actions]
        for t in range(max(min(args.warmup, 1000), 500)):
            b3[t % ring]()
        el, k_ms = median_region(lambda t: b3[t % ring](), reps=max(3, args.repeats))
        staggered = {"us_per_step": el / args.steps * 1e6, "kernel_us": k_ms * 1e3, "value": total_envs * args.steps / el,
                     "frac": n * B_ALG / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "resets_per_step": n / (e3.env.ep_len + 1.0),
                     "what": "the timed configuration with episode phases spread uniformly (st[tick] = i mod (ep_len + 1)): every step resets "
                             "n / (ep_len + 1) envs inside the launch; the median of %d regions" % max(3, args.repeats), "regions": max(3, args.repeats)}
        e3.env.close()
        # ... and the timed configuration held for seconds instead of the K-step burst (K = 20 is 1 ms of GPU work): the rate the clocks
        # settle at under this load -- and something a coarse outside sampler of GPU activity (the driver's is 5 s) can see
        sustained = None
        if args.sustain_s > 0:
            ks = max(args.steps, int(args.sustain_s / (kern_ms * 1e-3)))
            el_s, k_ms_s = timed_region(one_step, ks)
            sustained = {"steps": ks, "seconds": el_s, "us_per_step": el_s / ks * 1e6, "kernel_us": k_ms_s * 1e3, "value": total_envs * ks / el_s,
                         "frac": n * B_ALG / (k_ms_s * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "what": "the timed configuration for %d consecutive steps (about %.0f s) in ONE region, no re-priming: sustained clocks" % (ks, args.sustain_s)}
        # ... and the same kernel where the Infinity Cache cannot help: 2^22 envs (state 0.55 GB, 1.16 GB moved per step against a 256-MB cache).
        # At the metric's own N = 2^20 part of what a step reads is what the previous step wrote and is still in that cache; this is the
        # figure that leans on HBM alone ("the 2^20 point alone does not prove HBM")
        big = None
        if n == TOTAL_ENVS:
            try:
                nb = 4 * TOTAL_ENVS
                e4 = ShardedQuadrotorEnv(nb, **kw)
                e4.reset()
                a4 = [torch.rand((nb, 4), device=dev, generator=gen) * 2 - 1 for _ in range(2)]
                b4 = [e4.env.bind_step(a, e4.obs, e4.reward, e4.done) for a in a4]
                k4 = max(50, min(args.steps, 300))
                for t in range(k4):
                    b4[t % 2]()
                runs = sorted(timed_region(lambda t: b4[t % 2](), k4) for _ in range(3))
                el4, k_ms4 = runs[1]
                big = {"envs": nb, "us_per_step": el4 / k4 * 1e6, "kernel_us": k_ms4 * 1e3, "value": nb * k4 / el4,
                       "frac": nb * B_ALG / (k_ms4 * 1e-3) / 1e9 / HBM_PEAK_GBPS, "steps": k4, "regions": 3,
                       "what": "the timed configuration at 4 x the batch (state 0.55 GB: far above the 256-MB Infinity Cache); median of three regions"}
                e4.env.close()
                del e4, a4, b4
            except Exception as exc:        # (a box without the memory for it: the line says so instead of failing)
                big = {"error": repr(exc)}

    if rank == 0:
        env_steps_per_iter = total_envs * (roll if roll else 1)
        value = env_steps_per_iter * args.steps / elapsed
        b_alg = B_ALG + (128 if (args.randomize or args.model == "RandomQuad") else 0) + (24 * (args.swarm - 1) if args.swarm else 0)   # + neighbour obs words
        plain_kernel = not (args.swarm or args.no_noise or args.fp32 or args.reward != "quadrotor" or args.randomize_every)
        key = None
        if plain_kernel and n == TOTAL_ENVS:
            if not args.randomize and args.model == "DefaultQuad":
                key = "default_" + ("plain", "alias", "shadow")[env.state_layout]
            elif args.randomize and args.model == "Crazyflie":
                key = "c3_" + ("plain", "alias", "shadow")[env.state_layout]
        per_env, src, stale = pmc_traffic_per_env_step(key) if key else (None, None, False)
        kernel_name = "step_kernel<%d>" % env.launch_variant
        if roll and not args.graph:
            # a fused T-step launch reads state (+ parameters) once and writes it once; per step only the action
            # comes in (16 B) and obs + reward + done go out (72 + 4 + 1 B): SURVEY 8(d)'s words, amortised over T
            b_alg = 93.0 + (b_alg - 93.0) / roll
            per_env, src, kernel_name = None, None, "rollout_kernel"
        achieved = n * (roll if roll else 1) * b_alg / (kern_ms * 1e-3) / 1e9
        how = ("fp32 arithmetic, the fp32 obs tensor is the whole state (reduced precision: outside the parity bar)" if args.fp32
               else "fp64-grade split state with its fp32 head aliased to the obs tensor (layout 'alias', opt-in; the Python class's "
                    "default 'shadow' is timed beside it in `layouts`)" if env.state_layout == 1
               else "fp64-grade split state with library-owned heads + a copy to the obs tensor (layout 'shadow', the Python class's default)"
               if env.state_layout == 2 else "fp64 state planes + separate obs tensor (layout 'plain')")
        extras = (", per-env randomized params" if args.randomize else ", one random quadrotor per env (device sampler)" if args.model == "RandomQuad" else "") + \
                 (", re-randomised on the device every %d episodes" % args.randomize_every if args.randomize_every else "") + \
                 (", staggered episode phases" if args.stagger else "") + \
                 (", quadrotor_multi log-distance reward" if args.reward == "multi" and not args.swarm else "") + \
                 (", swarm worlds of %d agents: neighbour reward + observation terms, quadrotor_multi log-distance reward "
                  "(own specification, parity-unpinned)" % args.swarm if args.swarm else "")
        coll = {"packed": ", ONE RCCL gather per step of the packed [obs|reward|done] rows ([count,%d] fp32, written by the step launch) "
                          "to rank 0" % (D + 2),
                "obs": ", ONE RCCL gather per step of the obs tensor to rank 0", "none": ""}[gather] + \
               (", HIP graph of %d single-step launches per replay" % args.graph if args.graph else
                ", fused open-loop rollouts of T=%d steps per launch" % roll if roll else "")
        if rehearsal:
            coll = coll.replace("RCCL", "gloo") + " -- REHEARSAL: %d gloo ranks sharing ONE GPU, the rate means nothing" % world
        if world > 1 and scaling == "strong":
            shape = "N=%d %s envs in all (BASELINE config 4), sharded %d per GPU over %d GPUs" % (total_envs, args.model, n, world)
        else:
            shape = "N=%d %s envs per GPU (%d in all)" % (n, args.model, total_envs)
        # every GAQ_* override in effect goes into the line: a variant library, a forced kernel variant or an ablation must
        # not print a normal-looking result
        overrides = {k: v for k, v in sorted(os.environ.items())
                     if k.startswith("GAQ_") and k not in ("GAQ_BENCH_SPAWNED",) and v != ""}
        ablated = overrides.get("GAQ_ABLATE", "0") not in ("0", "")
        line = {
            "metric": "env-steps/sec (whole node) at N=2^20 Hummingbird; achieved HBM GB/s",
            "value": None if ablated else value, "unit": "env-steps/s", "n_gpus": world, "rccl_ranks": rccl_ranks, **({"rehearsal": "gloo ranks sharing one GPU: not a measurement"} if rehearsal else {}), "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32" if args.fp32 else "f64", "data": "synthetic", "primed_ms": args.prime_ms,
            "primed_what": "scratch GPU work (not steps) for primed_ms before the first timed region and primed_ms / 5 before each later one, each followed by the W warm-up steps; outside the timed regions",
            "config": {"workload": "%s, RawControl, sim_freq=200 sim_steps=2 ep_time=5, obs xyz_vxyz_R_omega, thrust noise %s, "
                                   "auto-reset, %s%s%s" % (shape, "off" if args.no_noise else "on (Philox OU)", how, extras, coll),
                       "envs_per_gpu": n, "total_envs": total_envs, "obs_dim": D, "gather": gather,
                       "layout": ("plain", "alias", "shadow")[env.state_layout], "class_default_layout": "shadow",
                       "kernel_variant": env.kernel_variant, "launch_variant": env.launch_variant, "parallelism": "env-shard x%d" % world, "overrides": overrides,
                       "measurement_build": bool(env._lib.gaq_is_diag_build())},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "algorithmic_frac": achieved / HBM_PEAK_GBPS,
                         "traffic": None if per_env is None else per_env * n,
                         "traffic_source": src, "traffic_stale": stale, "kernel": kernel_name, "kernel_ms": kern_ms,
                         "alg_bytes_per_launch": n * (roll if roll else 1) * b_alg, "alg_bytes_per_env_step": b_alg,
                         "what": "frac = algorithmic_frac = ALGORITHMIC bytes (SURVEY 8d: 352 B per env-step whatever the layout) / kernel "
                                 "time / peak -- NOT delivered bandwidth; measured_* = PMC bytes actually moved (an earlier rocprofv3 --pmc run "
                                 "of the same kernel sources) / this run's kernel time"},
        }
        if ablated:
            line["ablated"] = "GAQ_ABLATE=%s in a measurement build: the physics are wrong by construction, only ms_per_step / kernel_ms mean anything" % overrides["GAQ_ABLATE"]
        if per_env is not None:
            mg = per_env * n / (kern_ms * 1e-3) / 1e9
            line["roofline"].update(measured_gbps=mg, measured_frac=mg / HBM_PEAK_GBPS, traffic_bytes_per_env_step=per_env)
        if copy_gbps:
            # the same two fractions against what a plain 16-byte-per-lane copy reaches on THIS box in THIS run
            line["roofline"].update(peak_measured=copy_gbps, frac_of_measured=achieved / copy_gbps,
                                    peak_measured_what="gaq_hbm_copy_dev: 1-GiB copy, one 16-byte load + one 16-byte store per lane, read + written "
                                                       "bytes / HIP-event time, best of five launches right after priming, same process")
            if per_env is not None:
                line["roofline"]["measured_frac_of_measured"] = mg / copy_gbps
        rates = [env_steps_per_iter * args.steps / e for e in regions]
        line["repeats"] = {"values": rates, "median": float(np.median(rates)), "min": min(rates), "max": max(rates),
                           "what": "rate of every timed K-step region in order; `value` is the median"}
        if phases is not None:
            line["phases"] = phases
        if variants is not None:
            line["variants"] = variants
            # the SCALABLE form of the path is the one without a collective (policy sharded with the envs): beside `value` so that a first
            # 8-GPU reading below the 1-GPU number is read as rank 0's xGMI ingest, not as a kernel regression
            line["value_data_parallel"] = variants["gather_none"]["value"]
            if world > 1:
                row_bytes = (D + 2) * 4 if gather == "packed" else D * 4
                ingest = (world - 1) * n * row_bytes
                per_link = n * row_bytes / 153e9                     # every shard rides its own xGMI link into rank 0 (7 x ~153 GB/s)
                line["expected"] = {
                    "rank0_ingest_bytes_per_step": ingest, "xgmi_link_gbps": 153, "links_into_rank0": min(world - 1, 7),
                    "gather_floor_ms": per_link * 1e3, "gather_floor_value": total_envs / max(per_link, kern_ms * 1e-3),
                    "reading": "`value` includes ONE gather per step of every shard's rows to rank 0: %d bytes arrive there per step, each shard "
                               "over its own xGMI link (%.0f us at the link's peak; RCCL's grouped send/recv adds launch and protocol time on "
                               "top) -- the step kernel itself is phases.kernel_ms.  `value_data_parallel` (variants.gather_none) is the same "
                               "run without the collective: what scales with the number of GPUs" % (ingest, per_link * 1e6)}
        if layouts is not None:
            line["layouts"] = layouts
            line["staggered_episodes"] = staggered
            if sustained is not None:
                line["sustained"] = sustained
            if big is not None:
                line["beyond_infinity_cache"] = big
                if "frac" in big:
                    line["roofline"]["frac_beyond_infinity_cache"] = big["frac"]
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
            if not ablated and line["cpu_baseline"].get("value"):
                line["vs_cpu_baseline"] = value / line["cpu_baseline"]["value"]     # (vs_baseline stays null: BASELINE.md publishes no number)
        json_out.write(json.dumps(line) + "\n")
        json_out.flush()
    if dist.is_initialized():
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus < 1:
        sys.stderr.write("bench.py: --gpus must be >= 1\n")
        sys.exit(2)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args.gpus)
    worker(args)


if __name__ == "__main__":
    main()
