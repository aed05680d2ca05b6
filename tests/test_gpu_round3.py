"""-m gpu: what round 3 added -- the packed multi-GPU rows written by the step launch itself, the one-launch graph-safe step
counter, the sync-free torch step path, kernel selection reported by the handle against gaq_plan, the refused ablation variable,
RCCL under the driver's eyes (bench.py's N > 1 path on one rank)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from gym_art_amd import _lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [
    # (constructor kwargs, what it exercises)
    (dict(alias_obs=True), "alias layout: heads are the observation (step_kernel<20 | size bits>)"),
    (dict(alias_obs=None), "library-owned heads + copy"),
    (dict(alias_obs=False), "fp64 planes"),
    (dict(alias_obs=True, precision="fp32"), "fp32 state"),
    (dict(alias_obs=True, dynamics_params="Crazyflie"), "motor lag, mixed residual rows"),
    (dict(alias_obs=True, dynamics_params="Crazyflie", dyn_sampler_1={"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"}),
     "per-env parameters"),
    (dict(obs_repr="xyz_vxyz_R_omega_acc_act"), "packed observation, 25 words (27-word rows: 4-byte aligned only)"),
    (dict(obs_repr="xyzr_vxyzr_R_omega_h", alias_obs=False), "plain layout, 19 words"),
    (dict(sense_noise="default", init_random_state=True), "sensor noise (F_PACK)"),
    (dict(resample_goal=True), "generic-lite kernel: rows by the pack launch inside gaq_step_dev"),
    (dict(raw_control=False), "Mellinger"),
]


@pytest.mark.parametrize("n", [777, 4096])
def test_step_launch_writes_the_packed_rows(n):
    """gaq_set_packed_rows_dev: the [obs | reward | done] rows that leave with the step launch are bit for bit what gaq_pack_rows_dev
    makes of the step's outputs -- every layout, ragged last tile, in-kernel resets (done = 1 rows), nothing written past the end."""
    import torch
    from gym_art_amd import QuadrotorEnv
    dev = torch.device("cuda", 0)
    for kw, what in CASES:
        env = QuadrotorEnv(num_envs=n, ep_time=0.05, seed=4, **kw)
        D = env.obs_dim
        obs = torch.empty((n, D), device=dev); rew = torch.empty(n, device=dev); done = torch.empty(n, dtype=torch.uint8, device=dev)
        fused = torch.full((n + 3, D + 2), -7.0, device=dev)            # three canary rows behind the last env
        ref = torch.full((n, D + 2), -1.0, device=dev)
        env.reset_dev(obs)
        env.set_packed_rows(fused[:n])
        seen_done = 0
        for t in range(7):                                               # ep_len 5: every env reports done on the sixth step
            a = torch.rand((n, 4), device=dev) * 2 - 1
            if not env.raw_control:
                a = a * 0
            env.step_dev(a, obs, rew, done)
            env.pack_rows_dev(obs, rew, done, ref)
            torch.cuda.synchronize()
            assert torch.equal(fused[:n], ref), (what, t)
            assert torch.equal(fused[:n, :D], obs) and torch.equal(fused[:n, D + 1], done.float()), (what, t)
            seen_done += int(done.sum().item())
        assert seen_done == n, what
        assert bool((fused[n:] == -7.0).all()), what
        env.set_packed_rows(None)                                        # unregistered: the buffer is left alone
        fused.fill_(-3.0)
        env.step_dev(torch.zeros((n, 4), device=dev), obs, rew, done)
        torch.cuda.synchronize()
        assert bool((fused == -3.0).all()), what
        env.close()


def test_packed_rows_do_not_change_the_step():
    """Registering the row buffer changes nothing else: trajectories (noise, resets) are bit-identical with and without it."""
    import torch
    from gym_art_amd import QuadrotorEnv
    dev = torch.device("cuda", 0)
    n = 131072
    kw = dict(num_envs=n, ep_time=0.1, seed=11, alias_obs=True)
    a_env, b_env = QuadrotorEnv(**kw), QuadrotorEnv(**kw)
    assert a_env.kernel_variant == b_env.kernel_variant
    oa = torch.empty((n, 18), device=dev); ob = torch.empty((n, 18), device=dev)
    ra = torch.empty(n, device=dev); rb = torch.empty(n, device=dev)
    da = torch.empty(n, dtype=torch.uint8, device=dev); db = torch.empty(n, dtype=torch.uint8, device=dev)
    rows = torch.empty((n, 20), device=dev)
    a_env.reset_dev(oa); b_env.reset_dev(ob)
    b_env.set_packed_rows(rows)
    for t in range(25):
        act = torch.rand((n, 4), device=dev) * 2 - 1
        a_env.step_dev(act, oa, ra, da); b_env.step_dev(act, ob, rb, db)
    torch.cuda.synchronize()
    assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db)
    assert torch.equal(rows[:, :18], ob) and torch.equal(rows[:, 18], rb)
    a_env.close(); b_env.close()


@pytest.mark.parametrize("n,K,force_nt", [(65536, 60, False), (1 << 20, 24, False), (1 << 20, 24, True), (100, 30, False)])
def test_graph_replays_with_the_one_launch_step_counter(n, K, force_nt):
    """Graph-safe mode is ONE kernel node per step: the F_CTR twin of the step kernel reads the device-resident counter and checks in
    with non-returning atomics.  Replays equal eager steps bit for bit at one wave per SIMD (65 536 envs), with a single partly filled
    workgroup, at 2^20 envs (no twin at that size: the one-thread bump launch follows the step) and at 2^20 envs with the small-batch
    kernel FORCED (GAQ_NT=1) -- 16 384 waves scheduled in many rounds: a wave that starts late must still see THIS launch's index; the
    counter read back between replays is exact; and
    a kernel that reads the counter's first word alone (here: the F_ROWS twin, once packed rows are registered, followed by the bump
    launch) follows F_CTR launches correctly (the spread check-ins are folded first)."""
    import torch
    from gym_art_amd import QuadrotorEnv
    kw = dict(num_envs=n, ep_time=0.1, seed=21, alias_obs=True)
    if force_nt:
        os.environ["GAQ_NT"] = "1"
    try:
        eager, graphed = QuadrotorEnv(**kw), QuadrotorEnv(**kw)
    finally:
        os.environ.pop("GAQ_NT", None)
    F_NT, F_CTR = 256, 8192
    from tests.test_plan_cpu import base_cfg, plan
    p = plan(base_cfg(n, noise=1, obs_state_alias=1, auto_reset=1), cus=torch.cuda.get_device_properties(0).multi_processor_count)
    twin = bool(graphed.kernel_variant & F_NT)             # the self-counting twins exist for the small-batch kernels
    assert twin == (force_nt or n <= 131072)
    if not force_nt:
        assert p.step_variant == graphed.kernel_variant and (p.ctr_variant == (p.step_variant | F_CTR) if twin else p.ctr_variant == -1)
    dev = torch.device("cuda")
    o_e = torch.empty((n, 18), device=dev); r_e = torch.empty(n, device=dev); d_e = torch.empty(n, dtype=torch.uint8, device=dev)
    o_g = torch.empty((n, 18), device=dev); r_g = torch.empty(n, device=dev); d_g = torch.empty(n, dtype=torch.uint8, device=dev)
    a_g = torch.empty((n, 4), device=dev)
    eager.reset_dev(o_e); graphed.reset_dev(o_g)
    graphed.set_graph_safe(True)
    acts = torch.rand((4, n, 4), device=dev) * 2 - 1
    a_g.copy_(acts[0])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graphed.step_dev(a_g, o_g, r_g, d_g)
    torch.cuda.current_stream().wait_stream(side)
    eager.step_dev(acts[0], o_e, r_e, d_e)
    torch.cuda.synchronize()
    assert torch.equal(o_e, o_g)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for k in range(3):                                  # three steps per replay, the same action tensor
            graphed.step_dev(a_g, o_g, r_g, d_g)
    lib = _lib.load()
    ctr = _lib.GaqCounters()
    for t in range(K):
        a_g.copy_(acts[t % 4])
        g.replay()
        for k in range(3):
            eager.step_dev(acts[t % 4], o_e, r_e, d_e)
        if t % 8 == 7 or t == K - 1:
            torch.cuda.synchronize()
            assert torch.equal(o_e, o_g) and torch.equal(r_e, r_g) and torch.equal(d_e, d_g), t
            _lib.check(lib.gaq_get_counters(graphed._handle, C.byref(ctr), None, None))
            assert ctr.step_index == 1 + 3 * (t + 1)
    # a reset keyed by the device-resident index (it sums the counter's words) ...
    graphed.reset_dev(o_g); eager.reset_dev(o_e)
    # ... then packed rows are registered: the F_ROWS twin reads the counter's first word alone and is followed by the bump launch
    rows = torch.empty((n, 20), device=dev)
    graphed.set_packed_rows(rows)
    for k in range(3):
        graphed.step_dev(acts[k], o_g, r_g, d_g); eager.step_dev(acts[k], o_e, r_e, d_e)
    torch.cuda.synchronize()
    assert torch.equal(o_e, o_g) and torch.equal(r_e, r_g) and torch.equal(rows[:, :18], o_g) and torch.equal(rows[:, 18], r_g)
    graphed.set_packed_rows(None)                           # ... and back to the self-counting kernel
    for k in range(3):
        graphed.step_dev(acts[k], o_g, r_g, d_g); eager.step_dev(acts[k], o_e, r_e, d_e)
    _lib.check(lib.gaq_get_counters(graphed._handle, C.byref(ctr), None, None))
    assert ctr.step_index == 1 + 3 * K + 6
    # back to the host counter
    graphed.set_graph_safe(False)
    graphed.step_dev(acts[1], o_g, r_g, d_g); eager.step_dev(acts[1], o_e, r_e, d_e)
    torch.cuda.synchronize()
    assert torch.equal(o_e, o_g) and torch.equal(r_e, r_g)
    eager.close(); graphed.close()


def test_torch_step_path_does_not_synchronise():
    """QuadrotorEnv.step(torch tensor) with dynamics_randomize_every on the DEVICE randomizer must not touch `done` on the host: 20
    steps under torch's sync debug mode "error" (the read-back was a D2H copy + stream sync per step)."""
    import torch
    from gym_art_amd import QuadrotorEnv
    n = 8192
    env = QuadrotorEnv(dynamics_params="Crazyflie", num_envs=n, ep_time=0.05, seed=5, dynamics_randomize_every=1,
                       dyn_sampler_1={"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"})
    assert env._dev_rand
    dev = torch.device("cuda", 0)
    acts = torch.rand((n, 4), device=dev) * 2 - 1
    out = (torch.empty((n, env.obs_dim), device=dev), torch.empty(n, device=dev), torch.empty(n, dtype=torch.uint8, device=dev))
    env.step(acts, out=out)
    torch.cuda.synchronize()
    m0 = env.models["mass"].copy()
    torch.cuda.set_sync_debug_mode("error")
    try:
        for t in range(20):
            obs, rew, done, _ = env.step(acts, out=out)
            obs2, _, _, _ = env.step(acts)                  # fresh output tensors: allocation must not synchronise either
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    m1 = env.models["mass"]
    assert not np.array_equal(m0, m1)                       # ... and the envs WERE re-randomised along the way (ep_len 5)
    env.check_finite()
    env.close()


def test_handle_reports_the_kernel_gaq_plan_predicts():
    """gaq_kernel_variant of live handles == gaq_plan of their configurations with this device's compute-unit count: the CPU
    enumeration (tests/test_plan_cpu.py) speaks about the very selection that runs."""
    import torch
    from gym_art_amd import QuadrotorEnv
    from tests.test_plan_cpu import base_cfg, plan
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    F_PREDRAW, F_NT = 128, 256
    for n, bits in ((cus * 4 * 64, F_NT), (cus * 8 * 64, F_NT | F_PREDRAW), (cus * 64 * 64, F_PREDRAW)):
        env = QuadrotorEnv(num_envs=n, alias_obs=True, seed=1)
        assert env.kernel_variant == (20 | bits), (n, env.kernel_variant)
        p = plan(base_cfg(n, noise=1, obs_state_alias=1, auto_reset=1), cus=cus)
        assert p.step_variant == env.kernel_variant and p.state_layout == env.state_layout == 1
        env.close()
    env = QuadrotorEnv(num_envs=4096, raw_control=False, seed=1)           # Mellinger, uniform model: specialised (F_MELL), split state
    assert env.kernel_variant == (16384 | 16 | 4) and env.state_layout == 2
    env.close()
    env = QuadrotorEnv(num_envs=4096, raw_control=False, seed=1, obs_repr="xyz_vxyz_R_omega_h")      # ... a packed observation: F_MELL | F_PACK
    assert env.kernel_variant == (16384 | 1024 | 16 | 4) and env.state_layout == 2
    env.close()
    env = QuadrotorEnv(num_envs=4096, raw_control=False, seed=1, obs_repr="xyz_vxyz_quat_omega")     # ... a diagnostics-tier one: F_MELL | F_AUXP
    assert env.kernel_variant == (16384 | 65536 | 1024 | 16 | 4) and env.state_layout == 2
    env.close()
    env = QuadrotorEnv(num_envs=4096, raw_control=False, seed=1, obs_repr="xyz_vxyz_quat_omega", alias_obs=False)     # ... on fp64 planes: generic
    assert env.kernel_variant == (8 | 512) and env.state_layout == 0
    env.close()


def test_ablation_variable_is_refused_by_the_product_library():
    """GAQ_ABLATE (timing-only ablations: wrong physics by construction) exists in measurement builds only; the in-tree library fails
    gaq_create loudly instead of running with or silently ignoring it."""
    from gym_art_amd import QuadrotorEnv
    assert _lib.load().gaq_is_diag_build() == 0
    os.environ["GAQ_ABLATE"] = "1"
    try:
        with pytest.raises(ValueError, match="GAQ_ABLATE"):
            QuadrotorEnv(num_envs=256)
    finally:
        os.environ.pop("GAQ_ABLATE")
    QuadrotorEnv(num_envs=256).close()


def test_moved_episode_clocks_cannot_outrun_the_staged_parameters():
    """Per-episode re-randomisation stages every env's NEXT draw and refills it off the critical path; gaq_set_state moving the episode
    clocks so that envs finish twice inside one refill period must not make an env fly on a stale draw: the library refills before the
    next step, and the overrun counter -- now looked at by the synchronous entry points -- stays clear."""
    from gym_art_amd import QuadrotorEnv
    n = 4096
    env = QuadrotorEnv(dynamics_params="Crazyflie", num_envs=n, ep_time=0.5, seed=3, dynamics_randomize_every=1,
                       dyn_sampler_1={"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"})
    assert env._dev_rand and env.ep_len == 50
    a = np.zeros((n, 4), np.float32)
    lib = _lib.load()
    ctr = _lib.GaqCounters()
    res = np.zeros(n, np.uint32)
    for rep in range(4):
        st = env.get_state()
        st[37] = env.ep_len                     # every env finishes on the next step ...
        env.set_state(st)
        env.step(a)
        st = env.get_state()
        st[37] = env.ep_len                     # ... and again right away, long before the 51-step refill period is over
        env.set_state(st)
        env.step(a)
        _lib.check(lib.gaq_get_counters(env._handle, C.byref(ctr), None, _lib.ptr(res)))     # (checks the overrun counter)
        assert (res == 1 + 2 * (rep + 1)).all()
    env.check_finite()
    m = env.models                              # gaq_get_params: checks it too
    trees = env.sampled_trees()
    from gym_art_amd import quad_params as qp
    derived, _ = qp.derive_models(trees, False)
    assert np.allclose(derived["mass"], m["mass"], rtol=1e-12)
    env.close()


def _bench(args, env_extra):
    env = dict(os.environ, **env_extra)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_rccl_gather_of_the_packed_rows_runs_on_this_stack():
    """BASELINE config 4's collective under the driver's eyes: bench.py's N > 1 code path with ONE rank --
    librccl loads, the `nccl` backend initialises on the GPU, dist.gather moves the packed rows the step launch wrote -- in a child
    process; the line carries the phase timings and the gather variants that make an 8-GPU number attributable."""
    line = _bench(["--gpus", "1", "--envs", "131072", "--gather", "packed", "--steps", "200", "--warmup", "100", "--repeats", "2",
                   "--no-cpu-baseline"], {"GAQ_BENCH_FORCE_DIST": "1"})
    assert line["rccl_ranks"] == 1 and line["n_gpus"] == 1
    assert line["config"]["gather"] == "packed" and line["config"]["envs_per_gpu"] == 131072
    assert np.isfinite(line["value"]) and line["value"] > 1e8
    ph = line["phases"]
    assert set(ph) >= {"kernel_ms", "pack_ms", "gather_ms"} and ph["pack_ms"] == 0.0      # fused into the step launch
    assert line["config"]["launch_variant"] == line["config"]["kernel_variant"] | 4096     # ... by the F_ROWS twin of the shard's kernel
    assert 0.0 < ph["kernel_ms"] < 1.0 and ph["gather_ms"] >= 0.0
    assert set(line["variants"]) >= {"gather_none", "gather_obs", "packed_unfused"}
    for v in line["variants"].values():
        assert np.isfinite(v["value"]) and v["value"] > 0


def test_bench_line_reports_every_layout():
    """The N = 1 line times the class-default layout too: `layouts` holds alias / shadow / plain, one region each."""
    line = _bench(["--envs", "262144", "--steps", "150", "--warmup", "100", "--repeats", "2", "--no-cpu-baseline"], {})
    assert set(line["layouts"]) == {"alias", "shadow", "plain"}
    for k, v in line["layouts"].items():
        assert v["us_per_step"] > 0 and 0 < v["frac"] < 1.2, (k, v)
    assert "class default" in line["layouts"]["shadow"]["what"]
    assert line["layouts"]["alias"]["us_per_step"] == pytest.approx(line["ms_per_step"] * 1e3)       # the timed layout: the line's own measurement
    # ... and the same configuration with desynchronised episodes (resets in every launch): 262144 / 501 envs per step, at most a few % slower
    sg = line["staggered_episodes"]
    assert sg["resets_per_step"] == pytest.approx(262144 / 501.0) and 0.7 * line["ms_per_step"] * 1e3 < sg["us_per_step"] < 1.5 * line["ms_per_step"] * 1e3, sg
    assert line["config"]["overrides"] == {}


def test_fork_drop_in_class_against_the_reference():
    """gym_art_amd.quadrotor_multi.QuadrotorEnv = the fork's own class (quadrotor_multi.py:659-843): fixture G17's `multi` blocks -- the
    UNMODIFIED fork run with random constructor arguments -- through it with the very same keyword arguments and NO reward switch;
    the fork's defaults (ep_time 4, reward weights), its "type"-keyed sampler dicts and its info["rewards"] key set."""
    import pickle
    from gym_art_amd.quadrotor_multi import QuadrotorEnv as ForkEnv
    from tests import golden_util as gu
    from tests import hh
    d = gu.load("g17_random_constructor_arguments")
    ran = 0
    for blk in gu.env_blocks(d):
        if str(blk["module"]) == "quadrotor":
            continue
        kw = gu.kwargs_of(blk)
        env = ForkEnv(dynamics_change={"noise": {"thrust_noise_ratio": 0.}}, seed=0, **kw)
        assert env.ep_len == int(blk["ep_len"]) and json.loads(str(blk["rew_coeff_json"])) == env.rew_coeff
        st = hh.pack_state(blk["init_pos"], blk["init_vel"], blk["init_rot"], blk["init_omega"], blk["goal"])
        env.set_state(np.concatenate([st, np.zeros(3)])[:, None])
        f32 = bool(blk["as_f32"])
        for t in range(blk["obs"].shape[0]):
            o, r, dn, info = env.step(blk["actions"][t].astype(np.float32 if f32 else np.float64))
            assert gu.rel_err(o, blk["obs"][t]) <= 1e-6 and abs(r - blk["reward"][t]) <= 3e-7 and dn == bool(blk["done"][t]), (kw, t)
        assert sorted(info["rewards"]) == sorted(["rew_main", "rew_pos", "rew_action", "rew_crash", "rew_orient", "rew_yaw", "rew_rot",
                                                  "rew_attitude", "rew_spin", "rew_vel"])          # quadrotor_multi.py:627-640
        env.close()
        ran += 1
    assert ran >= 8
    # defaults: ep_time = 4 (quadrotor_multi.py:668), fork reward weights (:811-818); the default model name is the fork's own bug (:665)
    env = ForkEnv(dynamics_params="DefaultQuad", seed=1)
    assert env.ep_time == 4 and env.ep_len == 400 and env.rew_coeff["effort"] == 0.01 and env.rew_coeff["spin"] == 0.0
    assert env.rew_coeff["pos_log_weight"] == 1.0 and env.spec.max_episode_steps == 400
    env2 = pickle.loads(pickle.dumps(env))
    assert type(env2) is ForkEnv and env2.ep_len == 400 and env2.rew_coeff == env.rew_coeff
    env.close(); env2.close()
    with pytest.raises(AttributeError):
        ForkEnv()
    # sampler dicts keyed "type" (:759-768); "class" is the other module's key
    env = ForkEnv(dynamics_params="Crazyflie", num_envs=64, seed=2, randomize_on_device=False,
                  dyn_sampler_1={"type": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"})
    m = env.models["mass"]
    assert m.std() > 0 and abs(m.mean() / 0.028 - 1) < 0.1
    obs, rew, done, _ = env.step(np.zeros((64, 4), np.float32))
    assert np.isfinite(obs).all() and np.isfinite(rew).all()
    env.close()
    with pytest.raises(KeyError):
        ForkEnv(dynamics_params="Crazyflie", dyn_sampler_1={"class": "RelativeSampler", "noise_ratio": 0.2})
    with pytest.raises(TypeError):
        ForkEnv(dynamics_params="DefaultQuad", reward="quadrotor")


def test_specialised_mellinger_kernels_agree_with_the_generic_one_at_scale():
    """F_MELL (Mellinger in the specialised kernels, round 3) against the generic kernel it used to run in: 8192 envs from random initial
    states (init_random_state resets), 200 noise-free steps, all three layouts, Hummingbird and CrazyFlie -- every observation within
    1e-6 (the same arithmetic header; what differs is the state storage and hipcc's contraction choices in the controller)."""
    from gym_art_amd import QuadrotorEnv
    n = 8192
    for model in ("DefaultQuad", "Crazyflie"):
        kw = dict(dynamics_params=model, num_envs=n, raw_control=False, ep_time=5, seed=13, init_random_state=True, thrust_noise="off",
                  auto_reset=False)
        os.environ["GAQ_FORCE_GENERIC"] = "1"
        try:
            ref = QuadrotorEnv(alias_obs=False, **kw)
        finally:
            os.environ.pop("GAQ_FORCE_GENERIC")
        assert ref.kernel_variant == 8
        st0 = ref.get_state()
        envs = [QuadrotorEnv(alias_obs=a, **kw) for a in (False, True, None)]
        for e in envs:
            assert e.kernel_variant & 16384
            e.set_state(st0)
        a = np.zeros((n, 4), np.float32)
        worst = 0.0
        for t in range(200):
            o_ref, r_ref, d_ref, _ = ref.step(a)
            for e in envs:
                o, r, d, _ = e.step(a)
                worst = max(worst, float(np.max(np.abs(o - o_ref) / np.maximum(np.abs(o_ref), 1.0))))
                assert np.max(np.abs(r - r_ref)) <= 1e-6 and np.array_equal(d, d_ref)
        assert worst <= 1e-6, (model, worst)
        # ... and the controller is in the loop: from tumbling random states a good part of the batch is already near the goal after 2 s
        st = envs[1].get_state()
        assert np.mean(np.linalg.norm(st[0:3].T - np.array([0., 0., 2.]), axis=1) < 0.5) > 0.25
        ref.close()
        for e in envs:
            e.close()


@pytest.mark.parametrize("layout", [None, False])
def test_swarm_collision_response_against_its_specification(layout):
    """Swarm layer, PARITY-UNPINNED (the reference has no multi-agent env; own specification, include/gaq.h gaq_swarm): the collision
    RESPONSE.  Per-agent dynamics come from the pinned oracle (one step of the batch), the response from oracle.swarm_response; the
    device's post-step velocities have to be their sum, positions untouched; momentum of every world is conserved by it; switched
    off, agents pass through each other as in round 2."""
    from gym_art_amd import QuadrotorEnvMulti
    from oracle import quad_oracle as qo
    A, W = 8, 256
    n = A * W
    kw = dict(num_agents=A, num_worlds=W, ep_time=5, seed=23, thrust_noise="off", auto_reset=False, goal_radius=0.25, collision_dist=0.5,
              prox_dist=1.5, alias_obs=layout)       # None: the split-state swarm kernel (F_SWARM); False: the light generic kernel
    env = QuadrotorEnvMulti(**kw)
    off = QuadrotorEnvMulti(collision_response=False, **kw)
    assert env.swarm["collision_response"] is True and off.swarm["collision_response"] is False
    rng = np.random.RandomState(4)
    st = env.get_state()
    # crowd every world: agents within ~0.3 m of the world's centre, flying at up to 2 m/s in random directions
    st[0:3] = (np.array([0., 0., 3.])[:, None] + rng.uniform(-0.3, 0.3, (3, n)))
    st[3:6] = rng.uniform(-2, 2, (3, n))
    env.set_state(st); off.set_state(st)
    models = env.models
    p = qo.Params(n, mass=models["mass"], inertia=models["inertia"], thrust_max=models["thrust_max"], torque_max=models["torque_max"],
                  prop_pos=models["prop_pos"].reshape(n, 4, 3), damp_time_up=models["damp_time_up"], damp_time_down=models["damp_time_down"],
                  linearity=models["linearity"], arm=models["arm"], ou_sigma=models["ou_sigma"], vel_damp=models["vel_damp"],
                  damp_omega_quadratic=models["damp_omega_quadratic"], C_drag=models["c_drag"], C_roll=models["c_roll"])
    cfg = qo.Config(sim_freq=200., sim_steps=2, ep_time=5, reward_variant="multi")
    cfg.action_f32 = True
    hits = 0
    for t in range(6):
        before = env.get_state()
        s = qo.State(n)
        s.goal[:] = before[34:37].T
        s.set_state(before[0:3].T, before[3:6].T, before[6:15].T.reshape(n, 3, 3), before[15:18].T)
        s.omega = before[15:18].T.copy()                   # (not the first step after a set_state of the reference: keep fp64)
        s.tick[:] = 1
        a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
        qo.env_step(s, p, cfg, a.astype(np.float64))
        dv = qo.swarm_response(s.pos, s.vel, A, 0.5)
        hits += int((np.abs(dv).sum(axis=1) > 0).sum())
        off.set_state(before)
        obs, rew, done, _ = env.step(a)
        off.step(a)
        after, plain = env.get_state(), off.get_state()
        assert np.allclose(after[0:3].T, s.pos, atol=2e-6) and np.allclose(plain[0:3], after[0:3], atol=1e-9)
        assert np.allclose(plain[3:6].T, s.vel, atol=2e-6)                       # no response: the pinned per-agent dynamics alone
        assert np.allclose(after[3:6].T, s.vel + dv, atol=5e-6), np.abs(after[3:6].T - s.vel - dv).max()
        # momentum of a world (equal masses): unchanged by the response
        assert np.allclose(after[3:6].T.reshape(W, A, 3).sum(axis=1), plain[3:6].T.reshape(W, A, 3).sum(axis=1), atol=2e-5)
        # the observation and the neighbour block show the post-response velocities
        assert np.allclose(obs[:, 3:6], after[3:6].T, atol=2e-6) and np.allclose(obs[:, 18:], qo.swarm_obs(after[0:3].T, after[3:6].T, A), atol=4e-6)
    assert hits > 200
    env.close(); off.close()


def test_sharded_swarm_rehearsal_and_whole_world_shards():
    """BASELINE config 5's sharding, executed: `bench.py --swarm 8 --gpus 2` end to end with both ranks on this
    box's one GPU over gloo (a rehearsal: the rate means nothing) -- whole worlds per shard, the packed gather of 62-word rows -- and two
    shard handles keyed by global env index against ONE handle holding all the worlds, bit for bit."""
    import torch
    from gym_art_amd import QuadrotorEnv
    env = dict(os.environ, GAQ_BENCH_REHEARSAL="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--swarm", "8", "--envs", "32768", "--steps", "30",
                          "--warmup", "10", "--repeats", "2", "--prime-ms", "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    c = d["config"]
    assert d["n_gpus"] == 2 and c["total_envs"] == 32768 and c["envs_per_gpu"] == 16384 and c["obs_dim"] == 60 and c["gather"] == "packed"
    assert "REHEARSAL" in c["workload"] and "parity-unpinned" in c["workload"] and d["value"] > 0
    # shards of whole worlds == one handle (global env index keys the reset draws and the noise; neighbours never cross a shard)
    n, A = 4096, 8
    kw = dict(ep_time=0.1, seed=5, reward="multi", swarm=dict(agents=A), auto_reset=True)
    whole = QuadrotorEnv(num_envs=n, **kw)
    parts = [QuadrotorEnv(num_envs=n // 2, env_id_offset=k * n // 2, **kw) for k in range(2)]
    dev = torch.device("cuda", 0)
    D = whole.obs_dim
    ow = torch.empty((n, D), device=dev); rw = torch.empty(n, device=dev); dw = torch.empty(n, dtype=torch.uint8, device=dev)
    op = [torch.empty((n // 2, D), device=dev) for _ in range(2)]; rp = [torch.empty(n // 2, device=dev) for _ in range(2)]
    dp = [torch.empty(n // 2, dtype=torch.uint8, device=dev) for _ in range(2)]
    whole.reset_dev(ow)
    for k in range(2):
        parts[k].reset_dev(op[k])
    for t in range(25):                                     # ep_len 10: auto-resets inside
        a = torch.rand((n, 4), device=dev) * 2 - 1
        whole.step_dev(a, ow, rw, dw)
        for k in range(2):
            parts[k].step_dev(a[k * n // 2:(k + 1) * n // 2].contiguous(), op[k], rp[k], dp[k])
    torch.cuda.synchronize()
    assert torch.equal(ow, torch.cat(op)) and torch.equal(rw, torch.cat(rp)) and torch.equal(dw, torch.cat(dp))
    assert int(dw.sum().item()) == 0 and whole.get_state()[37].max() < 10
    whole.close()
    for e in parts:
        e.close()


def test_live_handles_select_what_gaq_plan_says_over_random_configurations():
    """The CPU enumeration of tests/test_plan_cpu.py goes through gaq_plan; this closes the loop on the device: 300 random configurations
    (the same option space) created as REAL handles report the variant, the layout and the observation width gaq_plan predicts for this
    device's compute-unit count, step once, and the step launches the instantiation gaq_launch_variant names (rows registered / graph-safe
    mode switch it to the twins gaq_plan lists)."""
    import torch
    from tests.test_plan_cpu import base_cfg, plan
    lib = _lib.load()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    rng = np.random.RandomState(7)
    dev = torch.device("cuda", 0)
    obs_sets = [0, 1, 2, 3, 4, 8, 12, 14, 15, 16, 18, 32, 64, 96, 33, 97]
    made = 0
    for it in range(300):
        n = int(rng.choice([64, 200, 4096, 65536, 131072]))
        kw = {"control": int(rng.randint(3)), "noise": int(rng.choice([0, 1])), "obs_flags": int(rng.choice(obs_sets)),
              "obs_state_alias": int(rng.randint(3)), "auto_reset": 1, "sim_steps": int(rng.choice([1, 2, 4]))}
        if rng.rand() < 0.3:
            kw.update({"model.damp_time_up": 0.15, "model.damp_time_down": 0.15})
        if rng.rand() < 0.25:
            kw.update({"sense.enabled": 1, "sense.pos_norm_std": 0.005, "sense.gyro_noise_density": 0.000175,
                       "sense.gyro_norm_std": 0.0 if rng.rand() < 0.6 else 0.01, "sense.gyro_bias_correlation_time": 1000.0})
        extra = rng.choice(["", "", "aux", "resample_goal", "excite", "swarm", "action_change", "fp32"])
        if extra == "aux":
            kw["aux_outputs"] = 1
        elif extra in ("resample_goal", "excite"):
            kw[extra] = 1
        elif extra == "swarm":
            if kw["obs_flags"] & 16:
                continue
            kw.update({"swarm.agents": 8, "swarm.goal_radius": 0.5, "swarm.collision_dist": 0.3, "swarm.prox_dist": 1.2, "swarm.response": 1})
        elif extra == "action_change":
            kw["rew.action_change"] = 0.1
        elif extra == "fp32":
            kw["fp32_state"] = 1
        cfg = base_cfg(n, **kw)
        p = plan(cfg, cus=cus)
        h = C.c_void_p()
        rc = lib.gaq_create(C.byref(cfg), C.byref(h))
        if kw.get("fp32_state") and p.state_layout == 0:
            assert rc == -1 and b"fp32_state" in lib.gaq_last_error(), kw          # refused, as the plan says (not launchable)
            continue
        assert rc == 0, (kw, lib.gaq_last_error())
        made += 1
        assert (lib.gaq_kernel_variant(h), lib.gaq_state_layout(h), lib.gaq_obs_dim(h)) == (p.step_variant, p.state_layout, p.obs_dim), (kw, n)
        D = p.obs_dim
        obs = torch.zeros((n, D), device=dev); rew = torch.zeros(n, device=dev); done = torch.zeros(n, dtype=torch.uint8, device=dev)
        act = torch.zeros((n, 4), device=dev)
        _lib.check(lib.gaq_reset_dev(h, None, _lib.ptr(obs), None))
        assert lib.gaq_launch_variant(h) == p.step_variant
        _lib.check(lib.gaq_step_dev(h, _lib.ptr(act), _lib.ptr(obs), _lib.ptr(rew), _lib.ptr(done), None))
        rows = torch.zeros((n, D + 2), device=dev)
        _lib.check(lib.gaq_set_packed_rows_dev(h, _lib.ptr(rows)))
        assert lib.gaq_launch_variant(h) == (p.rows_variant if p.rows_variant >= 0 else p.step_variant), kw
        _lib.check(lib.gaq_step_dev(h, _lib.ptr(act), _lib.ptr(obs), _lib.ptr(rew), _lib.ptr(done), None))
        _lib.check(lib.gaq_set_packed_rows_dev(h, None))
        _lib.check(lib.gaq_set_graph_safe(h, 1))
        assert lib.gaq_launch_variant(h) == (p.ctr_variant if p.ctr_variant >= 0 else p.step_variant), kw
        _lib.check(lib.gaq_step_dev(h, _lib.ptr(act), _lib.ptr(obs), _lib.ptr(rew), _lib.ptr(done), None))
        torch.cuda.synchronize()
        assert bool(torch.isfinite(obs).all()) and bool(torch.isfinite(rew).all()), kw
        ctr = _lib.GaqCounters()
        _lib.check(lib.gaq_get_counters(h, C.byref(ctr), None, None))
        assert ctr.step_index == 3, kw
        lib.gaq_destroy(h)
    assert made > 250


def test_terminal_observation_in_info_follows_the_vector_env_convention():
    """terminal_observation=True on an auto-reset batch: info["terminal_observation"][i] is what the reference would have returned with
    done=True for env i -- checked against an identically seeded env WITHOUT auto-reset -- while step() itself hands out the first
    observation of the new episode; NumPy and torch callers; nothing is fetched on steps that end no episode."""
    import torch
    from gym_art_amd import QuadrotorEnv
    n = 300
    kw = dict(num_envs=n, ep_time=0.05, seed=8, thrust_noise="off")
    auto = QuadrotorEnv(terminal_observation=True, **kw)
    manual = QuadrotorEnv(auto_reset=False, **kw)
    assert np.array_equal(auto.reset(), manual.reset())
    rng = np.random.RandomState(3)
    for t in range(6):                                       # ep_len 5: the sixth step ends every episode
        a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
        o_a, r_a, d_a, info = auto.step(a)
        o_m, r_m, d_m, _ = manual.step(a)
        assert np.array_equal(d_a, d_m) and np.array_equal(r_a, r_m)
        assert ("terminal_observation" in info) == bool(d_a.any())
    assert d_a.all() and info["terminal_observation"].shape == (n, 18)
    assert np.allclose(info["terminal_observation"], o_m, rtol=2.5e-7, atol=1e-30) and not np.allclose(o_a, o_m)
    # torch callers: the device tensor itself (rows valid where done), no synchronisation
    dev = torch.device("cuda", 0)
    tauto = QuadrotorEnv(terminal_observation=True, **kw)
    tman = QuadrotorEnv(auto_reset=False, **kw)
    rng = np.random.RandomState(3)
    for t in range(6):
        a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
        o, r, d, info = tauto.step(torch.as_tensor(a, device=dev))
        o_m, _, _, _ = tman.step(a)
    assert torch.is_tensor(info["terminal_observation"]) and bool(d.all())
    assert np.allclose(info["terminal_observation"].cpu().numpy(), o_m, rtol=2.5e-7, atol=1e-30)
    for e in (auto, manual, tauto, tman):
        e.close()


def test_observation_variants_on_the_light_diagnostics_tier():
    """The quaternion / t2w / t2t observation variants (pinned against the patched-import fixture G15 in the FULL diagnostics tier, with
    the reference's recorded draws) run on the light generic kernel for a uniform model on fp64 planes since round 3 (F_LITE | F_DIAG: 218
    VGPRs, two waves per SIMD instead of one) and on the SPLIT state since round 4 (F_AUXP: the class default layout): the same
    observations, rewards and noisy t2w / t2t draws as the full tier (GAQ_FORCE_GENERIC=1), sensor noise included -- the light tier to
    rounding, the split state to its 39-bit storage."""
    from gym_art_amd import QuadrotorEnv
    n = 2048
    rng = np.random.RandomState(12)
    for obs_repr in ("xyz_vxyz_quat_omega", "xyzr_vxyzr_quat_omega", "xyzr_vxyzr_quat_omega_h", "xyz_vxyz_R_omega_t2w", "xyzr_vxyzr_R_omega_t2w",
                     "xyz_vxyz_R_omega_t2w_t2t"):
        for sense in (None, "default"):
            kw = dict(num_envs=n, obs_repr=obs_repr, ep_time=0.1, seed=31, sense_noise=sense, dynamics_params="Crazyflie")
            split = QuadrotorEnv(**kw)                       # class default layout: the split state, F_AUXP | F_PACK | F_ALIAS | lag | noise
            light = QuadrotorEnv(alias_obs=False, **kw)      # fp64 planes: the light generic kernel
            os.environ["GAQ_FORCE_GENERIC"] = "1"
            try:
                full = QuadrotorEnv(alias_obs=False, **kw)
            finally:
                os.environ.pop("GAQ_FORCE_GENERIC")
            assert split.kernel_variant == (65536 | 1024 | 16 | 4 | 2) and split.state_layout == 2, split.kernel_variant
            assert light.kernel_variant == (8 | 64 | 512) and full.kernel_variant == (8 | 512), (light.kernel_variant, full.kernel_variant)
            o0 = full.reset()
            assert np.array_equal(light.reset(), o0) and np.allclose(split.reset(), o0, rtol=0, atol=2e-7)
            for t in range(25):                              # ep_len 10: in-kernel resets inside
                a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
                ol, rl, dl, _ = light.step(a)
                of, rf, df, _ = full.step(a)
                os_, rs, ds, _ = split.step(a)
                assert np.allclose(ol, of, rtol=0, atol=2e-7) and np.allclose(rl, rf, rtol=0, atol=1e-7) and np.array_equal(dl, df), (obs_repr, sense, t)
                # (R2quat divides by 4w, w = sqrt(1 + tr R) / 2 (quad_utils.py:101-108): near a half-turn -- yaw-only initial attitudes pass
                #  through it -- the quaternion amplifies the 2^-39 of the split storage by 1 / (2 (1 + tr R)); rows with |w| < 0.02 are
                #  compared on everything but the quaternion)
                ok = np.ones_like(of, dtype=bool)
                if "quat" in obs_repr:
                    ok[np.abs(of[:, 6]) < 0.02, 6:10] = False
                err = np.where(ok, np.abs(os_ - of) / np.maximum(np.abs(of), 1.0), 0.0)
                assert float(np.max(err)) <= 1e-6 and np.allclose(rs, rf, rtol=0, atol=1e-6) and np.array_equal(ds, df), (obs_repr, sense, t)
            split.close(); light.close(); full.close()


def test_output_ring_of_the_numpy_path():
    """out_ring=k: the NumPy step hands out its obs / reward / done arrays from a ring of k preallocated sets (fresh 75-MB arrays cost more
    in page faults than the step's whole PCIe round trip at N = 2^20): the same numbers as the default path, an array stays valid for k - 1
    further steps and is then reused; env.actions still reads as the reference's float64 pair."""
    from gym_art_amd import QuadrotorEnv
    n = 20000
    kw = dict(num_envs=n, ep_time=0.1, seed=6)
    plain, ring = QuadrotorEnv(**kw), QuadrotorEnv(out_ring=3, **kw)
    assert np.array_equal(plain.reset(), ring.reset())
    rng = np.random.RandomState(1)
    kept = []
    for t in range(7):
        a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
        o_p, r_p, d_p, _ = plain.step(a)
        o_r, r_r, d_r, _ = ring.step(a)
        assert np.array_equal(o_p, o_r) and np.array_equal(r_p, r_r) and np.array_equal(d_p, d_r) and d_r.dtype == np.bool_
        kept.append((o_r, o_p.copy()))
        if t >= 2:
            assert np.array_equal(kept[t - 2][0], kept[t - 2][1])          # two steps old: still intact (ring of three)
        if t >= 3:
            assert kept[t - 3][0] is o_r                                    # three steps old: the same array object, reused
    acts = ring.actions
    assert acts[0].dtype == np.float64 and np.array_equal(acts[0], a.astype(np.float64)) and acts[1].shape == (n, 4)
    plain.close(); ring.close()


def test_mellinger_with_packed_observations_on_the_split_state():
    """F_MELL | F_PACK: the Mellinger controller with the body-frame / appended-height / accelerometer + action observations and with sensor
    noise, on the split state, against the generic kernel these configurations ran in until round 3: 4096 envs from random states, 150
    steps, every observation and reward within 1e-6 (the noisy ones draw the same Philox streams)."""
    from gym_art_amd import QuadrotorEnv
    n = 4096
    for obs_repr, sense, model in (("xyz_vxyz_R_omega_h", None, "DefaultQuad"), ("xyzr_vxyzr_R_omega", None, "Crazyflie"),
                                   ("xyz_vxyz_R_omega_acc_act", None, "DefaultQuad"), ("xyz_vxyz_R_omega", "default", "DefaultQuad"),
                                   ("xyzr_vxyzr_R_omega_h", "default", "Crazyflie")):
        kw = dict(dynamics_params=model, num_envs=n, raw_control=False, ep_time=5, seed=17, init_random_state=True, thrust_noise="off",
                  auto_reset=False, obs_repr=obs_repr, sense_noise=sense)
        os.environ["GAQ_FORCE_GENERIC"] = "1"
        try:
            ref = QuadrotorEnv(**kw)
        finally:
            os.environ.pop("GAQ_FORCE_GENERIC")
        env = QuadrotorEnv(**kw)
        assert (env.kernel_variant & (16384 | 1024 | 16)) == (16384 | 1024 | 16) and ref.kernel_variant & 8, (env.kernel_variant, ref.kernel_variant)
        env.set_state(ref.get_state())
        a = np.zeros((n, 4), np.float32)
        for t in range(150):
            o_r, r_r, d_r, _ = ref.step(a)
            o, r, d, _ = env.step(a)
            assert np.max(np.abs(o - o_r) / np.maximum(np.abs(o_r), 1.0)) <= 1e-6 and np.max(np.abs(r - r_r)) <= 1e-6, (obs_repr, sense, t)
        ref.close(); env.close()
