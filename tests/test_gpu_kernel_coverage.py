"""GPU: EVERY step / rollout kernel instantiation of gaq_kernels.hpp flies once against the full generic kernel.

`tools/kernel_coverage.py` (what a test run launched against the instantiation lists) showed that round 3's suite never launched 42 of
the 101 step kernels and 9 of the 16 rollout kernels -- the fp32 forms, most per-episode re-randomisation (F_RZ) forms, the lag-capable
F_ROWS / F_CTR twins, three Mellinger ones.  Here each instantiated mask is turned back into the constructor arguments that select it
(the inverse of gaq.hip::select_kernel; the handle's own `launch_variant` has to agree) and flown for 60 steps of 2088 envs -- random
initial states, random actions, ten-step episodes, so in-kernel resets and parameter promotions happen -- beside the same configuration
in the FULL generic kernel (`GAQ_FORCE_GENERIC=1`: fp64 planes, every runtime flag honoured, and itself pinned to the oracle and the
golden fixtures by tests/test_gpu_parity.py / test_gpu_round2.py): observations and rewards within 1e-6, dones equal; the fp32 forms
within 5e-4 of the fp64 generic kernel (their state is fp32: DESIGN.md section 2).  Twins: the registered packed rows have to be the
step's own outputs (F_ROWS), graph-safe stepping has to give the same trajectory (F_CTR).  Rollouts: T fused steps against T single
steps of the same handle type."""
import contextlib
import os
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = int(os.environ.get("GAQ_FUZZ_SEED", "0"))       # tools/hunt.sh: the same flights from other seeds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PER_ENV, LAG, NOISE, GENERIC, ALIAS, FP32, LITE, PREDRAW, NT, DIAG, PACK, RZ, ROWS, CTR, MELL, SWARM, AUXP, ENVX, BIAS = (1 << k for k in range(19))
SAMPLER = {"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"}
N, STEPS = 2088, 60          # (2088 = 8 x 261 envs: 32 whole wave tiles and one of 40 lanes)


def instantiated(prefix):
    src = open(os.path.join(ROOT, "gym_art_amd", "csrc", "gaq_kernels.hpp")).read()
    out = set()
    for m in re.finditer(r"#define %s_PART\d\(X\)(.*)" % prefix, src):
        out |= {int(x) for x in re.findall(r"X\((\d+)u\)", m.group(1))}
    return sorted(out)


@contextlib.contextmanager
def environ(**kv):
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update({k: v for k, v in kv.items() if v is not None})
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def recipe(mask):
    """(swarm?, constructor kwargs, creation environment, twin, kwargs the REFERENCE drops, tolerance) of the configuration that selects
    step_kernel<mask> -- None for the two masks that ARE the reference (<8>, <9>: launched by every case below)."""
    base = mask & ~(ROWS | CTR)
    twin = "rows" if mask & ROWS else "ctr" if mask & CTR else None
    kw, env, ref_drop, tol, swarm = {}, {}, (), 1e-6, False
    if base & PER_ENV:
        kw["dyn_sampler_1"] = dict(SAMPLER)
    if base & RZ:
        kw["dynamics_randomize_every"] = 1
    if base & GENERIC:
        lite, diag = bool(base & LITE), bool(base & DIAG)
        if not lite and not diag:
            if base & RZ:
                kw["sense_noise"] = {"gyro_norm_std": 0.01}       # <2057>: the full generic kernel with promotions (the gyro-bias random
                return False, kw, env, twin, ref_drop, tol        # walk needs it; the forced reference is this very kernel)
            return None
        if lite and not diag:
            kw["resample_goal"] = True                 # per-env goals: the light tier (<72>, <73>, <2121>) -- on fp64 planes (on the split
            kw["alias_obs"] = False                    # state they fly <197648> ...: F_ENVX, round 4)
        elif lite and diag:
            kw["info"] = True                          # <584>: the aux row on a uniform RawControl model ...
            kw["alias_obs"] = False                    # ... on fp64 planes (on the split state it is <66576> ...: F_AUXP, round 4)
            ref_drop = ("info",)
        else:
            kw["info"] = True                          # <520>, <521>, <2569>: the aux row beside something heavy, on fp64 planes: per-env
            kw["alias_obs"] = False                    # models (on the split state they fly <66577> ...: F_AUXP with per-env models) ...
            if not base & PER_ENV:
                kw["raw_control"] = False              # ... or Mellinger on a uniform model (on the split state: <82960> ...)
            ref_drop = ("info",)
        return False, kw, env, twin, ref_drop, tol
    if base & LAG:
        kw["dynamics_params"] = "Crazyflie"
    kw["thrust_noise"] = "philox" if base & NOISE else "off"
    if base & MELL:
        kw["raw_control"] = False
    if base & SWARM:
        swarm = True
    if base & FP32:
        kw.update(precision="fp32", alias_obs=True)
        ref_drop, tol = ("precision",), 5e-4
    elif base & ENVX:
        kw.update(resample_goal=True, excite=True, alias_obs=None)          # <197648> ...: per-env goals (new ones every fifth tick too) on the
        if base & BIAS:                                                      # split state; <459792> ...: + the gyro-bias random walk
            kw["sense_noise"] = {"gyro_norm_std": 0.01}
    elif base & AUXP:
        kw.update(info=True, alias_obs=None)            # <66576> ...: the info dict's aux row on the split state
        ref_drop = ("info",)
    elif base & PACK:
        kw.update(obs_repr="xyz_vxyz_R_omega_h", alias_obs=None)
    elif base & ALIAS:
        kw["alias_obs"] = True
    else:
        kw["alias_obs"] = False
    if (base & ~(PREDRAW | NT)) in (20, 22, 23):       # the size-specific forms of the three alias kernels: overrides of the batch-size rule
        env = {"GAQ_PREDRAW": "1" if base & PREDRAW else "0", "GAQ_NT": "1" if base & NT else "0"}
    return swarm, kw, env, twin, ref_drop, tol


def make(swarm, kw, env, generic=False):
    from gym_art_amd import QuadrotorEnv, QuadrotorEnvMulti
    common = dict(ep_time=0.1, seed=29 + SEED, init_random_state=True, auto_reset=True)
    with environ(GAQ_FORCE_GENERIC="1" if generic else None, **env):
        if swarm:
            return QuadrotorEnvMulti(num_agents=8, num_worlds=N // 8, goal_radius=0.5, **common, **kw)
        return QuadrotorEnv(num_envs=N, **common, **kw)


def fly(e, actions, twin=None):
    import torch
    dev = actions.device
    D = e.obs_dim
    obs = torch.empty((N, D), device=dev); rew = torch.empty(N, device=dev); done = torch.empty(N, dtype=torch.uint8, device=dev)
    rows = None
    if twin == "rows":
        rows = torch.zeros((N, D + 2), device=dev)
        e.set_packed_rows(rows)
    elif twin == "ctr":
        e.set_graph_safe(True)
    e.reset_dev(obs)
    variant = e.launch_variant
    O, R, Dn = [obs.cpu().numpy().copy()], [], []
    for t in range(actions.shape[0]):
        e.step_dev(actions[t], obs, rew, done)
        O.append(obs.cpu().numpy().copy()); R.append(rew.cpu().numpy().copy()); Dn.append(done.cpu().numpy().copy())
        if rows is not None:
            r = rows.cpu().numpy()
            assert np.array_equal(r[:, :D], O[-1]) and np.array_equal(r[:, D], R[-1]) and np.array_equal(r[:, D + 1], Dn[-1].astype(np.float32)), t
    return variant, np.stack(O), np.stack(R), np.stack(Dn)


def close_enough(a, b, tol):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0))) <= tol


@pytest.mark.parametrize("variation", ["as_shipped", "damped", "all_parameter_planes", "other_rates_and_rewards"])
def test_every_step_kernel_instantiation_against_the_generic_kernel(variation):
    """`as_shipped`: the models as the reference ships them.  `damped`: dynamics_change gives every model linear velocity damping and
    quadratic angular damping (zero in every shipped model; with per-env parameters two more planes are then loaded, and a promotion
    moves all 45 planes instead of the 19 hot ones).  `all_parameter_planes`: GAQ_NO_COMPACT=1 -- per-env kernels load all 30 planes
    instead of rebuilding 10 of them from 5 (the generic reference keeps rebuilding: the two have to agree to the bit anyway).
    `other_rates_and_rewards`: one sub-step of 10 ms per env step, the [0, 1] action convention, the fork's log-distance reward with every
    optional term switched on (rot / attitude: the arccos path; yaw, vel)."""
    import torch
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev); gen.manual_seed(5 + SEED)
    actions = torch.rand((STEPS, N, 4), device=dev, generator=gen) * 2 - 1
    refs, flown, skipped = {}, [], []
    for mask in instantiated("GAQ_STEP"):
        rc = recipe(mask)
        if rc is None:
            skipped.append(mask)
            continue
        swarm, kw, env, twin, ref_drop, tol = rc
        if variation == "damped":
            kw = dict(kw, dynamics_change={"damp": {"vel": 0.05, "omega_quadratic": 0.01}})
        elif variation == "all_parameter_planes":
            if not mask & PER_ENV:
                continue
            env = dict(env, GAQ_NO_COMPACT="1")
        elif variation == "other_rates_and_rewards":
            kw = dict(kw, sim_freq=100., sim_steps=1, raw_control_zero_middle=False, reward="multi",
                      rew_coeff={"rot": 0.1, "attitude": 0.1, "yaw": 0.1, "vel": 0.05})
        e = make(swarm, kw, env)
        if variation == "damped":
            assert np.all(np.asarray(e.models["vel_damp"]) > 0) and np.all(np.asarray(e.models["damp_omega_quadratic"]) > 0), mask
        try:
            variant, O, R, Dn = fly(e, actions, twin)
        finally:
            e.close()
        assert variant == mask, "step_kernel<%d>: the recipe %r launched <%d>" % (mask, (kw, env, twin), variant)
        ref_kw = {k: v for k, v in kw.items() if k not in ref_drop}
        key = repr((swarm, sorted(ref_kw.items(), key=str)))
        if key not in refs:
            g = make(swarm, ref_kw, {}, generic=True)
            try:
                gv, gO, gR, gD = fly(g, actions)
            finally:
                g.close()
            assert gv & GENERIC and not gv & LITE, gv
            refs[key] = (gO, gR, gD)
        gO, gR, gD = refs[key]
        assert Dn.sum() > N, "no episode ended: the in-kernel reset / promotion path was not flown"
        assert np.array_equal(Dn, gD), "step_kernel<%d>: dones differ from the generic kernel's" % mask
        for t in range(STEPS + 1):
            assert close_enough(O[t], gO[t], tol), "step_kernel<%d>: observation %d differs from the generic kernel's (%r)" % (mask, t, kw)
        assert float(np.max(np.abs(R - gR))) <= max(tol, 2e-6) * 10, "step_kernel<%d>: rewards differ" % mask
        flown.append(mask)
    if variation == "all_parameter_planes":
        assert len(flown) >= 35, len(flown)
        return
    assert skipped == [8, 9] and len(flown) >= 99, (skipped, len(flown))
    # ... and both reference kernels were launched by the cases above
    from gym_art_amd import _lib
    import ctypes as C
    buf = (C.c_uint32 * 1024)()
    k = _lib.load().gaq_launched_variants(0, buf, 1024)
    launched = {int(buf[i]) for i in range(k)}
    assert set(instantiated("GAQ_STEP")) <= launched, sorted(set(instantiated("GAQ_STEP")) - launched)


def test_every_rollout_kernel_instantiation_against_single_steps():
    """rollout_kernel<F> (the alias kernels 16 ... 23 and their fp32 forms 48 ... 55): 24 fused steps = 24 single steps of a twin handle --
    observations within one fp32 ulp (the fused loop keeps fp64 state in registers between steps; fp32 forms: both round every step),
    rewards and dones likewise."""
    import torch
    from gym_art_amd import QuadrotorEnv
    dev = torch.device("cuda", 0)
    T = 24
    gen = torch.Generator(device=dev); gen.manual_seed(6 + SEED)
    actions = torch.rand((T, N, 4), device=dev, generator=gen) * 2 - 1
    for mask in instantiated("GAQ_ROLL"):
        kw = dict(num_envs=N, ep_time=0.1, seed=31 + SEED, init_random_state=True, auto_reset=True, alias_obs=True,
                  thrust_noise="philox" if mask & NOISE else "off")
        if mask & LAG:
            kw["dynamics_params"] = "Crazyflie"
        if mask & PER_ENV:
            kw["dyn_sampler_1"] = dict(SAMPLER)
        if mask & FP32:
            kw["precision"] = "fp32"
        with environ(GAQ_PREDRAW="0", GAQ_NT="0"):
            fused, single = QuadrotorEnv(**kw), QuadrotorEnv(**kw)
        D = fused.obs_dim
        oT = torch.empty((T, N, D), device=dev); rT = torch.empty((T, N), device=dev); dT = torch.empty((T, N), dtype=torch.uint8, device=dev)
        fused.reset_dev(oT[T - 1])             # alias layout: the state heads live where the LAST step's observation goes
        fused.step_many_dev(actions, oT, rT, dT)
        obs = torch.empty((N, D), device=dev); rew = torch.empty(N, device=dev); done = torch.empty(N, dtype=torch.uint8, device=dev)
        single.reset_dev(obs)
        assert single.kernel_variant == mask, (mask, single.kernel_variant)
        for t in range(T):
            single.step_dev(actions[t], obs, rew, done)
            tol = 2e-5 if mask & FP32 else 3e-7
            assert torch.allclose(oT[t], obs, rtol=tol, atol=tol), "rollout_kernel<%d>: observation of step %d" % (mask, t)
            assert torch.allclose(rT[t], rew, rtol=1e-5, atol=1e-5) and torch.equal(dT[t], done), "rollout_kernel<%d>: step %d" % (mask, t)
        assert int(dT.sum()) > N
        fused.close(); single.close()
    from gym_art_amd import _lib
    import ctypes as C
    buf = (C.c_uint32 * 64)()
    k = _lib.load().gaq_launched_variants(1, buf, 64)
    assert set(instantiated("GAQ_ROLL")) <= {int(buf[i]) for i in range(k)}


def test_host_managed_rerandomisation_of_scattered_envs_and_the_rarely_called_entry_points():
    """tools/kernel_coverage.py also counts the C-ABI calls a test run makes: gaq_set_params_indexed (host-managed per-episode
    re-randomisation of the envs that JUST finished, quadrotor.py:1063-1066 -- every other test lets the device randomiser do it),
    gaq_set_timing / gaq_last_kernel_ms, gaq_num_envs and gaq_stream were reached by none.  Episodes are desynchronised with a masked
    reset, so every `done` batch is a scattered subset: those envs and only those get new parameters, the device's parameter planes
    equal the host's arrays afterwards, and a second handle given the same parameters wholesale (gaq_set_params) flies the same step."""
    import ctypes as C
    from gym_art_amd import QuadrotorEnv, _lib
    n = 300
    kw = dict(dynamics_params="Crazyflie", num_envs=n, ep_time=0.05, seed=3 + SEED, dyn_sampler_1=dict(SAMPLER), thrust_noise="off",
              randomize_on_device=False)
    env = QuadrotorEnv(dynamics_randomize_every=1, **kw)
    lib = env._lib
    assert lib.gaq_num_envs(env._handle) == n and lib.gaq_stream(env._handle) is not None
    env.reset()
    rng = np.random.RandomState(1)
    for t in range(3):
        env.step(rng.uniform(-1, 1, (n, 4)).astype(np.float32))
    mask = np.zeros(n, np.uint8); mask[rng.permutation(n)[:n // 3]] = 1
    env.reset(mask=mask)                                   # a third of the envs start a new episode three steps late
    subsets = 0
    env.set_timing(True)
    for t in range(14):
        before = {k: v.copy() for k, v in env.models.items()}
        _, _, done, _ = env.step(rng.uniform(-1, 1, (n, 4)).astype(np.float32))
        assert 0.0 < env.last_kernel_ms() < 50.0
        changed = np.any(env.models["inertia"] != before["inertia"], axis=1)
        assert np.array_equal(changed, done.astype(bool)), t          # exactly the finished envs were re-drawn
        if done.any() and not done.all():
            subsets += 1
    assert subsets >= 3
    rows = np.empty((n, _lib.MODEL_DOUBLES), dtype=np.float64)
    _lib.check(lib.gaq_get_params(env._handle, _lib.ptr(rows), 0, n))
    host = _lib.models_to_rows(env.models)
    bad = np.abs(rows - host) > 1e-6 * np.abs(host) + 1e-15          # (ou_sigma is an fp32 plane on the device)
    assert not bad.any(), np.argwhere(bad)[:5]                        # the device flies with what the host sampled
    twin = QuadrotorEnv(**kw)
    twin.reset()
    _lib.check(lib.gaq_set_params(twin._handle, _lib.ptr(np.ascontiguousarray(rows)), 0, n))
    twin.set_state(env.get_state())
    a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
    env.dynamics_randomize_every = None                               # (keep this step's finished envs on their parameters)
    (o1, r1, d1, _), (o2, r2, d2, _) = env.step(a), twin.step(a)
    keep = ~d1.astype(bool)                                           # finished envs were reset (twin's reset keys differ: other episode counts)
    assert keep.sum() >= n // 4 and np.allclose(o1[keep], o2[keep], rtol=1e-6, atol=1e-6) and np.allclose(r1[keep], r2[keep], atol=1e-6) and np.array_equal(d1, d2)
    env.close(); twin.close()
