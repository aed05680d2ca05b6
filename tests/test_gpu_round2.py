"""-m gpu: what round 2 added, each against the reference's own recorded numbers through the C ABI:
float32-action arithmetic (G13), sensor noise with the reference's recorded draws on the DEVICE (G10), the complete
info dict (G14), the library-owned-heads state layout, the alias guard, the packed multi-GPU row, the mixed residual
rows at full size, scattered parameter updates."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from tests import golden_util as gu
from tests import gpu_util as G
from tests.test_gpu_parity import check_block, handle_for

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("alias", [0, 1, 2])
def test_float32_action_arrays(alias):
    """RawControl fed float32 ARRAYS forms 0.5*(a+1) in float32 (quadrotor_control.py:88-92; fixture G13): the
    `action_f32` mode reproduces the reference's trajectories; the float64-array arithmetic provably does not."""
    d = gu.load("g13_float32_actions")
    for i, blk in enumerate(gu.env_blocks(d)):
        kw = gu.kwargs_of(blk)
        control = 1 if kw.get("raw_control_zero_middle", True) is False else 0
        flags = 8 if kw.get("obs_repr", "").endswith("_act") else 0
        h = handle_for(blk, gu.sub(blk, "const_"), 3, alias=alias, action_f32=1, control=control, obs_flags=flags)
        outs, spread = G.run_blocks(h, [blk], 3)
        check_block(outs[0], blk)
        assert spread == 0.0
        h.close()
        if control == 0 and alias == 0:
            h = handle_for(blk, gu.sub(blk, "const_"), 1, alias=0, action_f32=0, control=control, obs_flags=flags)
            outs, _ = G.run_blocks(h, [blk], 1)
            assert gu.rel_err(outs[0]["obs"], blk["obs"]) > 1e-6      # the other arithmetic is a different trajectory
            # ... and gaq_set_action_dtype switches an existing handle
            from gym_art_amd import _lib
            _lib.check(h.lib.gaq_set_action_dtype(h.h, 1))
            outs, _ = G.run_blocks(h, [blk], 1)
            check_block(outs[0], blk)
            h.close()


@pytest.mark.parametrize("fixture", ["g10_sense_noise", "g16_sense_noise_param_sets"])
def test_sensor_noise_on_device_with_the_reference_draws(fixture):
    """SensorNoise.add_noise on the DEVICE, value for value (fixtures G10, G16): Gaussian + uniform position / velocity noise,
    the small-angle quaternion attitude noise (quat_norm_std != 0), both gyro models incl. the bias random walk over
    180 add_noise calls, accelerometer noise -- fed the draws the reference made (gaq_set_sense_input_dev).  G16: ten random
    parameter sets over the six working observation variants."""
    import torch
    from gym_art_amd import _lib
    d = gu.load(fixture)
    flags = {"xyz_vxyz_R_omega_acc_act": 12, "xyz_vxyz_R_omega": 0, "xyzr_vxyzr_R_omega_h": 3, "xyz_vxyz_R_omega_h": 2, "xyzr_vxyzr_R_omega": 1,
             "xyz_vxyz_R_omega_act": 8}
    for blk in gu.env_blocks(d):
        sn = json.loads(str(blk["sense_json"]))
        sense = {} if sn == "default" else dict(sn)
        n = 3
        h = handle_for(blk, gu.sub(blk, "const_"), n, sense=sense, sense_input=1, obs_flags=flags[str(blk["obs_repr"])])
        keep = []

        def feed(draws):          # [3 calls, 10 slots, 3] -> [3, 12, 3, n] (slots 10 / 11: the t2w / t2t normals, unused here)
            d12 = np.zeros((3, 12, 3), np.float32)
            d12[:, :10] = draws
            buf = torch.tensor(np.repeat(d12[..., None], n, axis=3), device="cuda")
            keep.append(buf)
            _lib.check(h.lib.gaq_set_sense_input_dev(h.h, _lib.ptr(buf)))

        # reset observation: one add_noise call on the reset state (quadrotor.py:1143); call slot 2 carries its draws
        st = np.zeros((42, n))
        st[:39] = G.hh.pack_state(blk["reset_pos"], blk["reset_vel"], blk["reset_rot"], blk["reset_omega"], blk["goal"])[:, None]
        st[39:42] = blk["ctor_gyro_bias"][:, None]
        h.set_state(st)
        draws0 = np.zeros((3, 10, 3))
        draws0[2] = blk["reset_draws"][0]
        feed(draws0)
        o0 = h.observe()
        assert gu.rel_err(o0[0], blk["reset_obs"]) <= 1e-6
        assert np.max(np.abs(h.get_state()[39:42, 0] - blk["reset_gyro_bias"])) <= 1e-7
        # the trajectory: three calls per step, all of which advance the gyro bias
        st[:39] = G.hh.pack_state(blk["init_pos"], blk["init_vel"], blk["init_rot"], blk["init_omega"], blk["goal"],
                                  svd_ctr=int(round(float(blk["init_svd"]) / float(blk["dt"]))))[:, None]
        st[39:42] = blk["reset_gyro_bias"][:, None]
        h.set_state(st)
        T = blk["obs"].shape[0]
        worst = 0.0
        for t in range(T):
            feed(blk["draws"][t])
            a = np.repeat(blk["actions"][t][None], n, axis=0)
            obs, rew, done = h.step(a)
            worst = max(worst, gu.rel_err(obs[0], blk["obs"][t]))
            assert np.array_equal(obs[0], obs[1]) and np.array_equal(obs[0], obs[2])
            assert abs(rew[0] - blk["reward"][t]) <= 2e-7
            assert np.max(np.abs(h.get_state()[39:42, 0] - blk["gyro_bias"][t])) <= 2e-7, t
        assert worst <= 1e-6, worst
        if sense.get("gyro_norm_std", 0) and sense.get("gyro_noise_density", 0.000175):
            assert np.abs(blk["gyro_bias"][-1]).max() > 0
        with pytest.raises(Exception):      # a step without fresh draws is a call-sequence error
            h.step(a)
        h.close()


def test_info_dict_matches_the_reference():
    """info["obs_comp"] / info["dyn_params"] (quadrotor.py:994-1025) through the env class, every numeric entry, for
    RawControl (Hummingbird, CrazyFlie) and the Mellinger controller (fixture G14)."""
    from gym_art_amd import QuadrotorEnv
    from tests import hh
    d = gu.load("g14_info_dict")
    for blk in gu.env_blocks(d):
        kw = gu.kwargs_of(blk)
        env = QuadrotorEnv(dynamics_params=str(blk["model"]), dynamics_change={"noise": {"thrust_noise_ratio": 0.}},
                           ep_time=5, seed=0, **kw)
        st = hh.pack_state(blk["init_pos"], blk["init_vel"], blk["init_rot"], blk["init_omega"], blk["goal"])
        env.set_state(np.concatenate([st, np.zeros(3)])[:, None])
        for t in range(blk["obs"].shape[0]):
            o, r, dn, info = env.step(blk["actions"][t])          # float64 arrays, like the fixture's calls
            assert gu.rel_err(o, blk["obs"][t]) <= 1e-6
            comp = info["obs_comp"]
            assert set(comp) == {"xyz", "vxyz", "acc", "omega", "omega_dot", "R", "act", "act_clipped", "act_filtered",
                                 "act_torque", "torque"}
            for k, v in comp.items():
                ref = blk["info_obs_comp_" + k][t]
                assert gu.rel_err(np.asarray(v[0]), ref) <= 1e-6, (k, t, v[0], ref)
        dyn = info["dyn_params"]
        assert set(dyn) == {"mass", "motor_linearity", "motor_time_up", "motor_time_down", "motor_assymetry", "motor_pos",
                            "motor_ccw", "t2w", "t2t", "t2i", "inertia", "thrust_max", "torque_max", "arm", "grav", "dt"}
        for k, v in dyn.items():
            assert gu.rel_err(np.asarray(v[0], dtype=np.float64), blk["info_dyn_params_" + k][0]) <= 1e-9, k
        env.close()


def test_library_owned_heads_layout_is_safe_and_equal():
    """obs_state_alias = 2 (the Python class's default): the split state lives in library-owned rows and the caller's tensor
    gets a copy -- scribbling on it changes nothing, and the trajectory equals the alias layout's bit for bit."""
    import torch
    from gym_art_amd import QuadrotorEnv
    n, T = 3000, 40
    kw = dict(num_envs=n, ep_time=0.1, seed=11)                       # auto-resets inside the run, noise on
    envs = {name: QuadrotorEnv(alias_obs=a, **kw) for name, a in (("shadow", None), ("alias", True), ("plain", False))}
    assert [envs[k].state_layout for k in ("shadow", "alias", "plain")] == [2, 1, 0]
    assert envs["alias"].obs_is_state and not envs["shadow"].obs_is_state
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    out = {}
    for name, env in envs.items():
        obs = torch.empty((n, 18), device=dev); rew = torch.empty(n, device=dev); done = torch.empty(n, dtype=torch.uint8, device=dev)
        env.reset_dev(obs)
        gen.manual_seed(3)
        frames = []
        for t in range(T):
            a = torch.rand((n, 4), device=dev, generator=gen) * 2 - 1
            env.step_dev(a, obs, rew, done)
            frames.append(torch.cat([obs, rew[:, None], done.float()[:, None]], 1).cpu().numpy())
            if name != "alias":
                obs.fill_(123.0)                                       # the caller's copy is the caller's
        out[name] = np.stack(frames)
    assert np.array_equal(out["shadow"], out["alias"])
    assert np.max(np.abs(out["shadow"][..., :18] - out["plain"][..., :18])) <= 5e-6    # truncated vs rounded heads
    # fused rollouts and host-pointer steps go through the same library-owned rows
    env = envs["shadow"]
    acts = torch.rand((8, n, 4), device=dev, generator=gen) * 2 - 1
    oT = torch.empty((8, n, 18), device=dev); rT = torch.empty((8, n), device=dev); dT = torch.empty((8, n), dtype=torch.uint8, device=dev)
    ref = envs["alias"]
    o2, r2, d2 = torch.empty_like(oT), torch.empty_like(rT), torch.empty_like(dT)
    env.step_many_dev(acts, oT, rT, dT)
    ref.step_many_dev(acts, o2, r2, d2)
    oT.fill_(7.0)
    a = np.zeros((n, 4), np.float32)
    x1, _, _, _ = env.step(a)
    x2, _, _, _ = ref.step(a)
    torch.cuda.synchronize()
    assert torch.equal(r2, rT) and np.array_equal(x1, x2)
    for e in envs.values():
        e.close()


def test_alias_guard_catches_a_modified_observation():
    """GAQ_CHECK_ALIAS=1: in alias mode 1 an in-place edit of the returned observation tensor is an error, not silent
    corruption of the physics (ADVICE r1 / VERDICT r1 item 9)."""
    import torch
    from gym_art_amd import QuadrotorEnv, _lib
    os.environ["GAQ_CHECK_ALIAS"] = "1"
    try:
        env = QuadrotorEnv(num_envs=1000, ep_time=5, seed=2, alias_obs=True)
    finally:
        os.environ.pop("GAQ_CHECK_ALIAS", None)
    dev = torch.device("cuda", 0)
    obs = torch.empty((1000, 18), device=dev); rew = torch.empty(1000, device=dev); done = torch.empty(1000, dtype=torch.uint8, device=dev)
    a = torch.zeros((1000, 4), device=dev)
    env.reset_dev(obs)
    for _ in range(3):
        env.step_dev(a, obs, rew, done)              # untouched: fine
    obs[17, 3] += 1e-3                                # e.g. an in-place normaliser
    with pytest.raises(_lib.GaqError, match="modified"):
        env.step_dev(a, obs, rew, done)
    env.close()


def test_packed_rows_kernel():
    """gaq_pack_rows_dev: [obs | reward | done] rows of the multi-GPU single collective, bit for bit."""
    import torch
    from gym_art_amd import QuadrotorEnv
    for kw, D in ((dict(), 18), (dict(obs_repr="xyz_vxyz_R_omega_acc_act"), 25)):
        n = 777
        env = QuadrotorEnv(num_envs=n, ep_time=0.05, seed=4, **kw)
        dev = torch.device("cuda", 0)
        obs = torch.empty((n, D), device=dev); rew = torch.empty(n, device=dev); done = torch.empty(n, dtype=torch.uint8, device=dev)
        rows = torch.full((n, D + 2), -1.0, device=dev)
        env.reset_dev(obs)
        for t in range(6):                 # ep_len 5: the sixth step reports done for every env
            env.step_dev(torch.rand((n, 4), device=dev) * 2 - 1, obs, rew, done)
        env.pack_rows_dev(obs, rew, done, rows)
        torch.cuda.synchronize()
        assert torch.equal(rows[:, :D], obs) and torch.equal(rows[:, D], rew) and torch.equal(rows[:, D + 1], done.float())
        assert done.sum().item() == n
        env.close()


def test_scattered_parameter_updates_touch_only_their_envs():
    """gaq_set_params_indexed on a few envs of a large handle: the other envs' parameters, SVD counters and OU states stay,
    the touched ones are replaced / cleared; kernel-selection flags follow running counts (no O(N) scan)."""
    from gym_art_amd import QuadrotorEnv
    n = 1 << 16
    env = QuadrotorEnv(dynamics_params="Crazyflie", num_envs=n, ep_time=5, seed=7,
                       dyn_sampler_1={"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"})
    rng = np.random.RandomState(0)
    a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
    for _ in range(3):
        env.step(a)
    st0 = env.get_state()
    mass0 = env.models["mass"].copy()
    ids = np.sort(rng.choice(n, 37, replace=False))
    env.resample_dynamics(env_ids=ids)
    st1 = env.get_state()
    other = np.ones(n, bool); other[ids] = False
    assert np.array_equal(st0[:, other], st1[:, other])
    assert np.all(st1[26:30, ids] == 0) and np.all(st1[38, ids] == 0) and np.all(st0[38, ids] == 6)
    assert np.all(env.models["mass"][ids] != mass0[ids]) and np.array_equal(env.models["mass"][other], mass0[other])
    o1, _, _, _ = env.step(a)
    assert np.isfinite(o1).all()
    env.close()


# ---- the parameter pipeline on the device (quad_params_dev.hpp in the rerandomize kernel) -----------------------------
def _tree_rows_of(prm_blocks):
    from gym_art_amd import quad_params as qp
    from tests.test_quad_params_dev import tree_from_flat_params
    return qp.flatten_tree(qp.batch_tree([tree_from_flat_params(p) for p in prm_blocks]))


def test_device_derivation_matches_the_reference_constants_and_trajectories():
    """QuadLink + update_model ON THE DEVICE (gaq_set_param_trees) for the reference's own parameter dicts: the three shipped
    models (G4b) and the 32 RelativeSampler draws of G4 -- derived constants read back with gaq_get_params <= 1e-12, then the
    G4 trajectories flown with those device-built planes (compact parameter path) against the reference."""
    from gym_art_amd import _lib
    d4b, g4 = gu.load("g4b_models"), gu.load("g4_randomized")
    blocks = gu.env_blocks(g4)
    prm = [gu.sub(d4b, nm + "_param_") for nm in ("DefaultQuad", "Crazyflie", "MediumQuad")] + [gu.sub(b, "param_") for b in blocks]
    const = [gu.sub(d4b, nm + "_const_") for nm in ("DefaultQuad", "Crazyflie", "MediumQuad")] + [gu.sub(b, "const_") for b in blocks]
    n = len(prm)
    trees = np.ascontiguousarray(_tree_rows_of(prm))
    b0 = blocks[0]
    h = G.Handle(n, float(b0["dt"]), int(b0["sim_steps"]), int(b0["ep_len"]), per_env=1)
    _lib.check(h.lib.gaq_set_param_trees(h.h, _lib.ptr(trees), 0, 0, n))
    rows = np.empty((n, _lib.MODEL_DOUBLES))
    _lib.check(h.lib.gaq_get_params(h.h, _lib.ptr(rows), 0, n))
    m = _lib.rows_to_models(rows)
    for i, c in enumerate(const):
        for key, ref in (("mass", c["mass"]), ("inertia", c["inertia"]), ("thrust_max", c["thrust_max"]), ("torque_max", c["torque_max"]),
                         ("prop_pos", np.asarray(c["prop_pos"]).reshape(12)), ("arm", c["arm"]), ("damp_time_up", c["damp_time_up"]),
                         ("damp_time_down", c["damp_time_down"]), ("linearity", c["motor_linearity"])):
            assert gu.rel_err(m[key][i], ref) <= 1e-12, (i, key)
        assert abs(m["ou_sigma"][i] - float(c["thrust_noise_sigma"])) <= 1e-8          # an fp32 plane
    with pytest.raises(_lib.GaqError):          # the two parameter paths do not mix
        _lib.check(h.lib.gaq_set_params(h.h, _lib.ptr(rows), 0, n))
    h.close()
    # trajectories of G4 with device-derived planes, both layouts
    for alias in (0, 1):
        nb = len(blocks)
        h = G.Handle(nb, float(b0["dt"]), int(b0["sim_steps"]), int(b0["ep_len"]), per_env=1, alias=alias)
        t2 = np.ascontiguousarray(_tree_rows_of([gu.sub(b, "param_") for b in blocks]))
        _lib.check(h.lib.gaq_set_param_trees(h.h, _lib.ptr(t2), 0, 0, nb))
        outs, _ = G.run_blocks(h, blocks, nb)
        for o, b in zip(outs, blocks):
            check_block(o, b)
        h.close()


def test_device_sampler_distribution_and_rerandomize_every():
    """RelativeSampler on the device through the env class: the distribution of derived constants equals the host
    pipeline's (KS); dynamics_randomize_every = 2 re-draws exactly the envs whose (k + 1) % 2 == 0 after their k-th
    finished episode, clears their SVD counter / OU state, leaves everybody else's parameters alone."""
    import torch
    from scipy import stats
    from gym_art_amd import QuadrotorEnv
    sampler = {"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"}
    n = 1 << 15
    dev_env = QuadrotorEnv(dynamics_params="Crazyflie", num_envs=n, ep_time=5, seed=21, dyn_sampler_1=sampler, thrust_noise="off",
                           auto_reset=False)
    host_env = QuadrotorEnv(dynamics_params="Crazyflie", num_envs=n, ep_time=5, seed=22, dyn_sampler_1=sampler,
                            randomize_on_device=False)
    assert dev_env._dev_rand and not host_env._dev_rand
    md, mh = dev_env.models, host_env.models
    for key in ("mass", "arm", "linearity", "damp_time_up", "ou_sigma"):
        assert stats.ks_2samp(md[key], mh[key]).pvalue > 1e-4, key
    for key, col in (("inertia", 0), ("inertia", 2), ("thrust_max", 1), ("torque_max", 3), ("prop_pos", 0), ("prop_pos", 4)):
        assert stats.ks_2samp(md[key][:, col], mh[key][:, col]).pvalue > 1e-4, (key, col)
    # the sampled trees read back are the ones the planes were derived from (host derivation of the same trees)
    from gym_art_amd import quad_params as qp
    again, extra = qp.derive_models(dev_env.sampled_trees())
    assert gu.rel_err(again["inertia"], md["inertia"]) <= 1e-12 and gu.rel_err(again["thrust_max"], md["thrust_max"]) <= 1e-12
    assert np.allclose(dev_env.models_extra["motor_assymetry"].sum(1), 4.0)
    # a step with device-derived parameters against the oracle fed the read-back constants
    from oracle import quad_oracle as qo
    k = 512
    p = qo.Params(k, mass=md["mass"][:k], inertia=md["inertia"][:k], thrust_max=md["thrust_max"][:k], torque_max=md["torque_max"][:k],
                  prop_pos=md["prop_pos"][:k].reshape(k, 4, 3), damp_time_up=md["damp_time_up"][:k], damp_time_down=md["damp_time_down"][:k],
                  linearity=md["linearity"][:k], arm=md["arm"][:k], ou_sigma=0 * md["ou_sigma"][:k], vel_damp=md["vel_damp"][:k],
                  damp_omega_quadratic=md["damp_omega_quadratic"][:k], C_drag=md["c_drag"][:k], C_roll=md["c_roll"][:k])
    st = dev_env.get_state()
    s = qo.State(k)
    s.goal[:] = st[34:37, :k].T
    s.set_state(st[0:3, :k].T, st[3:6, :k].T, st[6:15, :k].T.reshape(k, 3, 3), st[15:18, :k].T)
    cfg = qo.Config(sim_freq=200., sim_steps=2, ep_time=5)
    cfg.action_f32 = True                                    # float32 arrays in: float32 arithmetic on both sides
    rng = np.random.RandomState(4)
    for t in range(40):
        a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
        o, r, dn, _ = dev_env.step(a)
        o_ref, r_ref, _ = qo.env_step(s, p, cfg, a[:k].astype(np.float64))
        assert gu.rel_err(o[:k], o_ref) <= 1e-6 and np.max(np.abs(r[:k] - r_ref)) <= 2e-7, t
    dev_env.close(); host_env.close()

    # per-episode re-randomisation on the device
    n, every = 4096, 2
    env = QuadrotorEnv(dynamics_params="Crazyflie", num_envs=n, ep_time=0.05, seed=5, dyn_sampler_1=sampler,
                       dynamics_randomize_every=every, thrust_noise="off")          # ep_len 5 -> 6 steps per episode
    assert env._dev_rand and env.ep_len == 5
    dev = torch.device("cuda", 0)
    obs = torch.empty((n, 18), device=dev); rew = torch.empty(n, device=dev); done = torch.empty(n, dtype=torch.uint8, device=dev)
    env.reset_dev(obs)
    # stagger the episode phases so that different envs finish at different steps
    st = env.get_state(); st[37] = np.arange(n) % 6; env.set_state(st)
    mass_prev = env.models["mass"].copy()
    finished = np.zeros(n, dtype=np.int64)
    changed_total = 0
    for t in range(30):
        env.step_dev(torch.rand((n, 4), device=dev) * 2 - 1, obs, rew, done)
        torch.cuda.synchronize()
        dn = done.cpu().numpy().astype(bool)
        finished[dn] += 1
        due = dn & ((finished + 1) % every == 0)
        mass_now = env.models["mass"]
        assert np.all(mass_now[due] != mass_prev[due]) and np.array_equal(mass_now[~due], mass_prev[~due]), t
        if due.any():
            s = env.get_state()
            assert np.all(s[38, due] == 0) and np.all(s[26:30, due] == 0)
        changed_total += int(due.sum())
        mass_prev = mass_now.copy()
    assert changed_total > n            # everybody was re-drawn at least once on average
    assert np.isfinite(obs.cpu().numpy()).all()
    env.check_finite()
    env.close()


def test_swarm_at_the_largest_supported_world_size_and_desynchronised_ticks():
    """Swarm layer (own specification, parity-unpinned) at 16 agents per world: 108-word observation rows = 110 KB of LDS per
    workgroup, above the 64 KB a launch gets by default -- observation / terminal rows against the
    specification's restatement; then a world whose agents' episode clocks were desynchronised by set_state: the
    neighbour exchange (wave shuffles) must still see every agent (it used to sit in the finishing lanes' branch)."""
    import torch
    from gym_art_amd import QuadrotorEnvMulti
    from oracle import quad_oracle as qo
    A, W = 16, 40
    n = A * W
    env = QuadrotorEnvMulti(num_agents=A, num_worlds=W, ep_time=0.05, seed=23, thrust_noise="off")      # ep_len 5, auto-reset
    assert env.obs_dim == 18 + 6 * (A - 1)
    dev = torch.device("cuda", 0)
    obs = torch.empty((n, env.obs_dim), device=dev); rew = torch.empty(n, device=dev); done = torch.empty(n, dtype=torch.uint8, device=dev)
    term = torch.zeros((n, env.obs_dim), device=dev)
    env.set_terminal_obs(term)
    env.reset_dev(obs)
    st = env.get_state()
    st[37, 3] = 4; st[37, 21] = 2                       # two agents (worlds 0 and 1) run ahead of their worlds
    env.set_state(st)
    rng = np.random.RandomState(1)
    for t in range(9):
        before = env.get_state()
        a = torch.tensor(rng.uniform(-1, 1, (n, 4)).astype(np.float32), device=dev)
        env.step_dev(a, obs, rew, done)
        torch.cuda.synchronize()
        s = env.get_state()
        o = obs.cpu().numpy()
        assert np.allclose(o[:, 18:], qo.swarm_obs(s[0:3].T, s[3:6].T, A), atol=2e-6), t
        dn = done.cpu().numpy().astype(bool)
        lone = {1: 3, 3: 21}.get(t)                     # the agent whose clock was set ahead finishes alone in its world
        if lone is not None:
            w0 = (lone // A) * A
            assert dn[lone] and dn[w0:w0 + A].sum() == 1
            tr = term.cpu().numpy()[lone]
            goal = before[34:37, lone]
            pos_i = tr[0:3] + goal                      # the finishing agent's last position, from its own terminal row
            pos = s[0:3].T
            for j in range(1, A):                       # neighbour (a + j) mod A: still flying, its state is the current one
                nb = w0 + (lone - w0 + j) % A
                assert np.allclose(tr[18 + 6 * (j - 1):18 + 6 * (j - 1) + 3], pos[nb] - pos_i, atol=3e-6), (t, j)
        elif dn.any():
            assert np.isfinite(term.cpu().numpy()[dn]).all()
    assert np.isfinite(obs.cpu().numpy()).all()
    env.close()
    with pytest.raises(ValueError):
        QuadrotorEnvMulti(num_agents=32, num_worlds=4)


def test_size_specific_kernel_instantiations_are_bit_identical():
    """Batch-size-specific kernel instantiations -- OU normals drawn under the load latency (F_PREDRAW), non-temporal cache
    policy on the streaming loads / stores (F_NT) -- compute the same trajectories as the ordinary ones, bit for bit: same
    Philox blocks, same arithmetic.  Hummingbird, uniform CrazyFlie (motor lag), per-env CrazyFlie, with in-kernel resets."""
    from gym_art_amd import QuadrotorEnv
    sampler = {"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"}
    for kw in (dict(dynamics_params="DefaultQuad"), dict(dynamics_params="Crazyflie"),
               dict(dynamics_params="Crazyflie", dyn_sampler_1=sampler, randomize_on_device=False)):
        n = 5000
        outs = []
        for predraw, nt in (("0", "0"), ("1", "0"), ("0", "1"), ("1", "1")):
            os.environ["GAQ_PREDRAW"], os.environ["GAQ_NT"] = predraw, nt
            try:
                env = QuadrotorEnv(num_envs=n, ep_time=0.1, seed=31, alias_obs=True, **kw)
            finally:
                os.environ.pop("GAQ_PREDRAW", None); os.environ.pop("GAQ_NT", None)
            rng = np.random.RandomState(6)
            frames = [env.reset()]
            for t in range(25):
                o, r, d, _ = env.step(rng.uniform(-1, 1, (n, 4)).astype(np.float32))
                frames.append(o); frames.append(np.repeat(r[:, None], 18, 1)); frames.append(np.repeat(d[:, None].astype(np.float32), 18, 1))
            outs.append(np.stack(frames))
            env.close()
        for o in outs[1:]:
            assert np.array_equal(outs[0], o)
        assert outs[0][3::3].sum() > 0          # episodes ended inside the run


def test_patched_import_observation_variants_on_device():
    """The quaternion / t2w / t2t observation variants (get_state.py:276-384; complete in the reference but NameError as
    shipped -- PATCHED-IMPORT fixture G15) through the C ABI, with the reference's recorded draws."""
    import torch
    from gym_art_amd import _lib
    from tests import hh
    d = gu.load("g15_obs_variants_patched_imports")
    for blk in gu.env_blocks(d):
        sn = json.loads(str(blk["sense_json"]))
        sense = None if sn is None else ({} if sn == "default" else dict(sn))
        name = str(blk["obs_repr"])
        n = 2
        needs_draws = sense is not None or "t2w" in name        # (the plain quaternion observation draws nothing)
        h = handle_for(blk, gu.sub(blk, "const_"), n, sense=sense, sense_input=int(needs_draws), obs_flags=hh.OBS_FLAGS_PATCHED[name])
        assert h.D == blk["obs"].shape[1]
        st = np.zeros((42, n))
        st[:39] = hh.pack_state(blk["init_pos"], blk["init_vel"], blk["init_rot"], blk["init_omega"], blk["goal"],
                                svd_ctr=int(round(float(blk["init_svd"]) / float(blk["dt"]))))[:, None]
        st[39:42] = blk["init_gyro_bias"][:, None]
        h.set_state(st)
        keep = []
        for t in range(blk["obs"].shape[0]):
            if needs_draws:
                buf = torch.tensor(np.repeat(np.asarray(blk["draws"][t], np.float32)[..., None], n, axis=3), device="cuda")
                keep.append(buf)
                _lib.check(h.lib.gaq_set_sense_input_dev(h.h, _lib.ptr(buf)))
            obs, rew, done = h.step(np.repeat(blk["actions"][t][None], n, axis=0))
            tol = 1e-6 * max(1.0, 0.05 / float(blk["obs"][t][6]) ** 2) if "quat" in name else 1e-6
            assert gu.rel_err(obs[0], blk["obs"][t]) <= tol, (name, t)
            assert np.array_equal(obs[0], obs[1]) and abs(rew[0] - blk["reward"][t]) <= 2e-7
        if sn is not None:
            assert np.max(np.abs(h.get_state()[39:42, 0] - blk["gyro_bias"][-1])) <= 2e-7
        h.close()
    # through the env class: spaces and shapes, device RNG for the t2w noise
    from gym_art_amd import QuadrotorEnv
    env = QuadrotorEnv(obs_repr="xyz_vxyz_R_omega_t2w_t2t", num_envs=4096, ep_time=5, seed=3, t2w_std=0.1)
    assert env.obs_dim == 20 and env.observation_space.shape == (20,)
    o = env.reset()
    t2w = 2.8                                            # DefaultQuad (quad_models.py:45-85)
    assert abs(o[:, 18].mean() - (t2w - 1.5) / 8.5) < 2e-3 and abs(o[:, 18].std() - 0.05 * t2w / 8.5) < 1.5e-3
    assert np.all(o[:, 19] >= 0) and np.all(o[:, 19] <= 1)
    env.close()
    env = QuadrotorEnv(obs_repr="xyz_vxyz_quat_omega", num_envs=64, ep_time=5, seed=3)
    o, _, _, _ = env.step(np.zeros((64, 4), np.float32))
    assert o.shape == (64, 13) and np.allclose(np.linalg.norm(o[:, 6:10], axis=1), 1.0, atol=1e-6)
    env.close()


def test_bound_step_equals_step_dev():
    """QuadrotorEnv.bind_step (pointer / stream look-ups hoisted out of the loop) launches the same step as step_dev."""
    import torch
    from gym_art_amd import QuadrotorEnv
    n = 4096
    a_env, b_env = (QuadrotorEnv(num_envs=n, ep_time=0.1, seed=13) for _ in range(2))
    dev = torch.device("cuda", 0)
    bufs = [[torch.empty((n, 18), device=dev), torch.empty(n, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)] for _ in range(2)]
    acts = [torch.rand((n, 4), device=dev) * 2 - 1 for _ in range(4)]
    a_env.reset_dev(bufs[0][0]); b_env.reset_dev(bufs[1][0])
    bound = [b_env.bind_step(a, *bufs[1]) for a in acts]
    for t in range(30):
        a_env.step_dev(acts[t % 4], *bufs[0])
        bound[t % 4]()
    torch.cuda.synchronize()
    for x, y in zip(bufs[0], bufs[1]):
        assert torch.equal(x, y)
    a_env.close(); b_env.close()


def test_graph_captured_step_with_device_rerandomisation_equals_eager():
    """A captured step launch sequence -- step kernel (with the in-launch promotion), the refill pass, the step-index bump -- replayed K times
    equals K eager steps: the finished envs' new parameters are functions of (seed, env, resample count), all device-resident."""
    import torch
    from gym_art_amd import QuadrotorEnv
    sampler = {"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"}
    n, K = 4096, 30
    kw = dict(dynamics_params="Crazyflie", num_envs=n, ep_time=0.05, seed=17, dyn_sampler_1=sampler, dynamics_randomize_every=1)
    eager, graphed = QuadrotorEnv(**kw), QuadrotorEnv(**kw)
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev); gen.manual_seed(2)
    acts = torch.rand((K + 1, n, 4), device=dev, generator=gen) * 2 - 1
    mk = lambda: (torch.empty((n, 18), device=dev), torch.empty(n, device=dev), torch.empty(n, dtype=torch.uint8, device=dev))
    (o_e, r_e, d_e), (o_g, r_g, d_g) = mk(), mk()
    a_g = torch.empty((n, 4), device=dev)
    eager.reset_dev(o_e); graphed.reset_dev(o_g)
    graphed.set_graph_safe(True)
    a_g.copy_(acts[0])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graphed.step_dev(a_g, o_g, r_g, d_g)
    torch.cuda.current_stream().wait_stream(side)
    eager.step_dev(acts[0], o_e, r_e, d_e)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        graphed.step_dev(a_g, o_g, r_g, d_g)
    for t in range(1, K + 1):
        a_g.copy_(acts[t])
        g.replay()
        eager.step_dev(acts[t], o_e, r_e, d_e)
    torch.cuda.synchronize()
    assert torch.equal(o_e, o_g) and torch.equal(r_e, r_g) and torch.equal(d_e, d_g)
    graphed.set_graph_safe(False)
    me, mg = eager.models, graphed.models
    assert np.array_equal(me["mass"], mg["mass"]) and np.array_equal(me["inertia"], mg["inertia"])
    assert len(np.unique(me["mass"])) > 0.99 * n
    eager.close(); graphed.close()


@pytest.mark.parametrize("opts", [
    dict(obs_flags=1), dict(obs_flags=2), dict(obs_flags=3, sense={}), dict(obs_flags=12), dict(obs_flags=8, rew={"action_change": 0.2}),
    dict(sense={}), dict(obs_flags=1, sense={"quat_norm_std": 0.02, "pos_unif_range": 0.01}),
])
def test_split_state_with_a_packed_observation_matches_the_plain_layout(opts):
    """Observations that are not the 18 state heads (body frame, appended height / accelerometer / previous action, sensor
    noise) used to need the fp64 state planes; the F_PACK kernels keep the split state (library-owned heads + residual rows)
    and pack the observation beside it.  Same RNG keys, same arithmetic: equal to the plain layout up to the 39-bit storage."""
    from gym_art_amd import _lib
    from tests.test_gpu_properties import hummingbird_const, actions_for
    n, T = 4096, 30
    d3 = gu.load("g3_crazyflie")
    for const, noise in ((hummingbird_const(0.01), 1), (dict(gu.sub(d3, "const_")), 0)):
        split = G.Handle(n, 0.005, 2, 10, const=const, noise=noise, auto_reset=1, seed=41, alias=2, **opts)
        plain = G.Handle(n, 0.005, 2, 10, const=const, noise=noise, auto_reset=1, seed=41, alias=0, **opts)
        assert split.lib.gaq_state_layout(split.h) == 2 and plain.lib.gaq_state_layout(plain.h) == 0 and not split.alias
        assert split.D == plain.D
        os_, op = split.reset(), plain.reset()
        assert np.allclose(os_, op, rtol=0, atol=2e-6)
        assert np.allclose(split.observe(), plain.observe(), rtol=0, atol=2e-6) or opts.get("sense") is not None   # (noisy obs: new draws per call)
        for t in range(T):
            act = actions_for(t, n, seed=9)
            (os_, rs, ds), (op, rp, dp) = split.step(act), plain.step(act)
            assert np.allclose(os_, op, rtol=0, atol=5e-6) and np.allclose(rs, rp, rtol=0, atol=3e-7), t
            assert np.array_equal(ds, dp)
        assert np.allclose(split.get_state()[0:18], plain.get_state()[0:18], rtol=0, atol=2e-6)
        assert ds.sum() + dp.sum() >= 0
        split.close(); plain.close()


def test_plain_c_program_drives_the_library():
    """examples/c_abi_demo.c: include/gaq.h and libgaq.so from plain C (gcc; no Python, torch or HIP headers) -- create, reset,
    100 host-pointer steps with a hover-ish action, destroy.  The boundary is a C ABI and nothing else."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "c_abi_demo")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(root, "include"), "-o", exe,
                           os.path.join(root, "examples", "c_abi_demo.c"), "-L" + os.path.join(root, "gym_art_amd"), "-lgaq",
                           "-Wl,-rpath," + os.path.join(root, "gym_art_amd"), "-lm"])
    r = subprocess.run([exe, "4096", "100"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["obs_dim"] == 18 and out["state_layout"] == 2 and out["episodes_finished"] == 0
    assert -0.05 < out["mean_reward"] < 0.05 and all(abs(x) <= 1.0001 for x in out["R0_diag"])
    # ... and the same batch as one SHARDED handle (gaq_create_sharded, three shards on this box's GPU): the same line, digit for digit
    r3 = subprocess.run([exe, "4096", "100", "3"], capture_output=True, text=True, timeout=300)
    assert r3.returncode == 0, r3.stderr
    assert "shard 2: envs [2752, 4096) on device 0" in r3.stderr
    assert json.loads(r3.stdout.strip().splitlines()[-1]) == out


def test_random_quad_on_the_device():
    """dynamics_params="RandomQuad" (the reference's random-quadrotor sampler, one draw per env and per episode) with the
    sampler and the density-based QuadLink on the device: derived constants of the reference's own 16 RandomQuad draws (G4b)
    through gaq_set_param_trees(links_by_density=1) <= 1e-12; the class path: distribution = host pipeline's, trajectories of
    device-sampled quads = the oracle's with the read-back constants, per-episode re-randomisation."""
    import torch
    from scipy import stats
    from gym_art_amd import QuadrotorEnv, _lib, quad_params as qp
    from tests.test_quad_params_dev import tree_from_flat_params
    d = gu.load("g4b_models")
    nr = int(d["n_random"])
    trees = np.ascontiguousarray(qp.flatten_tree(qp.batch_tree([tree_from_flat_params(gu.sub(d, "rq%d_param_" % i)) for i in range(nr)])))
    h = G.Handle(nr, 0.005, 2, 500, per_env=1)
    _lib.check(h.lib.gaq_set_param_trees(h.h, _lib.ptr(trees), 1, 0, nr))
    rows = np.empty((nr, _lib.MODEL_DOUBLES))
    _lib.check(h.lib.gaq_get_params(h.h, _lib.ptr(rows), 0, nr))
    m = _lib.rows_to_models(rows)
    for i in range(nr):
        c = gu.sub(d, "rq%d_const_" % i)
        for key, ref in (("mass", c["mass"]), ("inertia", c["inertia"]), ("thrust_max", c["thrust_max"]), ("torque_max", c["torque_max"]),
                         ("prop_pos", np.asarray(c["prop_pos"]).reshape(12)), ("arm", c["arm"]), ("damp_time_up", c["damp_time_up"])):
            assert gu.rel_err(m[key][i], ref) <= 1e-12, (i, key)
    h.close()

    n = 1 << 14
    dev_env = QuadrotorEnv(dynamics_params="RandomQuad", num_envs=n, ep_time=5, seed=3, thrust_noise="off", auto_reset=False)
    host_env = QuadrotorEnv(dynamics_params="RandomQuad", num_envs=n, ep_time=5, seed=4, thrust_noise="off", auto_reset=False,
                            randomize_on_device=False)
    assert dev_env._dev_rand and not host_env._dev_rand
    md, mh = dev_env.models, host_env.models
    for key in ("mass", "arm", "damp_time_up"):
        assert stats.ks_2samp(md[key], mh[key]).pvalue > 1e-4, key
    for key, col in (("inertia", 0), ("inertia", 2), ("thrust_max", 2), ("prop_pos", 1)):
        assert stats.ks_2samp(md[key][:, col], mh[key][:, col]).pvalue > 1e-4, (key, col)
    again, _ = qp.derive_models(dev_env.sampled_trees())
    assert gu.rel_err(again["inertia"], md["inertia"]) <= 1e-12 and gu.rel_err(again["mass"], md["mass"]) <= 1e-12
    # trajectories against the oracle with the read-back constants
    from oracle import quad_oracle as qo
    k = 256
    p = qo.Params(k, mass=md["mass"][:k], inertia=md["inertia"][:k], thrust_max=md["thrust_max"][:k], torque_max=md["torque_max"][:k],
                  prop_pos=md["prop_pos"][:k].reshape(k, 4, 3), damp_time_up=md["damp_time_up"][:k], damp_time_down=md["damp_time_down"][:k],
                  linearity=md["linearity"][:k], arm=md["arm"][:k], ou_sigma=0 * md["ou_sigma"][:k], vel_damp=md["vel_damp"][:k],
                  damp_omega_quadratic=md["damp_omega_quadratic"][:k], C_drag=md["c_drag"][:k], C_roll=md["c_roll"][:k])
    st = dev_env.get_state()
    s = qo.State(k)
    s.goal[:] = st[34:37, :k].T
    s.set_state(st[0:3, :k].T, st[3:6, :k].T, st[6:15, :k].T.reshape(k, 3, 3), st[15:18, :k].T)
    cfg = qo.Config(sim_freq=200., sim_steps=2, ep_time=5)
    cfg.action_f32 = True
    rng = np.random.RandomState(4)
    for t in range(40):
        a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
        o, r, dn, _ = dev_env.step(a)
        o_ref, r_ref, _ = qo.env_step(s, p, cfg, a[:k].astype(np.float64))
        assert gu.rel_err(o[:k], o_ref) <= 1e-6 and np.max(np.abs(r[:k] - r_ref)) <= 2e-7, t
    dev_env.close(); host_env.close()
    # a new random quadrotor for every finished episode
    env = QuadrotorEnv(dynamics_params="RandomQuad", num_envs=2048, ep_time=0.05, seed=9, dynamics_randomize_every=1)
    dev = torch.device("cuda", 0)
    obs = torch.empty((2048, 18), device=dev); rew = torch.empty(2048, device=dev); done = torch.empty(2048, dtype=torch.uint8, device=dev)
    env.reset_dev(obs)
    m0 = env.models["mass"].copy()
    for t in range(6):
        env.step_dev(torch.rand((2048, 4), device=dev) * 2 - 1, obs, rew, done)
    torch.cuda.synchronize()
    assert done.all() and np.all(env.models["mass"] != m0)
    env.check_finite()
    env.close()


def test_staged_rerandomisation_parameter_sequences():
    """The in-launch promotion of dynamics_randomize_every: whatever the episode phases -- every env of a wave finishing at once (the
    plane-wise copy), one env per wave (the lane-parallel copy), clocks moved by set_state right before an episode ends -- env i
    flies draw k of ITS sampler stream during its (k * every)-th ... episode: the derived constants read back equal the host derivation
    of the trees read back, both handles agree on every env with the same number of finished episodes, and no env is ever promoted to
    planes that were not refilled (the device-side overrun check behind gaq_nan_count)."""
    import torch
    from gym_art_amd import QuadrotorEnv, quad_params as qp
    sampler = {"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"}
    n, ep_len = 8192, 9
    dev = torch.device("cuda", 0)
    kw = dict(dynamics_params="Crazyflie", num_envs=n, ep_time=ep_len / 100.0, seed=23, dyn_sampler_1=sampler, dynamics_randomize_every=2)
    sync, stag = QuadrotorEnv(**kw), QuadrotorEnv(**kw)
    assert sync.ep_len == ep_len
    obs = torch.empty((n, 18), device=dev); rew = torch.empty(n, device=dev); done = torch.empty(n, dtype=torch.uint8, device=dev)
    finished = {}
    for name, env in (("sync", sync), ("stag", stag)):
        env.reset_dev(obs)
        if name == "stag":
            st = env.get_state()
            st[37] = np.arange(n) % (ep_len + 1)              # one finishing env in most waves per step
            env.set_state(st)
        cnt = np.zeros(n, dtype=np.int64)
        gen = torch.Generator(device=dev); gen.manual_seed(3)
        for t in range(5 * (ep_len + 1) + 3):
            env.step_dev(torch.rand((n, 4), device=dev, generator=gen) * 2 - 1, obs, rew, done)
            cnt += done.cpu().numpy().astype(np.int64)
            if name == "stag" and t == 17:                    # move every clock to the last step of its episode: all finish next step
                st = env.get_state()
                st[37] = ep_len
                env.set_state(st)
        env.check_finite()                                    # raises if an env consumed planes that had not been refilled
        finished[name] = cnt
        m = env.models
        again, _ = qp.derive_models(env.sampled_trees())
        for key in ("mass", "inertia", "thrust_max", "torque_max", "prop_pos", "arm", "damp_time_up", "linearity"):
            assert gu.rel_err(again[key], m[key]) <= 1e-12, (name, key)
    ms, mt = sync.models, stag.models
    draws = lambda c: (c + 1) // 2               # every = 2: due when (k + 1) % 2 == 0 after the k-th finished episode
    assert np.all(draws(finished["sync"]) == 3) and np.all(draws(finished["stag"]) == 3)
    for key in ("mass", "inertia", "thrust_max", "arm"):
        assert np.array_equal(ms[key], mt[key]), key          # coalesced and lane-parallel promotions: the same draw, bit for bit
    gen = torch.Generator(device=dev); gen.manual_seed(4)
    cnt = finished["stag"].copy()
    for t in range(ep_len + 1):                               # one more episode of the staggered handle: some envs reach draw 4
        stag.step_dev(torch.rand((n, 4), device=dev, generator=gen) * 2 - 1, obs, rew, done)
        cnt += done.cpu().numpy().astype(np.int64)
    stag.check_finite()
    mt = stag.models
    same = draws(cnt) == 3
    assert same.sum() > 100 and (~same).sum() > 100
    assert np.array_equal(ms["mass"][same], mt["mass"][same]) and np.all(ms["mass"][~same] != mt["mass"][~same])
    sync.close(); stag.close()
    # at the C ABI a period needs in-kernel resets (otherwise a finished env reports done on every step)
    from gym_art_amd import _lib
    h = G.Handle(64, 0.005, 2, 10, per_env=1, auto_reset=0)
    rz = _lib.GaqRandomizer()
    rz.sampler, rz.every = 2, 1
    assert h.lib.gaq_set_randomizer(h.h, C.byref(rz)) == -1          # GAQ_ERR_INVALID
    rz.every = 0
    assert h.lib.gaq_set_randomizer(h.h, C.byref(rz)) == 0
    h.close()


def test_two_handles_on_two_streams_are_independent():
    """Two handles driven from two torch streams at once, with host-pointer calls (gaq_get_state -> the handle's own synchronisation,
    not a device-wide one) on one while the other has work in flight: each computes what it computes alone."""
    import torch
    from gym_art_amd import QuadrotorEnv
    dev = torch.device("cuda", 0)
    n, T = 1 << 16, 40
    kw_a = dict(dynamics_params="DefaultQuad", num_envs=n, ep_time=0.1, seed=3, alias_obs=True)
    kw_b = dict(dynamics_params="Crazyflie", num_envs=n, ep_time=0.07, seed=4)
    gen = torch.Generator(device=dev); gen.manual_seed(9)
    acts = torch.rand((T, n, 4), device=dev, generator=gen) * 2 - 1

    def run(env, stream, peek=None):
        obs = torch.empty((n, 18), device=dev); rew = torch.empty(n, device=dev); done = torch.empty(n, dtype=torch.uint8, device=dev)
        with torch.cuda.stream(stream):
            env.reset_dev(obs)
            for t in range(T):
                env.step_dev(acts[t], obs, rew, done)
                if peek is not None and t % 7 == 3:
                    peek.get_state()                      # a host-pointer call on the OTHER handle, mid-flight
        return obs, rew, done

    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for s in (sa, sb):
        s.wait_stream(torch.cuda.current_stream())
    alone_a = run(QuadrotorEnv(**kw_a), sa); torch.cuda.synchronize()
    alone_b = run(QuadrotorEnv(**kw_b), sb); torch.cuda.synchronize()
    ea, eb = QuadrotorEnv(**kw_a), QuadrotorEnv(**kw_b)
    obs_a = torch.empty((n, 18), device=dev); rew_a = torch.empty(n, device=dev); done_a = torch.empty(n, dtype=torch.uint8, device=dev)
    obs_b = torch.empty((n, 18), device=dev); rew_b = torch.empty(n, device=dev); done_b = torch.empty(n, dtype=torch.uint8, device=dev)
    with torch.cuda.stream(sa):
        ea.reset_dev(obs_a)
    with torch.cuda.stream(sb):
        eb.reset_dev(obs_b)
    for t in range(T):                                    # interleaved launches on the two streams
        with torch.cuda.stream(sa):
            ea.step_dev(acts[t], obs_a, rew_a, done_a)
        with torch.cuda.stream(sb):
            eb.step_dev(acts[t], obs_b, rew_b, done_b)
        if t % 7 == 3:
            ea.get_state(); eb.get_state()
    torch.cuda.synchronize()
    assert torch.equal(alone_a[0], obs_a) and torch.equal(alone_a[1], rew_a) and torch.equal(alone_a[2], done_a)
    assert torch.equal(alone_b[0], obs_b) and torch.equal(alone_b[1], rew_b) and torch.equal(alone_b[2], done_b)
    ea.close(); eb.close()


@pytest.mark.parametrize("case", ["hummingbird_alias", "crazyflie_device_randomised", "crazyflie_host_randomised", "sense_noise_bias_walk"])
def test_checkpoint_resume_is_bit_exact(case):
    """state_dict() / load_state_dict(): an env rebuilt from its constructor arguments and the checkpoint continues bit for bit --
    thrust noise, in-kernel resets, per-episode re-randomisation (device: parameters rebuilt from the resample counts; host: the
    models travel in the checkpoint), the gyro-bias walk -- through pickle, like a training job would store it."""
    import pickle
    import torch
    from gym_art_amd import QuadrotorEnv
    sampler = {"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"}
    n = 4096
    kw = {"hummingbird_alias": dict(dynamics_params="DefaultQuad", alias_obs=True, ep_time=0.12),
          "crazyflie_device_randomised": dict(dynamics_params="Crazyflie", dyn_sampler_1=sampler, dynamics_randomize_every=2, ep_time=0.08),
          "crazyflie_host_randomised": dict(dynamics_params="Crazyflie", dyn_sampler_1=sampler, randomize_on_device=False, ep_time=0.08),
          "sense_noise_bias_walk": dict(dynamics_params="DefaultQuad", ep_time=0.1,
                                        sense_noise={"gyro_norm_std": 0.01, "quat_norm_std": 0.01})}[case]
    kw.update(num_envs=n, seed=7)
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    acts = torch.rand((70, n, 4), device=dev, generator=gen) * 2 - 1
    mk = lambda D: (torch.empty((n, D), device=dev), torch.empty(n, device=dev), torch.empty(n, dtype=torch.uint8, device=dev))
    a = QuadrotorEnv(**kw)
    oa, ra, da = mk(a.obs_dim)
    a.reset_dev(oa)
    st = a.get_state(); st[37] = np.arange(n) % (a.ep_len + 1); a.set_state(st)      # staggered phases
    if case == "hummingbird_alias":
        a.reset_dev(oa, mask=torch.zeros(n, dtype=torch.uint8, device=dev))          # a second reset call: the reset-call counter matters
    for t in range(40):
        a.step_dev(acts[t], oa, ra, da)
    blob = pickle.dumps(a.state_dict())
    b = QuadrotorEnv(**kw).load_state_dict(pickle.loads(blob))
    ob, rb, db = mk(b.obs_dim)
    if b.state_layout == 1:                     # heads aliased to the caller's tensor: hand it one (the state planes hold the values)
        b.reset_dev(ob, mask=torch.zeros(n, dtype=torch.uint8, device=dev))
        b.load_state_dict(pickle.loads(blob))
    for t in range(40, 70):
        a.step_dev(acts[t], oa, ra, da)
        b.step_dev(acts[t], ob, rb, db)
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db), t
    assert np.array_equal(a.get_state(), b.get_state())
    if a._per_env:
        assert np.array_equal(a.models["mass"], b.models["mass"]) and np.array_equal(a.models["inertia"], b.models["inertia"])
    a.check_finite(); b.check_finite()
    a.close(); b.close()


def test_bench_two_rank_rehearsal_on_one_gpu():
    """`python bench.py --gpus 2` end to end on the 1-GPU test box: GAQ_BENCH_REHEARSAL=1 puts both ranks on GPU 0 over gloo (RCCL
    refuses two ranks per device), so that everything N > 1 in bench.py except RCCL itself runs -- the parent starting its ranks,
    config 4's split of the TOTAL batch, the packed single gather per step, barriers, the max over ranks, rank 0's one JSON line."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GAQ_BENCH_REHEARSAL="1")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--envs", "16384", "--steps", "30", "--warmup", "10",
                          "--repeats", "2", "--prime-ms", "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["rccl_ranks"] == 0 and d["value"] > 0
    c = d["config"]
    assert c["total_envs"] == 16384 and c["envs_per_gpu"] == 8192 and c["gather"] == "packed" and "REHEARSAL" in c["workload"]
    assert "cpu_baseline" not in d or d["cpu_baseline"] is None or d["n_gpus"] == 1
    # the way the driver launches N > 1: under torch.distributed.run (the ranks exist already; bench.py must not start its own)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "2", "--envs", "16384", "--steps", "30",
                          "--warmup", "10", "--repeats", "2", "--prime-ms", "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["total_envs"] == 16384 and d["config"]["envs_per_gpu"] == 8192


@pytest.mark.parametrize("model", ["hummingbird", "crazyflie"])
def test_device_rng_streams_against_the_cpu_build_of_the_same_header(model):
    """Thrust noise (Philox OU) and in-kernel resets on the device against oracle/cpu_native.cpp -- the same arithmetic header compiled
    by g++ with its own driver loop: same seed, same actions, three episode ends inside the run.  What statistics cannot see -- a wrong
    stream id, env index or step counter in a Philox key on the device -- would show here at once; what remains is the difference
    between the device's fast log / sin / cos and libm's in the Box-Muller step (1e-6 of a normal)."""
    from oracle import cpu_native as cn
    n, T, ep_len = 2048, 40, 12
    d = gu.load("g2_hummingbird_raw" if model == "hummingbird" else "g3_crazyflie")
    const = dict(gu.sub(d, "const_"))
    const["thrust_noise_sigma"] = np.float64(0.01)
    rng = np.random.RandomState(5)
    st = np.zeros((42, n))
    st[0:3] = (rng.uniform(-2, 2, (n, 3)) + [0, 0, 2]).astype(np.float32).T
    st[2] = np.maximum(st[2], 0.3)
    st[3:6] = rng.uniform(-1, 1, (3, n)).astype(np.float32)
    q, r = np.linalg.qr(rng.normal(size=(n, 3, 3)))
    q = q * np.sign(np.einsum("nii->ni", r))[:, None, :]
    q[np.linalg.det(q) < 0, :, 0] *= -1
    st[6:15] = q.astype(np.float32).reshape(n, 9).T
    st[15:18] = rng.uniform(-3, 3, (3, n)).astype(np.float32)
    st[34:37] = np.array([[0.], [0.], [2.]])
    st[37] = np.arange(n) % (ep_len + 1)                       # staggered episode clocks: resets on every step
    for alias in (0, 1):
        h = G.Handle(n, 0.005, 2, ep_len, const=const, noise=1, auto_reset=1, seed=99, alias=alias)
        h.set_state(st)
        b = cn.Batch(n, const, dt=0.005, sim_steps=2, ep_len=ep_len, noise=1, auto_reset=1, seed=99)
        b.set_state(st)
        worst = 0.0
        for t in range(T):
            a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
            od, rd, dd = h.step(a)
            oc, rc, dc = b.step(a, threads=4)
            assert np.array_equal(dd, dc), t
            worst = max(worst, gu.rel_err(od, oc))
            assert np.max(np.abs(rd - rc)) <= 2e-6, t
        assert worst <= 5e-6, worst
        assert np.max(np.abs(h.get_state()[26:30] - 0) ) > 0      # the OU state is alive
        h.close(); b.close()


def test_device_sampler_streams_against_the_host_build():
    """The device's parameter draws (perturb_tree / random_quad_tree in the rerandomize kernel) against the g++ build of the same header,
    leaf by leaf, for the first and for later draws of envs with a global-index offset: the Philox keys (seed, GLOBAL env index, resample
    count) are the device's own; the only difference left is fast log / sin / cos vs libm in the normals (1e-6 of a standard deviation)."""
    import ctypes as C
    import torch
    from gym_art_amd import QuadrotorEnv, _lib, quad_models, quad_params as qp
    from tests import hh
    L = hh.lib()
    n, off, seed = 512, 1 << 20, 31
    dev = torch.device("cuda", 0)
    obs = torch.empty((n, 18), device=dev); rew = torch.empty(n, device=dev); done = torch.empty(n, dtype=torch.uint8, device=dev)
    sampler = {"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"}
    for kw, rq in ((dict(dynamics_params="Crazyflie", dyn_sampler_1=sampler), False), (dict(dynamics_params="RandomQuad"), True)):
        env = QuadrotorEnv(num_envs=n, env_id_offset=off, seed=seed, ep_time=0.03, dynamics_randomize_every=1, **kw)
        env.reset_dev(obs)
        for draws in (0, 3):                                   # after the initial draw, and after three finished episodes
            if draws:
                for t in range(draws * (env.ep_len + 1)):
                    env.step_dev(torch.zeros((n, 4), device=dev), obs, rew, done)
            rows = np.empty((n, qp.TREE_DOUBLES))
            _lib.check(env._lib.gaq_get_param_trees(env._handle, _lib.ptr(rows), 0, n))
            ref = np.zeros((n, 40))
            if rq:
                for i in range(n):
                    L.hh_random_quad_tree(C.c_uint64(seed), C.c_uint64(off + i), C.c_uint64(draws), ref[i].ctypes.data_as(C.POINTER(C.c_double)))
            else:
                base = np.ascontiguousarray(qp.flatten_tree(qp.broadcast_tree(quad_models.model_params("crazyflie"), 1))[0])
                ratio = np.ascontiguousarray(qp.ratio_rows(qp.broadcast_tree(quad_models.model_params("crazyflie"), 1), 0.2, None)[0])
                for i in range(n):
                    L.hh_perturb_tree(base.ctypes.data_as(C.POINTER(C.c_double)), ratio.ctypes.data_as(C.POINTER(C.c_double)), 0,
                                      C.c_uint64(seed), C.c_uint64(off + i), C.c_uint64(draws), ref[i].ctypes.data_as(C.POINTER(C.c_double)))
            scale = np.maximum(np.abs(ref).max(axis=0, keepdims=True), 1e-9)     # per leaf: a draw near zero is compared on the leaf's scale
            assert np.max(np.abs(rows - ref) / scale) <= 2e-6, (rq, draws, float(np.max(np.abs(rows - ref) / scale)))
        env.close()


def test_device_reset_draws_against_the_host_build():
    """gaq_reset with init_random_state + resample_goal on the device against reset_env of the same header compiled by g++, env by env:
    the Philox key of a reset call is (seed, GLOBAL env index, step index + (reset calls << 44)).  Second reset call after some steps
    included; fast sin / cos vs libm leaves 1e-6."""
    import ctypes as C
    from tests import hh
    L = hh.lib()
    L.hh_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
    n, off, seed = 300, 4096, 17
    const = dict(gu.sub(gu.load("g2_hummingbird_raw"), "const_"))
    h = G.Handle(n, 0.005, 2, 500, const=const, seed=seed, env_id_offset=off, init_random_state=1, resample_goal=1)
    m = hh.make_model(const)
    c = hh.make_cfg(0.005, 2, 500, m)
    c.seed, c.init_random_state, c.resample_goal, c.per_env_goal = seed, 1, 1, 1
    steps_done = 0
    for call in (1, 2):
        h.reset()
        st = h.get_state()
        for i in range(n):
            ref = np.zeros(39)
            ref[6], ref[10], ref[14] = 1.0, 1.0, 1.0
            L.hh_reset(C.byref(c), ref.ctypes.data, C.c_uint64(off + i), C.c_uint64(steps_done + (call << 44)))
            assert np.max(np.abs(st[0:6, i] - ref[0:6])) <= 1e-6 and np.max(np.abs(st[15:18, i] - ref[15:18])) <= 1e-6, (call, i)
            assert np.max(np.abs(st[6:15, i] - ref[6:15])) <= 3e-5, (call, i)          # the random attitude goes through fast sin / cos
            assert np.max(np.abs(st[34:37, i] - ref[34:37])) <= 1e-6, (call, i)
        assert st[0:3].std() > 0.5 and np.ptp(st[36]) > 1.0            # random positions, resampled goal heights
        for t in range(3):
            h.step(np.zeros((n, 4), np.float32))
            steps_done += 1
    h.close()


@pytest.mark.parametrize("walk", [False, True])
def test_device_sensor_noise_streams_against_the_host_build(walk):
    """Sensor noise drawn ON the device (Philox: three add_noise calls per step, keyed by global env index and step) against the g++ build
    of the same header running each env on its own -- white-noise gyro in the split-state kernel with a packed observation (the
    "default" model), and the gyro-bias walk with quaternion + uniform noise in the generic kernel.  The arithmetic is pinned to the
    reference by recorded draws elsewhere (G10); this pins the device's own draws: keys, call order, the composite bias step."""
    from tests import hh
    n, off, seed, T, dt = 192, 1 << 18, 23, 12, 0.005
    const = dict(gu.sub(gu.load("g2_hummingbird_raw"), "const_"))
    sense = {"gyro_norm_std": 0.01, "quat_norm_std": 0.01, "pos_unif_range": 0.01, "vel_unif_range": 0.02, "quat_unif_range": 0.005} if walk else {}
    h = G.Handle(n, dt, 2, 500, const=const, seed=seed, env_id_offset=off, sense=sense, obs_flags=2)       # the _h observation: height appended
    rng = np.random.RandomState(8)
    st = np.zeros((42, n))
    st[0:3] = (rng.uniform(-2, 2, (n, 3)) + [0, 0, 2]).astype(np.float32).T
    st[2] = np.maximum(st[2], 0.3)
    st[3:6] = rng.uniform(-1, 1, (3, n)).astype(np.float32)
    q, r = np.linalg.qr(rng.normal(size=(n, 3, 3)))
    q = q * np.sign(np.einsum("nii->ni", r))[:, None, :]
    q[np.linalg.det(q) < 0, :, 0] *= -1
    st[6:15] = q.astype(np.float32).reshape(n, 9).T
    st[15:18] = rng.uniform(-3, 3, (3, n)).astype(np.float32)
    st[34:37] = np.array([[0.], [0.], [2.]])
    if walk:
        st[39:42] = rng.uniform(-0.01, 0.01, (3, n)).astype(np.float32)
    h.set_state(st)
    acts = rng.uniform(-1, 1, (T, n, 4)).astype(np.float32)
    dev = np.stack([h.step(acts[t])[0] for t in range(T)])            # [T, n, D]
    bias_dev = h.get_state()[39:42]
    m = hh.make_model(const)
    c = hh.make_cfg(dt, 2, 500, m, obs_repr="xyz_vxyz_R_omega_h")
    c.seed = seed
    prm = dict(pos_norm_std=0.005, pos_unif_range=0., vel_norm_std=0.01, vel_unif_range=0., quat_norm_std=0., quat_unif_range=0.,
               gyro_noise_density=0.000175, acc_static_noise_std=0.002, acc_dynamic_noise_ratio=0.005, gyro_norm_std=0.,
               gyro_random_walk=0.0105, gyro_bias_correlation_time=1000.)
    prm.update(sense)
    c.sense.enabled = 1
    for k, v in prm.items():
        setattr(c.sense, k, float(v))
    if walk:
        tau = prm["gyro_bias_correlation_time"]
        sg = prm["gyro_noise_density"] / np.sqrt(dt)
        sb = np.sqrt(-(sg ** 2) * (tau / 2) * (np.exp(-2 * dt / tau) - 1))
        pi = np.exp(-dt / tau)
        c.gyro_bias, c.gyro_pi, c.gyro_sigma = 1, pi, sb
        c.gyro_pi_step, c.gyro_sigma_step = pi ** 3, sb * np.sqrt(1 + pi ** 2 + pi ** 4)
    worst, worst_bias = 0.0, 0.0
    for i in range(n):
        c.env_offset = off + i
        out = hh.rollout(c, m, st[:39, i], acts[:, i], variant=8, want_traj=False, gyro_bias=st[39:42, i])
        worst = max(worst, gu.rel_err(dev[:, i], out["obs"]))
        worst_bias = max(worst_bias, float(np.max(np.abs(out["gyro_bias"] - bias_dev[:, i]))))
    assert worst <= 3e-6, worst
    assert worst_bias <= 1e-7, worst_bias
    h.close()


def test_device_excite_goals_against_the_host_build():
    """excite=True (a new goal from U(-0.5, 0.5)^2 x U(1.5, 2.5) whenever tick % 5 == 0, quadrotor.py:957-963): the goals the device draws
    and the observations that follow, env by env against the g++ build of the same header (Philox key: seed, global env index, step)."""
    from gym_art_amd import QuadrotorEnv
    from tests import hh
    n, off, seed, T = 96, 777, 41, 14
    env = QuadrotorEnv(num_envs=n, env_id_offset=off, seed=seed, ep_time=5, thrust_noise="off", auto_reset=False, excite=True)
    st0 = env.get_state()
    rng = np.random.RandomState(2)
    acts = rng.uniform(-1, 1, (T, n, 4)).astype(np.float32)
    dev = np.stack([env.step(acts[t])[0] for t in range(T)])
    goals_dev = env.get_state()[34:37]
    const = dict(gu.sub(gu.load("g2_hummingbird_raw"), "const_"))
    const["thrust_noise_sigma"] = np.float64(0.0)
    m = hh.make_model(const)
    c = hh.make_cfg(0.005, 2, 500, m, action_f32=1)
    c.seed, c.excite, c.per_env_goal = seed, 1, 1
    worst = 0.0
    for i in range(n):
        c.env_offset = off + i
        out = hh.rollout(c, m, st0[:39, i], acts[:, i], variant=8, want_traj=False)
        worst = max(worst, gu.rel_err(dev[:, i], out["obs"]))
        assert np.max(np.abs(out["state"][34:37] - goals_dev[:, i])) <= 1e-6, i
    assert worst <= 1e-6, worst
    assert np.ptp(goals_dev[0]) > 0.5                       # the goals did move
    env.close()


def test_class_level_random_configurations_against_the_oracle():
    """Random CONSTRUCTOR arguments through gym_art_amd.QuadrotorEnv -- model, controller flags, observation variant, reward variant and
    weights, sim_freq x sim_steps, ep_time, layout, init_random_state, batch size incl. 1 -- a few steps each from the env's own reset
    state against the oracle configured from the same arguments: the Python layer's mapping of keywords to gaq_config (observation
    flags, reward coefficients, control mode, rates, action dtype) has to be the reference's.  GAQ_FUZZ_CONFIGS / GAQ_FUZZ_SEED for more."""
    from gym_art_amd import QuadrotorEnv
    from gym_art_amd.quadrotor import OBS_FLAGS
    from oracle import quad_oracle as qo
    rng = np.random.RandomState(int(os.environ.get("GAQ_FUZZ_SEED", "4242")))
    reprs = [k for k in OBS_FLAGS if "t2w" not in k and "quat" not in k]
    for c in range(int(os.environ.get("GAQ_FUZZ_CONFIGS", "30"))):
        model = ["DefaultQuad", "Crazyflie", "MediumQuad", "RandomQuad"][rng.randint(4)]
        raw = bool(rng.randint(4))                       # one in four flies the Mellinger controller
        zero_middle = bool(rng.randint(2))
        freq, steps = [(200., 2), (100., 4), (400., 1), (250., 3)][rng.randint(4)]
        obs_repr = reprs[rng.randint(len(reprs))]
        module = ["quadrotor", "multi"][rng.randint(2)]
        keys = ("pos", "effort", "action_change", "crash", "orient", "yaw", "rot", "attitude", "spin", "vel")
        rew = {k: float(rng.uniform(0, 1)) for k in keys if rng.rand() < 0.4}
        n = [1, 3, 64, 130][rng.randint(4)]
        layout = [None, True, False][rng.randint(3)]
        f32 = bool(rng.randint(2))
        kw = dict(dynamics_params=model, dynamics_change={"noise": {"thrust_noise_ratio": 0.}}, raw_control=raw,
                  raw_control_zero_middle=zero_middle, sim_freq=freq, sim_steps=steps, ep_time=float(rng.choice([1.0, 5.0, 7.0])),
                  obs_repr=obs_repr, rew_coeff=rew, init_random_state=bool(rng.randint(2)), num_envs=n, seed=int(rng.randint(1 << 30)),
                  alias_obs=layout, auto_reset=False)
        if module == "multi":
            kw["reward"] = "multi"
        if model != "RandomQuad" and rng.rand() < 0.4:   # per-env parameters: sampled on the host or on the device, as the class decides or is told
            kw["dyn_sampler_1"] = {"class": "RelativeSampler", "noise_ratio": float(rng.choice([0.1, 0.2])), "sampler": str(rng.choice(["normal", "uniform"]))}
        if rng.rand() < 0.5:
            kw["randomize_on_device"] = bool(rng.randint(2))
        try:
            env = QuadrotorEnv(**kw)
        except ValueError as ex:                         # asked for the device pipeline where it cannot be used: refused loudly, never silently
            assert kw.get("randomize_on_device") is True and "randomize_on_device=True is not possible" in str(ex), (kw, ex)
            kw.pop("randomize_on_device")
            env = QuadrotorEnv(**kw)
        env.reset()
        st = env.get_state()
        m = env.models
        p = qo.Params(n, mass=m["mass"], inertia=m["inertia"], thrust_max=m["thrust_max"], torque_max=m["torque_max"],
                      prop_pos=m["prop_pos"].reshape(n, 4, 3), damp_time_up=m["damp_time_up"], damp_time_down=m["damp_time_down"],
                      linearity=m["linearity"], arm=m["arm"], ou_sigma=0 * m["ou_sigma"], vel_damp=m["vel_damp"],
                      damp_omega_quadratic=m["damp_omega_quadratic"], C_drag=m["c_drag"], C_roll=m["c_roll"])
        control = "mellinger" if not raw else ("raw_zero_middle" if zero_middle else "raw")
        if control == "mellinger":
            p.jacobian_inverse()
        cfg = qo.Config(sim_freq=freq, sim_steps=steps, ep_time=kw["ep_time"], control=control, obs_repr=obs_repr, rew_coeff=rew,
                        reward_variant=module)
        cfg.action_f32 = f32
        assert cfg.ep_len == env.ep_len, (c, cfg.ep_len, env.ep_len)
        s = qo.State(n)
        s.goal[:] = st[34:37].T
        s.set_state(st[0:3].T, st[3:6].T, st[6:15].T.reshape(n, 3, 3), st[15:18].T)
        for t in range(6):
            a = rng.uniform(-1.1, 1.1, (n, 4)).astype(np.float32)
            act = a if f32 else a.astype(np.float64)
            o, r, d, info = env.step(act[0] if n == 1 else act)
            o_ref, r_ref, d_ref = qo.env_step(s, p, cfg, a.astype(np.float64))
            o, r, d = np.atleast_2d(o), np.atleast_1d(r), np.atleast_1d(d)
            assert o.shape == o_ref.shape, (c, obs_repr, o.shape, o_ref.shape)
            assert gu.rel_err(o, o_ref) <= 1e-6, (c, t, kw)
            assert np.max(np.abs(r - r_ref)) <= 2e-6, (c, t, kw)
            assert np.array_equal(d.astype(bool), d_ref), (c, t)
            if n == 1 and t == 2:                       # the drop-in loop's info dict: the reward terms sum to the reward
                assert abs(sum(v for k, v in info["rewards"].items() if k.startswith("rew_") and k != "rew_main") * env.dt
                           - float(r[0])) <= 1e-6 * max(1.0, abs(float(r[0]))), (c, info["rewards"], r)      # (:601: reward = -dt * sum of costs)
        env.close()


def test_promoting_kernel_instantiation_physics_against_the_oracle():
    """The F_RZ instantiation (per-episode re-randomisation handled inside the step launch) is the per-env kernel plus an epilogue; here
    its physics is held against the oracle directly: randomised CrazyFlies (motor lag), every episode re-drawn, thrust noise off; after
    a few episodes the device state and the read-back parameters seed the oracle, and the envs that do not finish in the next steps
    have to agree."""
    import torch
    from gym_art_amd import QuadrotorEnv
    from oracle import quad_oracle as qo
    sampler = {"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"}
    n, ep_len = 4096, 11
    env = QuadrotorEnv(dynamics_params="Crazyflie", num_envs=n, ep_time=ep_len * 0.01, seed=13, dyn_sampler_1=sampler, dynamics_randomize_every=1,
                       thrust_noise="off", alias_obs=True)
    assert env.ep_len == ep_len and env._dev_rand
    dev = torch.device("cuda", 0)
    obs = torch.empty((n, 18), device=dev); rew = torch.empty(n, device=dev); done = torch.empty(n, dtype=torch.uint8, device=dev)
    env.reset_dev(obs)
    st = env.get_state(); st[37] = np.arange(n) % (ep_len + 1); env.set_state(st)
    gen = torch.Generator(device=dev); gen.manual_seed(6)
    for t in range(3 * (ep_len + 1) + 2):
        env.step_dev(torch.rand((n, 4), device=dev, generator=gen) * 2 - 1, obs, rew, done)
    st, m = env.get_state(), env.models
    keep = st[37] + 4 <= ep_len
    assert keep.sum() > 1000
    p = qo.Params(n, mass=m["mass"], inertia=m["inertia"], thrust_max=m["thrust_max"], torque_max=m["torque_max"],
                  prop_pos=m["prop_pos"].reshape(n, 4, 3), damp_time_up=m["damp_time_up"], damp_time_down=m["damp_time_down"],
                  linearity=m["linearity"], arm=m["arm"], ou_sigma=0 * m["ou_sigma"], vel_damp=m["vel_damp"],
                  damp_omega_quadratic=m["damp_omega_quadratic"], C_drag=m["c_drag"], C_roll=m["c_roll"])
    cfg = qo.Config(sim_freq=200., sim_steps=2, ep_time=ep_len * 0.01)
    cfg.action_f32 = True
    s = qo.State(n)
    s.goal[:] = st[34:37].T
    s.set_state(st[0:3].T, st[3:6].T, st[6:15].T.reshape(n, 3, 3), st[15:18].T)
    s.thrust_rot_damp[:] = st[18:22].T
    s.thrust_cmds_damp[:] = st[22:26].T
    s.since_last_svd[:] = st[38] * 0.005
    s.tick[:] = st[37].astype(np.int64)
    for t in range(3):
        a = (torch.rand((n, 4), device=dev, generator=gen) * 2 - 1)
        env.step_dev(a, obs, rew, done)
        o_ref, r_ref, _ = qo.env_step(s, p, cfg, a.cpu().numpy().astype(np.float64))
        o, r = obs.cpu().numpy(), rew.cpu().numpy()
        assert gu.rel_err(o[keep], o_ref[keep]) <= 1e-6, t
        assert np.max(np.abs(r[keep] - r_ref[keep])) <= 1e-6, t
    env.check_finite()
    env.close()


def test_fused_rollouts_equal_single_steps_over_random_configurations():
    """gaq_step_many_dev (state in registers across T steps) against T gaq_step_dev calls over random configurations of everything the
    fused kernel honours -- model (uniform Hummingbird / CrazyFlie, per-env CrazyFlie or RandomQuad), reward variant and weights incl. the
    rot / attitude terms, thrust noise (same Philox keys in both), in-kernel resets with init_random_state and short episodes, layout A1 /
    A2, ragged batch sizes, chunk lengths: two kernels, one trajectory (to the FMA-contraction differences between two instantiations)."""
    import torch
    from gym_art_amd import QuadrotorEnv
    dev = torch.device("cuda", 0)
    rng = np.random.RandomState(int(os.environ.get("GAQ_FUZZ_SEED", "99")))
    sampler = {"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"}
    for c in range(int(os.environ.get("GAQ_FUZZ_CONFIGS", "24"))):
        kind = ["DefaultQuad", "Crazyflie", "Crazyflie_rand", "RandomQuad"][rng.randint(4)]
        n = int(rng.choice([64, 130, 1000, 4096]))
        keys = ("pos", "effort", "crash", "orient", "yaw", "rot", "attitude", "spin", "vel")
        kw = dict(dynamics_params=kind.split("_")[0], num_envs=n, seed=int(rng.randint(1 << 30)), ep_time=float(rng.choice([0.06, 0.11, 5.0])),
                  rew_coeff={k: float(rng.uniform(0, 1)) for k in keys if rng.rand() < 0.4}, init_random_state=bool(rng.randint(2)),
                  thrust_noise=str(rng.choice(["philox", "off"])), alias_obs=[True, None][rng.randint(2)])
        if kind == "Crazyflie_rand":
            kw["dyn_sampler_1"] = sampler
        if rng.randint(2):
            kw["reward"] = "multi"
        a_env, b_env = QuadrotorEnv(**kw), QuadrotorEnv(**kw)
        T = int(rng.choice([2, 7, 16, 33]))
        chunks = int(rng.choice([1, 3]))
        oa = torch.empty((n, 18), device=dev); ra = torch.empty(n, device=dev); da = torch.empty(n, dtype=torch.uint8, device=dev)
        ob0 = torch.empty((n, 18), device=dev)
        a_env.reset_dev(oa); b_env.reset_dev(ob0)
        assert torch.equal(oa, ob0)
        gen = torch.Generator(device=dev); gen.manual_seed(c)
        prev = ob0
        for k in range(chunks):
            acts = torch.rand((T, n, 4), device=dev, generator=gen) * 2.4 - 1.2
            OB = torch.empty((T, n, 18), device=dev); RB = torch.empty((T, n), device=dev); DB = torch.empty((T, n), dtype=torch.uint8, device=dev)
            b_env.step_many_dev(acts, OB, RB, DB)
            for t in range(T):
                a_env.step_dev(acts[t], oa, ra, da)
                assert torch.equal(da, DB[t]), (c, k, t, kw)
                e = float((oa - OB[t]).abs().div(OB[t].abs().clamp(min=1.0)).max())
                assert e <= 2e-6, (c, k, t, e, kw)
                assert float((ra - RB[t]).abs().max()) <= 2e-6, (c, k, t, kw)
        a_env.close(); b_env.close()


def test_c_abi_survives_random_configurations():
    """Random, mostly invalid gaq_config records (sizes, versions, negative and huge counts, NaN / inf / zero rates and weights, flag bits
    that do not exist, swarm sizes that are not powers of two, non-finite models) at gaq_create, and a few calls on whatever comes back: an
    error is an error code with a message and no handle, a handle works and is destroyed -- the library never crashes, hangs or faults
    the GPU on bad input.  In a child process, so that a crash would be this test's failure only."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "aux", "abi_fuzz_worker.py"), os.environ.get("GAQ_FUZZ_SEED", "5"),
                          os.environ.get("GAQ_FUZZ_CONFIGS", "250")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0, (out.returncode, out.stderr[-3000:])
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["ok"] >= 20 and d["refused"] >= 20 and d["stepped"] >= 10, d


def test_handles_give_all_their_memory_back():
    """Create / use / destroy in a loop -- the heaviest handle kinds: per-env parameters with the staged rows of per-episode
    re-randomisation, episode tracking and the done list, and the host-pointer path's staging with aux rows and pinned mirrors -- and
    the free HBM the driver reports does not drift: after a warm-up it moves within the driver's own pooling noise (a few hundred MB
    either way, measured over 1200 cycles) while 400 more handles of 0.1-0.4 GB each come and go; any per-env array left behind
    (>= 1 MB per handle here) would add up to >= 400 MB on top of that."""
    import torch
    from gym_art_amd import QuadrotorEnv
    sampler = {"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"}
    dev = torch.device("cuda", 0)
    n = 1 << 18

    def cycle(k):
        if k % 2:
            env = QuadrotorEnv(dynamics_params="Crazyflie", num_envs=n, ep_time=0.05, seed=k, dyn_sampler_1=sampler, dynamics_randomize_every=1,
                               compact_done=True)
            env.track_episodes(True)
        else:
            env = QuadrotorEnv(num_envs=n, ep_time=0.05, seed=k)
        obs = torch.empty((n, 18), device=dev); rew = torch.empty(n, device=dev); done = torch.empty(n, dtype=torch.uint8, device=dev)
        a = torch.zeros((n, 4), device=dev)
        env.reset_dev(obs)
        for t in range(4):
            env.step_dev(a, obs, rew, done)
        if k % 8 == 0:
            env.get_state(); env.models
        env.close()
        if k % 16 == 0:
            small = QuadrotorEnv(num_envs=64, seed=k, info=True)       # host-pointer path: staging, pinned mirrors, aux rows
            small.reset(); small.step(np.zeros((64, 4), np.float32)); small.close()
        del obs, rew, done, a
        torch.cuda.empty_cache()

    def free():
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info(0)[0]

    for k in range(100):
        cycle(k)
    lows = [free()]
    for block in range(4):
        for k in range(100):
            cycle(1000 + 100 * block + k)
        lows.append(free())
    assert lows[0] - min(lows[1:]) < (512 << 20), lows


@pytest.mark.parametrize("layout", [None, True, False])
def test_random_constructor_arguments_against_the_reference(layout):
    """Fixture G17 -- 24 random combinations of model, controller, observation variant, reward variant and weights, rates and action
    dtype run through the UNMODIFIED reference -- through gym_art_amd.QuadrotorEnv with the very same keyword arguments, in all three
    state layouts: what the class-level fuzz checks against the oracle, here against the reference's own numbers."""
    from gym_art_amd import QuadrotorEnv
    from tests import hh
    d = gu.load("g17_random_constructor_arguments")
    worst = 0.0
    for blk in gu.env_blocks(d):
        kw = gu.kwargs_of(blk)
        if str(blk["module"]) != "quadrotor":
            kw["reward"] = "multi"
        env = QuadrotorEnv(dynamics_change={"noise": {"thrust_noise_ratio": 0.}}, seed=0, alias_obs=layout, **kw)
        assert env.ep_len == int(blk["ep_len"]) and env.obs_dim == blk["obs"].shape[1]
        st = hh.pack_state(blk["init_pos"], blk["init_vel"], blk["init_rot"], blk["init_omega"], blk["goal"])
        env.set_state(np.concatenate([st, np.zeros(3)])[:, None])
        f32 = bool(blk["as_f32"])
        for t in range(blk["obs"].shape[0]):
            a = blk["actions"][t].astype(np.float32 if f32 else np.float64)
            o, r, dn, info = env.step(a)
            worst = max(worst, gu.rel_err(o, blk["obs"][t]))
            assert abs(r - blk["reward"][t]) <= 3e-7, (kw, t)
            assert dn == bool(blk["done"][t]) and bool(env.crashed) == bool(blk["crashed"][t])
        env.close()
    assert worst <= 1e-6, worst


def test_dynamics_change_through_the_class_against_the_reference():
    """Fixture G18 through gym_art_amd.QuadrotorEnv(dynamics_params=..., dynamics_change=...): the constants the handle flies with (read
    back through env.dynamics) are the reference's update_model constants, limits of resample_dynamics included."""
    from gym_art_amd import QuadrotorEnv
    d = gu.load("g18_dynamics_change")
    for blk in gu.env_blocks(d):
        change = json.loads(str(blk["change_json"]))
        env = QuadrotorEnv(dynamics_params=str(blk["model"]), dynamics_change=change, seed=0)
        c = gu.sub(blk, "const_")
        m = env.models
        for mine, ref in (("mass", "mass"), ("inertia", "inertia"), ("thrust_max", "thrust_max"), ("torque_max", "torque_max"), ("arm", "arm"),
                          ("damp_time_up", "damp_time_up"), ("linearity", "motor_linearity"), ("c_drag", "C_rot_drag"), ("vel_damp", "vel_damp")):
            assert gu.rel_err(m[mine][0], c[ref]) <= 1e-12, (str(blk["model"]), mine)
        assert gu.rel_err(env.dynamics.thrust_max, c["thrust_max"]) <= 1e-12 and gu.rel_err(env.dynamics.mass, c["mass"]) <= 1e-12
        o = env.reset()
        o, r, dn, info = env.step(np.zeros(4, np.float32))
        assert np.isfinite(o).all() and np.isfinite(r)
        env.close()


def test_resampled_goals_on_the_device():
    """Fixture G19 through the class with resample_goal=True: the goal the reference drew is put into the state (the device draws its own),
    then the trajectory -- observation relative to that goal, reward, Mellinger flying to it -- has to be the reference's."""
    from gym_art_amd import QuadrotorEnv
    from tests import hh
    d = gu.load("g19_resampled_goals")
    for blk in gu.env_blocks(d):
        kw = gu.kwargs_of(blk)
        env = QuadrotorEnv(dynamics_change={"noise": {"thrust_noise_ratio": 0.}}, seed=0, resample_goal=True, **kw)
        st = hh.pack_state(blk["init_pos"], blk["init_vel"], blk["init_rot"], blk["init_omega"], blk["goal"])
        env.set_state(np.concatenate([st, np.zeros(3)])[:, None])
        worst = 0.0
        for t in range(blk["obs"].shape[0]):
            o, r, dn, info = env.step(blk["actions"][t])
            worst = max(worst, gu.rel_err(o, blk["obs"][t]))
            assert abs(r - blk["reward"][t]) <= 3e-7 and dn == bool(blk["done"][t]), (kw, t)
        assert worst <= 1e-6, (kw, worst)
        assert np.allclose(info["obs_comp"]["xyz"][0], blk["pos"][-1], atol=1e-6)
        env.close()


def test_gravity_argument_on_the_device():
    """Fixture G20: `gravity` = 5 / 12 through the class -- only the accelerometer words of the observation see it."""
    from gym_art_amd import QuadrotorEnv
    from tests import hh
    d = gu.load("g20_gravity_argument")
    for blk in gu.env_blocks(d):
        kw = gu.kwargs_of(blk)
        env = QuadrotorEnv(dynamics_change={"noise": {"thrust_noise_ratio": 0.}}, seed=0, **kw)
        o0 = env.reset()
        assert gu.rel_err(o0[18:21], blk["reset_obs_acc"]) <= 1e-6
        st = hh.pack_state(blk["init_pos"], blk["init_vel"], blk["init_rot"], blk["init_omega"], blk["goal"])
        env.set_state(np.concatenate([st, np.zeros(3)])[:, None])
        worst = 0.0
        for t in range(blk["obs"].shape[0]):
            o, r, dn, info = env.step(blk["actions"][t])
            worst = max(worst, gu.rel_err(o, blk["obs"][t]))
            assert abs(r - blk["reward"][t]) <= 3e-7, (kw, t)
        assert worst <= 1e-6, (kw, worst)
        env.close()


def test_observation_and_action_spaces_equal_the_reference():
    """Fixture G7 keeps the reference's observation_space / action_space bounds for every working obs_repr, both RawControl conventions, other
    rates and both modules; G15 for the patched variants: the spaces of gym_art_amd.QuadrotorEnv are the same arrays (Garage's normalisers and
    the policy's input / output sizes are built from them)."""
    from gym_art_amd import QuadrotorEnv
    seen = 0
    for name in ("g7_obs_reward_variants", "g15_obs_variants_patched_imports"):
        d = gu.load(name)
        for blk in gu.env_blocks(d):
            if "obs_low" not in blk:
                continue
            kw = gu.kwargs_of(blk) if "kwargs_json" in blk else {}
            if name.startswith("g15"):
                kw = dict(obs_repr=str(blk["obs_repr"]))
                sn = json.loads(str(blk["sense_json"]))
                if sn is not None:
                    kw["sense_noise"] = sn
            if "module" in blk and str(blk["module"]) != "quadrotor":
                kw["reward"] = "multi"
            env = QuadrotorEnv(seed=0, **kw)
            assert np.array_equal(np.asarray(env.observation_space.low, dtype=np.float64), blk["obs_low"]), kw
            assert np.array_equal(np.asarray(env.observation_space.high, dtype=np.float64), blk["obs_high"]), kw
            if "act_low" in blk:
                assert np.array_equal(np.asarray(env.action_space.low, dtype=np.float64), blk["act_low"]), kw
                assert np.array_equal(np.asarray(env.action_space.high, dtype=np.float64), blk["act_high"]), kw
            env.close()
            seen += 1
    assert seen >= 12


def test_step_into_caller_provided_tensors():
    """env.step(actions, out=(obs, rew, done)) on the torch path: the same numbers as the allocating form, written into the caller's tensors."""
    import torch
    from gym_art_amd import QuadrotorEnv
    dev = torch.device("cuda", 0)
    n = 4096
    a_env, b_env = QuadrotorEnv(num_envs=n, seed=5, ep_time=0.1), QuadrotorEnv(num_envs=n, seed=5, ep_time=0.1)
    a_env.reset(); b_env.reset()
    obs = torch.empty((n, 18), device=dev); rew = torch.empty(n, device=dev); done = torch.empty(n, dtype=torch.uint8, device=dev)
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    for t in range(30):
        act = torch.rand((n, 4), device=dev, generator=gen) * 2 - 1
        o1, r1, d1, _ = a_env.step(act)
        o2, r2, d2, _ = b_env.step(act, out=(obs, rew, done))
        assert o2 is obs and r2 is rew and d2 is done
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2)
    a_env.close(); b_env.close()
