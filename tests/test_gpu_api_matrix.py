"""GPU: the env-level operations (get_state / set_state, masked reset, observe, compacted done list, checkpoint / resume) against EVERY
kind of handle -- the three state layouts, fp32 state, packed observations with sensor noise, the generic tiers, Mellinger, per-env
models re-randomised every episode (fp64 and fp32), the swarm kernel.  The other test files check each operation in depth on one or two
kinds; the kernel-coverage run (tests/test_gpu_kernel_coverage.py) showed what such a gap can hide.  Everything here is an exact
property of twins built from the same constructor arguments and seed, so it needs no oracle:
  * set_state(get_state()) changes nothing: the twin that went through it keeps producing the same bits;
  * a masked reset touches the masked envs only (whole worlds for the swarm), and they start a fresh episode;
  * observe() returns the observation the last step returned (kinds without sensor noise: that one is drawn per call);
  * the compacted done list is nonzero(done);
  * state_dict() -> pickle -> a NEW env -> load_state_dict() continues bit for bit across episode ends."""
import pickle

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SAMPLER = {"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"}
N = 640            # ten wave tiles; 80 worlds of 8 agents
KINDS = {
    "alias": dict(alias_obs=True),
    "shadow_class_default": dict(),
    "plain_fp64_planes": dict(alias_obs=False),
    "fp32_state": dict(alias_obs=True, precision="fp32"),
    "crazyflie_alias": dict(alias_obs=True, dynamics_params="Crazyflie"),
    "packed_obs_sensor_noise": dict(sense_noise="default", obs_repr="xyz_vxyz_R_omega_acc_act"),
    "packed_obs_body_frame_h": dict(obs_repr="xyzr_vxyzr_R_omega_h", dynamics_params="Crazyflie"),
    "generic_lite_resampled_goals": dict(resample_goal=True),
    "generic_full_bias_walk": dict(sense_noise={"gyro_norm_std": 0.01, "quat_norm_std": 0.01}),
    "generic_diag_quaternion_obs": dict(obs_repr="xyz_vxyz_quat_omega"),
    "mellinger_crazyflie": dict(raw_control=False, dynamics_params="Crazyflie"),
    "mellinger_packed_obs": dict(raw_control=False, obs_repr="xyz_vxyz_R_omega_h"),
    "per_env_rerandomised": dict(dynamics_params="Crazyflie", dyn_sampler_1=dict(SAMPLER), dynamics_randomize_every=1, alias_obs=True),
    "per_env_rerandomised_class_default": dict(dynamics_params="Crazyflie", dyn_sampler_1=dict(SAMPLER), dynamics_randomize_every=2),
    "per_env_rerandomised_fp32": dict(dynamics_params="Crazyflie", dyn_sampler_1=dict(SAMPLER), dynamics_randomize_every=1, alias_obs=True,
                                      precision="fp32"),
    "random_quads_every_episode": dict(dynamics_params="RandomQuad", dynamics_randomize_every=1),
    "swarm": "swarm",
    "swarm_generic": "swarm_plain",
}


def build(kind, **extra):
    from gym_art_amd import QuadrotorEnv, QuadrotorEnvMulti
    common = dict(ep_time=0.08, seed=41, init_random_state=True, auto_reset=True)       # ep_len 8: an episode ends every nine steps
    common.update(extra)
    kw = KINDS[kind]
    if isinstance(kw, str):
        return QuadrotorEnvMulti(num_agents=8, num_worlds=N // 8, goal_radius=0.5, alias_obs=None if kw == "swarm" else False, **common)
    return QuadrotorEnv(num_envs=N, **common, **kw)


class Flight:
    def __init__(self, env, dev):
        import torch
        self.env = env
        D = env.obs_dim
        self.obs = torch.empty((N, D), device=dev); self.rew = torch.empty(N, device=dev); self.done = torch.empty(N, dtype=torch.uint8, device=dev)

    def reset(self, mask=None):
        self.env.reset_dev(self.obs, mask=mask)
        return self

    def step(self, a):
        self.env.step_dev(a, self.obs, self.rew, self.done)
        return self.obs.cpu().numpy().copy(), self.rew.cpu().numpy().copy(), self.done.cpu().numpy().copy()


def same(x, y):
    return all(np.array_equal(p, q) for p, q in zip(x, y))


@pytest.mark.parametrize("kind", sorted(KINDS))
def test_env_operations_on_every_kind_of_handle(kind):
    import torch
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev); gen.manual_seed(3)
    acts = torch.rand((80, N, 4), device=dev, generator=gen) * 2 - 1
    a, b, c = (Flight(build(kind), dev).reset() for _ in range(3))
    noisy_obs = bool(a.env._sense)
    is_swarm = isinstance(KINDS[kind], str)
    # desynchronise the episodes (by whole worlds for the swarm), the same way in all three
    phase = (np.arange(N) // (8 if is_swarm else 1)) % (a.env.ep_len + 1)
    for f in (a, b, c):
        st = f.env.get_state(); st[37] = phase; f.env.set_state(st)
    t = 0
    for _ in range(6):
        ra, rb = a.step(acts[t]), b.step(acts[t]); c.step(acts[t]); t += 1
        assert same(ra, rb), (kind, "twins differ before anything was done to them", t)
    # ---- set_state(get_state()) is the identity
    b.env.set_state(b.env.get_state())
    for _ in range(12):
        ra, rb = a.step(acts[t]), b.step(acts[t]); c.step(acts[t]); t += 1
        assert same(ra, rb), (kind, "set_state(get_state()) changed the flight", t)
    assert np.array_equal(a.env.get_state(), b.env.get_state())
    # ---- observe() = the observation the last step returned
    if not noisy_obs:
        assert np.array_equal(np.asarray(a.env.observe(), dtype=np.float32), ra[0]), (kind, "observe()")
    # ---- a masked reset: a and b take it, c does not
    rng = np.random.RandomState(5)
    groups = N // 8 if is_swarm else N
    pick = rng.permutation(groups)[:groups // 3]
    mask = np.zeros(N, np.uint8)
    if is_swarm:
        mask.reshape(-1, 8)[pick] = 1
    else:
        mask[pick] = 1
    m = torch.from_numpy(mask).to(dev)
    a.reset(mask=m); b.reset(mask=m)
    st = a.env.get_state()
    assert np.all(st[37][mask == 1] == 0) and np.array_equal(st[37][mask == 0], c.env.get_state()[37][mask == 0]), (kind, "ticks after the masked reset")
    keep = mask == 0
    per_env_models = bool(getattr(a.env, "_per_env", False))
    for k in range(7):
        ra, rb, rc = a.step(acts[t]), b.step(acts[t]), c.step(acts[t]); t += 1
        assert same(ra, rb), (kind, "twins differ after the same masked reset", t)
        if not (per_env_models and a.env.dynamics_randomize_every):     # (a reset of a re-randomising handle may re-draw parameters: quadrotor.py:1063)
            assert np.array_equal(ra[0][keep], rc[0][keep]) and np.array_equal(ra[1][keep], rc[1][keep]) and \
                np.array_equal(ra[2][keep], rc[2][keep]), (kind, "the masked reset touched other envs", t)
    assert np.isfinite(ra[0]).all() and np.isfinite(ra[1]).all()
    # ---- checkpoint -> pickle -> a new env
    blob = pickle.dumps(a.env.state_dict())
    d = Flight(build(kind).load_state_dict(pickle.loads(blob)), dev)
    if d.env.state_layout == 1:                 # heads aliased to the caller's tensor: hand it one, then load again (the state planes hold the values)
        d.env.reset_dev(d.obs, mask=torch.zeros(N, dtype=torch.uint8, device=dev))
        d.env.load_state_dict(pickle.loads(blob))
    ends = 0
    for _ in range(25):
        ra, rd = a.step(acts[t]), d.step(acts[t]); t += 1
        assert same(ra, rd), (kind, "the resumed env differs", t)
        ends += int(ra[2].sum())
    assert ends > N and np.array_equal(a.env.get_state(), d.env.get_state())
    for f in (a, b, c, d):
        f.env.check_finite()
        f.env.close()


@pytest.mark.parametrize("kind", ["alias", "fp32_state", "packed_obs_sensor_noise", "generic_lite_resampled_goals", "mellinger_crazyflie",
                                  "per_env_rerandomised_fp32", "swarm"])
def test_compacted_done_list_on_every_kind_of_handle(kind):
    import torch
    dev = torch.device("cuda", 0)
    f = Flight(build(kind, compact_done=True), dev).reset()
    st = f.env.get_state(); st[37] = (np.arange(N) // 8) % (f.env.ep_len + 1); f.env.set_state(st)
    gen = torch.Generator(device=dev); gen.manual_seed(4)
    seen = 0
    for t in range(20):
        _, _, done = f.step(torch.rand((N, 4), device=dev, generator=gen) * 2 - 1)
        idx = np.sort(np.asarray(f.env.done_indices()))
        assert np.array_equal(idx, np.nonzero(done)[0]), (kind, t)
        seen += len(idx)
    assert seen > N
    f.env.close()
