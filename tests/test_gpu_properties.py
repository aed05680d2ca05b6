"""-m gpu: size-independent properties of the HIP path at BASELINE sizes (N = 65 536 and 2^20), where the
oracle is too slow to follow: determinism, shard invariance of the RNG, rigid-body invariants, episode
structure with in-kernel auto-reset, reset distribution vs the reference (fixture G8), OU-noise statistics,
done-list compaction, and the specialised vs generic kernel instantiations."""
import numpy as np
import pytest

from tests import golden_util as gu
from tests import gpu_util as G

pytestmark = pytest.mark.gpu


def hummingbird_const(noise_sigma=0.01):
    c = gu.sub(gu.load("g2_hummingbird_raw"), "const_")
    c = dict(c)
    c["thrust_noise_sigma"] = np.float64(noise_sigma)
    return c


def actions_for(step, n, seed=0):
    rng = np.random.RandomState(1000 * seed + step)
    return rng.uniform(-1, 1, size=(n, 4)).astype(np.float32)


@pytest.mark.parametrize("alias", [0, 1])
def test_determinism_and_shard_invariance(alias):
    """Same seed -> identical bits; envs [k, k+m) of a big batch == a separate handle with env_id_offset k
    (RNG keyed by the GLOBAL env index: results do not depend on how the batch is sharded over GPUs)."""
    n, k, m, T = 65536, 40000, 4096, 30
    const = hummingbird_const()
    a = G.Handle(n, 0.005, 2, 12, const=const, noise=1, auto_reset=1, seed=7, alias=alias)
    b = G.Handle(n, 0.005, 2, 12, const=const, noise=1, auto_reset=1, seed=7, alias=alias)
    c = G.Handle(m, 0.005, 2, 12, const=const, noise=1, auto_reset=1, seed=7, env_id_offset=k, alias=alias)
    oa, ob, oc = a.reset(), b.reset(), c.reset()
    assert np.array_equal(oa, ob) and np.array_equal(oa[k:k + m], oc)
    for t in range(T):      # ep_len 12 -> two auto-resets inside the window
        act = actions_for(t, n)
        ra, rb, rc = a.step(act), b.step(act), c.step(act[k:k + m])
        for x, y, z in zip(ra, rb, rc):
            assert np.array_equal(x, y)
            assert np.array_equal(x[k:k + m], z)
    assert np.array_equal(a.get_state()[:, k:k + m], c.get_state())


@pytest.mark.parametrize("alias", [0, 1])
def test_full_size_invariants_and_episode_structure(alias):
    """N = 2^20 Hummingbird envs (the BASELINE metric's size), thrust noise on, in-kernel auto-reset."""
    n, ep_len = 1 << 20, 20
    h = G.Handle(n, 0.005, 2, ep_len, const=hummingbird_const(), noise=1, auto_reset=1, seed=3, compact_done=1, alias=alias)
    obs = h.reset()
    assert obs.shape == (n, 18) and np.all(np.isfinite(obs))
    rng = np.random.RandomState(0)
    base = rng.uniform(-1, 1, size=(4096, 4)).astype(np.float32)
    done_steps = []
    for t in range(2 * (ep_len + 1)):
        act = np.roll(np.tile(base, (n // 4096, 1)), t, axis=0)
        obs, rew, done = h.step(act)
        assert np.all(np.isfinite(obs)) and np.all(np.isfinite(rew))
        if done.any():
            assert done.all()                                # synchronous episodes: everybody finishes together
            assert np.array_equal(h.done_list(), np.arange(n, dtype=np.uint32))
            done_steps.append(t)
        else:
            assert len(h.done_list()) == 0
    assert done_steps == [ep_len, 2 * ep_len + 1]            # done = tick > ep_len -> ep_len + 1 steps per episode
    st = h.get_state()
    R = st[6:15].T.reshape(n, 3, 3)
    ortho = np.abs(np.einsum("nij,nkj->nik", R, R) - np.eye(3)).max()
    assert ortho < 1e-9, ortho                               # fp64-grade rotation chain stays orthonormal
    assert np.all(np.abs(np.linalg.det(R[:: 997]) - 1) < 1e-9)
    assert np.all(np.abs(st[15:18]) <= 40.0)                 # omega clip
    assert np.all(st[0:2] >= -10) and np.all(st[0:3] <= 10) and np.all(st[2] >= 0)   # room box
    assert np.all(st[37] == 0)                               # tick reset by the auto-reset of the last step
    # obs is [pos - goal, vel, R, omega] of that very state
    assert np.allclose(obs[:, 0:3], (st[0:3] - st[34:37]).T, atol=1e-6)
    assert np.allclose(obs[:, 6:15], st[6:15].T, atol=1e-7)


def test_reset_distribution_matches_reference():
    """Device reset vs 4000 resets of the reference (fixture G8): position box, floor clamp, yaw window."""
    from scipy import stats
    d = gu.load("g8_reset_distribution")
    n = 16384
    h = G.Handle(n, 0.005, 2, 500, const=hummingbird_const(0.0), seed=5, alias=1)
    h.reset()
    st = h.get_state()
    pos, R = st[0:3].T, st[6:15].T.reshape(n, 3, 3)
    for k in range(3):
        assert stats.ks_2samp(pos[:, k], d["pos"][:, k]).pvalue > 1e-3
    assert np.all(pos[:, 2] >= 0.25) and abs(np.mean(pos[:, 2] == 0.25) - np.mean(d["pos"][:, 2] == 0.25)) < 0.02

    def yaw_offset(p, r):
        return np.angle(np.exp(1j * (np.arctan2(r[:, 1, 0], r[:, 0, 0]) - np.arctan2(-p[:, 1], -p[:, 0]))))
    a, b = yaw_offset(pos, R), yaw_offset(d["pos"], d["rot"])
    assert np.max(np.abs(a)) <= np.pi / 3 + 1e-6
    assert stats.ks_2samp(a, b).pvalue > 1e-3
    assert np.abs(np.einsum("nij,nkj->nik", R, R) - np.eye(3)).max() < 5e-12      # 40-bit split state (alias)
    assert np.all(st[3:6] == 0) and np.all(st[15:18] == 0) and np.all(st[18:26] == 0)
    # a second reset draws new states; a masked reset leaves the others alone
    h2 = h.get_state()
    mask = np.zeros(n, np.uint8)
    mask[::2] = 1
    h.reset(mask)
    st2 = h.get_state()
    assert np.array_equal(st2[:, 1::2], h2[:, 1::2]) and not np.array_equal(st2[0:3, ::2], h2[0:3, ::2])
    # init_random_state / resample_goal variant
    hr = G.Handle(n, 0.005, 2, 500, const=hummingbird_const(0.0), seed=6, init_random_state=1, resample_goal=1)
    hr.reset()
    sr = hr.get_state()
    Rr = sr[6:15].T.reshape(n, 3, 3)
    assert np.abs(np.einsum("nij,nkj->nik", Rr, Rr) - np.eye(3)).max() < 1e-12
    assert stats.ks_2samp(np.linalg.norm(sr[3:6], axis=0), np.linalg.norm(d["vel_rs"], axis=1)).pvalue > 1e-3
    assert stats.ks_2samp(np.linalg.norm(sr[15:18], axis=0), np.linalg.norm(d["omega_rs"], axis=1)).pvalue > 1e-3
    assert stats.ks_2samp(sr[36], d["goal_rs"][:, 2]).pvalue > 1e-3
    assert stats.ks_2samp(Rr[:, 2, 2], d["rot_rs"][:, 2, 2]).pvalue > 1e-3
    assert stats.ks_2samp(Rr[:, 0, 1], d["rot_rs"][:, 0, 1]).pvalue > 1e-3


def test_ou_noise_statistics():
    """Philox/Box-Muller OU thrust noise: stationary std sigma / sqrt(1 - 0.85^2), zero mean, and
    independence across envs and motors (quad_utils.py:197-201 with theta = 0.15)."""
    n = 65536
    h = G.Handle(n, 0.005, 2, 5000, const=hummingbird_const(0.01), noise=1, seed=9)
    h.reset()
    act = np.zeros((n, 4), np.float32)
    for t in range(40):          # 80 OU updates >> 1/theta
        h.step(act)
    ou = h.get_state()[26:30]
    want = 0.01 / np.sqrt(1 - 0.85 ** 2)
    assert abs(ou.std() - want) / want < 0.02
    assert abs(ou.mean()) < 4 * want / np.sqrt(ou.size)
    c = np.corrcoef(ou)
    assert np.max(np.abs(c - np.eye(4))) < 0.02
    assert abs(np.corrcoef(ou[0, :-1], ou[0, 1:])[0, 1]) < 0.02
    from scipy import stats
    assert stats.kstest(ou[1] / ou[1].std(), "norm").pvalue > 1e-4


def test_specialised_kernels_match_generic_kernel():
    """The feature-specialised instantiations (F = 0, lag, noise) compute exactly what the generic one does."""
    import ctypes as C
    n, T = 8192, 25
    d3 = gu.load("g3_crazyflie")
    for const, noise in ((hummingbird_const(0.01), 1), (hummingbird_const(0.0), 0), (dict(gu.sub(d3, "const_")), 1)):
        fast = G.Handle(n, 0.005, 2, 10, const=const, noise=noise, auto_reset=1, seed=21)
        gen = G.Handle(n, 0.005, 2, 10, const=const, noise=noise, auto_reset=1, seed=21, obs_flags=8, force_generic=True)
        of, og = fast.reset(), gen.reset()
        assert np.array_equal(of, og[:, :18])
        for t in range(T):
            act = actions_for(t, n, seed=4)
            (of, rf, df), (og, rg, dg) = fast.step(act), gen.step(act)
            # two instantiations of one template: the compiler may contract a*b+c differently, so allow
            # last-bit differences of the fp64 chain (far below the fp32 output rounding almost always)
            assert np.allclose(of, og[:, :18], rtol=0, atol=2e-6) and np.allclose(rf, rg, rtol=0, atol=1e-7)
            assert np.array_equal(df, dg)
        assert gen.D == 22 and np.allclose(og[:, 18:22], actions_for(T - 2, n, seed=4), atol=0)   # appended: the previous action
        assert np.allclose(fast.get_state()[0:18], gen.get_state()[0:18], rtol=0, atol=1e-9)


@pytest.mark.parametrize("opts", [
    dict(reward_mode=1),                                                        # quadrotor_multi log-distance reward
    dict(rew={"rot": 0.3, "attitude": 0.2, "yaw": 0.1}),                        # acos terms
    dict(init_random_state=1),                                                  # random-state resets (in-kernel and explicit)
    dict(sense={}),                                                             # SensorNoise() defaults, white-noise gyro
    dict(sense={"quat_norm_std": 0.01, "pos_unif_range": 0.02, "vel_unif_range": 0.01, "quat_unif_range": 0.005},
         init_random_state=1, reward_mode=1, rew={"rot": 0.1}),
    dict(obs_flags=1),                                                          # body-frame observation (xyzr_vxyzr_R_omega)
    dict(obs_flags=2), dict(obs_flags=3, sense={}),                             # `_h` variants (19 words)
    dict(obs_flags=4), dict(rew={"action_change": 0.2}),                        # accelerometer words; action-change term
    dict(obs_flags=1, sense={"quat_norm_std": 0.02}),
])
def test_options_moved_into_the_specialised_kernels_match_the_generic_kernel(opts):
    """Options the specialised kernels honour through wave-uniform branches against the generic
    instantiation of the same template (forced by the flag-bearing `_h` observation, compared on the first 18 words):
    same RNG keys, same arithmetic -- equal up to FMA-contraction differences between two instantiations."""
    n, T = 4096, 25
    d3 = gu.load("g3_crazyflie")
    for const, noise in ((hummingbird_const(0.01), 1), (dict(gu.sub(d3, "const_")), 0)):
        gopts = dict(opts, obs_flags=opts.get("obs_flags", 0) | 8, force_generic=True)   # + `_act` words, generic kernel
        fast = G.Handle(n, 0.005, 2, 10, const=const, noise=noise, auto_reset=1, seed=33, **opts)
        gen = G.Handle(n, 0.005, 2, 10, const=const, noise=noise, auto_reset=1, seed=33, **gopts)
        of, og = fast.reset(), gen.reset()
        Df = fast.D
        fl = opts.get("obs_flags", 0)
        assert Df == 18 + (1 if fl & 2 else 0) + (3 if fl & 4 else 0) and gen.D == Df + 4
        assert np.allclose(of, og[:, :Df], rtol=0, atol=1e-6)
        for t in range(T):
            act = actions_for(t, n, seed=5)
            (of, rf, df), (og, rg, dg) = fast.step(act), gen.step(act)
            assert np.allclose(of, og[:, :Df], rtol=0, atol=3e-6) and np.allclose(rf, rg, rtol=0, atol=2e-7), t
            assert np.array_equal(df, dg)
        assert np.allclose(fast.get_state()[0:18], gen.get_state()[0:18], rtol=0, atol=1e-9)
        fast.close(); gen.close()


def test_alias_mode_equals_plain_mode_and_survives_buffer_changes():
    """obs_state_alias: same trajectories as the plain fp64 layout (the split keeps 39 of 53 mantissa bits), whichever observation buffers the caller passes: a fresh one per step, the same one in place, or a
    [T,N,D] rollout tensor through gaq_step_many_dev."""
    import torch
    from gym_art_amd import _lib
    n, T = 5000, 24          # not a multiple of 64: partial last tile
    const = hummingbird_const(0.01)
    plain = G.Handle(n, 0.005, 2, 9, const=const, noise=1, auto_reset=1, seed=13)
    alias = G.Handle(n, 0.005, 2, 9, const=const, noise=1, auto_reset=1, seed=13, alias=1)
    inplace = G.Handle(n, 0.005, 2, 9, const=const, noise=1, auto_reset=1, seed=13, alias=1)
    many = G.Handle(n, 0.005, 2, 9, const=const, noise=1, auto_reset=1, seed=13, alias=1)
    assert alias.alias and not plain.alias
    o0p = plain.reset()
    o0 = alias.reset()
    assert np.allclose(o0, o0p, rtol=2.5e-7, atol=1e-30)    # alias obs words are the state truncated (not rounded) to fp32
    lib = plain.lib
    dev = torch.device("cuda")
    guard = 4096             # canary floats after the observation buffer: nothing may be written past N*D
    ob_in = torch.full((n * 18 + guard,), 7.0, device=dev)
    rew_t, done_t = torch.empty(n, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)
    _lib.check(lib.gaq_reset_dev(inplace.h, None, _lib.ptr(ob_in), None))
    torch.cuda.synchronize()
    assert np.array_equal(ob_in[:n * 18].cpu().numpy().reshape(n, 18), o0) and bool((ob_in[n * 18:] == 7.0).all())
    acts = [actions_for(t, n, seed=8) for t in range(T)]
    big_obs = torch.zeros((T, n, 18), device=dev)
    big_rew, big_done = torch.zeros((T, n), device=dev), torch.zeros((T, n), dtype=torch.uint8, device=dev)
    many.reset()
    a_all = torch.tensor(np.stack(acts), device=dev)
    _lib.check(lib.gaq_step_many_dev(many.h, T, _lib.ptr(a_all), _lib.ptr(big_obs), _lib.ptr(big_rew), _lib.ptr(big_done), None))
    torch.cuda.synchronize()
    for t in range(T):
        op, rp, dp = plain.step(acts[t])
        oa, ra, da = alias.step(acts[t])                    # host path: library-owned device buffer
        a_t = torch.tensor(acts[t], device=dev)
        _lib.check(lib.gaq_step_dev(inplace.h, _lib.ptr(a_t), _lib.ptr(ob_in), _lib.ptr(rew_t), _lib.ptr(done_t), None))
        torch.cuda.synchronize()
        oi = ob_in[:n * 18].cpu().numpy().reshape(n, 18)
        assert np.allclose(oa, op, rtol=0, atol=2e-6) and np.allclose(ra, rp, atol=1e-7) and np.array_equal(da, dp)
        assert np.array_equal(oi, oa) and np.array_equal(rew_t.cpu().numpy(), ra)
        # the fused T-step rollout keeps the state in fp64 registers between steps (no 39-bit store per step): its
        # observations can differ from T single steps in the last fp32 bit of a few words
        assert np.allclose(big_obs[t].cpu().numpy(), oa, rtol=3e-7, atol=1e-9)
        assert np.allclose(big_rew[t].cpu().numpy(), ra, rtol=0, atol=1e-8)
        assert np.array_equal(big_done[t].cpu().numpy().astype(bool), da)
        assert bool((ob_in[n * 18:] == 7.0).all())
    assert np.allclose(many.get_state()[0:18], alias.get_state()[0:18], rtol=0, atol=1e-9)
    sp, sa = plain.get_state(), alias.get_state()
    assert np.allclose(sa[0:18], sp[0:18], rtol=0, atol=1e-9) and np.array_equal(sa[26:30], sp[26:30])
    # set_state / get_state round trip in alias mode keeps 48 bits
    alias.set_state(sp)
    assert np.allclose(alias.get_state()[0:18], sp[0:18], rtol=4e-12, atol=4e-12)      # 39-bit split state
    assert np.allclose(alias.observe(), plain.observe(), rtol=2.5e-7, atol=1e-30)      # truncated vs rounded fp32 head


def test_fp32_mode_is_fast_but_outside_the_parity_bar():
    """gaq_config.fp32_state: fp32 arithmetic, the observation rows are the whole state.  Held to what DESIGN.md
    section 2 says about fp32: the free-running error against the reference's G2 trajectories is orders above the
    fp64 path's (>= 1e-6) yet bounded (<= 2e-3 over 500 steps), the observation is bit-for-bit the state, single steps
    and fused rollouts agree to fp32 round-off, and unsupported configurations are refused instead of silently downgraded."""
    import torch
    from gym_art_amd import _lib
    d = gu.load("g2_hummingbird_raw")
    blocks = gu.env_blocks(d)
    const = gu.sub(d, "const_")
    n = 6 * 64
    b0 = blocks[0]
    mk = lambda **kw: G.Handle(n, float(b0["dt"]), int(b0["sim_steps"]), int(b0["ep_len"]), const=const, **kw)
    h32, h64 = mk(fp32=1), mk(alias=1)
    assert h32.alias and h64.alias
    st0 = G.planes_from_blocks(blocks, n)
    h32.set_state(st0); h64.set_state(st0)
    outs32, spread = G.run_blocks(h32, blocks, n)
    outs64, _ = G.run_blocks(h64, blocks, n)
    assert spread == 0.0
    e32 = max(gu.rel_err(o["obs"], b["obs"]) for o, b in zip(outs32, blocks))
    e64 = max(gu.rel_err(o["obs"], b["obs"]) for o, b in zip(outs64, blocks))
    print("fp32 mode max rel err over 500 steps: %.2e (fp64 path: %.2e)" % (e32, e64))
    assert e64 <= 1e-6 and 1e-6 < e32 <= 2e-3
    assert all(np.array_equal(o["done"], b["done"]) for o, b in zip(outs32, blocks))
    # observation == state, exactly
    st = h32.get_state()
    a = np.zeros((n, 4), np.float32)
    obs, _, _ = h32.step(a)
    st = h32.get_state()
    assert np.array_equal(obs[:, 3:18].astype(np.float64), st[3:18].T)
    assert np.array_equal(obs[:, 0:3], (st[0:3].T - st[34:37].T).astype(np.float32))
    # fused rollout vs single steps
    ha, hb = mk(fp32=1, noise=1, seed=3), mk(fp32=1, noise=1, seed=3)
    ha.set_state(st0); hb.set_state(st0)
    T = 12
    dev = torch.device("cuda")
    acts = (torch.rand((T, n, 4), device=dev) * 2 - 1)
    o_T, r_T, d_T = torch.zeros((T, n, 18), device=dev), torch.zeros((T, n), device=dev), torch.zeros((T, n), dtype=torch.uint8, device=dev)
    _lib.check(ha.lib.gaq_step_many_dev(ha.h, T, _lib.ptr(acts), _lib.ptr(o_T), _lib.ptr(r_T), _lib.ptr(d_T), None))
    o1, r1, d1 = torch.zeros((n, 18), device=dev), torch.zeros(n, device=dev), torch.zeros(n, dtype=torch.uint8, device=dev)
    for t in range(T):
        _lib.check(hb.lib.gaq_step_dev(hb.h, _lib.ptr(acts[t]), _lib.ptr(o1), _lib.ptr(r1), _lib.ptr(d1), None))
        torch.cuda.synchronize()
        # (same arithmetic, but the two kernels are separate instantiations: the compiler contracts a*b+c into FMAs
        #  differently, so agreement is to fp32 round-off growing slowly with t, not bit-for-bit)
        assert torch.allclose(o1, o_T[t], rtol=2e-5, atol=2e-5) and torch.allclose(r1, r_T[t], rtol=1e-4, atol=1e-6), t
    # a configuration that needs the generic kernel cannot run in fp32: refused, not downgraded
    for h in (h32, h64, ha, hb):
        h.close()
    with pytest.raises(ValueError, match="fp32_state"):
        mk(fp32=1, obs_flags=8)
    # through the env class: reset kernel, in-kernel auto-reset and state export in fp32 mode
    from gym_art_amd import QuadrotorEnv
    env = QuadrotorEnv(num_envs=1000, ep_time=0.05, seed=2, precision="fp32")          # ep_len 5
    assert env.obs_is_state and env.ep_len == 5
    obs = env.reset()
    st = env.get_state()
    assert np.array_equal(obs[:, 3:18].astype(np.float64), st[3:18].T) and np.all(st[37] == 0)
    R = st[6:15].T.reshape(1000, 3, 3)
    assert np.abs(np.einsum("nij,nkj->nik", R, R) - np.eye(3)).max() < 1e-6          # fp32-rounded rotation matrices
    seen_done = 0
    for t in range(13):
        obs, rew, done, _ = env.step(np.random.RandomState(t).uniform(-1, 1, (1000, 4)).astype(np.float32))
        assert np.all(np.isfinite(obs)) and np.all(np.isfinite(rew))
        seen_done += int(done.sum())
        st = env.get_state()
        assert np.array_equal(obs[:, 3:18].astype(np.float64), st[3:18].T)
    assert seen_done == 2 * 1000


def test_soak_six_default_episodes_at_full_size():
    """N = 2^20, default episode length (501 steps), 3100 steps = six episodes with noise and in-kernel resets, entirely
    on device tensors with fresh full-scale actions every step; then the invariants of a healthy simulation and the
    device-side episode bookkeeping."""
    import torch
    from gym_art_amd import QuadrotorEnv
    n, steps = 1 << 20, 3100
    env = QuadrotorEnv(num_envs=n, ep_time=5, seed=8)
    assert env.state_layout == 2 and env.ep_len == 500
    env.track_episodes(True)
    dev = torch.device("cuda")
    obs = torch.empty((n, 18), device=dev); rew = torch.empty(n, device=dev); done = torch.empty(n, dtype=torch.uint8, device=dev)
    env.reset_dev(obs)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    ring = [torch.rand((n, 4), device=dev, generator=gen) * 2 - 1 for _ in range(16)]
    n_done = 0
    for t in range(steps):
        env.step_dev(ring[t % 16], obs, rew, done)
        if t % 501 == 500:
            n_done += int(done.sum().item())
    torch.cuda.synchronize()
    env.check_finite()                                      # no non-finite reward in 3.25e9 env steps
    assert n_done == 6 * n                                  # 3006 = 6 * 501: everybody finished exactly six episodes
    stats = env.episode_stats()
    assert stats["episodes"] == 6 * n and abs(stats["mean_length"] - 501.0) < 1e-9
    assert np.isfinite(stats["mean_return"]) and -60.0 < stats["mean_return"] < -5.0      # random flailing: about -30 per episode
    st = env.get_state()
    assert np.all(np.isfinite(st))
    R = st[6:15].T.reshape(n, 3, 3)
    assert np.abs(np.einsum("nij,nkj->nik", R, R) - np.eye(3)).max() < 1e-9      # re-orthonormalised every 50 steps
    assert np.all(np.abs(st[15:18]) <= 40.0) and np.all(st[2] >= 0) and np.all(np.abs(st[0:3]) <= 10)
    assert np.all(st[37] == steps - 6 * 501)                # tick of the seventh episode
    assert np.all(st[38] < 100)                             # SVD counter keeps cycling
    o = obs.cpu().numpy()
    assert np.allclose(o[:, 0:3], (st[0:3] - st[34:37]).T, atol=2e-6) and np.allclose(o[:, 6:15], st[6:15].T, atol=2e-7)


def _windows_match_small_handles(kwargs, n, windows, steps, check_params=False):
    """Envs [k, k+m) of an n-env handle == a separate m-env handle with env_id_offset k, bit for bit (obs, reward, done):
    the property that makes a maximum-size launch checkable -- every 32-bit offset, buffer range and tile index of the big
    launch has to be right for the last window to agree."""
    import torch
    from gym_art_amd import QuadrotorEnv
    dev = torch.device("cuda", 0)
    big = QuadrotorEnv(num_envs=n, alias_obs=True, **kwargs)
    obs = torch.empty((n, 18), device=dev); rew = torch.empty(n, device=dev); done = torch.empty(n, dtype=torch.uint8, device=dev)
    act = torch.empty((n, 4), device=dev)
    small = []
    for k, m in windows:
        e = QuadrotorEnv(num_envs=m, alias_obs=True, env_id_offset=k, **kwargs)
        o = torch.empty((m, 18), device=dev); r = torch.empty(m, device=dev); d = torch.empty(m, dtype=torch.uint8, device=dev)
        e.reset_dev(o)
        small.append((k, m, e, o, r, d))
    big.reset_dev(obs)
    for k, m, e, o, r, d in small:
        assert torch.equal(obs[k:k + m], o), ("reset", k)
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    n_done = 0
    for t in range(steps):
        act.uniform_(-1, 1, generator=gen)
        big.step_dev(act, obs, rew, done)
        for k, m, e, o, r, d in small:
            e.step_dev(act[k:k + m].clone(), o, r, d)
            assert torch.equal(obs[k:k + m], o) and torch.equal(rew[k:k + m], r) and torch.equal(done[k:k + m], d), (t, k)
        n_done += int(done[-1].item())
    assert n_done >= 2                                        # auto-resets happened inside the window
    assert bool(torch.isfinite(obs[:: 4099]).all()) and bool(torch.isfinite(rew).all())
    big.check_finite()
    if check_params:
        for k, m, e, o, r, d in small:
            from gym_art_amd import _lib
            rows_b = np.empty((m, _lib.MODEL_DOUBLES)); rows_s = np.empty((m, _lib.MODEL_DOUBLES))
            _lib.check(big._lib.gaq_get_params(big._handle, _lib.ptr(rows_b), k, m))
            _lib.check(e._lib.gaq_get_params(e._handle, _lib.ptr(rows_s), 0, m))
            assert np.array_equal(rows_b, rows_s), k
    for s in small:
        s[2].close()
    big.close()
    del obs, act, rew, done
    torch.cuda.empty_cache()


def test_maximum_size_handle_uniform_model():
    """N = 2^27 envs in ONE handle (the documented maximum, 128 x the BASELINE metric's batch; ~20 GB of the 288 GB): first,
    middle and last windows against small handles."""
    n = 1 << 27
    _windows_match_small_handles(dict(ep_time=0.03, seed=11, init_random_state=True), n,
                                 [(0, 128), ((1 << 26) + 64 * 1001, 192), (n - 192, 192)], steps=16)


def test_large_handle_device_randomised_models():
    """N = 2^25 CrazyFlies with per-env parameters sampled and derived on the device, a new draw per episode: windows against
    small handles (the parameter planes, 45 x 8 B per env = 12 GB, and the sampler keyed by the global env index)."""
    n = 1 << 25
    kw = dict(dynamics_params="Crazyflie", dyn_sampler_1={"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"},
              dynamics_randomize_every=1, ep_time=0.03, seed=12)
    _windows_match_small_handles(kw, n, [(0, 128), ((1 << 24) + 64 * 77, 128), (n - 128, 128)], steps=16, check_params=True)
