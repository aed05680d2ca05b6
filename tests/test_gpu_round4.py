"""-m gpu: what round 4 added or repaired -- gaq_get_params as a pure read, captured graphs of kernels that read the
graph-safe counter's first word alone, the caller's action array left alone by the host step path."""
import ctypes as C
import os

import numpy as np
import pytest

from gym_art_amd import _lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAMPLER = {"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"}


def _rows(lib, handle, first, count):
    rows = np.empty((count, _lib.MODEL_DOUBLES), dtype=np.float64)
    _lib.check(lib.gaq_get_params(handle, _lib.ptr(rows), first, count))
    return rows


@pytest.mark.parametrize("layout", ["class_default", "fp32"])
def test_reading_the_parameters_changes_nothing(layout):
    """gaq_get_params is a READ: twins with per-episode re-randomisation on the device -- staggered ten-step episodes, so
    promotions that move only the hot planes happen on every step -- step side by side; one of them is asked for parameters all the
    time (one env, a window, everything).  States, counters and the rows read stay bit-identical to the twin that is never asked, and
    what is read equals what the host derives from the sampled trees (the cold planes of a promoted env are a draw behind in memory)."""
    import torch
    from gym_art_amd import QuadrotorEnv
    from gym_art_amd import quad_params as qp
    dev = torch.device("cuda", 0)
    n, ep = 1000, 0.05
    kw = dict(dynamics_params="Crazyflie", num_envs=n, ep_time=ep, seed=11, dyn_sampler_1=dict(SAMPLER), dynamics_randomize_every=1)
    if layout == "fp32":
        kw.update(precision="fp32", alias_obs=True)
    a, b = QuadrotorEnv(**kw), QuadrotorEnv(**kw)
    lib = a._lib
    bufs = [tuple((torch.empty((n, 18), device=dev), torch.empty(n, device=dev), torch.empty(n, dtype=torch.uint8, device=dev))) for _ in range(2)]
    for env, (o, _, _) in zip((a, b), bufs):
        env.reset_dev(o)
        st = env.get_state()
        st[37] = np.arange(n) % (env.ep_len + 1)             # staggered episode clocks: some envs finish on every step
        env.set_state(st)
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    rng = np.random.RandomState(2)
    for t in range(40):
        act = torch.rand((n, 4), device=dev, generator=gen) * 2 - 1
        for env, (o, r, d) in zip((a, b), bufs):
            env.step_dev(act, o, r, d)
        # reads on twin b only: one env, a ragged window across tiles, sometimes everything
        f = int(rng.randint(0, n))
        _rows(lib, b._handle, f, 1)
        f = int(rng.randint(0, n - 150))
        _rows(lib, b._handle, f, 150)
        if t % 7 == 3:
            _rows(lib, b._handle, 0, n)
        torch.cuda.synchronize()
        assert torch.equal(bufs[0][0], bufs[1][0]) and torch.equal(bufs[0][1], bufs[1][1]) and torch.equal(bufs[0][2], bufs[1][2]), t
    a.check_finite(); b.check_finite()
    sa, sb = a.get_state(), b.get_state()
    assert np.array_equal(sa, sb)
    ra, rb = _rows(lib, a._handle, 0, n), _rows(lib, b._handle, 0, n)
    assert np.array_equal(ra, rb)
    # ... and the rows are the models of the trees the envs fly with NOW (hot and cold planes alike)
    again, _ = qp.derive_models(b.sampled_trees())
    got = _lib.rows_to_models(rb)
    for key in ("mass", "inertia", "thrust_max", "torque_max", "prop_pos", "arm", "damp_time_up", "damp_time_down", "linearity"):
        err = np.max(np.abs(np.asarray(again[key]) - np.asarray(got[key])) / np.maximum(np.abs(np.asarray(again[key])), 1e-300))
        assert err <= 1e-12, (key, err)
    # every env has been promoted at least three times by now: the read really went through the re-derivation path
    cnt = _lib.GaqCounters()
    res = np.empty(n, np.uint32)
    _lib.check(lib.gaq_get_counters(b._handle, C.byref(cnt), None, _lib.ptr(res)))
    assert res.min() >= 3
    # switching the period off leaves the cold planes in memory a draw behind: the read still returns the rows of the
    # draw every env flies with
    rz = _lib.GaqRandomizer()
    base = b.dynamics_params_batched
    rz.sampler, rz.every = 0, 0
    rz.ratio[:] = list(qp.ratio_rows(base, 0.2, None)[0])
    C.memmove(C.byref(rz.base), qp.flatten_tree(base)[0].ctypes.data, C.sizeof(rz.base))
    _lib.check(lib.gaq_set_randomizer(b._handle, C.byref(rz)))
    assert np.array_equal(_rows(lib, b._handle, 0, n), rb)
    a.close(); b.close()


def test_reading_parameters_the_caller_supplied_before_any_draw():
    """A C-ABI caller whose planes differ from the randomizer's base (gaq_set_param_trees of its own, then gaq_set_randomizer with a
    period): until an env's first promotion its planes are the caller's, gaq_get_params returns exactly those, and asking does not
    replace them with the base model's.  Twins again: the one that is read from flies the same trajectory."""
    from tests import gpu_util as G, golden_util as gu
    from gym_art_amd import quad_params as qp, quad_models
    n, dt, ep_len = 256, 0.005, 9
    cf = qp.broadcast_tree(quad_models.crazyflie_params(), 1)
    mine = qp.broadcast_tree(quad_models.crazyflie_params(), n)
    r9 = np.random.RandomState(9)
    mine["motor"]["thrust_to_weight"] = mine["motor"]["thrust_to_weight"] * (1.0 + 0.05 * r9.uniform(-1, 1, n))
    mine["motor"]["damp_time_up"] = mine["motor"]["damp_time_up"] * (1.0 + 0.05 * r9.uniform(-1, 1, n))
    trees = np.ascontiguousarray(qp.flatten_tree(mine))
    rng = np.random.RandomState(4)
    const = dict(gu.sub(gu.load("g3_crazyflie"), "const_"))
    const["thrust_noise_sigma"] = np.float64(0.01)
    hs = []
    for _ in range(2):
        h = G.Handle(n, dt, 2, ep_len, per_env=1, auto_reset=1, noise=1, seed=8, const=const)
        _lib.check(h.lib.gaq_set_param_trees(h.h, _lib.ptr(trees), 0, 0, n))
        rz = _lib.GaqRandomizer()
        rz.sampler, rz.every = 0, 1
        rz.ratio[:] = list(qp.ratio_rows(cf, 0.2, None)[0])
        C.memmove(C.byref(rz.base), qp.flatten_tree(cf)[0].ctypes.data, C.sizeof(rz.base))
        _lib.check(h.lib.gaq_set_randomizer(h.h, C.byref(rz)))
        h.reset()
        st = h.get_state()
        st[37] = np.arange(n) % (ep_len + 1)
        h.set_state(st)
        hs.append(h)
    ha, hb = hs
    given = _rows(ha.lib, ha.h, 0, n)
    assert np.ptp(given[:, 4]) > 0                  # thrust_max differs between envs: the caller's trees, not the base
    want, _ = qp.derive_models(mine)
    assert np.max(np.abs(given[:, 4:8] - want["thrust_max"]) / want["thrust_max"]) <= 1e-12
    seen_first = np.zeros(n, bool)
    for t in range(2 * (ep_len + 1)):
        act = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
        before = _rows(hb.lib, hb.h, 0, n)          # a read before every step of twin b
        oa, ra_, da = ha.step(act)
        ob, rb_, db = hb.step(act)
        assert np.array_equal(oa, ob) and np.array_equal(ra_, rb_) and np.array_equal(da, db), t
        # envs that have not finished an episode yet still hold the caller's planes, bit for bit
        assert np.array_equal(before[~seen_first], given[~seen_first]), t
        seen_first |= db
    assert seen_first.all()
    after = _rows(hb.lib, hb.h, 0, n)
    assert np.all(after[:, 0] != given[:, 0])       # every env flies a drawn model now (the masses of the links were perturbed)
    assert np.array_equal(after, _rows(ha.lib, ha.h, 0, n))
    assert np.array_equal(ha.get_state(), hb.get_state())
    ha.close(); hb.close()


def test_host_step_keeps_its_own_copy_of_the_callers_action_array():
    """env.actions[1] is the PREVIOUS action (quadrotor.py:943-944).  For big batches the class keeps the float32 array it was given
    instead of converting 16 MB to float64 on every step -- its own copy of it: a sampler that refills ONE action buffer in place
    must not see the previous action change under it."""
    from gym_art_amd import QuadrotorEnv
    n = 8192                                                          # above the 4096-env threshold of the float32 shortcut
    env = QuadrotorEnv(num_envs=n, ep_time=1, seed=1)
    env.reset()
    buf = np.random.RandomState(0).uniform(-1, 1, (n, 4)).astype(np.float32)
    first = buf.copy()
    env.step(buf)
    buf[:] = 0.25                                                     # the sampler refills its buffer in place
    env.step(buf)
    acts = env.actions
    assert np.array_equal(np.asarray(acts[0], dtype=np.float32), np.full((n, 4), 0.25, np.float32))
    assert np.array_equal(np.asarray(acts[1], dtype=np.float32), first)
    env.close()


# ---- one env object over several devices in one process (SURVEY 8b `device_ids`, 8e; include/gaq.h gaq_sharded) -----------------------
MULTI_CASES = [
    # (label, constructor kwargs, env count) -- device lists are given per test
    ("class default layout", dict(), 1024),
    ("heads in the caller's tensor, ragged", dict(alias_obs=True), 1000),
    ("fp64 planes, one env in the last shard", dict(alias_obs=False), 129),
    ("sensor noise + random initial states", dict(sense_noise="default", init_random_state=True), 777),
    ("CrazyFlie, re-randomised on the device every episode", dict(dynamics_params="Crazyflie", dyn_sampler_1=dict(SAMPLER),
                                                                  dynamics_randomize_every=1), 1000),
    ("Mellinger", dict(raw_control=False), 640),
    ("gyro-bias walk (state that survives resets), resampled goals", dict(sense_noise={"gyro_norm_std": 0.01}, resample_goal=True), 320),
]


@pytest.mark.parametrize("force_copy", [False, True])
@pytest.mark.parametrize("case", MULTI_CASES, ids=[c[0] for c in MULTI_CASES])
def test_one_process_multi_device_env_equals_one_handle(case, force_copy):
    """QuadrotorEnv(num_envs=N, device_ids=[...]) against QuadrotorEnv(num_envs=N): the stacked observations / rewards / dones of the
    NumPy step() loop and of the torch step_dev() loop, the state planes and the parameters are bit-equal, through resets (six-step
    episodes) -- the shards' random streams are keyed by the global env index.  On a one-GPU box the device list names device 0 four
    times: the shards then write the caller's tensors in place; GAQ_SHARDED_FORCE_COPY=1 sends every shard through the staging buffers
    and copies a remote device's shard takes."""
    import torch
    from gym_art_amd import QuadrotorEnv
    label, kw, n = case
    kw = dict(kw, num_envs=n, ep_time=0.05, seed=17)
    one = QuadrotorEnv(**kw)
    if force_copy:
        os.environ["GAQ_SHARDED_FORCE_COPY"] = "1"
    try:
        many = QuadrotorEnv(device_ids=[0, 0, 0, 0], **kw)
    finally:
        os.environ.pop("GAQ_SHARDED_FORCE_COPY", None)
    assert type(many).__name__ == "QuadrotorEnvOnDevices" and isinstance(many, QuadrotorEnv)
    assert many.num_envs == n and sum(c for _, c in many.shard_ranges) == n and many.obs_dim == one.obs_dim
    assert all(f % 64 == 0 for f, _ in many.shard_ranges) and 2 <= len(many.shards) <= 4
    assert many.observation_space.shape == one.observation_space.shape and many.spec.max_episode_steps == one.spec.max_episode_steps
    rng = np.random.RandomState(3)
    o1, o2 = one.reset(), many.reset()
    assert np.array_equal(o1, o2), label
    for t in range(9):
        a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
        if not one.raw_control:
            a *= 0
        (oa, ra, da, _), (ob, rb, db, _) = one.step(a), many.step(a)
        assert np.array_equal(oa, ob) and np.array_equal(ra, rb) and np.array_equal(da, db), (label, t)
    assert np.array_equal(one.get_state(), many.get_state())
    assert many.traj_count == one.traj_count > 0
    ma, mb = one.models, many.models
    assert all(np.array_equal(ma[k], mb[k]) for k in ma)
    # device tensors on device_ids[0]
    dev = torch.device("cuda", 0)
    D = one.obs_dim
    t1 = (torch.empty((n, D), device=dev), torch.empty(n, device=dev), torch.empty(n, dtype=torch.uint8, device=dev))
    t2 = (torch.empty((n, D), device=dev), torch.empty(n, device=dev), torch.empty(n, dtype=torch.uint8, device=dev))
    mask = torch.as_tensor((np.arange(n) % 3 == 0).astype(np.uint8), device=dev)
    one.reset_dev(t1[0], mask); many.reset_dev(t2[0], mask)
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    for t in range(9):
        a = torch.rand((n, 4), device=dev, generator=gen) * 2 - 1
        if not one.raw_control:
            a = a * 0
        one.step_dev(a, *t1); many.step_dev(a, *t2)
        torch.cuda.synchronize()
        assert torch.equal(t1[0], t2[0]) and torch.equal(t1[1], t2[1]) and torch.equal(t1[2], t2[2]), (label, t)
    one.check_finite(); many.check_finite()
    assert torch.cuda.current_device() == 0
    one.close(); many.close()


def test_multi_device_env_pickles_by_constructor_arguments_and_keeps_subclasses():
    """Pickling (quadrotor.py:688: by constructor arguments) of a multi-device env gives back a multi-device env of the USER's class with
    the same device list, which flies the same episode; the fork's class and the swarm class keep their own behaviour (worlds(), the
    log-distance reward) when they are spread over devices; what is not built for several devices says so."""
    import pickle
    from gym_art_amd import QuadrotorEnv, QuadrotorEnvMulti
    from gym_art_amd.quadrotor_multi import QuadrotorEnv as ForkEnv
    n = 512
    env = QuadrotorEnv(num_envs=n, device_ids=[0, 0], ep_time=0.1, seed=5)
    twin = pickle.loads(pickle.dumps(env))
    assert type(twin) is type(env) and twin.device_ids == [0, 0] and twin.num_envs == n and len(twin.shards) == 2
    rng = np.random.RandomState(1)
    assert np.array_equal(env.reset(), twin.reset())
    for t in range(4):
        a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
        r1, r2 = env.step(a), twin.step(a)
        assert all(np.array_equal(x, y) for x, y in zip(r1[:3], r2[:3]))
    # checkpoint -> new env -> continue
    sd = env.state_dict()
    cont = QuadrotorEnv(num_envs=n, device_ids=[0, 0], ep_time=0.1, seed=5).load_state_dict(sd)
    a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
    r1, r2 = env.step(a), cont.step(a)
    assert all(np.array_equal(x, y) for x, y in zip(r1[:3], r2[:3]))
    for e in (env, twin, cont):
        e.close()
    # the swarm class: 40 worlds of 8 agents over three shards of whole worlds, against one handle
    sw1 = QuadrotorEnvMulti(num_agents=8, num_worlds=40, ep_time=0.05, seed=2)
    sw3 = QuadrotorEnvMulti(num_agents=8, num_worlds=40, ep_time=0.05, seed=2, device_ids=[0, 0, 0])
    assert isinstance(sw3, QuadrotorEnvMulti) and all(c % 8 == 0 for _, c in sw3.shard_ranges) and len(sw3.shards) == 3
    assert np.array_equal(sw1.reset(), sw3.reset())
    for t in range(8):
        a = rng.uniform(-1, 1, (320, 4)).astype(np.float32)
        r1, r3 = sw1.step(a), sw3.step(a)
        assert all(np.array_equal(x, y) for x, y in zip(r1[:3], r3[:3])), t
    assert sw3.worlds(r3[0]).shape == (40, 8, sw3.obs_dim)
    assert type(pickle.loads(pickle.dumps(sw3))).__name__ == "QuadrotorEnvMultiOnDevices"
    sw1.close(); sw3.close()
    # the fork's class: its own reward
    f1 = ForkEnv(dynamics_params="DefaultQuad", num_envs=256, seed=3)
    f2 = ForkEnv(dynamics_params="DefaultQuad", num_envs=256, seed=3, device_ids=[0, 0])
    assert np.array_equal(f1.reset(), f2.reset())
    a = rng.uniform(-1, 1, (256, 4)).astype(np.float32)
    r1, r2 = f1.step(a), f2.step(a)
    assert np.array_equal(r1[1], r2[1]) and f2.rew_coeff["effort"] == 0.01
    f1.close(); f2.close()
    # a single id is just `device`; a device list with per-device facilities says where they live
    single = QuadrotorEnv(num_envs=64, device_ids=[0])
    assert type(single) is QuadrotorEnv and single.device == 0
    single.close()
    with pytest.raises(NotImplementedError, match="info=True"):
        QuadrotorEnv(num_envs=128, device_ids=[0, 0], info=True)
    m = QuadrotorEnv(num_envs=128, device_ids=[0, 0])
    with pytest.raises(NotImplementedError, match="per-device facility"):
        m.set_graph_safe(True)
    m.close()
    with pytest.raises(ValueError, match="out of range"):
        QuadrotorEnv(num_envs=128, device_ids=[0, 99])


def test_c_level_sharded_handle_equals_one_handle():
    """include/gaq.h for a plain-C caller: gaq_create_sharded / gaq_step_sharded(_dev) / gaq_reset_sharded(_dev) against gaq_create /
    gaq_step / gaq_reset on the same configuration -- host pointers and device pointers, a masked reset, a ragged batch, the ranges the
    library reports, shard handles borrowed for a state read-back."""
    import torch
    from tests.test_plan_cpu import base_cfg
    lib = _lib.load()
    n = 64 * 7 + 13
    cfg = base_cfg(n, noise=1, auto_reset=1, obs_state_alias=2, ep_len=5, seed=77)
    one = C.c_void_p()
    _lib.check(lib.gaq_create(C.byref(cfg), C.byref(one)))
    ids = (C.c_int32 * 3)(0, 0, 0)
    sh = C.c_void_p()
    _lib.check(lib.gaq_create_sharded(C.byref(cfg), ids, 3, C.byref(sh)))
    assert lib.gaq_sharded_num_shards(sh) == 3 and lib.gaq_sharded_num_envs(sh) == n
    f, c, d = C.c_int64(0), C.c_int64(0), C.c_int32(-1)
    spans = []
    for k in range(3):
        _lib.check(lib.gaq_sharded_range(sh, k, C.byref(f), C.byref(c), C.byref(d)))
        spans.append((f.value, c.value)); assert d.value == 0
    assert spans == [(0, 192), (192, 192), (384, n - 384)]
    o1, o2 = np.empty((n, 18), np.float32), np.empty((n, 18), np.float32)
    r1, r2, d1, d2 = np.empty(n, np.float32), np.empty(n, np.float32), np.empty(n, np.uint8), np.empty(n, np.uint8)
    _lib.check(lib.gaq_reset(one, None, _lib.ptr(o1))); _lib.check(lib.gaq_reset_sharded(sh, None, _lib.ptr(o2)))
    assert np.array_equal(o1, o2)
    rng = np.random.RandomState(0)
    finished = 0
    for t in range(8):
        a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
        _lib.check(lib.gaq_step(one, _lib.ptr(a), _lib.ptr(o1), _lib.ptr(r1), _lib.ptr(d1)))
        _lib.check(lib.gaq_step_sharded(sh, _lib.ptr(a), _lib.ptr(o2), _lib.ptr(r2), _lib.ptr(d2)))
        assert np.array_equal(o1, o2) and np.array_equal(r1, r2) and np.array_equal(d1, d2), t
        finished += int(d1.sum())
    assert finished == n                             # ep_len 5: the eight steps crossed one episode end (in-kernel resets)
    mask = (np.arange(n) % 2).astype(np.uint8)
    _lib.check(lib.gaq_reset(one, _lib.ptr(mask), _lib.ptr(o1))); _lib.check(lib.gaq_reset_sharded(sh, _lib.ptr(mask), _lib.ptr(o2)))
    assert np.array_equal(o1, o2)
    # device pointers, on a side stream
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream()
    ta = torch.empty((n, 4), device=dev)
    to1, tr1, td1 = torch.empty((n, 18), device=dev), torch.empty(n, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)
    to2, tr2, td2 = torch.empty((n, 18), device=dev), torch.empty(n, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)
    for t in range(8):
        ta.copy_(torch.as_tensor(rng.uniform(-1, 1, (n, 4)).astype(np.float32)))
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            st = C.c_void_p(side.cuda_stream)
            _lib.check(lib.gaq_step_dev(one, _lib.ptr(ta), _lib.ptr(to1), _lib.ptr(tr1), _lib.ptr(td1), st))
            _lib.check(lib.gaq_step_sharded_dev(sh, _lib.ptr(ta), _lib.ptr(to2), _lib.ptr(tr2), _lib.ptr(td2), st))
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert torch.equal(to1, to2) and torch.equal(tr1, tr2) and torch.equal(td1, td2), t
    # the shards are ordinary handles: their states stack to the one handle's
    st1 = np.empty((42, n))
    _lib.check(lib.gaq_get_state(one, _lib.ptr(st1)))
    parts = []
    for k, (fk, ck) in enumerate(spans):
        p = np.empty((42, ck))
        _lib.check(lib.gaq_get_state(C.c_void_p(lib.gaq_sharded_shard(sh, k)), _lib.ptr(p)))
        parts.append(p)
    assert np.array_equal(st1, np.concatenate(parts, axis=1))
    _lib.check(lib.gaq_synchronize_sharded(sh))
    _lib.check(lib.gaq_destroy_sharded(sh)); _lib.check(lib.gaq_destroy(one))
    bad = (C.c_int32 * 2)(0, 99)
    assert lib.gaq_create_sharded(C.byref(cfg), bad, 2, C.byref(sh)) == -1 and b"out of range" in lib.gaq_last_error()


@pytest.mark.parametrize("n,force_nt", [(64 * 1000, False), ((1 << 21) + 64 * 37, True)])
def test_self_advancing_step_counter_over_many_replays_at_odd_wave_counts(n, force_nt):
    """the one-launch graph-safe counter (gaq_kernels.hpp step_counter_checkin) at wave counts that are NOT powers of
    two -- 1000 waves (2^shift = 1024: the first wave adds 25) and 32 808 waves at 2^21 + 2368 envs with the small-batch kernel forced
    (2^shift = 65 536: the first wave adds 32 729; the waves are scheduled in many rounds, late ones must still see THIS launch's
    index) -- over 200 single-step replays of one captured graph, against eager steps of a twin: bit-equal, and the counter read back is
    exact."""
    import torch
    from gym_art_amd import QuadrotorEnv
    from tests.test_plan_cpu import base_cfg, plan
    kw = dict(num_envs=n, ep_time=0.1, seed=23, alias_obs=True)
    if force_nt:
        os.environ["GAQ_NT"] = "1"
    try:
        eager, graphed = QuadrotorEnv(**kw), QuadrotorEnv(**kw)
    finally:
        os.environ.pop("GAQ_NT", None)
    p = plan(base_cfg(n, noise=1, obs_state_alias=1, auto_reset=1))
    assert p.ctr_waves & (p.ctr_waves - 1) and p.ctr_inc0 > 1 and (1 << p.ctr_shift) - p.ctr_inc0 == p.ctr_waves - 1
    dev = torch.device("cuda")
    bufs = [(torch.empty((n, 18), device=dev), torch.empty(n, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)) for _ in range(2)]
    a_g = torch.empty((n, 4), device=dev)
    eager.reset_dev(bufs[0][0]); graphed.reset_dev(bufs[1][0])
    graphed.set_graph_safe(True)
    assert graphed.launch_variant & 8192                                  # the self-counting twin (F_CTR)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    acts = torch.rand((5, n, 4), device=dev, generator=gen) * 2 - 1
    a_g.copy_(acts[0])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graphed.step_dev(a_g, *bufs[1])
    torch.cuda.current_stream().wait_stream(side)
    eager.step_dev(acts[0], *bufs[0])
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        graphed.step_dev(a_g, *bufs[1])
    lib = _lib.load()
    ctr = _lib.GaqCounters()
    K = 200
    for t in range(K):
        a_g.copy_(acts[t % 5])
        g.replay()
        eager.step_dev(acts[t % 5], *bufs[0])
        if t % 50 == 49:
            torch.cuda.synchronize()
            assert all(torch.equal(x, y) for x, y in zip(bufs[0], bufs[1])), t
            _lib.check(lib.gaq_get_counters(graphed._handle, C.byref(ctr), None, None))
            assert ctr.step_index == 2 + t
    eager.close(); graphed.close()


def test_a_captured_graph_of_a_first_word_kernel_survives_self_counting_launches():
    """a step kernel WITHOUT the self-counting twin reads the graph-safe counter's first word alone; whether the spread
    check-ins of earlier F_CTR launches must be folded into it first was decided when the launch was enqueued -- a graph captured while
    nothing was spread held no fold node and, replayed after F_CTR launches, keyed its noise with a stale index.  A captured launch now
    always carries the fold.  Here: packed rows registered (the F_ROWS twin: a first-word kernel) -> capture; rows unregistered -> eager
    F_CTR steps; rows registered again -> replay.  Against a twin that steps eagerly throughout."""
    import torch
    from gym_art_amd import QuadrotorEnv
    n = 65536
    kw = dict(num_envs=n, ep_time=0.1, seed=31, alias_obs=True)
    eager, graphed = QuadrotorEnv(**kw), QuadrotorEnv(**kw)
    dev = torch.device("cuda")
    be = (torch.empty((n, 18), device=dev), torch.empty(n, device=dev), torch.empty(n, dtype=torch.uint8, device=dev))
    bg = (torch.empty((n, 18), device=dev), torch.empty(n, device=dev), torch.empty(n, dtype=torch.uint8, device=dev))
    rows = torch.empty((n, 20), device=dev)
    a_g = torch.empty((n, 4), device=dev)
    gen = torch.Generator(device=dev); gen.manual_seed(2)
    acts = torch.rand((6, n, 4), device=dev, generator=gen) * 2 - 1
    eager.reset_dev(be[0]); graphed.reset_dev(bg[0])
    graphed.set_graph_safe(True)
    graphed.set_packed_rows(rows)
    assert graphed.launch_variant & 4096 and not graphed.launch_variant & 8192     # F_ROWS, not F_CTR: reads the first word alone
    a_g.copy_(acts[0])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graphed.step_dev(a_g, *bg)                                                  # warm-up outside the capture
    torch.cuda.current_stream().wait_stream(side)
    eager.step_dev(acts[0], *be)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):                                                       # captured while no check-in is spread
        graphed.step_dev(a_g, *bg)
    graphed.set_packed_rows(None)
    assert graphed.launch_variant & 8192
    for t in range(1, 4):                                                           # eager self-counting steps: check-ins spread over 64 words
        graphed.step_dev(acts[t], *bg); eager.step_dev(acts[t], *be)
    graphed.set_packed_rows(rows)
    for t in range(4, 6):                                                           # the old graph, twice
        a_g.copy_(acts[t])
        g.replay()
        eager.step_dev(acts[t], *be)
        torch.cuda.synchronize()
        assert all(torch.equal(x, y) for x, y in zip(be, bg)), t
        assert torch.equal(rows[:, :18], bg[0])
    ctr = _lib.GaqCounters()
    _lib.check(_lib.load().gaq_get_counters(graphed._handle, C.byref(ctr), None, None))
    assert ctr.step_index == 6
    eager.close(); graphed.close()


# ---- per-env goals and the gyro-bias walk on the split state (quad_core.hpp F_ENVX) ----------------------------------------------------
ENVX_CASES = [
    ("resample_goal", dict(resample_goal=True), 0),
    ("excite", dict(excite=True), 0),
    ("resample_goal + excite, Crazyflie", dict(resample_goal=True, excite=True, dynamics_params="Crazyflie"), 2),
    ("gyro-bias walk", dict(sense_noise={"gyro_norm_std": 0.01, "quat_norm_std": 0.01, "pos_unif_range": 0.01}), 0),
    ("gyro-bias walk + resample_goal + the aux row, thrust noise off", dict(sense_noise={"gyro_norm_std": 0.02}, resample_goal=True, info=True,
                                                                          thrust_noise="off"), -4),
    ("excite + body-frame observation with the height", dict(excite=True, obs_repr="xyzr_vxyzr_R_omega_h"), 0),
    ("resample_goal + quaternion observation", dict(resample_goal=True, obs_repr="xyz_vxyz_quat_omega"), 0),
    ("resample_goal + excite, per-env randomised Crazyflie re-randomised every episode",
     dict(resample_goal=True, excite=True, dynamics_params="Crazyflie", dyn_sampler_1=dict(SAMPLER), dynamics_randomize_every=1), 1 + 2 + 2048),
    ("excite, one random quadrotor per env, t2w / t2t observed", dict(excite=True, dynamics_params="RandomQuad", obs_repr="xyz_vxyz_R_omega_t2w_t2t"), 1 + 2),
    # ... and under the Mellinger controller, which is what excite is for (F_MELL | F_ENVX | F_AUXP)
    ("excite, Mellinger, sensor noise, height observed", dict(excite=True, raw_control=False, sense_noise="default", obs_repr="xyz_vxyz_R_omega_h"), 16384),
    ("resample_goal + the aux row, Mellinger, Crazyflie", dict(resample_goal=True, info=True, raw_control=False, dynamics_params="Crazyflie"), 16384 + 2),
]


@pytest.mark.parametrize("case", ENVX_CASES, ids=[c[0] for c in ENVX_CASES])
def test_per_env_goals_and_gyro_bias_on_the_split_state_against_the_generic_kernel(case):
    """resample_goal / excite (quadrotor.py:1078-1081, :957-963) and SensorNoise's gyro-bias random walk (sensor_noise.py:160-168) for a
    uniform model in the class default layout: step_kernel<197648 ...> / <459792 ...> (F_ENVX [| F_BIAS] | F_AUXP | F_PACK | F_ALIAS: fp32 heads holding pos - the
    env's OWN goal + residual rows, a goal plane and a bias plane beside them) against the same configuration in the full generic kernel
    on fp64 planes (GAQ_FORCE_GENERIC=1, itself pinned to the host build of the arithmetic header and to fixture G19 by
    tests/test_gpu_round2.py): goals bit-equal, observations / rewards within 1e-6 through resets (ten-step episodes), dones equal; the
    state read back through get_state (goal and bias planes included) agrees, and a state written with set_state -- other goals -- flies on
    from there like the generic handle's."""
    import torch
    from gym_art_amd import QuadrotorEnv
    label, kw, dmask = case
    n, steps = 2088, 45
    kw = dict(kw, num_envs=n, ep_time=0.1, seed=41, init_random_state=True, auto_reset=True)
    split = QuadrotorEnv(**kw)
    os.environ["GAQ_FORCE_GENERIC"] = "1"
    try:
        ref = QuadrotorEnv(**{k: v for k, v in kw.items()})
    finally:
        os.environ.pop("GAQ_FORCE_GENERIC", None)
    base = 131072 | 65536 | 1024 | 16 | 4 | (262144 if isinstance(kw.get("sense_noise"), dict) else 0)
    assert split.kernel_variant == base + dmask and split.state_layout == 2, (split.kernel_variant, split.state_layout)
    assert ref.kernel_variant & 8 and not ref.kernel_variant & 64, ref.kernel_variant
    o1, o2 = split.reset(), ref.reset()
    assert np.allclose(o1, o2, rtol=1e-6, atol=1e-6), label
    assert np.array_equal(np.asarray(split.goal), np.asarray(ref.goal))
    rng = np.random.RandomState(8)
    finished = 0
    for t in range(steps):
        a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
        (oa, ra, da, ia), (ob, rb, db, ib) = split.step(a), ref.step(a)
        assert np.array_equal(da, db), (label, t)
        err = np.abs(oa - ob) / np.maximum(np.abs(ob), 1.0)
        if "quat" in kw.get("obs_repr", ""):
            err = err[np.abs(ob[:, 6]) > 0.05]          # (R2quat divides by 4w: ill-conditioned near half-turns, tests/test_gpu_api_matrix.py)
        assert float(err.max()) <= 1e-6, (label, t, float(err.max()))
        assert float(np.max(np.abs(ra - rb))) <= 2e-5, (label, t)
        assert np.array_equal(np.asarray(split.goal), np.asarray(ref.goal)), (label, t)
        finished += int(da.sum())
    assert finished > n
    sa, sb = split.get_state(), ref.get_state()
    assert np.allclose(sa, sb, rtol=1e-6, atol=1e-6), label
    if kw.get("resample_goal") or kw.get("excite"):
        g = np.asarray(split.goal)
        assert g.shape == (n, 3) and np.unique(g[:, 2]).size > n // 4            # per-env goals indeed
    # a state written from outside: the generic handle's state with every goal moved
    st = sb.copy()
    moved = kw.get("resample_goal") or kw.get("excite")
    if moved:
        st[34:37, :] += rng.uniform(-0.2, 0.2, (3, n)).astype(np.float32)
    split.set_state(st); ref.set_state(st)
    assert np.allclose(split.get_state(), ref.get_state(), rtol=1e-6, atol=1e-6)
    for t in range(4):
        a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
        (oa, ra, da, _), (ob, rb, db, _) = split.step(a), ref.step(a)
        keep = np.ones(n, bool)
        if "quat" in kw.get("obs_repr", ""):
            keep = np.abs(ob[:, 6]) > 0.05
        assert np.array_equal(da, db) and np.allclose(oa[keep], ob[keep], rtol=1e-6, atol=1e-6) and np.allclose(ra, rb, atol=2e-5), (label, t)
    split.check_finite(); ref.check_finite()
    split.close(); ref.close()


# ---- gaq_step on small batches: the step launch reads / writes mapped host memory itself (no copies) -------------------------------------
@pytest.mark.parametrize("kw", [dict(), dict(alias_obs=True), dict(alias_obs=False), dict(raw_control=False, info=True),
                                dict(sense_noise="default", obs_repr="xyz_vxyz_R_omega_acc_act", info=True),
                                dict(dynamics_params="Crazyflie", dyn_sampler_1=dict(SAMPLER), dynamics_randomize_every=1)],
                         ids=["class_default", "alias", "fp64_planes", "mellinger_info", "sense_noise_acc_act_info", "per_env_rerandomised"])
def test_host_path_without_copies_equals_the_copying_one(kw):
    """gaq_step(host pointers) for batches whose staging block fits 1 MiB: actions are read from and observation / reward / done (and the
    info dict's state planes + aux rows) written into mapped host memory by the launches themselves.  GAQ_ZERO_COPY=0 (read when the
    handle first steps) keeps round 3's path -- one pinned copy in, one out: both have to return the same bits, through resets, for the
    single-env drop-in call shapes too."""
    from gym_art_amd import QuadrotorEnv
    for n in (1, 333):
        a_kw = dict(kw, num_envs=n, ep_time=0.05, seed=23)
        os.environ["GAQ_ZERO_COPY"] = "0"
        try:
            ref = QuadrotorEnv(**a_kw)
            o_ref = ref.reset()
            rng = np.random.RandomState(4)
            acts = [rng.uniform(-1, 1, (n, 4)).astype(np.float32) for _ in range(25)]
            out_ref = [ref.step(a[0] if n == 1 else a) for a in acts]
            st_ref = ref.get_state()
            ref.close()
        finally:
            os.environ.pop("GAQ_ZERO_COPY", None)
        env = QuadrotorEnv(**a_kw)
        assert np.array_equal(env.reset(), o_ref)
        for t, a in enumerate(acts):
            o, r, d, info = env.step(a[0] if n == 1 else a)
            o2, r2, d2, info2 = out_ref[t]
            assert np.array_equal(o, o2) and np.array_equal(r, r2) and np.array_equal(d, d2), (kw, n, t)
            assert sorted(info) == sorted(info2)
            for k in info:
                va, vb = info[k], info2[k]
                if isinstance(va, dict):
                    assert sorted(va) == sorted(vb) and all(np.array_equal(np.asarray(va[q]), np.asarray(vb[q])) for q in va), (k, t)
                else:
                    assert np.array_equal(np.asarray(va), np.asarray(vb)), (k, t)
        assert np.array_equal(env.get_state(), st_ref)
        env.close()


# ---- Mellinger on models the DEVICE samples and re-samples every episode (gaq.hip jinv_kernel) ------------------------------------------
@pytest.mark.parametrize("model,variant", [("Crazyflie", 16384 | 2048 | 16 | 2 | 1), ("DefaultQuad", 16384 | 2048 | 16 | 1)])
def test_mellinger_with_device_sampled_models_rebuilds_its_inverse_jacobians(model, variant):
    """The reference's test_rollout / benchmark() mode with -drr / -dre (quadrotor.py:1187-1230: the Mellinger controller on dynamics
    re-randomised every episode) as a batch: the device samples every env's model, the step launch promotes a finished env to its next
    draw, and the controller of that env needs the inverse jacobian of the NEW model from the next step on (quadrotor_control.py:290-291).
    The library rebuilds it on the device from the parameter planes the kernels fly with (jinv_kernel, after every launch that can
    have promoted).  Checked two ways: (a) against the same configuration in the full generic kernel (same draws, same pass) through
    five episodes; (b) against a twin on the HOST parameter pipeline -- whose inverse jacobians the host computes (gaq_set_params) -- given
    the device's parameters and state right after a round of promotions: the next steps agree."""
    from gym_art_amd import QuadrotorEnv, _lib
    n = 1500
    # (thrust noise off: the twin of (b) starts its step count -- the key of the noise draws -- at zero)
    kw = dict(num_envs=n, dynamics_params=model, dyn_sampler_1=dict(SAMPLER), raw_control=False, ep_time=0.1, seed=7, init_random_state=True,
              thrust_noise="off")
    env = QuadrotorEnv(dynamics_randomize_every=1, **kw)
    os.environ["GAQ_FORCE_GENERIC"] = "1"
    try:
        gen = QuadrotorEnv(dynamics_randomize_every=1, **kw)
    finally:
        os.environ.pop("GAQ_FORCE_GENERIC", None)
    assert env.kernel_variant == variant and env.state_layout == 2, env.kernel_variant
    assert gen.kernel_variant & 8 and gen.kernel_variant & 2048
    o1, o2 = env.reset(), gen.reset()
    assert np.allclose(o1, o2, rtol=1e-6, atol=1e-6)
    zero = np.zeros((n, 4), np.float32)
    first_models = {k: np.array(v) for k, v in env.models.items()}
    finished = 0
    for t in range(58):                                   # five episodes of eleven steps and a bit
        (oa, ra, da, _), (ob, rb, db, _) = env.step(zero), gen.step(zero)
        assert np.array_equal(da, db), t
        err = np.abs(oa - ob) / np.maximum(np.abs(ob), 1.0)
        assert float(err.max()) <= 1e-6 and float(np.max(np.abs(ra - rb))) <= 2e-5, (t, float(err.max()))
        finished += int(da.sum())
    assert finished >= 5 * n
    env.check_finite(); gen.check_finite()
    now_models = env.models
    assert np.all(now_models["mass"] != first_models["mass"])           # every env flies its sixth model
    # (b) the host pipeline's inverse jacobians for the device's parameters
    rows = np.empty((n, _lib.MODEL_DOUBLES), dtype=np.float64)
    _lib.check(env._lib.gaq_get_params(env._handle, _lib.ptr(rows), 0, n))
    twin = QuadrotorEnv(randomize_on_device=False, **kw)
    twin.reset()
    _lib.check(twin._lib.gaq_set_params(twin._handle, _lib.ptr(np.ascontiguousarray(rows)), 0, n))
    twin.set_state(env.get_state())
    for t in range(6):                                     # (stays inside the running episodes: 58 = 5 x 11 + 3)
        (oa, ra, da, _), (ob, rb, db, _) = env.step(zero), twin.step(zero)
        assert not da.any() and not db.any()
        assert np.allclose(oa, ob, rtol=1e-6, atol=1e-6) and np.allclose(ra, rb, atol=2e-5), t
    # the controllers do their job on the sampled models: nobody has left the room's middle after the sixth episode's first steps
    assert float(np.abs(oa[:, :3]).max()) < 6.0
    env.close(); gen.close(); twin.close()
