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
