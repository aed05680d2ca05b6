"""Worker of test_c_abi_survives_random_configurations: throws random -- mostly invalid -- gaq_config records at gaq_create and, when a
handle comes back, drives it for a few calls.  Runs in its own process so that a crash of the library is a test failure, not the end of
the test run.  Prints one JSON line: counts by outcome."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gym_art_amd import _lib  # noqa: E402
from tests import gpu_util as G  # noqa: E402
from tests import golden_util as gu  # noqa: E402

seed, count = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.RandomState(seed)
lib = _lib.load()
const = dict(gu.sub(gu.load("g2_hummingbird_raw"), "const_"))
outcomes = {"ok": 0, "refused": 0, "stepped": 0}


def all_finite(struct):
    """every float / double field of a ctypes struct (nested structs and arrays included) is finite"""
    for name, typ in struct._fields_:
        v = getattr(struct, name)
        if isinstance(v, C.Structure):
            if not all_finite(v):
                return False
        elif isinstance(v, C.Array):
            if any(isinstance(x, float) and not np.isfinite(x) for x in v):
                return False
        elif isinstance(v, float) and not np.isfinite(v):
            return False
    return True


def weird_int(lo, hi):
    r = rng.rand()
    return int(rng.randint(lo, hi + 1)) if r < 0.95 else int(rng.choice([-1, 0, 2 ** 31 - 1, -2 ** 31, 65, 1000]))


def weird_float(lo, hi):
    r = rng.rand()
    return float(rng.uniform(lo, hi)) if r < 0.97 else float(rng.choice([0.0, -1.0, np.nan, np.inf, 1e-30, 1e30]))


for c in range(count):
    cfg = _lib.GaqConfig()
    cfg.struct_size = C.sizeof(cfg) if rng.rand() < 0.99 else int(rng.randint(0, 2000))
    cfg.abi_version = _lib.ABI_VERSION if rng.rand() < 0.99 else int(rng.randint(0, 10))
    cfg.num_envs = int(rng.choice([1, 2, 63, 64, 65, 130, 1000, 4096])) if rng.rand() < 0.97 else int(rng.choice([0, -5, (1 << 27) + 1]))
    cfg.env_id_offset = int(rng.choice([0, 0, 0, 8, 64, 1 << 40, -1, 3]))
    cfg.device = 0 if rng.rand() < 0.99 else int(rng.choice([-1, 7, 100]))
    cfg.seed = int(rng.randint(0, 1 << 62))
    cfg.sim_freq = weird_float(20.0, 1000.0)
    cfg.sim_steps = weird_int(1, 8)
    cfg.ep_len = weird_int(1, 700)
    cfg.room_size, cfg.gravity = weird_float(1.0, 20.0), 9.81
    cfg.t2w_std, cfg.t2t_std = 0.005, 0.0005
    cfg.control, cfg.noise, cfg.reward_mode = weird_int(0, 2), weird_int(0, 1), weird_int(0, 1)
    cfg.obs_flags = int(rng.choice([0, 0, 1, 2, 3, 4, 8, 12, 16, 17, 19, 32, 33, 96, int(rng.randint(0, 128))])) if rng.rand() < 0.97 else int(rng.choice([-1, 255, 1 << 20]))
    cfg.auto_reset, cfg.init_random_state, cfg.resample_goal = rng.randint(2), rng.randint(2), rng.randint(2)
    cfg.per_env_params, cfg.compact_done = rng.randint(2), rng.randint(2)
    cfg.obs_state_alias = weird_int(0, 2)
    cfg.fp32_state, cfg.excite, cfg.aux_outputs, cfg.action_f32, cfg.sense_input = (int(rng.rand() < 0.2) for _ in range(5))
    if rng.rand() < 0.15:
        cfg.swarm.agents = int(rng.choice([0, 1, 2, 3, 4, 8, 16, 32, 64, -2]))
        cfg.swarm.goal_radius, cfg.swarm.collision_dist, cfg.swarm.prox_dist = weird_float(0, 2), weird_float(0, 1), weird_float(0, 2)
        cfg.swarm.w_collision, cfg.swarm.w_prox = weird_float(0, 2), weird_float(0, 2)
    for k, _t in _lib.GaqRewCoeff._fields_:
        setattr(cfg.rew, k, weird_float(0.0, 1.0) if rng.rand() < 0.5 else 0.0)
    if rng.rand() < 0.4:
        cfg.sense.enabled = 1
        for k, _t in _lib.GaqSenseNoise._fields_[1:]:
            setattr(cfg.sense, k, weird_float(0.0, 0.05))
    cfg.model = _lib.row_to_model(G.model_row(const))
    if rng.rand() < 0.2:
        cfg.model.mass = weird_float(0.1, 2.0)
        cfg.model.inertia[int(rng.randint(3))] = weird_float(1e-4, 1e-2)
        cfg.model.damp_time_up = weird_float(0.0, 0.3)
    h = C.c_void_p()
    rc = lib.gaq_create(C.byref(cfg), C.byref(h))
    if rc != 0:
        assert not h.value, "an error must not hand back a handle"
        assert lib.gaq_last_error(), "an error must leave a message"
        outcomes["refused"] += 1
        continue
    outcomes["ok"] += 1
    n, D = int(cfg.num_envs), lib.gaq_obs_dim(h)
    assert 13 <= D <= 18 + 8 + 6 * 63, D
    obs = np.empty((n, D), np.float32); rew = np.empty(n, np.float32); done = np.empty(n, np.uint8)
    r1 = lib.gaq_reset(h, None, _lib.ptr(obs))
    if r1 == 0:
        for t in range(3):
            a = rng.uniform(-1.5, 1.5, (n, 4)).astype(np.float32)
            r2 = lib.gaq_step(h, _lib.ptr(a), _lib.ptr(obs), _lib.ptr(rew), _lib.ptr(done))
            if r2 != 0:
                break
        else:
            outcomes["stepped"] += 1
    st = np.empty((_lib.STATE_PLANES, n))
    lib.gaq_get_state(h, _lib.ptr(st))
    if rng.rand() < 0.5:    # garbage in the state planes: NaN, inf, huge values, rotation blocks that are no rotations, odd counters
        bad = st.copy()
        for _ in range(int(rng.randint(1, 6))):
            pl, col = int(rng.randint(0, 26)), int(rng.randint(0, n))      # pos, vel, R, omega, motor filter: what a reset re-creates
            bad[pl, col] = rng.choice([np.nan, np.inf, -np.inf, 1e300, -1e30, 1e-300, 0.0, 12345.678])
        if rng.rand() < 0.3:
            bad[6:15] = rng.normal(size=(9, n)) * rng.choice([1e-3, 1.0, 1e6])
        if rng.rand() < 0.3:
            bad[37] = rng.choice([0, 1, 65535, 70000, -3, 0.5])
        rs = lib.gaq_set_state(h, _lib.ptr(np.ascontiguousarray(bad)))
        outcomes["garbage_state_refused" if rs else "garbage_state_taken"] = outcomes.get("garbage_state_refused" if rs else "garbage_state_taken", 0) + 1
        for t in range(2):
            a = rng.uniform(-1.5, 1.5, (n, 4)).astype(np.float32)
            if rng.rand() < 0.2:
                a[int(rng.randint(0, n)), int(rng.randint(0, 4))] = rng.choice([np.nan, np.inf, 1e30])
            lib.gaq_step(h, _lib.ptr(a), _lib.ptr(obs), _lib.ptr(rew), _lib.ptr(done))     # GAQ_OK or GAQ_ERR_NAN: both fine, a crash is not
        lib.gaq_reset(h, None, _lib.ptr(obs))
        r3 = lib.gaq_step(h, _lib.ptr(np.zeros((n, 4), np.float32)), _lib.ptr(obs), _lib.ptr(rew), _lib.ptr(done))
        if r1 == 0 and r3 == 0 and cfg.noise != 2 and not cfg.sense_input and all_finite(cfg):
            assert np.isfinite(obs).all() and np.isfinite(rew).all(), "a reset has to bring a poisoned env back"

    cnt = C.c_int64(0)
    lib.gaq_nan_count(h, C.byref(cnt))
    assert lib.gaq_destroy(h) == 0
print(json.dumps(outcomes))
