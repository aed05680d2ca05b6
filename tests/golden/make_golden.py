#!/usr/bin/env python
"""Generate the golden fixtures in this directory from the UNMODIFIED reference.

Container-only: needs /root/reference (never present on the GPU box) and the
import shims in ref_shims.py.  Run as

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden.py

Every scenario drives the reference's own `QuadrotorEnv.step` /
`QuadrotorDynamics.step1` (gym_art/quadrotor/quadrotor.py:942-1028, :273-436)
with scripted, fp32-representable inputs (actions, initial states) and records
inputs and outputs.  Only data is written (npz); no reference source travels.

Scenario ids follow SURVEY.md §8c (G1..G9).
"""
import copy
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()

import gym_art.quadrotor.quadrotor as refq  # noqa: E402
import gym_art.quadrotor.quad_utils as ref_utils  # noqa: E402
import gym_art.quadrotor.quadrotor_randomization as ref_rand  # noqa: E402

NOISE_OFF = {"noise": {"thrust_noise_ratio": 0.0}}


def f32(x):
    """Round to fp32-representable doubles (so GPU and reference start bit-identical)."""
    return np.asarray(x, dtype=np.float32).astype(np.float64)


def rand_rot(rng):
    """Uniform random rotation (QR of a Gaussian), entries then fp32-rounded."""
    q, r = np.linalg.qr(rng.normal(size=(3, 3)))
    q = q * np.sign(np.diag(r))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return f32(q)


def yaw_rot(psi):
    c, s = np.cos(psi), np.sin(psi)
    return f32(np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]]))


def derived_constants(dyn):
    """The constants QuadrotorDynamics.update_model derives (quadrotor.py:142-208)."""
    return dict(
        mass=np.float64(dyn.mass),
        inertia=np.array(dyn.inertia, dtype=np.float64),
        thrust_max=np.array(dyn.thrust_max, dtype=np.float64),
        torque_max=np.array(dyn.torque_max, dtype=np.float64),
        prop_pos=np.array(dyn.prop_pos, dtype=np.float64),
        prop_crossproducts=np.array(dyn.prop_crossproducts, dtype=np.float64),
        motor_assymetry=np.array(dyn.motor_assymetry, dtype=np.float64),
        arm=np.float64(dyn.arm),
        motor_linearity=np.float64(dyn.motor_linearity),
        damp_time_up=np.float64(dyn.motor_damp_time_up),
        damp_time_down=np.float64(dyn.motor_damp_time_down),
        thrust_noise_sigma=np.float64(dyn.thrust_noise.sigma),
        vel_damp=np.float64(dyn.vel_damp),
        damp_omega_quadratic=np.float64(dyn.damp_omega_quadratic),
        C_rot_drag=np.float64(dyn.C_rot_drag),
        C_rot_roll=np.float64(dyn.C_rot_roll),
        thrust_to_weight=np.float64(dyn.thrust_to_weight),
        torque_to_thrust=np.float64(dyn.torque_to_thrust),
        torque_to_inertia=np.array(dyn.torque_to_inertia, dtype=np.float64),
        com=np.array(dyn.model.com, dtype=np.float64),
    )


def flatten_params(p, prefix=""):
    out = {}
    for k, v in p.items():
        if isinstance(v, dict):
            out.update(flatten_params(v, prefix + k + "."))
        else:
            out[prefix + k] = np.array(v, dtype=np.float64)
    return out


class NormalRecorder(object):
    """Proxy for the `nr` (numpy.random) module object used by OUNoise.noise
    (quad_utils.py:197-201); records every randn draw."""

    def __init__(self, rng):
        self.rng = rng
        self.draws = []

    def randn(self, n):
        x = self.rng.randn(n)
        self.draws.append(x.copy())
        return x


def make_env(module=refq, **kw):
    kw.setdefault("dynamics_params", "DefaultQuad")
    kw.setdefault("sim_freq", 200.0)
    kw.setdefault("sim_steps", 2)
    kw.setdefault("ep_time", 5)
    env = module.QuadrotorEnv(**kw)
    return env


def set_state(env, pos, vel, rot, omega, svd=0.0, tick=0):
    d = env.dynamics
    d.set_state(np.array(pos, dtype=np.float64), np.array(vel, dtype=np.float64),
                np.array(rot, dtype=np.float64), np.array(omega, dtype=np.float64))
    d.reset()
    d.since_last_svd = svd
    env.crashed = False
    env.tick = tick
    env.actions = [np.zeros([4, ]), np.zeros([4, ])]


def rollout(env, actions, record_ctrl=False, as_f32=False, record_info=False):
    """Step the reference env through `actions` [T,4]; record everything per step.  `as_f32`: hand the reference
    float32 arrays (what action_space.sample() and most policies produce) instead of float64 ones.  `record_info`: keep the
    numeric entries of info["obs_comp"] / info["dyn_params"] (quadrotor.py:994-1025)."""
    T = actions.shape[0]
    d = env.dynamics
    rec = {k: [] for k in ("obs", "reward", "done", "crashed", "pos", "vel", "rot", "omega",
                           "thrust_rot_damp", "thrust_cmds_damp", "accelerometer", "omega_dot",
                           "torque", "since_last_svd", "rew_raw", "ctrl")}
    raw_keys = ["rewraw_pos", "rewraw_action", "rewraw_crash", "rewraw_orient", "rewraw_yaw",
                "rewraw_rot", "rewraw_attitude", "rewraw_spin", "rewraw_act_change", "rewraw_vel"]
    for t in range(T):
        obs, rew, done, info = env.step(actions[t].astype(np.float32) if as_f32 else actions[t].copy())
        if record_info:
            for grp in ("obs_comp", "dyn_params"):
                if grp == "dyn_params" and t > 0:       # constants of the model: the first step's copy is enough
                    continue
                for k, v in info[grp].items():
                    rec.setdefault("info_%s_%s" % (grp, k), []).append(np.array(v[0], dtype=np.float64))
        rec["obs"].append(np.array(obs, dtype=np.float64))
        rec["reward"].append(float(rew))
        rec["done"].append(bool(done))
        rec["crashed"].append(bool(env.crashed))
        rec["pos"].append(d.pos.copy())
        rec["vel"].append(d.vel.copy())
        rec["rot"].append(d.rot.copy())
        rec["omega"].append(np.array(d.omega, dtype=np.float64))
        rec["thrust_rot_damp"].append(d.thrust_rot_damp.copy())
        rec["thrust_cmds_damp"].append(d.thrust_cmds_damp.copy())
        rec["accelerometer"].append(np.array(d.accelerometer, dtype=np.float64))
        rec["omega_dot"].append(np.array(d.omega_dot, dtype=np.float64))
        rec["torque"].append(np.array(d.torque, dtype=np.float64))
        rec["since_last_svd"].append(float(d.since_last_svd))
        rr = info["rewards"]
        if "rewraw_pos" in rr:
            rec["rew_raw"].append(np.array([-rr[k] for k in raw_keys], dtype=np.float64))
        if record_ctrl:
            rec["ctrl"].append(np.array(env.controller.action, dtype=np.float64))
    out = {}
    for k, v in rec.items():
        if len(v):
            out[k] = np.array(v)
    return out


def init_block(env, pos, vel, rot, omega, svd=0.0):
    return dict(init_pos=np.array(pos, dtype=np.float64), init_vel=np.array(vel, dtype=np.float64),
                init_rot=np.array(rot, dtype=np.float64), init_omega=np.array(omega, dtype=np.float64),
                init_svd=np.float64(svd), goal=np.array(env.goal, dtype=np.float64),
                dt=np.float64(env.dt), sim_steps=np.int64(env.sim_steps), ep_len=np.int64(env.ep_len))


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %-28s %7.1f kB" % (name + ".npz", os.path.getsize(path) / 1e3))


def pack(prefix, d):
    return {prefix + k: v for k, v in d.items()}


def random_init(rng, goal, vel_scale=0.0, omega_scale=0.0, full_rot=False):
    pos = f32(rng.uniform(-2.0, 2.0, size=3) + goal)
    pos[2] = max(pos[2], 0.25)
    vel = f32(vel_scale * rng.uniform(-1, 1, size=3))
    omega = f32(omega_scale * rng.uniform(-1, 1, size=3))
    rot = rand_rot(rng) if full_rot else yaw_rot(rng.uniform(-np.pi, np.pi))
    return pos, vel, rot, omega


# --------------------------------------------------------------------------- scenarios
def g9_kat():
    """Single-step known-answer test."""
    env = make_env(dynamics_change=NOISE_OFF)
    pos, vel, rot, omega = [0.5, -0.25, 1.5], [0.1, 0.2, -0.3], yaw_rot(0.3), [0.4, -0.5, 0.6]
    pos, vel, omega = f32(pos), f32(vel), f32(omega)
    set_state(env, pos, vel, rot, omega)
    act = f32([[0.25, -0.5, 0.75, 0.0]])
    arrays = init_block(env, pos, vel, rot, omega)
    arrays.update(rollout(env, act))
    arrays["actions"] = act
    arrays.update(pack("const_", derived_constants(env.dynamics)))
    save("g9_kat", **arrays)


def g1_mellinger():
    """C1: Hummingbird + Mellinger controller, one full episode (501 steps), noise off."""
    env = make_env(dynamics_change=NOISE_OFF, raw_control=False, tf_control=False)
    rng = np.random.RandomState(101)
    arrays = {}
    for i in range(2):
        pos, vel, rot, omega = random_init(rng, env.goal)
        set_state(env, pos, vel, rot, omega)
        T = env.ep_len + 1
        act = np.zeros((T, 4))
        blk = init_block(env, pos, vel, rot, omega)
        blk.update(rollout(env, act, record_ctrl=True))
        arrays.update(pack("e%d_" % i, blk))
    arrays["Jinv"] = np.array(env.controller.Jinv, dtype=np.float64)
    arrays.update(pack("const_", derived_constants(env.dynamics)))
    save("g1_mellinger", **arrays)


def g1b_mellinger_other_models():
    """The Mellinger controller on the other shipped models (CrazyFlie: motor lag; MediumQuad), one episode each."""
    arrays = {}
    rng = np.random.RandomState(111)
    for i, model in enumerate(("Crazyflie", "MediumQuad")):
        env = make_env(dynamics_params=model, dynamics_change=NOISE_OFF, raw_control=False, tf_control=False)
        pos, vel, rot, omega = random_init(rng, env.goal, vel_scale=0.3)
        set_state(env, pos, vel, rot, omega)
        T = env.ep_len + 1
        blk = init_block(env, pos, vel, rot, omega)
        blk.update(rollout(env, np.zeros((T, 4)), record_ctrl=True))
        blk["Jinv"] = np.array(env.controller.Jinv, dtype=np.float64)
        blk.update(pack("const_", derived_constants(env.dynamics)))
        arrays.update(pack("e%d_" % i, blk))
    arrays["n_envs"] = np.int64(2)
    save("g1b_mellinger_other_models", **arrays)


def g2_hummingbird_raw():
    """C2 numerics: Hummingbird RawControl, random fp32 actions, 3 action scales, 500 steps."""
    env = make_env(dynamics_change=NOISE_OFF)
    rng = np.random.RandomState(202)
    arrays = {}
    i = 0
    for scale in (1.0, 0.3, 0.05):
        for rep in range(2):
            pos, vel, rot, omega = random_init(rng, env.goal, vel_scale=0.5 * rep, omega_scale=1.0 * rep,
                                               full_rot=bool(rep))
            set_state(env, pos, vel, rot, omega)
            act = f32(scale * rng.uniform(-1, 1, size=(500, 4)))
            blk = init_block(env, pos, vel, rot, omega)
            blk.update(rollout(env, act))
            blk["actions"] = act
            blk["scale"] = np.float64(scale)
            arrays.update(pack("e%d_" % i, blk))
            i += 1
    arrays["n_envs"] = np.int64(i)
    arrays.update(pack("const_", derived_constants(env.dynamics)))
    save("g2_hummingbird_raw", **arrays)


def g2b_episode_boundary():
    """done/tick semantics and SVD-counter persistence over 2 short episodes
    (quadrotor.py:986-987, :438-440): ep_time=0.5 -> ep_len=50, 51 steps/episode."""
    env = make_env(dynamics_change=NOISE_OFF, ep_time=0.5)
    rng = np.random.RandomState(212)
    arrays = {}
    svd = 0.0
    for ep in range(3):
        pos, vel, rot, omega = random_init(rng, env.goal)
        set_state(env, pos, vel, rot, omega, svd=svd)
        T = env.ep_len + 3  # step past done: the reference keeps returning done=True
        act = f32(rng.uniform(-1, 1, size=(T, 4)))
        blk = init_block(env, pos, vel, rot, omega, svd=svd)
        blk.update(rollout(env, act))
        blk["actions"] = act
        arrays.update(pack("e%d_" % ep, blk))
        svd = env.dynamics.since_last_svd  # counter survives reset()
    arrays["n_envs"] = np.int64(3)
    arrays.update(pack("const_", derived_constants(env.dynamics)))
    save("g2b_episode_boundary", **arrays)


def g3_crazyflie():
    """C3 base model: CrazyFlie (motor lag tau=0.1333, up/down switch), 500 steps."""
    env = make_env(dynamics_params="Crazyflie", dynamics_change=NOISE_OFF)
    rng = np.random.RandomState(303)
    arrays = {}
    i = 0
    for scale in (1.0, 0.2):
        pos, vel, rot, omega = random_init(rng, env.goal, full_rot=(scale < 1))
        set_state(env, pos, vel, rot, omega)
        act = f32(scale * rng.uniform(-1, 1, size=(500, 4)))
        blk = init_block(env, pos, vel, rot, omega)
        blk.update(rollout(env, act))
        blk["actions"] = act
        arrays.update(pack("e%d_" % i, blk))
        i += 1
    arrays["n_envs"] = np.int64(i)
    arrays.update(pack("const_", derived_constants(env.dynamics)))
    save("g3_crazyflie", **arrays)


def g3b_asym_lag():
    """Different up/down motor time constants + linearity<1 + MediumQuad."""
    chg = {"noise": {"thrust_noise_ratio": 0.0},
           "motor": {"damp_time_up": 0.15, "damp_time_down": 0.3, "linearity": 0.424}}
    arrays = {}
    rng = np.random.RandomState(313)
    for i, model in enumerate(("Crazyflie", "MediumQuad")):
        env = make_env(dynamics_params=model, dynamics_change=chg)
        pos, vel, rot, omega = random_init(rng, env.goal)
        set_state(env, pos, vel, rot, omega)
        act = f32(rng.uniform(-1, 1, size=(300, 4)))
        blk = init_block(env, pos, vel, rot, omega)
        blk.update(rollout(env, act))
        blk["actions"] = act
        blk.update(pack("const_", derived_constants(env.dynamics)))
        arrays.update(pack("e%d_" % i, blk))
    arrays["n_envs"] = np.int64(2)
    save("g3b_asym_lag", **arrays)


def g4_randomized():
    """C3: CrazyFlie + RelativeSampler(noise_ratio=0.2) x 32 parameter sets:
    sampled parameter dict -> derived constants (a2) and 120-step trajectories."""
    np.random.seed(404)
    rng = np.random.RandomState(405)
    sampler = {"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"}
    env = make_env(dynamics_params="Crazyflie", dynamics_change=NOISE_OFF, dyn_sampler_1=sampler)
    arrays = {}
    n = 32
    for i in range(n):
        env.resample_dynamics()
        prm = copy.deepcopy(env.dynamics_params)
        pos, vel, rot, omega = random_init(rng, env.goal, full_rot=(i % 2 == 1))
        set_state(env, pos, vel, rot, omega)
        T = 500 if i < 4 else 120
        act = f32(rng.uniform(-1, 1, size=(T, 4)))
        blk = init_block(env, pos, vel, rot, omega)
        r = rollout(env, act)
        for k in ("obs", "reward", "done", "crashed", "thrust_cmds_damp", "thrust_rot_damp"):
            blk[k] = r[k]
        blk["actions"] = act
        blk.update(pack("const_", derived_constants(env.dynamics)))
        blk.update(pack("param_", flatten_params(prm)))
        arrays.update(pack("e%d_" % i, blk))
    arrays["n_envs"] = np.int64(n)
    save("g4_randomized", **arrays)


def g4b_models():
    """Derived constants (a2) of every shipped model + RandomQuad samples
    (quad_models.py; quadrotor_randomization.py:142-243; inertia.py:182-310)."""
    arrays = {}
    for name in ("DefaultQuad", "Crazyflie", "MediumQuad"):
        env = make_env(dynamics_params=name)
        arrays.update(pack(name + "_const_", derived_constants(env.dynamics)))
        arrays.update(pack(name + "_param_", flatten_params(env.dynamics_params)))
    np.random.seed(414)
    env = make_env(dynamics_params="RandomQuad")
    n = 16
    for i in range(n):
        env.resample_dynamics()
        arrays.update(pack("rq%d_const_" % i, derived_constants(env.dynamics)))
        arrays.update(pack("rq%d_param_" % i, flatten_params(env.dynamics_params)))
    arrays["n_random"] = np.int64(n)
    save("g4b_models", **arrays)


def g4c_simplified():
    """dynamics_simplification=True: QuadLinkSimplified (inertia.py:312-440) -- two rods carrying the whole mass,
    massless propellers -- for every shipped model, plus one short trajectory."""
    arrays = {}
    for name in ("DefaultQuad", "Crazyflie", "MediumQuad"):
        env = make_env(dynamics_params=name, dynamics_simplification=True)
        arrays.update(pack(name + "_const_", derived_constants(env.dynamics)))
        arrays.update(pack(name + "_param_", flatten_params(env.dynamics_params)))
    # RandomQuad cannot be simplified in the reference: its arms carry no "l" and BoxLink(**arms) raises TypeError
    try:
        make_env(dynamics_params="RandomQuad", dynamics_simplification=True)
        arrays["randomquad_raises"] = np.array("")
    except TypeError as e:
        arrays["randomquad_raises"] = np.array(str(e))
    rng = np.random.RandomState(434)
    env = make_env(dynamics_params="Crazyflie", dynamics_simplification=True, dynamics_change=NOISE_OFF)
    pos, vel, rot, omega = random_init(rng, env.goal, vel_scale=0.3, omega_scale=0.5)
    set_state(env, pos, vel, rot, omega)
    act = f32(0.4 * rng.uniform(-1, 1, size=(200, 4)))
    blk = init_block(env, pos, vel, rot, omega)
    blk.update(rollout(env, act))
    blk["actions"] = act
    blk.update(pack("const_", derived_constants(env.dynamics)))
    arrays.update(pack("e0_", blk))
    arrays["n_envs"] = np.int64(1)
    save("g4c_simplified", **arrays)


def g5_drag_damp():
    """a6 branch: C_drag, C_roll != 0, plus vel / omega_quadratic damping."""
    chg = {"noise": {"thrust_noise_ratio": 0.0},
           "motor": {"C_drag": 0.0028, "C_roll": 0.003},
           "damp": {"vel": 0.001, "omega_quadratic": 0.015}}
    arrays = {}
    rng = np.random.RandomState(505)
    for i, model in enumerate(("DefaultQuad", "Crazyflie")):
        env = make_env(dynamics_params=model, dynamics_change=chg)
        pos, vel, rot, omega = random_init(rng, env.goal, vel_scale=1.0, omega_scale=2.0, full_rot=True)
        set_state(env, pos, vel, rot, omega)
        act = f32((0.3 + 0.7 * i) * rng.uniform(-1, 1, size=(400, 4)))
        blk = init_block(env, pos, vel, rot, omega)
        blk.update(rollout(env, act))
        blk["actions"] = act
        blk.update(pack("const_", derived_constants(env.dynamics)))
        arrays.update(pack("e%d_" % i, blk))
    arrays["n_envs"] = np.int64(2)
    save("g5_drag_damp", **arrays)


def g6_noise_injected():
    """Thrust noise ON (default ratio 0.05 -> OU sigma 0.01); the 4 normals drawn per
    sub-step are recorded so that the device path can be fed the identical noise."""
    arrays = {}
    rng = np.random.RandomState(606)
    for i, model in enumerate(("DefaultQuad", "Crazyflie")):
        env = make_env(dynamics_params=model)
        recorder = NormalRecorder(np.random.RandomState(616 + i))
        saved = ref_utils.nr
        ref_utils.nr = recorder
        try:
            pos, vel, rot, omega = random_init(rng, env.goal)
            set_state(env, pos, vel, rot, omega)
            env.dynamics.thrust_noise.reset()
            act = f32(0.5 * rng.uniform(-1, 1, size=(300, 4)))
            blk = init_block(env, pos, vel, rot, omega)
            blk.update(rollout(env, act))
        finally:
            ref_utils.nr = saved
        blk["actions"] = act
        blk["normals"] = np.array(recorder.draws).reshape(300, env.sim_steps, 4)
        blk["ou_state_final"] = np.array(env.dynamics.thrust_noise.state, dtype=np.float64)
        blk.update(pack("const_", derived_constants(env.dynamics)))
        arrays.update(pack("e%d_" % i, blk))
    arrays["n_envs"] = np.int64(2)
    save("g6_noise_injected", **arrays)


def g11_other_rates():
    """Other integration rates than the default 200 Hz x 2: (100 Hz, 4 sub-steps), (400 Hz, 1), (250 Hz, 3) -- the
    re-orthonormalisation period (`since_last_svd += dt` against 0.5 s, quadrotor.py:381-386), the motor time constants
    4 dt / T and the episode length all depend on dt.  Per-block constants (one model each)."""
    arrays = {}
    rng = np.random.RandomState(1111)
    cases = [("DefaultQuad", 100.0, 4, 3), ("Crazyflie", 400.0, 1, 2), ("Crazyflie", 250.0, 3, 2), ("DefaultQuad", 50.0, 1, 4)]
    for i, (model, freq, steps, ep_time) in enumerate(cases):
        env = make_env(dynamics_params=model, dynamics_change=NOISE_OFF, sim_freq=freq, sim_steps=steps, ep_time=ep_time)
        pos, vel, rot, omega = random_init(rng, env.goal, vel_scale=0.5, omega_scale=1.0, full_rot=(i % 2 == 1))
        set_state(env, pos, vel, rot, omega)
        T = env.ep_len + 1
        act = f32(0.7 * rng.uniform(-1, 1, size=(T, 4)))
        blk = init_block(env, pos, vel, rot, omega)
        blk.update(rollout(env, act))
        blk["actions"] = act
        blk.update(pack("const_", derived_constants(env.dynamics)))
        arrays.update(pack("e%d_" % i, blk))
    arrays["n_envs"] = np.int64(len(cases))
    save("g11_other_rates", **arrays)


def g12_edge_cases():
    """Corner states with a constant action for 40 steps: actions far outside [-1,1], floor / room-corner starts, omega at
    the clip, omega exactly zero, upside down, the crash height; and a 1-step episode (ep_len = 0)."""
    Rx180 = np.diag([1.0, -1.0, -1.0])
    cases = [
        ([0, 0, 2], [0, 0, 0], np.eye(3), [0, 0, 0], [5, -5, 3, -0.2]),
        ([0, 0, 0.0], [0, 0, -3], np.eye(3), [0, 0, 0], [-1, -1, -1, -1]),
        ([10, -10, 10], [4, -4, 4], np.eye(3), [1, 2, 3], [1, 1, 1, 1]),
        ([1, 1, 3], [0, 0, 0], np.eye(3), [40, -40, 40], [1, -1, 1, -1]),
        ([1, 1, 3], [0, 0, 0], np.eye(3), [0, 0, 0], [0, 0, 0, 0]),
        ([-2, 2, 5], [1, 0, 0], Rx180, [0.5, 0, 0], [0.3, 0.3, 0.3, 0.3]),
        ([0, 0, 0.1697], [0, 0, 0], np.eye(3), [0, 0, 0], [-0.2, -0.2, -0.2, -0.2]),
    ]
    arrays = {}
    i = 0
    for ep_time, T in ((5, 40), (0.005, 1)):
        env = make_env(dynamics_change=NOISE_OFF, ep_time=ep_time)
        for pos, vel, rot, omega, a in cases:
            pos, vel, omega = f32(pos), f32(vel), f32(omega)
            set_state(env, pos, vel, rot, omega)
            act = np.tile(f32(a), (T, 1))
            blk = init_block(env, pos, vel, rot, omega)
            blk.update(rollout(env, act))
            blk["actions"] = act
            arrays.update(pack("e%d_" % i, blk))
            i += 1
    # a smaller room (room_size = 4, quadrotor.py:723): the quad runs into every wall and the ceiling
    env = make_env(dynamics_change=NOISE_OFF, room_size=4)
    rng = np.random.RandomState(1212)
    for k in range(3):
        pos, vel, rot, omega = random_init(rng, env.goal, vel_scale=3.0, omega_scale=1.0)
        pos = f32(np.clip(pos, [-4, -4, 0], [4, 4, 4]))
        set_state(env, pos, vel, rot, omega)
        act = f32(rng.uniform(0.2, 1.0, size=(200, 4)))
        blk = init_block(env, pos, vel, rot, omega)
        blk.update(rollout(env, act))
        blk["actions"] = act
        blk["room_size"] = np.float64(4.0)
        arrays.update(pack("e%d_" % i, blk))
        i += 1
    arrays["n_envs"] = np.int64(i)
    arrays.update(pack("const_", derived_constants(env.dynamics)))
    save("g12_edge_cases", **arrays)


class SenseDrawRecorder(object):
    """Stands in for numpy.random.normal / uniform inside sensor_noise.py (imported there by name, :3-4):
    same arithmetic as numpy's (loc + scale*z, low + (high-low)*u) on a private stream, recording the standard
    draws z / u of every call so that the oracle can be fed the identical sequence."""

    def __init__(self, rng):
        self.rng, self.calls = rng, []

    def normal(self, loc=0., scale=1., size=None):
        z = self.rng.standard_normal(size)
        self.calls.append(("n", np.array(z, dtype=np.float64)))
        return loc + scale * z

    def uniform(self, low=0., high=1., size=None):
        u = self.rng.random_sample(size)
        self.calls.append(("u", np.array(u, dtype=np.float64)))
        return low + (high - low) * u

    def pop_add_noise_calls(self, bias_model):
        """Group the recorded draws into add_noise calls -> [calls, 10, 3] in the oracle's slot order."""
        per = 10 if bias_model else 9
        assert len(self.calls) % per == 0
        out = []
        for c in range(len(self.calls) // per):
            d = [v for _, v in self.calls[c * per:(c + 1) * per]]
            kinds = "".join(k for k, _ in self.calls[c * per:(c + 1) * per])
            z = np.zeros((10, 3))
            if bias_model:      # pos n,u  vel n,u  bias n, white n, quat n,u, acc n,n
                assert kinds == "nununnnunn"
                z[:] = d
            else:               # pos n,u  vel n,u  gyro n, quat n,u, acc n,n
                assert kinds == "nununnunn"
                z[0:5] = d[0:5]; z[6:10] = d[5:9]
            out.append(z)
        self.calls = []
        return np.array(out)


def g10_sense_noise():
    """SensorNoise on the observation path (sensor_noise.py:100-170), default gyro model and the gyro-bias
    random walk, with every draw recorded.  Three add_noise calls per env.step (quadrotor.py:946, :966-970, :988)
    and one per reset (:1143)."""
    cases = [("default", "xyz_vxyz_R_omega_acc_act", False),
             ({"gyro_norm_std": 0.01, "quat_norm_std": 0.01, "quat_unif_range": 0.005, "pos_unif_range": 0.02,
               "vel_unif_range": 0.01, "gyro_bias_correlation_time": 50.0, "gyro_noise_density": 0.002}, "xyz_vxyz_R_omega", True),
             ({"gyro_norm_std": 1.0}, "xyzr_vxyzr_R_omega_h", True)]
    _sense_noise_cases("g10_sense_noise", cases, 1010, 60)


def g16_sense_noise_param_sets():
    """Ten random SensorNoise parameter sets (any subset of the Gaussian / uniform position, velocity and attitude terms, both gyro
    models with different correlation times, accelerometer terms) over the six working observation variants, every draw recorded:
    pins the oracle's add_noise between G10's three sets."""
    rng = np.random.RandomState(1600)
    reprs = ["xyz_vxyz_R_omega", "xyz_vxyz_R_omega_h", "xyzr_vxyzr_R_omega", "xyzr_vxyzr_R_omega_h", "xyz_vxyz_R_omega_acc_act",
             "xyz_vxyz_R_omega_act"]
    cases = []
    for c in range(10):
        sn = {}
        for k, hi in (("pos_norm_std", 0.02), ("pos_unif_range", 0.02), ("vel_norm_std", 0.05), ("vel_unif_range", 0.05),
                      ("quat_norm_std", 0.03), ("quat_unif_range", 0.02), ("gyro_noise_density", 0.002), ("acc_static_noise_std", 0.01),
                      ("acc_dynamic_noise_ratio", 0.02), ("gyro_random_walk", 0.02)):
            sn[k] = float(rng.uniform(0, hi)) if rng.rand() < 0.6 else 0.0
        walk = bool(c % 2)
        sn["gyro_norm_std"] = float(rng.uniform(0.001, 0.02)) if walk else 0.0
        sn["gyro_bias_correlation_time"] = float(rng.choice([0.5, 10.0, 1000.0]))
        cases.append((sn, reprs[c % len(reprs)], walk))
    _sense_noise_cases("g16_sense_noise_param_sets", cases, 1610, 15)


def _sense_noise_cases(name, cases, seed0, T):
    import gym_art.quadrotor.sensor_noise as ref_sn
    arrays = {}
    rng = np.random.RandomState(seed0)
    saved = (ref_sn.normal, ref_sn.uniform)
    for i, (sn, obs_repr, bias_model) in enumerate(cases):
        rec = SenseDrawRecorder(np.random.RandomState(seed0 + 10 + i))
        ref_sn.normal, ref_sn.uniform = rec.normal, rec.uniform
        try:
            env = make_env(dynamics_change=NOISE_OFF, sense_noise=sn, obs_repr=obs_repr)
            ctor_draws = rec.pop_add_noise_calls(bias_model)     # __init__ ends with a _reset(): one add_noise call
            assert ctor_draws.shape[0] == 1
            ctor_bias = np.array(env.sense_noise.gyro_bias, dtype=np.float64)
            np.random.seed(seed0 + 20 + i)
            obs_reset = np.array(env.reset(), dtype=np.float64)
            blk = {"ctor_gyro_bias": ctor_bias, "reset_obs": obs_reset, "reset_draws": rec.pop_add_noise_calls(bias_model),
                   "reset_pos": env.dynamics.pos.copy(), "reset_vel": env.dynamics.vel.copy(),
                   "reset_rot": env.dynamics.rot.copy(), "reset_omega": np.array(env.dynamics.omega, dtype=np.float64),
                   "reset_gyro_bias": np.array(env.sense_noise.gyro_bias, dtype=np.float64)}
            pos, vel, rot, omega = random_init(rng, env.goal, vel_scale=0.5, omega_scale=1.0, full_rot=(i % 2 == 1))
            set_state(env, pos, vel, rot, omega)
            act = f32(0.6 * rng.uniform(-1, 1, size=(T, 4)))
            blk.update(init_block(env, pos, vel, rot, omega))
            bias = []
            roll = {}
            for t in range(T):
                r1 = rollout(env, act[t:t + 1])
                for k, v in r1.items():
                    roll.setdefault(k, []).append(v[0])
                bias.append(np.array(env.sense_noise.gyro_bias, dtype=np.float64))
            blk.update({k: np.array(v) for k, v in roll.items()})
            blk["draws"] = rec.pop_add_noise_calls(bias_model).reshape(T, 3, 10, 3)
            blk["gyro_bias"] = np.array(bias)
        finally:
            ref_sn.normal, ref_sn.uniform = saved
        blk["actions"] = act
        blk["obs_repr"] = np.array(obs_repr)
        blk["sense_json"] = np.array(json.dumps(sn))
        blk.update(pack("const_", derived_constants(env.dynamics)))
        arrays.update(pack("e%d_" % i, blk))
    arrays["n_envs"] = np.int64(len(cases))
    save(name, **arrays)


def g13_float32_actions():
    """RawControl called with float32 action ARRAYS (quadrotor_control.py:88-92): `0.5 * (action + 1.0)` is then float32
    arithmetic.  Same scenarios as G2/G3 in spirit (small action scales make the float32 sum lose the most bits), plus the
    `_act` observation (the raw float32 action is part of it) and the [0,1] action convention."""
    arrays = {}
    rng = np.random.RandomState(1313)
    cases = [("DefaultQuad", 1.0, {}), ("DefaultQuad", 0.05, {}), ("Crazyflie", 0.3, {}), ("Crazyflie", 1.0, {}),
             ("DefaultQuad", 0.5, {"obs_repr": "xyz_vxyz_R_omega_act"}), ("Crazyflie", 1.0, {"raw_control_zero_middle": False})]
    for i, (model, scale, kw) in enumerate(cases):
        env = make_env(dynamics_params=model, dynamics_change=NOISE_OFF, **kw)
        pos, vel, rot, omega = random_init(rng, env.goal, vel_scale=0.3 * (i % 2), omega_scale=1.0 * (i % 2), full_rot=bool(i % 2))
        set_state(env, pos, vel, rot, omega)
        lo = 0.0 if kw.get("raw_control_zero_middle", True) is False else -1.0
        act = f32(scale * rng.uniform(lo, 1, size=(400, 4)))
        blk = init_block(env, pos, vel, rot, omega)
        blk.update(rollout(env, act, record_ctrl=True, as_f32=True))
        blk["actions"] = act
        blk["kwargs_json"] = np.array(json.dumps(kw))
        blk.update(pack("const_", derived_constants(env.dynamics)))
        arrays.update(pack("e%d_" % i, blk))
    arrays["n_envs"] = np.int64(len(cases))
    save("g13_float32_actions", **arrays)


def g14_info_dict():
    """The per-step info dict (quadrotor.py:994-1025): every numeric entry of obs_comp and dyn_params, for RawControl
    (Hummingbird, CrazyFlie with motor lag) and the Mellinger controller, 60 steps each."""
    arrays = {}
    rng = np.random.RandomState(1414)
    cases = [("DefaultQuad", dict()), ("Crazyflie", dict()), ("DefaultQuad", dict(raw_control=False, tf_control=False))]
    for i, (model, kw) in enumerate(cases):
        env = make_env(dynamics_params=model, dynamics_change=NOISE_OFF, **kw)
        pos, vel, rot, omega = random_init(rng, env.goal, vel_scale=0.5, omega_scale=1.0, full_rot=(i == 1))
        set_state(env, pos, vel, rot, omega)
        act = f32(0.7 * rng.uniform(-1, 1, size=(60, 4)))
        blk = init_block(env, pos, vel, rot, omega)
        blk.update(rollout(env, act, record_ctrl=True, record_info=True))
        blk["actions"] = act
        blk["kwargs_json"] = np.array(json.dumps(kw))
        blk["model"] = np.array(model)
        if not kw.get("raw_control", True):
            blk["Jinv"] = np.array(env.controller.Jinv, dtype=np.float64)
        blk.update(pack("const_", derived_constants(env.dynamics)))
        arrays.update(pack("e%d_" % i, blk))
    arrays["n_envs"] = np.int64(len(cases))
    save("g14_info_dict", **arrays)


def g15_obs_variants_patched_imports():
    """PATCHED-IMPORT FIXTURE.  Six observation functions of get_state.py whose arithmetic is complete but which raise
    NameError as shipped because the module never imports the names they use (`normal`, `R2quat`): the t2w / t2t
    variants (:325-384) and the quaternion variants (:276-322).  Here those two names -- numpy.random.normal (through a
    recording proxy) and quad_utils.R2quat, the functions the code plainly means -- are injected into the module's
    namespace; NO reference source is changed or copied.  Everything else about these runs is the unmodified reference.
    The remaining broken variants are not restatable this way: their component names are missing from
    make_observation_space (KeyError in the constructor) or they misuse np.array / undefined variables."""
    import gym_art.quadrotor.get_state as gs
    import gym_art.quadrotor.sensor_noise as ref_sn
    arrays = {}
    rng = np.random.RandomState(1515)
    quat_sense = {"quat_norm_std": 0.01, "quat_unif_range": 0.004, "gyro_norm_std": 0.01, "gyro_bias_correlation_time": 50.0}
    cases = [("DefaultQuad", "xyz_vxyz_R_omega_t2w", None), ("Crazyflie", "xyzr_vxyzr_R_omega_t2w", None),
             ("MediumQuad", "xyz_vxyz_R_omega_t2w_t2t", "default"), ("DefaultQuad", "xyz_vxyz_quat_omega", None),
             ("Crazyflie", "xyzr_vxyzr_quat_omega", None), ("DefaultQuad", "xyzr_vxyzr_quat_omega_h", quat_sense)]
    saved = (ref_sn.normal, ref_sn.uniform)
    had = {k: getattr(gs, k, None) for k in ("normal", "R2quat")}
    for i, (model, obs_repr, sn) in enumerate(cases):
        rec = SenseDrawRecorder(np.random.RandomState(1520 + i))
        ref_sn.normal, ref_sn.uniform = rec.normal, rec.uniform
        gs.normal, gs.R2quat = rec.normal, ref_utils.R2quat
        try:
            env = make_env(dynamics_params=model, dynamics_change=NOISE_OFF, sense_noise=sn, obs_repr=obs_repr)
            rec.calls = []
            bias_model = isinstance(sn, dict) and sn.get("gyro_norm_std", 0.) != 0.
            per = 0 if sn is None else (10 if bias_model else 9)
            extra = ("t2w" in obs_repr) + ("t2t" in obs_repr)
            pos, vel, rot, omega = random_init(rng, env.goal, vel_scale=0.5, omega_scale=1.0, full_rot=(i % 2 == 1))
            set_state(env, pos, vel, rot, omega)
            bias0 = np.array(env.sense_noise.gyro_bias, dtype=np.float64) if sn is not None else np.zeros(3)
            T = 80
            act = f32(0.6 * rng.uniform(-1, 1, size=(T, 4)))
            blk = init_block(env, pos, vel, rot, omega)
            roll, draws, bias = {}, [], []
            for t in range(T):
                r1 = rollout(env, act[t:t + 1])
                for k, v in r1.items():
                    roll.setdefault(k, []).append(v[0])
                # this step's draws: add_noise alone (:946), then two state_vector calls = add_noise + the t2w / t2t normals
                seq = rec.calls
                rec.calls = []
                assert len(seq) == 3 * per + 2 * extra, (len(seq), per, extra)
                z = np.zeros((3, 12, 3))
                pos_in = 0
                for c in range(3):
                    if per:
                        d = [v for _, v in seq[pos_in:pos_in + per]]
                        if bias_model:
                            z[c, 0:10] = d
                        else:
                            z[c, 0:5] = d[0:5]; z[c, 6:10] = d[5:9]
                        pos_in += per
                    if c > 0:
                        for e in range(extra):
                            z[c, 10 + e, 0] = float(np.asarray(seq[pos_in][1]).reshape(-1)[0])
                            pos_in += 1
                assert pos_in == len(seq)
                draws.append(z)
                bias.append(np.array(env.sense_noise.gyro_bias, dtype=np.float64) if sn is not None else np.zeros(3))
            blk.update({k: np.array(v) for k, v in roll.items()})
            blk["draws"] = np.array(draws)
            blk["gyro_bias"] = np.array(bias)
            blk["init_gyro_bias"] = bias0
        finally:
            ref_sn.normal, ref_sn.uniform = saved
            for k, v in had.items():
                if v is None:
                    if hasattr(gs, k):
                        delattr(gs, k)
                else:
                    setattr(gs, k, v)
        blk["actions"] = act
        blk["obs_repr"] = np.array(obs_repr)
        blk["model"] = np.array(model)
        blk["sense_json"] = np.array(json.dumps(sn))
        blk["t2w_t2t"] = np.array([env.dynamics.thrust_to_weight, env.dynamics.torque_to_thrust], dtype=np.float64)
        blk["t2w_params"] = np.array([env.t2w_std, env.t2w_min, env.t2w_max, env.t2t_std, env.t2t_min, env.t2t_max], dtype=np.float64)
        blk["obs_low"] = np.array(env.observation_space.low, dtype=np.float64)
        blk["obs_high"] = np.array(env.observation_space.high, dtype=np.float64)
        blk.update(pack("const_", derived_constants(env.dynamics)))
        arrays.update(pack("e%d_" % i, blk))
    arrays["n_envs"] = np.int64(len(cases))
    arrays["provenance"] = np.array("patched-import fixture: get_state.normal and get_state.R2quat injected (missing imports in the "
                                    "reference as shipped); no reference source modified")
    save("g15_obs_variants_patched_imports", **arrays)


def g7_obs_reward_variants():
    """Every working obs_repr (get_state.py:5,134,147,219,236,249), non-default reward
    weights, non-zero-middle raw control, other sim_steps / sim_freq, and the
    quadrotor_multi log-distance reward (quadrotor_multi.py:550-650)."""
    import gym_art.quadrotor_multi.quadrotor_multi as refm
    arrays = {}
    rng = np.random.RandomState(707)
    cases = []
    for obs_repr in ("xyz_vxyz_R_omega", "xyz_vxyz_R_omega_h", "xyzr_vxyzr_R_omega", "xyzr_vxyzr_R_omega_h",
                     "xyz_vxyz_R_omega_acc_act", "xyz_vxyz_R_omega_act"):
        cases.append(dict(module="quadrotor", kw=dict(obs_repr=obs_repr)))
    cases.append(dict(module="quadrotor", kw=dict(
        rew_coeff={"pos": 0.7, "effort": 0.1, "action_change": 0.3, "crash": 2.0, "orient": 0.5, "yaw": 0.2,
                   "rot": 0.4, "attitude": 0.6, "spin": 0.25, "vel": 0.15})))
    cases.append(dict(module="quadrotor", kw=dict(raw_control_zero_middle=False)))
    cases.append(dict(module="quadrotor", kw=dict(sim_steps=1, sim_freq=100.0)))
    cases.append(dict(module="quadrotor", kw=dict(sim_steps=4, sim_freq=400.0, ep_time=1)))
    cases.append(dict(module="multi", kw=dict()))
    cases.append(dict(module="multi", kw=dict(rew_coeff={"spin": 0.1, "pos_offset": 0.05, "pos_log_weight": 0.8})))
    for i, c in enumerate(cases):
        module = refq if c["module"] == "quadrotor" else refm
        kw = dict(c["kw"])
        env = make_env(module=module, dynamics_change=NOISE_OFF, **kw)
        pos, vel, rot, omega = random_init(rng, env.goal, full_rot=(i % 3 == 2))
        set_state(env, pos, vel, rot, omega)
        T = 150
        lo = 0.0 if kw.get("raw_control_zero_middle", True) is False else -1.0
        act = f32(rng.uniform(lo, 1, size=(T, 4)))
        blk = init_block(env, pos, vel, rot, omega)
        blk.update(rollout(env, act))
        blk["actions"] = act
        blk["rew_coeff_json"] = np.array(json.dumps(env.rew_coeff))
        blk["kwargs_json"] = np.array(json.dumps(kw))
        blk["module"] = np.array(c["module"])
        blk["obs_low"] = np.array(env.observation_space.low, dtype=np.float64)
        blk["obs_high"] = np.array(env.observation_space.high, dtype=np.float64)
        blk["act_low"] = np.array(env.action_space.low, dtype=np.float64)
        blk["act_high"] = np.array(env.action_space.high, dtype=np.float64)
        arrays.update(pack("e%d_" % i, blk))
    arrays["n_envs"] = np.int64(len(cases))
    env = make_env(dynamics_change=NOISE_OFF)
    arrays.update(pack("const_", derived_constants(env.dynamics)))
    save("g7_obs_reward_variants", **arrays)


def g17_random_constructor_arguments():
    """Twenty-four random CONSTRUCTOR-argument sets through the reference -- model (DefaultQuad / Crazyflie / MediumQuad), controller
    (RawControl zero-middle / [0,1], Mellinger), observation variant, module (quadrotor / quadrotor_multi) and random reward weights, sim_freq x
    sim_steps, float32 or float64 action arrays -- 40 steps each from a random state: pins the oracle on COMBINATIONS of the features that
    G1-G14 pin one at a time."""
    import gym_art.quadrotor_multi.quadrotor_multi as refm
    arrays = {}
    rng = np.random.RandomState(1700)
    reprs = ["xyz_vxyz_R_omega", "xyz_vxyz_R_omega_h", "xyzr_vxyzr_R_omega", "xyzr_vxyzr_R_omega_h", "xyz_vxyz_R_omega_acc_act",
             "xyz_vxyz_R_omega_act"]
    keys = ("pos", "effort", "action_change", "crash", "orient", "yaw", "rot", "attitude", "spin", "vel")
    n_cases = 24
    for i in range(n_cases):
        module = ["quadrotor", "multi"][rng.randint(2)]
        raw = bool(rng.randint(4))
        freq, steps = [(200.0, 2), (100.0, 4), (400.0, 1), (250.0, 3)][rng.randint(4)]
        kw = dict(dynamics_params=["DefaultQuad", "Crazyflie", "MediumQuad"][rng.randint(3)], raw_control=raw,
                  raw_control_zero_middle=bool(rng.randint(2)), sim_freq=freq, sim_steps=steps, ep_time=float(rng.choice([1.0, 5.0, 7.0])),
                  obs_repr=reprs[rng.randint(len(reprs))],
                  rew_coeff={k: float(rng.uniform(0, 1)) for k in keys if rng.rand() < 0.4})
        as_f32 = bool(rng.randint(2))
        env = make_env(module=refq if module == "quadrotor" else refm, dynamics_change=NOISE_OFF, tf_control=False, **kw)
        pos, vel, rot, omega = random_init(rng, env.goal, full_rot=(i % 3 == 2))
        set_state(env, pos, vel, rot, omega)
        T = 40
        lo = 0.0 if (raw and not kw["raw_control_zero_middle"]) else -1.0
        act = f32(rng.uniform(lo, 1, size=(T, 4)))
        blk = init_block(env, pos, vel, rot, omega)
        blk.update(rollout(env, act, as_f32=as_f32))
        blk["actions"] = act
        blk["as_f32"] = np.array(as_f32)
        blk["rew_coeff_json"] = np.array(json.dumps(env.rew_coeff))
        blk["kwargs_json"] = np.array(json.dumps(kw))
        blk["module"] = np.array(module)
        blk.update(pack("const_", derived_constants(env.dynamics)))
        arrays.update(pack("e%d_" % i, blk))
    arrays["n_envs"] = np.int64(n_cases)
    save("g17_random_constructor_arguments", **arrays)


def g18_dynamics_change():
    """`dynamics_change` dicts (quadrotor.py:852-894 -> dict_update_existing, quad_utils.py:172-177) touching random subsets of the
    parameter tree -- link sizes and masses, motor and payload positions, arm angle, damping, thrust-to-weight, asymmetry, time constants,
    drag -- on the three shipped models: the derived constants the reference's update_model ends up with."""
    from gym_art.quadrotor.quad_models import crazyflie_params, defaultquad_params, mediumquad_params
    arrays = {}
    rng = np.random.RandomState(1800)
    base = {"Crazyflie": crazyflie_params, "DefaultQuad": defaultquad_params, "MediumQuad": mediumquad_params}
    i = 0
    for model in ("DefaultQuad", "Crazyflie", "MediumQuad"):
        for rep in range(6):
            tree = base[model]()
            change = {}
            for grp, sub in tree.items():
                for key, val in sub.items():
                    leaves = val.items() if isinstance(val, dict) else [(None, val)]
                    for leaf, v in leaves:
                        if rng.rand() > 0.25:
                            continue
                        if isinstance(v, (list, tuple, np.ndarray)):
                            new = [float(x) * rng.uniform(0.8, 1.2) if x != 0 else float(rng.uniform(-0.002, 0.002)) for x in v]
                        elif leaf == "z_sign":
                            new = int(rng.choice([-1, 1]))
                        elif v == 0:
                            new = float(rng.uniform(0, 0.02))
                        else:
                            new = float(v) * float(rng.uniform(0.8, 1.2))
                        if grp == "motor" and key == "linearity":
                            new = float(min(new, 1.0))
                        if leaf is None:
                            change.setdefault(grp, {})[key] = new
                        else:
                            change.setdefault(grp, {}).setdefault(key, {})[leaf] = new
            env = make_env(dynamics_params=model, dynamics_change=copy.deepcopy(change))
            blk = {"model": np.array(model), "change_json": np.array(json.dumps(change))}
            blk.update(pack("const_", derived_constants(env.dynamics)))
            arrays.update(pack("e%d_" % i, blk))
            i += 1
    arrays["n_envs"] = np.int64(i)
    save("g18_dynamics_change", **arrays)


def g19_resampled_goals():
    """resample_goal=True (quadrotor.py:1078-1079: goal z ~ U(0.5, 2) at every reset): the goal the reference drew, then 120 steps of
    RawControl / Mellinger flight towards it -- observations and rewards relative to a goal other than (0, 0, 2)."""
    arrays = {}
    rng = np.random.RandomState(1900)
    cases = [("DefaultQuad", True, "xyz_vxyz_R_omega"), ("DefaultQuad", False, "xyz_vxyz_R_omega"), ("Crazyflie", True, "xyzr_vxyzr_R_omega_h"),
             ("Crazyflie", False, "xyz_vxyz_R_omega_h"), ("MediumQuad", False, "xyzr_vxyzr_R_omega"), ("DefaultQuad", True, "xyz_vxyz_R_omega_acc_act")]
    for i, (model, raw, obs_repr) in enumerate(cases):
        env = make_env(dynamics_params=model, dynamics_change=NOISE_OFF, raw_control=raw, tf_control=False, resample_goal=True, obs_repr=obs_repr)
        np.random.seed(1910 + i)
        env.reset()                                   # draws the goal
        assert abs(float(env.goal[2]) - 2.0) > 1e-3
        pos, vel, rot, omega = random_init(rng, env.goal, full_rot=(i % 2 == 1))
        set_state(env, pos, vel, rot, omega)
        T = 120
        act = f32(rng.uniform(-1, 1, size=(T, 4)))
        blk = init_block(env, pos, vel, rot, omega)
        blk.update(rollout(env, act))
        blk["actions"] = act
        blk["kwargs_json"] = np.array(json.dumps(dict(dynamics_params=model, raw_control=raw, obs_repr=obs_repr)))
        blk.update(pack("const_", derived_constants(env.dynamics)))
        arrays.update(pack("e%d_" % i, blk))
    arrays["n_envs"] = np.int64(len(cases))
    save("g19_resampled_goals", **arrays)


def g20_gravity_argument():
    """The `gravity` constructor argument: in the reference it reaches ONLY the accelerometer reading (quadrotor.py:436) -- the dynamics,
    thrust_max and the controllers use the module constant GRAV (:175, :221, :426; quadrotor_control.py:323).  gravity = 5.0 and 12.0 with the
    accelerometer in the observation."""
    arrays = {}
    rng = np.random.RandomState(2000)
    cases = [(5.0, True), (12.0, False), (5.0, False)]
    for i, (g, raw) in enumerate(cases):
        env = make_env(dynamics_change=NOISE_OFF, raw_control=raw, tf_control=False, gravity=g, obs_repr="xyz_vxyz_R_omega_acc_act")
        np.random.seed(2010 + i)
        reset_obs = np.array(env.reset(), dtype=np.float64)
        pos, vel, rot, omega = random_init(rng, env.goal, full_rot=(i == 1))
        set_state(env, pos, vel, rot, omega)
        T = 60
        act = f32(rng.uniform(-1, 1, size=(T, 4)))
        blk = init_block(env, pos, vel, rot, omega)
        blk.update(rollout(env, act))
        blk["actions"] = act
        blk["reset_obs_acc"] = reset_obs[18:21]
        blk["kwargs_json"] = np.array(json.dumps(dict(raw_control=raw, gravity=g, obs_repr="xyz_vxyz_R_omega_acc_act")))
        blk.update(pack("const_", derived_constants(env.dynamics)))
        arrays.update(pack("e%d_" % i, blk))
    arrays["n_envs"] = np.int64(len(cases))
    save("g20_gravity_argument", **arrays)


def g21_sampler_streams():
    """The reference's parameter samplers draw by draw: RelativeSampler (normal / uniform, two noise ratios, a noise_ratio_custom tree) around
    the three shipped models and RandomQuad, each with numpy's global generator seeded -- the sampled trees.  A sampler that consumes a
    RandomState(seed) the way the reference consumes np.random.seed(seed) reproduces them exactly (n = 1)."""
    from gym_art.quadrotor.quad_models import crazyflie_params, defaultquad_params, mediumquad_params
    base = {"Crazyflie": crazyflie_params, "DefaultQuad": defaultquad_params, "MediumQuad": mediumquad_params}
    custom = {"motor": {"thrust_to_weight": 0.4, "damp_time_up": 0.1}, "geom": {"body": {"m": 0.05}}}
    arrays = {}
    i = 0
    for model in ("Crazyflie", "DefaultQuad", "MediumQuad"):
        for sampler, ratio, cust in (("normal", 0.2, None), ("uniform", 0.2, None), ("normal", 0.05, custom), ("uniform", 0.3, custom)):
            seed = 2100 + i
            np.random.seed(seed)
            tree = ref_rand.RelativeSampler(base[model](), noise_ratio=ratio, noise_ratio_custom=cust, sampler=sampler).sample(base[model]())
            blk = {"kind": np.array("relative"), "model": np.array(model), "sampler": np.array(sampler), "ratio": np.float64(ratio),
                   "custom_json": np.array(json.dumps(cust)), "seed": np.int64(seed)}
            blk.update(pack("param_", flatten_params(tree)))
            arrays.update(pack("e%d_" % i, blk))
            i += 1
    for k in range(6):
        seed = 2150 + k
        np.random.seed(seed)
        tree = ref_rand.RandomQuad().sample()
        blk = {"kind": np.array("randomquad"), "seed": np.int64(seed)}
        blk.update(pack("param_", flatten_params(tree)))
        arrays.update(pack("e%d_" % i, blk))
        i += 1
    arrays["n_envs"] = np.int64(i)
    save("g21_sampler_streams", **arrays)


def g8_reset_distribution():
    """Reset distribution (quadrotor.py:1059-1144): 4000 default resets
    (pos, yaw) and 4000 init_random_state resets (vel, omega, rot)."""
    np.random.seed(808)
    env = make_env(dynamics_change=NOISE_OFF)
    env._seed(809)
    n = 4000
    pos = np.zeros((n, 3))
    rot = np.zeros((n, 3, 3))
    for i in range(n):
        env.reset()
        pos[i], rot[i] = env.dynamics.pos, env.dynamics.rot
    env2 = make_env(dynamics_change=NOISE_OFF, init_random_state=True, resample_goal=True)
    env2._seed(810)
    pos2, vel2, om2, goal2 = (np.zeros((n, 3)) for _ in range(4))
    rot2 = np.zeros((n, 3, 3))
    for i in range(n):
        env2.reset()
        d = env2.dynamics
        pos2[i], vel2[i], om2[i], rot2[i], goal2[i] = d.pos, d.vel, d.omega, d.rot, env2.goal
    save("g8_reset_distribution", pos=pos, rot=rot, pos_rs=pos2, vel_rs=vel2, omega_rs=om2, rot_rs=rot2,
         goal_rs=goal2)


def timing():
    """Reference CPU step rate in THIS container (for BASELINE.md / DESIGN.md)."""
    out = {}
    for name, kw in (("raw", dict()), ("mellinger", dict(raw_control=False, tf_control=False))):
        env = make_env(**kw)
        rng = np.random.RandomState(0)
        n = 0
        t0 = time.perf_counter()
        for ep in range(3):
            env.reset()
            done = False
            while not done:
                _, _, done, _ = env.step(rng.uniform(-1, 1, size=4))
                n += 1
        out[name] = n / (time.perf_counter() - t0)
    print("reference env-steps/s (1 core, this container):", out)
    with open(os.path.join(HERE, "reference_timing.json"), "w") as f:
        json.dump({"env_steps_per_s": out, "host": "build container, 8 vCPU", "cores_used": 1}, f, indent=1)


if __name__ == "__main__":
    only = [a for a in sys.argv[1:] if a.startswith("g")]
    if only:                        # regenerate the named scenarios only, e.g. `make_golden.py g13_float32_actions`
        for name in only:
            globals()[name]()
        sys.exit(0)
    g9_kat()
    g1_mellinger()
    g1b_mellinger_other_models()
    g2_hummingbird_raw()
    g2b_episode_boundary()
    g3_crazyflie()
    g3b_asym_lag()
    g4_randomized()
    g4b_models()
    g4c_simplified()
    g5_drag_damp()
    g6_noise_injected()
    g7_obs_reward_variants()
    g8_reset_distribution()
    g10_sense_noise()
    g11_other_rates()
    g12_edge_cases()
    g13_float32_actions()
    g14_info_dict()
    g15_obs_variants_patched_imports()
    g16_sense_noise_param_sets()
    g17_random_constructor_arguments()
    g18_dynamics_change()
    g19_resampled_goals()
    g20_gravity_argument()
    g21_sampler_streams()
    if "--time" in sys.argv:
        timing()
