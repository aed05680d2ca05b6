"""Helpers shared by the parity tests: load golden fixtures (tests/golden/*.npz,
generated from the unmodified reference by tests/golden/make_golden.py) and drive
the CPU oracle through them."""
import json
import os

import numpy as np

from oracle import quad_oracle as qo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def sub(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def env_blocks(d):
    n = int(d["n_envs"]) if "n_envs" in d else None
    out = []
    i = 0
    while True:
        b = sub(d, "e%d_" % i)
        if not b:
            break
        out.append(b)
        i += 1
    assert n is None or n == len(out)
    return out


def rel_err(a, b):
    """The parity metric of SURVEY.md §7.3: max |a-b| / max(|b|, 1)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0))) if a.size else 0.0


def cfg_from_block(blk, **kw):
    dt = float(blk["dt"])
    sim_steps = int(blk["sim_steps"])
    ep_len = int(blk["ep_len"])
    if "room_size" in blk:
        kw = dict(kw, room_size=float(blk["room_size"]))
    cfg = qo.Config(sim_freq=1.0 / dt, sim_steps=sim_steps, **kw)
    cfg.dt = dt
    cfg.ep_len = ep_len
    return cfg


def oracle_rollout(blk, const, cfg, normals=None, need_jinv=False, sense=None, sense_draws=None):
    """Run the oracle from the block's initial state through its actions."""
    p = qo.Params.from_golden_const(1, const)
    if need_jinv:
        p.jacobian_inverse()
    s = qo.State(1)
    s.goal[:] = blk["goal"]
    s.set_state(blk["init_pos"], blk["init_vel"], blk["init_rot"], blk["init_omega"], svd=float(blk["init_svd"]))
    T = blk["obs"].shape[0]
    actions = blk["actions"] if "actions" in blk else np.zeros((T, 4))
    out = {k: [] for k in ("obs", "reward", "done", "crashed", "pos", "vel", "rot", "omega", "thrust_rot_damp",
                           "thrust_cmds_damp", "accelerometer", "omega_dot", "torque", "since_last_svd",
                           "rew_raw", "ctrl")}
    for t in range(T):
        nrm = None if normals is None else normals[t][:, None, :]
        sd = None if sense is None else sense_draws[t][:, None]
        obs, rew, done = qo.env_step(s, p, cfg, actions[t][None], nrm, sense, sd)
        out["obs"].append(obs[0]); out["reward"].append(rew[0]); out["done"].append(done[0])
        out["crashed"].append(s.crashed[0])
        for k in ("pos", "vel", "rot", "omega", "thrust_rot_damp", "thrust_cmds_damp", "accelerometer",
                  "omega_dot", "torque", "since_last_svd", "rew_raw", "ctrl"):
            out[k].append(np.array(getattr(s, k)[0]))
    return {k: np.array(v) for k, v in out.items()}, s


def kwargs_of(blk):
    return json.loads(str(blk["kwargs_json"])) if "kwargs_json" in blk else {}
