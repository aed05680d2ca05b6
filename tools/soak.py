#!/usr/bin/env python3
"""Long-run invariants on the device: tens of thousands of steps of the full-size batch (thousands of episodes per env in all), every
feature on -- thrust noise, in-kernel resets, per-episode re-randomisation with staggered phases -- then the state is checked: finite,
R orthonormal to 1e-9 (the fp64-grade rotation chain does not drift), omega and position inside their clips, the device-side episode
statistics consistent with steps / (ep_len + 1), no NaN reward ever, no staged-parameter overrun.  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_art_amd import QuadrotorEnv  # noqa: E402

dev = torch.device("cuda", 0)
sampler = {"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"}
out = {}
for label, n, steps, kw, stagger in (
        ("hummingbird_alias_2^20", 1 << 20, 100000, dict(dynamics_params="DefaultQuad", alias_obs=True), False),
        ("crazyflie_randomised_every_episode_staggered_2^20", 1 << 20, 40000,
         dict(dynamics_params="Crazyflie", dyn_sampler_1=sampler, dynamics_randomize_every=1, alias_obs=True), True),
        ("random_quad_every_episode_2^18_class_default_layout", 1 << 18, 40000, dict(dynamics_params="RandomQuad", dynamics_randomize_every=1), True),
        ("mellinger_2^18_class_default_layout", 1 << 18, 20000, dict(dynamics_params="DefaultQuad", raw_control=False), False),
        ("mellinger_crazyflie_sense_noise_obs_h_2^18", 1 << 18, 20000,
         dict(dynamics_params="Crazyflie", raw_control=False, sense_noise="default", obs_repr="xyz_vxyz_R_omega_h"), False),
        ("crazyflie_fp32_randomised_every_episode_staggered_2^18", 1 << 18, 20000,
         dict(dynamics_params="Crazyflie", dyn_sampler_1=sampler, dynamics_randomize_every=1, alias_obs=True, precision="fp32"), True),
        # round 4's kernels: the aux row + t2w observation on a re-randomised per-env batch (F_AUXP | F_RZ), sensor noise with the scalar model
        # reload, and one batch as four shards of a one-process multi-device env (all on this GPU)
        ("aux_row_t2w_obs_crazyflie_randomised_every_episode_staggered_2^18", 1 << 18, 20000,
         dict(dynamics_params="Crazyflie", dyn_sampler_1=sampler, dynamics_randomize_every=1, obs_repr="xyz_vxyz_R_omega_t2w", info=True), True),
        ("sense_noise_quaternion_obs_2^18", 1 << 18, 20000, dict(sense_noise="default", obs_repr="xyz_vxyz_quat_omega", init_random_state=True), False),
        ("four_shards_one_process_2^20", 1 << 20, 10000, dict(device_ids=[0, 0, 0, 0]), True),
        # ... per-env goals and the gyro-bias walk on the split state (F_ENVX, F_BIAS), Mellinger on per-env models (F_MELL, odd twins)
        ("resample_goal_excite_gyro_bias_walk_staggered_2^18", 1 << 18, 20000,
         dict(resample_goal=True, excite=True, sense_noise={"gyro_norm_std": 0.01}), True),
        ("resample_goal_crazyflie_randomised_every_episode_staggered_2^18", 1 << 18, 20000,
         dict(resample_goal=True, dynamics_params="Crazyflie", dyn_sampler_1=sampler, dynamics_randomize_every=1), True),
        ("mellinger_crazyflie_randomised_per_env_2^18", 1 << 18, 10000,
         dict(raw_control=False, dynamics_params="Crazyflie", dyn_sampler_1=sampler), False),
        ("mellinger_crazyflie_rerandomised_on_the_device_every_episode_staggered_2^18", 1 << 18, 20000,
         dict(raw_control=False, dynamics_params="Crazyflie", dyn_sampler_1=sampler, dynamics_randomize_every=1), True),
        ("mellinger_random_quads_every_episode_staggered_2^18", 1 << 18, 20000,
         dict(raw_control=False, dynamics_params="RandomQuad", dynamics_randomize_every=1), True)):
    env = QuadrotorEnv(num_envs=n, ep_time=5, seed=1, **kw)
    D = env.obs_dim
    obs = torch.empty((n, D), device=dev); rew = torch.empty(n, device=dev); done = torch.empty(n, dtype=torch.uint8, device=dev)
    env.reset_dev(obs)
    if stagger:
        st = env.get_state(); st[37] = np.arange(n) % (env.ep_len + 1); env.set_state(st)
    ring = [torch.rand((n, 4), device=dev) * 2 - 1 for _ in range(8)]
    steps_fn = [env.bind_step(a, obs, rew, done) for a in ring]
    done_total = torch.zeros((), dtype=torch.int64, device=dev)
    t0 = time.perf_counter()
    for t in range(steps):
        steps_fn[t & 7]()
        if (t & 63) == 0:
            done_total += done.sum()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    env.check_finite()                                     # NaN rewards since the start / staged-parameter overrun -> raises
    st = env.get_state()
    R = st[6:15].T.reshape(n, 3, 3)
    sub = slice(None, None, 37)
    res = {
        "envs": n, "steps": steps, "episodes_per_env": steps / (env.ep_len + 1.0), "env_steps_per_s_incl_host_loop": n * steps / wall,
        "all_finite": bool(np.isfinite(st[:37]).all() and torch.isfinite(obs).all().item() and torch.isfinite(rew).all().item()),
        "max_orthonormality_error": float(np.abs(np.einsum("nij,nkj->nik", R[sub], R[sub]) - np.eye(3)).max()),
        "max_abs_det_minus_1": float(np.abs(np.linalg.det(R[sub]) - 1).max()),
        "max_abs_omega": float(np.abs(st[15:18]).max()), "max_abs_xy": float(np.abs(st[0:2]).max()), "z_range": [float(st[2].min()), float(st[2].max())],
        "max_tick": int(st[37].max()), "done_sampled_every_64_steps": int(done_total.item()),
    }
    res["kernel_variant"] = env.kernel_variant
    ortho_tol = 1e-5 if kw.get("precision") == "fp32" else 1e-9     # (fp32 handles keep an fp32 rotation matrix)
    assert res["all_finite"] and res["max_orthonormality_error"] < ortho_tol and res["max_abs_omega"] <= 40.0 and res["max_abs_xy"] <= 10.0
    assert 0.0 <= res["z_range"][0] and res["z_range"][1] <= 10.0 and res["max_tick"] <= env.ep_len
    out[label] = res
    env.close()
    del obs, rew, done, ring, steps_fn
    torch.cuda.empty_cache()
    sys.stderr.write("%s ok (%.1f s)\n" % (label, wall))
print(json.dumps(out, indent=1))
