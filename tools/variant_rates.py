#!/usr/bin/env python3
"""Step rate of a few realistic configurations that need different kernel instantiations (device tensors, N = 2^20)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from gym_art_amd import QuadrotorEnv  # noqa: E402

n = 1 << 20
dev = torch.device("cuda")
cases = {
    "default configuration, alias_obs=True (heads in the obs tensor; what bench.py times)": dict(alias_obs=True),
    "default configuration, class default layout (library-owned heads + obs copy)": {},
    "default configuration, alias_obs=False (fp64 planes)": dict(alias_obs=False),
    "init_random_state (alias kernel)": dict(init_random_state=True, alias_obs=True),
    "sense_noise=default (split state + packed observation, F_PACK)": dict(sense_noise="default"),
    "sense_noise + init_random_state + rot/attitude reward terms (F_PACK)":
        dict(sense_noise="default", init_random_state=True, rew_coeff={"rot": 0.1, "attitude": 0.1}),
    "obs xyz_vxyz_R_omega_acc_act (D=25; F_PACK)": dict(obs_repr="xyz_vxyz_R_omega_acc_act"),
    "obs xyz_vxyz_R_omega_act + action_change reward term (F_PACK)": dict(obs_repr="xyz_vxyz_R_omega_act", rew_coeff={"action_change": 0.1}),
    "Crazyflie + sense_noise=default (lag kernel, F_PACK)": dict(dynamics_params="Crazyflie", sense_noise="default"),
    "Crazyflie + sense_noise=default, thrust noise off (lag kernel, F_PACK)": dict(dynamics_params="Crazyflie", sense_noise="default", thrust_noise="off"),
    "resample_goal=True, Crazyflie, thrust noise off (F_ENVX, motor lag)": dict(resample_goal=True, dynamics_params="Crazyflie", thrust_noise="off"),
    "excite=True with the Mellinger controller, Crazyflie, thrust noise off (F_MELL | F_ENVX, motor lag)":
        dict(excite=True, raw_control=False, dynamics_params="Crazyflie", thrust_noise="off"),
    "Mellinger controller, class default layout (F_MELL, library-owned heads)": dict(raw_control=False),
    "Mellinger controller, alias_obs=True (F_MELL)": dict(raw_control=False, alias_obs=True),
    "Mellinger controller, alias_obs=False (F_MELL, fp64 planes)": dict(raw_control=False, alias_obs=False),
    "Mellinger controller, Crazyflie (F_MELL, motor lag), class default layout": dict(raw_control=False, dynamics_params="Crazyflie"),
    "Mellinger controller, obs xyz_vxyz_R_omega_h (F_MELL | F_PACK)": dict(raw_control=False, obs_repr="xyz_vxyz_R_omega_h"),
    "Mellinger controller with per-env randomized Crazyflie (the reference's benchmark() mode with -drr; F_MELL with per-env models), class default layout":
        dict(raw_control=False, dynamics_params="Crazyflie", dyn_sampler_1={"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"}),
    "Mellinger controller with per-env randomized Crazyflie, re-randomised on the device every episode (the reference's benchmark() mode with -drr -dre 1; F_MELL | F_RZ + the inverse-jacobian pass)":
        dict(raw_control=False, dynamics_params="Crazyflie", dyn_sampler_1={"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"},
             dynamics_randomize_every=1),
    "Mellinger controller with per-env randomized Crazyflie + sense_noise=default (F_MELL | F_PACK with per-env models)":
        dict(raw_control=False, sense_noise="default", dynamics_params="Crazyflie", dyn_sampler_1={"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"}),
    "info=True: aux row for the info dict (class default layout: split state, F_AUXP)": dict(info=True),
    "resample_goal=True (per-env goals; class default layout: split state, F_ENVX)": dict(resample_goal=True),
    "resample_goal=True on fp64 planes (per-env goals: light generic kernel)": dict(resample_goal=True, alias_obs=False),
    "excite=True (a new goal every fifth tick; class default layout: split state, F_ENVX)": dict(excite=True),
    "resample_goal=True with per-env randomized Crazyflie, re-randomised every episode (class default layout: split state, F_ENVX with per-env models)":
        dict(resample_goal=True, dynamics_params="Crazyflie", dyn_sampler_1={"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"},
             dynamics_randomize_every=1),
    "resample_goal=True with per-env randomized Crazyflie, re-randomised every episode, on fp64 planes (light generic kernel, per-env models)":
        dict(resample_goal=True, alias_obs=False, dynamics_params="Crazyflie", dyn_sampler_1={"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"},
             dynamics_randomize_every=1),
    "sense_noise with the gyro-bias random walk (class default layout: split state, F_ENVX)": dict(sense_noise={"gyro_norm_std": 0.01}),
    "sense_noise with the gyro-bias random walk on fp64 planes (full generic kernel)": dict(sense_noise={"gyro_norm_std": 0.01}, alias_obs=False),
    "info=True on fp64 planes (light generic kernel + aux row)": dict(info=True, alias_obs=False),
    "info=True with the Mellinger controller (class default layout: split state, F_MELL | F_AUXP)": dict(info=True, raw_control=False),
    "info=True with the Mellinger controller on fp64 planes (full diagnostics tier)": dict(info=True, raw_control=False, alias_obs=False),
    "excite=True with the Mellinger controller (what excite is for: the controller chases a moving goal; F_MELL | F_ENVX)": dict(excite=True, raw_control=False),
    "excite=True with the Mellinger controller on fp64 planes (full generic kernel)": dict(excite=True, raw_control=False, alias_obs=False),
    "excite=True with the Mellinger controller, Crazyflie (F_MELL | F_ENVX, motor lag)": dict(excite=True, raw_control=False, dynamics_params="Crazyflie"),
    "info=True with the Mellinger controller, Crazyflie, thrust noise off (F_MELL | F_AUXP, motor lag)":
        dict(info=True, raw_control=False, dynamics_params="Crazyflie", thrust_noise="off"),
    "info=True with per-env randomized Crazyflie (class default layout: split state, F_AUXP with per-env models)":
        dict(info=True, dynamics_params="Crazyflie", dyn_sampler_1={"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"}),
    "info=True with per-env randomized Crazyflie on fp64 planes (full diagnostics tier, per-env models)":
        dict(info=True, alias_obs=False, dynamics_params="Crazyflie", dyn_sampler_1={"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"}),
    "obs xyz_vxyz_R_omega_t2w with per-env randomized Crazyflie, re-randomised every episode (domain randomisation with the ratio observed)":
        dict(obs_repr="xyz_vxyz_R_omega_t2w", dynamics_params="Crazyflie", dyn_sampler_1={"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"},
             dynamics_randomize_every=1),
    "one process, four shards on this GPU: device_ids=[0, 0, 0, 0], alias_obs=True (host cost of the fan-out: four launches per step)":
        dict(device_ids=[0, 0, 0, 0], alias_obs=True),
    "Crazyflie uniform (lag kernel, mixed residual rows), alias_obs=True": dict(dynamics_params="Crazyflie", alias_obs=True),
    "Crazyflie per-env randomized on the device, re-randomised every episode, class default layout":
        dict(dynamics_params="Crazyflie", dyn_sampler_1={"class": "RelativeSampler", "noise_ratio": 0.2, "sampler": "normal"},
             dynamics_randomize_every=1),
    "obs xyz_vxyz_quat_omega (patched-import variant; split state, F_AUXP)": dict(obs_repr="xyz_vxyz_quat_omega"),
    "obs xyz_vxyz_R_omega_t2w_t2t (patched-import variant; split state, F_AUXP)": dict(obs_repr="xyz_vxyz_R_omega_t2w_t2t"),
}
# python3 tools/variant_rates.py [substring [steps]]: only the cases whose name contains the substring (profiling runs: rocprofv3 ... -- python3 tools/variant_rates.py "sense_noise=default" 40)
only = sys.argv[1] if len(sys.argv) > 1 else None
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
out = {}
for name, kw in cases.items():
    if only is not None and only not in name:
        continue
    env = QuadrotorEnv(num_envs=n, ep_time=5, seed=0, **kw)
    D = env.obs_dim
    obs = torch.empty((n, D), device=dev); rew = torch.empty(n, device=dev); done = torch.empty(n, dtype=torch.uint8, device=dev)
    acts = [torch.rand((n, 4), device=dev) * 2 - 1 for _ in range(4)]
    env.reset_dev(obs)
    for t in range(30):
        env.step_dev(acts[t % 4], obs, rew, done)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = nsteps
    for t in range(steps):
        env.step_dev(acts[t % 4], obs, rew, done)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out[name] = {"us_per_step": dt / steps * 1e6, "env_steps_per_s": n * steps / dt, "obs_dim": D, "state_layout": int(env.state_layout),
                 "kernel_variant": env.kernel_variant, "frac_of_8TBps_on_352B": 352.0 * n / (dt / steps) / 8e12}
    env.close()
    del env, obs, rew, done, acts
print(json.dumps(out, indent=1))
