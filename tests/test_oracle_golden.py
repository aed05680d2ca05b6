"""The CPU oracle (oracle/quad_oracle.py) against the golden vectors recorded from the
unmodified reference.  Pins the oracle; runs without a GPU."""
import json

import numpy as np
import pytest

from oracle import quad_oracle as qo
from tests import golden_util as gu

TOL = 1e-12
STATE_KEYS = ("obs", "reward", "pos", "vel", "rot", "omega", "thrust_rot_damp", "thrust_cmds_damp",
              "accelerometer", "omega_dot", "torque", "since_last_svd")


def check(out, blk, keys=STATE_KEYS, tol=TOL):
    for k in keys:
        if k in blk:
            assert gu.rel_err(out[k], blk[k]) <= tol, k
    assert np.array_equal(out["done"], blk["done"])
    assert np.array_equal(out["crashed"], blk["crashed"])


def test_g9_known_answer():
    d = gu.load("g9_kat")
    cfg = gu.cfg_from_block(d)
    out, _ = gu.oracle_rollout(d, gu.sub(d, "const_"), cfg)
    check(out, d)
    # the literal values quoted in SURVEY.md §8c (there with an un-rounded initial velocity;
    # the fixture rounds it to fp32, hence 1e-9 rather than 1e-15 on the reward)
    assert abs(out["reward"][0] - 0.00029892250343235355) < 1e-9
    assert np.allclose(out["thrust_cmds_damp"][0], [0.625, 0.25, 0.875, 0.5], atol=1e-15)
    assert np.allclose(out["omega_dot"][0], [89.87491564928222, 0.23384080450430697, -34.171170084602565], rtol=1e-6)


def test_g1_mellinger_episode():
    d = gu.load("g1_mellinger")
    const = gu.sub(d, "const_")
    for i in range(2):
        blk = gu.sub(d, "e%d_" % i)
        cfg = gu.cfg_from_block(blk, control="mellinger")
        p = qo.Params.from_golden_const(1, const)
        assert np.allclose(p.jacobian_inverse()[0], d["Jinv"], rtol=1e-12, atol=1e-15)
        out, _ = gu.oracle_rollout(blk, const, cfg, need_jinv=True)
        check(out, blk, keys=STATE_KEYS + ("ctrl",), tol=1e-10)
        assert blk["done"][-1] and not blk["done"][-2] and len(blk["done"]) == int(blk["ep_len"]) + 1


def test_g1b_mellinger_on_crazyflie_and_mediumquad():
    for blk in gu.env_blocks(gu.load("g1b_mellinger_other_models")):
        const = gu.sub(blk, "const_")
        cfg = gu.cfg_from_block(blk, control="mellinger")
        p = qo.Params.from_golden_const(1, const)
        assert np.allclose(p.jacobian_inverse()[0], blk["Jinv"], rtol=1e-12, atol=1e-15)
        out, _ = gu.oracle_rollout(blk, const, cfg, need_jinv=True)
        check(out, blk, keys=STATE_KEYS + ("ctrl",), tol=1e-10)


@pytest.mark.parametrize("name", ["g2_hummingbird_raw", "g2b_episode_boundary", "g3_crazyflie", "g12_edge_cases"])
def test_raw_control_trajectories(name):
    d = gu.load(name)
    const = gu.sub(d, "const_")
    for blk in gu.env_blocks(d):
        cfg = gu.cfg_from_block(blk)
        out, _ = gu.oracle_rollout(blk, const, cfg)
        check(out, blk)
        assert gu.rel_err(out["rew_raw"], blk["rew_raw"]) <= TOL


@pytest.mark.parametrize("name", ["g3b_asym_lag", "g5_drag_damp", "g11_other_rates"])
def test_per_block_constants(name):
    d = gu.load(name)
    for blk in gu.env_blocks(d):
        cfg = gu.cfg_from_block(blk)
        out, _ = gu.oracle_rollout(blk, gu.sub(blk, "const_"), cfg)
        check(out, blk, tol=1e-11)


def test_g4_randomized_parameter_sets():
    d = gu.load("g4_randomized")
    blocks = gu.env_blocks(d)
    assert len(blocks) == 32
    lin = []
    for blk in blocks:
        cfg = gu.cfg_from_block(blk)
        const = gu.sub(blk, "const_")
        lin.append(float(const["motor_linearity"]))
        out, _ = gu.oracle_rollout(blk, const, cfg)
        check(out, blk, keys=("obs", "reward", "thrust_cmds_damp", "thrust_rot_damp"), tol=1e-11)
    assert min(lin) < 1.0  # RelativeSampler does perturb linearity below 1


def test_g6_injected_noise():
    d = gu.load("g6_noise_injected")
    for blk in gu.env_blocks(d):
        cfg = gu.cfg_from_block(blk)
        out, s = gu.oracle_rollout(blk, gu.sub(blk, "const_"), cfg, normals=blk["normals"])
        check(out, blk)
        assert gu.rel_err(s.ou_state[0], blk["ou_state_final"]) <= TOL


def test_g7_obs_and_reward_variants():
    d = gu.load("g7_obs_reward_variants")
    const = gu.sub(d, "const_")
    seen = set()
    for blk in gu.env_blocks(d):
        kw = gu.kwargs_of(blk)
        variant = "quadrotor" if str(blk["module"]) == "quadrotor" else "multi"
        control = "raw" if kw.get("raw_control_zero_middle", True) is False else "raw_zero_middle"
        cfg = gu.cfg_from_block(blk, control=control, obs_repr=kw.get("obs_repr", "xyz_vxyz_R_omega"),
                                rew_coeff=kw.get("rew_coeff"), reward_variant=variant)
        assert cfg.rew_coeff == json.loads(str(blk["rew_coeff_json"]))
        out, _ = gu.oracle_rollout(blk, const, cfg)
        check(out, blk)
        assert out["obs"].shape[1] == qo.OBS_REPRS[cfg.obs_repr][0]
        seen.add(cfg.obs_repr)
    assert seen == set(qo.OBS_REPRS)


@pytest.mark.parametrize("fixture", ["g10_sense_noise", "g16_sense_noise_param_sets"])
def test_g10_g16_sensor_noise_with_recorded_draws(fixture):
    """SensorNoise.add_noise incl. the gyro-bias random walk, fed the reference's own draws: 3 calls per step.  G10: the default
    model and two bias-walk sets over 60 steps; G16: ten random parameter sets over the six working observation variants."""
    d = gu.load(fixture)
    for blk in gu.env_blocks(d):
        const = gu.sub(blk, "const_")
        sn = json.loads(str(blk["sense_json"]))
        sense = qo.SenseNoise(1, **({} if sn == "default" else sn))
        cfg = gu.cfg_from_block(blk, obs_repr=str(blk["obs_repr"]))
        # the observation returned by reset() (one add_noise call on the freshly reset state)
        s0 = qo.State(1)
        s0.set_state(blk["reset_pos"], blk["reset_vel"], blk["reset_rot"], blk["reset_omega"])
        s0.goal[:] = blk["goal"]
        assert blk["reset_draws"].shape[0] == 1
        sense.gyro_bias[:] = blk["ctor_gyro_bias"]                  # the constructor's own _reset() already drew once
        o0 = qo.observe(s0, cfg, np.zeros((1, 4)), sense, blk["reset_draws"][0][None])
        assert gu.rel_err(o0[0], blk["reset_obs"]) <= TOL
        assert gu.rel_err(sense.gyro_bias[0], blk["reset_gyro_bias"]) <= TOL
        bias = []
        p = qo.Params.from_golden_const(1, const)
        s = qo.State(1)
        s.goal[:] = blk["goal"]
        s.set_state(blk["init_pos"], blk["init_vel"], blk["init_rot"], blk["init_omega"], svd=float(blk["init_svd"]))
        for t in range(blk["obs"].shape[0]):
            obs, rew, done = qo.env_step(s, p, cfg, blk["actions"][t][None], None, sense, blk["draws"][t][:, None])
            assert gu.rel_err(obs[0], blk["obs"][t]) <= TOL and gu.rel_err(rew[0], blk["reward"][t]) <= TOL, t
            bias.append(sense.gyro_bias[0].copy())
        assert gu.rel_err(np.array(bias), blk["gyro_bias"]) <= TOL
        assert gu.rel_err(s.pos[0], blk["pos"][-1]) <= TOL          # the true state never sees the noise
        if sense.gyro_norm_std != 0 and sense.gyro_noise_density != 0:      # (the walk's increments scale with the noise density)
            assert np.abs(blk["gyro_bias"][-1]).max() > 0


def test_g13_float32_action_arrays():
    """RawControl on float32 ARRAYS computes 0.5*(a+1) in float32 (quadrotor_control.py:88-92): the oracle's
    `action_f32` mode reproduces those trajectories; the float64-array arithmetic does NOT (up to 4e-5 away)."""
    d = gu.load("g13_float32_actions")
    apart = []
    for blk in gu.env_blocks(d):
        kw = gu.kwargs_of(blk)
        control = "raw" if kw.get("raw_control_zero_middle", True) is False else "raw_zero_middle"
        cfg = gu.cfg_from_block(blk, control=control, obs_repr=kw.get("obs_repr", "xyz_vxyz_R_omega"))
        cfg.action_f32 = True
        out, _ = gu.oracle_rollout(blk, gu.sub(blk, "const_"), cfg)
        assert gu.rel_err(out["obs"], blk["obs"]) <= TOL and gu.rel_err(out["ctrl"], blk["ctrl"]) <= TOL
        assert np.max(np.abs(out["reward"] - blk["reward"])) <= 1e-9       # (the reference's effort norm is a float32 one)
        cfg.action_f32 = False
        out64, _ = gu.oracle_rollout(blk, gu.sub(blk, "const_"), cfg)
        apart.append(gu.rel_err(out64["obs"], blk["obs"]))
    assert max(apart[:5]) > 1e-5 and min(apart[:5]) > 1e-6      # the two arithmetics are distinguishable on every zero-middle block
    assert apart[5] <= TOL                                       # [0,1] convention: scale 1, bias 0 -> no float32 rounding


def test_g14_info_dict_entries():
    """info["obs_comp"] (quadrotor.py:994-1006) from the oracle state, and t2i from the host parameter pipeline."""
    from gym_art_amd import quad_params as qp, quad_models
    d = gu.load("g14_info_dict")
    for blk in gu.env_blocks(d):
        kw = gu.kwargs_of(blk)
        const = gu.sub(blk, "const_")
        mell = kw.get("raw_control", True) is False
        cfg = gu.cfg_from_block(blk, control="mellinger" if mell else "raw_zero_middle")
        p = qo.Params.from_golden_const(1, const)
        if mell:
            p.jacobian_inverse()
        s = qo.State(1)
        s.goal[:] = blk["goal"]
        s.set_state(blk["init_pos"], blk["init_vel"], blk["init_rot"], blk["init_omega"], svd=float(blk["init_svd"]))
        for t in range(blk["obs"].shape[0]):
            qo.env_step(s, p, cfg, blk["actions"][t][None])
            comp = qo.info_obs_comp(s, blk["actions"][t][None])
            for k, v in comp.items():
                assert gu.rel_err(v[0], blk["info_obs_comp_" + k][t]) <= 1e-9, (k, t)
        tree = qp.batch_tree([quad_models.model_params(str(blk["model"]).lower())])
        models, extra = qp.derive_models(tree)
        assert gu.rel_err(extra["torque_to_inertia"][0], blk["info_dyn_params_t2i"][0]) <= TOL
        assert gu.rel_err(extra["torque_to_inertia"][0], const["torque_to_inertia"]) <= TOL
        assert gu.rel_err(np.mean(models["thrust_max"][0]), blk["info_dyn_params_thrust_max"][0]) <= TOL


def test_g15_patched_import_observation_variants():
    """The t2w / t2t and quaternion observation functions (get_state.py:276-384), which run in the reference once the two
    names the module forgot to import are supplied (PATCHED-IMPORT fixture G15, see make_golden.py)."""
    d = gu.load("g15_obs_variants_patched_imports")
    assert "patched-import" in str(d["provenance"])
    for blk in gu.env_blocks(d):
        const = gu.sub(blk, "const_")
        sn = json.loads(str(blk["sense_json"]))
        sense = None if sn is None else qo.SenseNoise(1, **({} if sn == "default" else sn))
        cfg = gu.cfg_from_block(blk, obs_repr=str(blk["obs_repr"]))
        assert [cfg.t2w_std, cfg.t2w_min, cfg.t2w_max, cfg.t2t_std, cfg.t2t_min, cfg.t2t_max] == list(blk["t2w_params"])
        p = qo.Params.from_golden_const(1, const)
        assert gu.rel_err(p.t2w[0], blk["t2w_t2t"][0]) <= TOL and gu.rel_err(p.t2t[0], blk["t2w_t2t"][1]) <= TOL
        p_derived = qo.Params.from_golden_const(1, {k: v for k, v in const.items() if k not in ("thrust_to_weight", "torque_to_thrust")})
        assert gu.rel_err(p_derived.t2w[0], blk["t2w_t2t"][0]) <= 1e-14 and gu.rel_err(p_derived.t2t[0], blk["t2w_t2t"][1]) <= 1e-14
        s = qo.State(1)
        s.goal[:] = blk["goal"]
        s.set_state(blk["init_pos"], blk["init_vel"], blk["init_rot"], blk["init_omega"], svd=float(blk["init_svd"]))
        if sense is not None:
            sense.gyro_bias[:] = blk["init_gyro_bias"]
        dim = qo.OBS_REPRS_PATCHED[cfg.obs_repr][0]
        assert blk["obs"].shape[1] == dim == len(blk["obs_low"])
        for t in range(blk["obs"].shape[0]):
            obs, rew, done = qo.env_step(s, p, cfg, blk["actions"][t][None], None, sense, blk["draws"][t][:, None])
            # R2quat divides by 4w with w = sqrt(1 + trace)/2: near a half-turn (w -> 0) the reference's own formula amplifies
            # the 1e-16 differences in R by 1/w^2
            tol = TOL * max(1.0, 0.05 / float(blk["obs"][t][6]) ** 2) if "quat" in cfg.obs_repr else TOL
            assert gu.rel_err(obs[0], blk["obs"][t]) <= tol and gu.rel_err(rew[0], blk["reward"][t]) <= TOL, (cfg.obs_repr, t)
        if sense is not None:
            assert gu.rel_err(sense.gyro_bias[0], blk["gyro_bias"][-1]) <= TOL


def test_svd_period_replay():
    assert qo.svd_period(0.005) == 100
    # other rates (fixture G11): the period is whatever the reference's fp64 accumulation of dt against 0.5 gives
    for blk in gu.env_blocks(gu.load("g11_other_rates")):
        dt, steps = float(blk["dt"]), int(blk["sim_steps"])
        fired = np.where(np.diff(blk["since_last_svd"]) < 0)[0]
        if len(fired) >= 2:
            assert np.all(np.diff(fired) * steps == qo.svd_period(dt)) or qo.svd_period(dt) % steps != 0
    d = gu.load("g2_hummingbird_raw")
    blk = gu.env_blocks(d)[0]
    ssvd = blk["since_last_svd"]
    fired = np.where(np.diff(ssvd) < 0)[0]
    assert len(fired) >= 9 and np.all(np.diff(fired) == 50)   # every 100 step1 = 50 env steps


def test_reset_distribution_matches_reference():
    """Statistical parity of the oracle's reset with 4000 reference resets (G8)."""
    from scipy import stats
    d = gu.load("g8_reset_distribution")
    n = 4000
    rng = np.random.RandomState(5)
    cfg = qo.Config(ep_time=5)
    s = qo.State(n)
    qo.reset(s, None, cfg, rng)
    for k in range(3):
        assert stats.ks_2samp(s.pos[:, k], d["pos"][:, k]).pvalue > 1e-3
    def yaw_offset(pos, rot):
        psi = np.arctan2(rot[:, 1, 0], rot[:, 0, 0])
        psi0 = np.arctan2(-pos[:, 1], -pos[:, 0])
        return np.angle(np.exp(1j * (psi - psi0)))
    a, b = yaw_offset(s.pos, s.rot), yaw_offset(d["pos"], d["rot"])
    assert np.max(np.abs(b)) <= np.pi / 3 + 1e-9 and np.max(np.abs(a)) <= np.pi / 3 + 1e-9
    assert stats.ks_2samp(a, b).pvalue > 1e-3
    assert np.all(d["pos"][:, 2] >= 0.25) and np.all(s.pos[:, 2] >= 0.25)
    # init_random_state variant
    s2 = qo.State(n)
    qo.reset(s2, None, cfg, rng, init_random_state=True, resample_goal=True)
    assert stats.ks_2samp(np.linalg.norm(s2.vel, axis=1), np.linalg.norm(d["vel_rs"], axis=1)).pvalue > 1e-3
    assert stats.ks_2samp(np.linalg.norm(s2.omega, axis=1), np.linalg.norm(d["omega_rs"], axis=1)).pvalue > 1e-3
    assert stats.ks_2samp(s2.goal[:, 2], d["goal_rs"][:, 2]).pvalue > 1e-3
    assert stats.ks_2samp(s2.rot[:, 2, 2], d["rot_rs"][:, 2, 2]).pvalue > 1e-3


def test_g17_random_constructor_arguments():
    """24 random combinations of model, controller, observation variant, reward variant and weights, rate and action dtype through the
    reference (fixture G17): the oracle on combinations of what G1-G14 pin feature by feature."""
    d = gu.load("g17_random_constructor_arguments")
    seen = set()
    for blk in gu.env_blocks(d):
        kw = gu.kwargs_of(blk)
        variant = "quadrotor" if str(blk["module"]) == "quadrotor" else "multi"
        control = "mellinger" if not kw["raw_control"] else ("raw_zero_middle" if kw["raw_control_zero_middle"] else "raw")
        cfg = gu.cfg_from_block(blk, control=control, obs_repr=kw["obs_repr"], rew_coeff=kw["rew_coeff"], reward_variant=variant)
        cfg.action_f32 = bool(blk["as_f32"])
        assert cfg.rew_coeff == json.loads(str(blk["rew_coeff_json"]))
        out, _ = gu.oracle_rollout(blk, gu.sub(blk, "const_"), cfg, need_jinv=(control == "mellinger"))
        check(out, blk)
        seen.add((control, kw["obs_repr"], variant, kw["dynamics_params"]))
    assert len(seen) >= 20


def test_g19_resampled_goals():
    """resample_goal=True: goals other than (0, 0, 2) in the observation (pos - goal), the reward and the Mellinger controller."""
    d = gu.load("g19_resampled_goals")
    for blk in gu.env_blocks(d):
        kw = gu.kwargs_of(blk)
        control = "raw_zero_middle" if kw["raw_control"] else "mellinger"
        cfg = gu.cfg_from_block(blk, control=control, obs_repr=kw["obs_repr"])
        out, _ = gu.oracle_rollout(blk, gu.sub(blk, "const_"), cfg, need_jinv=(control == "mellinger"))
        check(out, blk)
        assert abs(float(blk["goal"][2]) - 2.0) > 1e-3


def test_g20_gravity_argument():
    """The `gravity` constructor argument reaches only the accelerometer reading; dynamics and controllers keep GRAV = 9.81."""
    d = gu.load("g20_gravity_argument")
    for blk in gu.env_blocks(d):
        kw = gu.kwargs_of(blk)
        control = "raw_zero_middle" if kw["raw_control"] else "mellinger"
        cfg = gu.cfg_from_block(blk, control=control, obs_repr=kw["obs_repr"], gravity=kw["gravity"])
        out, _ = gu.oracle_rollout(blk, gu.sub(blk, "const_"), cfg, need_jinv=(control == "mellinger"))
        check(out, blk)
        assert gu.rel_err(blk["reset_obs_acc"], [0., 0., 9.81]) <= 1e-15        # after a reset: the module constant (:221)
