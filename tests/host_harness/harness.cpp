// TEST INFRASTRUCTURE -- not part of the product and never loaded by gym_art_amd.
//
// Compiles the device arithmetic header (gym_art_amd/csrc/quad_core.hpp) for the HOST with g++ so that
//   * the kernel's per-env arithmetic can be checked against the golden vectors in the GPU-less
//     build container (tests/test_core_host.py), under -fsanitize=address,undefined (GPU sanitizers
//     are not available on the pool), and
//   * candidate numeric schemes (fp32 vs fp64 arithmetic / storage) can be measured against the
//     reference's trajectories (DESIGN.md "Numerics").
// It steps ONE env at a time with a scalar loop: it is neither fast nor a fallback.
#include <cstdint>
#include <cstring>

#include "../../gym_art_amd/csrc/quad_core.hpp"
#include "../../gym_art_amd/csrc/quad_params_dev.hpp"

using namespace gaq;

struct HHModel {   // mirrors gaq_model (include/gaq.h)
  double mass, inertia[3], thrust_max[4], torque_max[4], prop_pos[12];
  double damp_time_up, damp_time_down, linearity, arm, ou_sigma, vel_damp, damp_omega_quadratic, c_drag, c_roll;
};

template <typename T>
static void derive(const HHModel& g, double dt, Model<T>& m) {
  m.mass = T(g.mass); m.inv_mass = T(1.0 / g.mass);
  for (int j = 0; j < 3; ++j) { m.inertia[j] = T(g.inertia[j]); m.inv_inertia[j] = T(1.0 / g.inertia[j]); }
  for (int j = 0; j < 4; ++j) {
    m.thrust_max[j] = T(g.thrust_max[j]); m.torque_max[j] = T(g.torque_max[j]);
    m.prop_x[j] = T(g.prop_pos[3 * j]); m.prop_y[j] = T(g.prop_pos[3 * j + 1]); m.prop_z[j] = T(g.prop_pos[3 * j + 2]);
  }
  m.tau_up = T(4 * dt / (g.damp_time_up + 1e-6));
  m.tau_down = T(4 * dt / (g.damp_time_down + 1e-6));
  m.linearity = T(g.linearity); m.arm = T(g.arm); m.vel_damp = T(g.vel_damp); m.damp_omega_q = T(g.damp_omega_quadratic);
  m.c_drag = T(g.c_drag); m.c_roll = T(g.c_roll); m.ou_sigma = (float)g.ou_sigma;
  m.jinv = nullptr;   // uniform model: StepCfg::jinv
}

// state layout = planes 0-38 of gaq_get_state (include/gaq.h), one env
template <typename T>
static void unpack(const double* st, EnvState<T>& s) {
  for (int j = 0; j < 3; ++j) { s.pos[j] = T(st[j]); s.vel[j] = T(st[3 + j]); s.omega[j] = T(st[15 + j]); s.goal[j] = T(st[34 + j]); }
  for (int j = 0; j < 9; ++j) s.rot[j] = T(st[6 + j]);
  for (int j = 0; j < 4; ++j) {
    s.rot_damp[j] = T(st[18 + j]); s.cmds_damp[j] = (float)st[22 + j]; s.ou[j] = (float)st[26 + j]; s.act_prev[j] = (float)st[30 + j];
  }
  s.tick = (uint32_t)st[37]; s.svd_ctr = (uint32_t)st[38];
  for (int j = 0; j < 3; ++j) s.gyro_bias[j] = 0.0f;   // sensor noise is not exercised by this harness
}
template <typename T>
static void pack(const EnvState<T>& s, double* st) {
  for (int j = 0; j < 3; ++j) { st[j] = (double)s.pos[j]; st[3 + j] = (double)s.vel[j]; st[15 + j] = (double)s.omega[j]; st[34 + j] = (double)s.goal[j]; }
  for (int j = 0; j < 9; ++j) st[6 + j] = (double)s.rot[j];
  for (int j = 0; j < 4; ++j) { st[18 + j] = (double)s.rot_damp[j]; st[22 + j] = s.cmds_damp[j]; st[26 + j] = s.ou[j]; st[30 + j] = s.act_prev[j]; }
  st[37] = s.tick; st[38] = s.svd_ctr;
}

// store_f32 == 1: round the integrator state to fp32 after every env step (emulates fp32 state planes);
// store_f32 == 2: the alias layout's mixed residual rows (39 bits for pos / vel / R, omega exact)
template <typename T, uint32_t F>
static void rollout(const StepCfg& cfg0, const HHModel& g, double* state, int T_steps, const float* actions,
                    const float* normals, int store_f32, float* obs, float* rew, uint8_t* done, double* traj,
                    const float* sense, float* gyro_bias) {
  StepCfg cfg = cfg0;
  Model<T> m; derive(g, cfg.dt, m);
  EnvState<T> s; unpack(state, s);
  for (int j = 0; j < 3; ++j) s.gyro_bias[j] = gyro_bias ? gyro_bias[j] : 0.0f;
  const int D = cfg.obs_dim;
  for (int t = 0; t < T_steps; ++t) {
    StepOut out;
    float* row = obs + (size_t)t * D;
    const float* nz = normals ? normals + (size_t)t * cfg.sim_steps * 4 : nullptr;
    const float* sd = sense ? sense + (size_t)t * 3 * 12 * 3 : nullptr;     // [3 calls][12 slots][3] of this step
    env_step<T, F>(s, m, cfg, actions + 4 * t, cfg.env_offset, [&](int k, int c) { return nz ? nz[k * 4 + c] : 0.0f; }, out,
                   [&](int k, float v, int) { row[k] = v; }, nullptr, NoSwarm(),
                   [&](int c, int slot, int j) { return sd ? sd[(c * 12 + slot) * 3 + j] : 0.0f; });
    rew[t] = out.reward; done[t] = out.done;
    if (store_f32 == 2) {
      // the alias layout's mixed residual rows (gaq_kernels.hpp kLoMix): pos - goal, vel and R keep 39 significant bits
      // (fp32 head truncated toward zero + 16 residual bits), omega and the motor filter state stay exact
      auto q39 = [](double v) { return split_decode(split_hi(v), split_lo(v)); };
      for (int j = 0; j < 3; ++j) { s.pos[j] = T(q39((double)s.pos[j] - (double)s.goal[j]) + (double)s.goal[j]); s.vel[j] = T(q39((double)s.vel[j])); }
      for (int j = 0; j < 9; ++j) s.rot[j] = T(q39((double)s.rot[j]));
    } else if (store_f32 >= 100) {
      // experiment: every integrator word keeps the truncated fp32 head + (store_f32 - 100) residual bits
      const uint32_t keep = 0xFFFFu & ~((1u << (16 - (store_f32 - 100))) - 1u);
      auto qn = [keep](double v) { return split_decode(split_hi(v), split_lo(v) & keep); };
      for (int j = 0; j < 3; ++j) { s.pos[j] = T(qn((double)s.pos[j] - (double)s.goal[j]) + (double)s.goal[j]); s.vel[j] = T(qn((double)s.vel[j])); s.omega[j] = T(qn((double)s.omega[j])); }
      for (int j = 0; j < 9; ++j) s.rot[j] = T(qn((double)s.rot[j]));
    } else if (store_f32) {
      for (int j = 0; j < 3; ++j) { s.pos[j] = T((float)s.pos[j]); s.vel[j] = T((float)s.vel[j]); s.omega[j] = T((float)s.omega[j]); }
      for (int j = 0; j < 9; ++j) s.rot[j] = T((float)s.rot[j]);
      for (int j = 0; j < 4; ++j) s.rot_damp[j] = T((float)s.rot_damp[j]);
    }
    if (traj) pack(s, traj + (size_t)t * 39);
    cfg.step_index += 1;
  }
  pack(s, state);
  if (gyro_bias) for (int j = 0; j < 3; ++j) gyro_bias[j] = s.gyro_bias[j];
}

extern "C" {
// arith: 0 = double, 1 = float.  variant: gaq::Feature mask (8 = generic).
int hh_rollout(const StepCfg* cfg, const HHModel* model, double* state39, int T_steps, const float* actions,
               const float* normals, int arith, int variant, int store_f32, float* obs, float* rew, uint8_t* done,
               double* traj, const float* sense, float* gyro_bias) {
#define RUN(TT, FF) rollout<TT, FF>(*cfg, *model, state39, T_steps, actions, normals, store_f32, obs, rew, done, traj, sense, gyro_bias)
  if (arith == 0) {
    switch (variant) {
      case 0: RUN(double, 0u); break; case 2: RUN(double, 2u); break; case 4: RUN(double, 4u); break;
      case 6: RUN(double, 6u); break; case 8: RUN(double, 8u); break; case 520: RUN(double, 520u); break; default: return -1;
    }
  } else {
    switch (variant) {
      case 0: RUN(float, 0u); break; case 2: RUN(float, 2u); break; case 8: RUN(float, 8u); break; default: return -1;
    }
  }
#undef RUN
  return 0;
}
int hh_sizeof_cfg(void) { return (int)sizeof(StepCfg); }
int hh_sizeof_model(void) { return (int)sizeof(HHModel); }
void hh_reset(const StepCfg* cfg, double* state39, uint64_t env_global, uint64_t key) {
  EnvState<double> s; unpack(state39, s);
  reset_env<double, F_GENERIC>(s, *cfg, env_global, key);
  pack(s, state39);
}
// the device-side parameter pipeline (quad_params_dev.hpp), on the host: tree [40] -> derived constants, and the sampler
int hh_sizeof_derived(void) { return (int)sizeof(DerivedModel); }
void hh_derive_tree(const double* tree40, int clip, int by_density, DerivedModel* out) {
  ParamTree t;
  for (int k = 0; k < TL_COUNT; ++k) t.v[k] = tree40[k];
  if (clip) clip_tree(t, nullptr);
  derive_tree(t, *out, by_density != 0);
}
void hh_random_quad_tree(uint64_t seed, uint64_t env, uint64_t rc, double* out40) {
  ParamTree o;
  random_quad_tree(seed, env, rc, o);
  for (int k = 0; k < TL_COUNT; ++k) out40[k] = o.v[k];
}
void hh_perturb_tree(const double* base40, const double* ratio40, int sampler, uint64_t seed, uint64_t env, uint64_t rc, double* out40) {
  ParamTree b, o;
  for (int k = 0; k < TL_COUNT; ++k) b.v[k] = base40[k];
  perturb_tree(b, ratio40, sampler, seed, env, rc, o);
  for (int k = 0; k < TL_COUNT; ++k) out40[k] = o.v[k];
}
void hh_philox(uint64_t seed, uint64_t env, uint64_t step, uint32_t stream, uint32_t out[4]) {
  Philox p(seed, env, step, stream);
  for (int i = 0; i < 4; ++i) out[i] = p.c[i];
}
void hh_normals(uint64_t seed, uint64_t env, uint64_t step, uint32_t stream, float out[4]) {
  Philox p(seed, env, step, stream);
  normals4(p, out);
}
// ten normals out of two Philox blocks (the sensor-noise draws): `count` consecutive env indices, out [count][10]
void hh_normals10(uint64_t seed, uint64_t env0, uint64_t step, uint32_t stream, int64_t count, float* out) {
  for (int64_t k = 0; k < count; ++k) {
    Philox a(seed, env0 + (uint64_t)k, step, stream), b(seed, env0 + (uint64_t)k, step, stream + 1u);
    normals10(a, b, out + 10 * k);
  }
}
}
