// cpu_native.cpp -- compiled, multi-threaded CPU port of the hot path: bench.py's `cpu_baseline` leg ("cpu-c++" of
// BASELINE.md section 4.3).
//
// TEST / MEASUREMENT INFRASTRUCTURE ONLY -- never loaded by the product (gym_art_amd has no CPU path and fails
// loudly without its HIP library; tests/test_abi_cpu.py enforces that it imports nothing from oracle/).
//
// What runs: the very arithmetic header the GPU kernels are instantiated from (gym_art_amd/csrc/quad_core.hpp, fp64
// integrator chain: QuadrotorEnv._step = RawControl -> step1 x sim_steps -> crash -> reward -> tick/done -> obs, with the
// in-place auto-reset and Philox OU thrust noise), compiled for the host by g++ and driven by an OpenMP loop over a
// struct-of-arrays batch.  The same host build of that header is pinned to the reference's golden vectors by
// tests/test_core_host.py (fixtures G2/G3/G6/G7, <= 1e-9) and this file's batch driver to the NumPy oracle by
// tests/test_cpu_native.py.  References: /root/reference/gym_art/quadrotor/quadrotor.py:942-1028 (_step), :273-436
// (step1), :544-638 (reward), get_state.py:5-15.
#include <omp.h>

#include <chrono>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../gym_art_amd/csrc/quad_core.hpp"

using namespace gaq;

namespace {

struct NativeModel {   // mirrors gaq_model (include/gaq.h)
  double mass, inertia[3], thrust_max[4], torque_max[4], prop_pos[12];
  double damp_time_up, damp_time_down, linearity, arm, ou_sigma, vel_damp, damp_omega_quadratic, c_drag, c_roll;
};

void derive(const NativeModel& g, double dt, Model<double>& m) {
  m.mass = g.mass; m.inv_mass = 1.0 / g.mass;
  for (int j = 0; j < 3; ++j) { m.inertia[j] = g.inertia[j]; m.inv_inertia[j] = 1.0 / g.inertia[j]; }
  for (int j = 0; j < 4; ++j) {
    m.thrust_max[j] = g.thrust_max[j]; m.torque_max[j] = g.torque_max[j];
    m.prop_x[j] = g.prop_pos[3 * j]; m.prop_y[j] = g.prop_pos[3 * j + 1]; m.prop_z[j] = g.prop_pos[3 * j + 2];
  }
  m.tau_up = 4 * dt / (g.damp_time_up + 1e-6);
  m.tau_down = 4 * dt / (g.damp_time_down + 1e-6);
  m.linearity = g.linearity; m.arm = g.arm; m.vel_damp = g.vel_damp; m.damp_omega_q = g.damp_omega_quadratic;
  m.c_drag = g.c_drag; m.c_roll = g.c_roll; m.ou_sigma = (float)g.ou_sigma;
  m.jinv = nullptr;
}

struct Batch {
  int64_t n = 0;
  StepCfg cfg;
  Model<double> model;
  std::vector<EnvState<double>> s;     // array of structs: one env is one cache-resident record on the CPU
};

// one env step of the whole batch: actions [n,4] fp32 -> obs [n,D] fp32, reward [n], done [n]
template <uint32_t F>
void step_batch(Batch* b, const float* actions, float* obs, float* reward, uint8_t* done, int threads) {
  const int64_t n = b->n;
  const int D = b->cfg.obs_dim;
  const StepCfg cfg = b->cfg;
  const Model<double> m = b->model;
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    StepOut out;
    out.reward = 0.0f; out.done = 0; out.crashed = 0;
    float* row = obs + (size_t)i * D;
    env_step<double, F>(b->s[(size_t)i], m, cfg, actions + 4 * i, cfg.env_offset + (uint64_t)i, [](int, int) { return 0.0f; }, out,
                        [&](int k, float v, int) { row[k] = v; });
    reward[i] = out.reward; done[i] = out.done;
  }
  b->cfg.step_index += 1;
}

}  // namespace

extern "C" {

int cn_sizeof_cfg(void) { return (int)sizeof(StepCfg); }
int cn_sizeof_model(void) { return (int)sizeof(NativeModel); }
int cn_max_threads(void) { return omp_get_max_threads(); }

void* cn_create(const StepCfg* cfg, const NativeModel* model, int64_t n) {
  Batch* b = new Batch();
  b->n = n; b->cfg = *cfg;
  derive(*model, cfg->dt, b->model);
  b->s.resize((size_t)n);
  std::memset(b->s.data(), 0, sizeof(EnvState<double>) * (size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    EnvState<double>& s = b->s[(size_t)i];
    s.rot[0] = s.rot[4] = s.rot[8] = 1.0;
    for (int j = 0; j < 3; ++j) s.goal[j] = cfg->goal_default[j];
  }
  return b;
}
void cn_destroy(void* h) { delete static_cast<Batch*>(h); }

// state planes 0-38 of gaq_get_state, [39][n] doubles
void cn_set_state(void* h, const double* planes) {
  Batch* b = static_cast<Batch*>(h);
  const int64_t n = b->n;
  for (int64_t i = 0; i < n; ++i) {
    EnvState<double>& s = b->s[(size_t)i];
    for (int j = 0; j < 3; ++j) { s.pos[j] = planes[j * n + i]; s.vel[j] = planes[(3 + j) * n + i]; s.omega[j] = planes[(15 + j) * n + i];
                                  s.goal[j] = planes[(34 + j) * n + i]; }
    for (int j = 0; j < 9; ++j) s.rot[j] = planes[(6 + j) * n + i];
    for (int j = 0; j < 4; ++j) { s.rot_damp[j] = planes[(18 + j) * n + i]; s.cmds_damp[j] = (float)planes[(22 + j) * n + i];
                                  s.ou[j] = (float)planes[(26 + j) * n + i]; s.act_prev[j] = (float)planes[(30 + j) * n + i]; }
    s.tick = (uint32_t)planes[37 * n + i]; s.svd_ctr = (uint32_t)planes[38 * n + i];
  }
}

void cn_reset(void* h, int threads) {
  Batch* b = static_cast<Batch*>(h);
  const int64_t n = b->n;
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int64_t i = 0; i < n; ++i)
    reset_env<double, F_GENERIC>(b->s[(size_t)i], b->cfg, b->cfg.env_offset + (uint64_t)i, (1ull << 44));
}

int cn_step(void* h, const float* actions, float* obs, float* reward, uint8_t* done, int threads) {
  Batch* b = static_cast<Batch*>(h);
  const bool lag = b->cfg.motor_lag != 0, noise = b->cfg.noise == NOISE_PHILOX;
  if (b->cfg.noise == NOISE_INPUT || b->cfg.control == CTRL_MELLINGER || b->cfg.drag) { step_batch<F_GENERIC>(b, actions, obs, reward, done, threads); return 0; }
  if (lag && noise) step_batch<F_LAG | F_NOISE>(b, actions, obs, reward, done, threads);
  else if (lag) step_batch<F_LAG>(b, actions, obs, reward, done, threads);
  else if (noise) step_batch<F_NOISE>(b, actions, obs, reward, done, threads);
  else step_batch<0u>(b, actions, obs, reward, done, threads);
  return 0;
}

// timed loop for bench.py: a ring of `ring` pre-generated action tensors [ring][n][4]; runs whole batch steps until
// `seconds` have passed (at least 3 steps).  Returns the number of batch steps; *elapsed = wall seconds.
int64_t cn_run(void* h, const float* action_ring, int ring, double seconds, int threads, double* elapsed) {
  Batch* b = static_cast<Batch*>(h);
  const int64_t n = b->n;
  std::vector<float> obs((size_t)n * b->cfg.obs_dim), rew((size_t)n);
  std::vector<uint8_t> done((size_t)n);
  const auto t0 = std::chrono::steady_clock::now();
  int64_t steps = 0;
  double el = 0.0;
  for (;;) {
    cn_step(h, action_ring + (size_t)(steps % ring) * n * 4, obs.data(), rew.data(), done.data(), threads);
    ++steps;
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (el > seconds && steps >= 3) break;
  }
  *elapsed = el;
  return steps;
}

}  // extern "C"
