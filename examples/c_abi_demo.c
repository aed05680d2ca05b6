/* c_abi_demo.c -- libgaq.so driven from plain C through include/gaq.h only: no Python, no torch, no HIP headers.
 * This is the whole drop-in boundary: what a maintainer of the reference would bind.
 *
 *   gcc -O2 -Iinclude -o examples/c_abi_demo examples/c_abi_demo.c -Lgym_art_amd -lgaq -Wl,-rpath,'$ORIGIN/../gym_art_amd' -lm
 *   examples/c_abi_demo [num_envs] [steps] [shards]
 *
 * shards > 1: the same batch as ONE sharded handle over `shards` shards (gaq_create_sharded; shard k on device k mod the number of
 * devices -- on a one-GPU box they share it), stepped with the same two calls: BASELINE config 4 for a plain-C caller.  The numbers
 * printed do not depend on the split (the random streams are keyed by the global env index).
 *
 * Hummingbird ("DefaultQuad") constants as QuadrotorDynamics.update_model derives them (quadrotor.py:142-208; values:
 * SURVEY.md 8a2), RawControl zero-middle, sim_freq 200, sim_steps 2, ep_time 5: a hover-ish constant action for `steps`
 * env steps through the host-pointer entry points, printing the mean reward and the first env's observation. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gaq.h"

#define CHECK(call)                                                         \
  do {                                                                      \
    int rc_ = (call);                                                       \
    if (rc_ != GAQ_OK) {                                                    \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, gaq_last_error());      \
      return 1;                                                             \
    }                                                                       \
  } while (0)

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 4096;
  const int steps = argc > 2 ? atoi(argv[2]) : 100;
  const int shards = argc > 3 ? atoi(argv[3]) : 1;
  gaq_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.struct_size = sizeof(cfg);
  cfg.abi_version = GAQ_ABI_VERSION;
  cfg.num_envs = n;
  cfg.seed = 1;
  cfg.sim_freq = 200.0; cfg.sim_steps = 2; cfg.ep_len = 500;      /* int(ep_time / (dt * sim_steps)), quadrotor.py:792 */
  cfg.room_size = 10.0; cfg.gravity = 9.81;
  cfg.t2w_std = 0.005; cfg.t2t_std = 0.0005;
  cfg.control = GAQ_CTRL_RAW_ZERO_MIDDLE;
  cfg.noise = GAQ_NOISE_PHILOX;
  cfg.reward_mode = GAQ_REW_QUADROTOR;
  cfg.auto_reset = 1;
  cfg.obs_state_alias = 2;                                         /* library-owned state heads: the obs buffer is ours */
  cfg.action_f32 = 1;
  cfg.rew.pos = 1.0f; cfg.rew.effort = 0.05f; cfg.rew.crash = 1.0f; cfg.rew.orient = 1.0f; cfg.rew.spin = 0.1f;   /* :799-806 */
  gaq_model* m = &cfg.model;
  m->mass = 0.816;
  m->inertia[0] = 3.746575e-3; m->inertia[1] = 3.746575e-3; m->inertia[2] = 6.149342e-3;
  const double sx[4] = {1, -1, -1, 1}, sy[4] = {-1, -1, 1, 1};
  for (int j = 0; j < 4; ++j) {
    m->thrust_max[j] = 5.603472; m->torque_max[j] = 0.2801736;
    m->prop_pos[3 * j] = 0.12 * sx[j]; m->prop_pos[3 * j + 1] = 0.12 * sy[j]; m->prop_pos[3 * j + 2] = 7.174e-3;
  }
  m->linearity = 1.0; m->arm = 0.169706; m->ou_sigma = 0.01;

  /* which kernel will this configuration run?  Pure host logic -- works without a device (tests/test_plan_cpu.py enumerates it) */
  gaq_plan_info plan;
  CHECK(gaq_plan(&cfg, -1, -1, 0, 256, &plan));
  fprintf(stderr, "plan: step_kernel<%d> (instantiated: %d), state layout %d, obs_dim %d, LDS %d B per wave\n", plan.step_variant,
          plan.step_instantiated, plan.state_layout, plan.obs_dim, plan.lds_per_wave);
  if (!plan.step_instantiated || !plan.launchable) return 4;
  if (gaq_num_devices() <= 0) { fprintf(stderr, "no HIP device: libgaq has no CPU path\n"); return 2; }

  gaq_env* env = NULL;
  gaq_sharded* sh = NULL;
  if (shards > 1) {
    int32_t devs[64];
    if (shards > 64) return 5;
    for (int k = 0; k < shards; ++k) devs[k] = k % gaq_num_devices();
    CHECK(gaq_create_sharded(&cfg, devs, shards, &sh));
    env = gaq_sharded_shard(sh, 0);                                  /* (borrowed: obs width / layout queries) */
    for (int k = 0; k < gaq_sharded_num_shards(sh); ++k) {
      int64_t first, count; int32_t dev;
      CHECK(gaq_sharded_range(sh, k, &first, &count, &dev));
      fprintf(stderr, "shard %d: envs [%lld, %lld) on device %d\n", k, (long long)first, (long long)(first + count), (int)dev);
    }
  } else {
    CHECK(gaq_create(&cfg, &env));
  }
  const int D = gaq_obs_dim(env);
  float* obs = malloc(sizeof(float) * n * D);
  float* act = malloc(sizeof(float) * n * 4);
  float* rew = malloc(sizeof(float) * n);
  unsigned char* done = malloc(n);
  if (sh) CHECK(gaq_reset_sharded(sh, NULL, obs)); else CHECK(gaq_reset(env, NULL, obs));
  for (long i = 0; i < 4 * n; ++i) act[i] = -0.28f;                /* 0.5 (a + 1) = 0.36 ~ hover thrust at t2w = 2.8 */
  double mean = 0.0;
  long finished = 0;
  for (int t = 0; t < steps; ++t) {
    if (sh) CHECK(gaq_step_sharded(sh, act, obs, rew, done)); else CHECK(gaq_step(env, act, obs, rew, done));
    for (long i = 0; i < n; ++i) { mean += rew[i]; finished += done[i]; }
  }
  mean /= (double)n * steps;
  printf("{\"num_envs\": %ld, \"steps\": %d, \"obs_dim\": %d, \"state_layout\": %d, \"mean_reward\": %.6g, \"episodes_finished\": %ld, "
         "\"obs0\": [%.5f, %.5f, %.5f], \"R0_diag\": [%.5f, %.5f, %.5f]}\n",
         n, steps, D, gaq_state_layout(env), mean, finished, obs[0], obs[1], obs[2], obs[6], obs[10], obs[14]);
  const int ok = isfinite(mean) && D == 18 && plan.obs_dim == D && plan.state_layout == gaq_state_layout(env) && !gaq_is_diag_build() &&
                 (!sh || gaq_sharded_num_envs(sh) == n);
  if (sh) CHECK(gaq_destroy_sharded(sh)); else CHECK(gaq_destroy(env));
  free(obs); free(act); free(rew); free(done);
  return ok ? 0 : 3;
}
