// gaq.hip -- fused HIP kernels (gfx950) + the C ABI of include/gaq.h.
//
// One lane = one environment, 64 envs per wavefront, 256 per workgroup.  State lives in HBM as
// struct-of-arrays planes [component][Npad] so every state load/store is a fully coalesced
// 512 B (fp64) / 256 B (fp32) wave transaction; the only array-of-structs tensors are the
// caller-facing actions [N,4] (one 16 B load per lane) and obs [N,D], which is transposed through
// an LDS tile and written as one contiguous run per workgroup.  No MFMA: the largest contraction is
// 3x3.3x3.  See DESIGN.md for the byte accounting and quad_core.hpp for the arithmetic.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/gaq.h"
#include "quad_core.hpp"

namespace {

constexpr int kBlock = 256;
constexpr int kP64 = 22;   // fp64 state planes: pos3 vel3 rot9 omega3 rot_damp4
constexpr int kP32 = 15;   // fp32 state planes: ou4 cmds_damp4 act_prev4 goal3
constexpr int kPar = 37;   // fp64 per-env parameter planes
enum ParPlane { PP_MASS = 0, PP_INV_MASS = 1, PP_INERTIA = 2, PP_INV_INERTIA = 5, PP_THRUST_MAX = 8, PP_TORQUE_MAX = 12,
                PP_PROP_X = 16, PP_PROP_Y = 20, PP_PROP_Z = 24, PP_TAU_UP = 28, PP_TAU_DOWN = 29, PP_LINEARITY = 30,
                PP_ARM = 31, PP_VEL_DAMP = 32, PP_DAMP_Q = 33, PP_C_DRAG = 34, PP_C_ROLL = 35, PP_OU_SIGMA = 36 };

struct DevPtrs {
  double* s64;       // [kP64][npad]
  float* s32;        // [kP32][npad]
  uint32_t* ctr;     // [npad]  tick | svd_ctr << 16
  const double* par; // [kPar][npad] or nullptr
  const float* noise_in;  // [sim_steps][4][n] or nullptr
  uint32_t* done_list;    // [npad] or nullptr
  uint32_t* done_count;   // [2] (ping-pong by step parity)
  uint32_t* nan_count;    // [1]
  int64_t n, npad;
};

using gaq::EnvState;
using gaq::Model;
using gaq::StepCfg;

// ---- plane access --------------------------------------------------------------------------------------
// Every SoA plane is addressed through a buffer resource built from wave-uniform values (plane base in
// SGPRs) plus ONE per-lane byte offset, so the ~45 planes a step touches cost no address VGPRs (flat
// addressing kept a 64-bit pointer per plane live from the loads to the stores: +40 VGPRs).  Out-of-range
// lanes are dropped by the hardware bounds check.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct Planes64 {
  const double* base; int64_t np;
  __device__ __forceinline__ double ld(int plane, uint32_t off8) const {
    auto r = __builtin_amdgcn_make_buffer_rsrc((void*)(base + plane * np), 0, (int)(np * 8), 0x00020000);
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, off8, 0, 0));
  }
  __device__ __forceinline__ void st(int plane, uint32_t off8, double v) const {
    auto r = __builtin_amdgcn_make_buffer_rsrc((void*)(base + plane * np), 0, (int)(np * 8), 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, off8, 0, 0);
  }
};
struct Planes32 {
  const void* base; int64_t np;
  __device__ __forceinline__ float ldf(int plane, uint32_t off4) const {
    auto r = __builtin_amdgcn_make_buffer_rsrc((void*)((const float*)base + plane * np), 0, (int)(np * 4), 0x00020000);
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off4, 0, 0));
  }
  __device__ __forceinline__ uint32_t ldu(int plane, uint32_t off4) const {
    auto r = __builtin_amdgcn_make_buffer_rsrc((void*)((const float*)base + plane * np), 0, (int)(np * 4), 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b32(r, off4, 0, 0);
  }
  __device__ __forceinline__ void stf(int plane, uint32_t off4, float v) const {
    auto r = __builtin_amdgcn_make_buffer_rsrc((void*)((const float*)base + plane * np), 0, (int)(np * 4), 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), r, off4, 0, 0);
  }
  __device__ __forceinline__ void stu(int plane, uint32_t off4, uint32_t v) const {
    auto r = __builtin_amdgcn_make_buffer_rsrc((void*)((const float*)base + plane * np), 0, (int)(np * 4), 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(v, r, off4, 0, 0);
  }
};

template <uint32_t F>
__device__ __forceinline__ void load_state(const DevPtrs& p, const StepCfg& cfg, uint32_t i, EnvState<double>& s) {
  const Planes64 a{p.s64, p.npad};
  const Planes32 b{p.s32, p.npad};
  const Planes32 c{p.ctr, p.npad};
  const uint32_t o8 = i * 8u, o4 = i * 4u;
#pragma unroll
  for (int j = 0; j < 3; ++j) s.pos[j] = a.ld(0 + j, o8);
#pragma unroll
  for (int j = 0; j < 3; ++j) s.vel[j] = a.ld(3 + j, o8);
#pragma unroll
  for (int j = 0; j < 9; ++j) s.rot[j] = a.ld(6 + j, o8);
#pragma unroll
  for (int j = 0; j < 3; ++j) s.omega[j] = a.ld(15 + j, o8);
  if (gaq::has_lag<F>(cfg)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { s.rot_damp[j] = a.ld(18 + j, o8); s.cmds_damp[j] = b.ldf(4 + j, o4); }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) { s.rot_damp[j] = 0.0; s.cmds_damp[j] = 0.0f; }
  }
  if (gaq::noise_mode<F>(cfg) != gaq::NOISE_OFF) {
#pragma unroll
    for (int j = 0; j < 4; ++j) s.ou[j] = b.ldf(0 + j, o4);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) s.ou[j] = 0.0f;
  }
  if (gaq::has_act_prev<F>(cfg)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) s.act_prev[j] = b.ldf(8 + j, o4);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) s.act_prev[j] = 0.0f;
  }
  if (gaq::has_env_goal<F>(cfg)) {
#pragma unroll
    for (int j = 0; j < 3; ++j) s.goal[j] = (double)b.ldf(12 + j, o4);
  } else {
#pragma unroll
    for (int j = 0; j < 3; ++j) s.goal[j] = cfg.goal_default[j];
  }
  const uint32_t cw = c.ldu(0, o4);
  s.tick = cw & 0xFFFFu;
  s.svd_ctr = cw >> 16;
}

template <uint32_t F>
__device__ __forceinline__ void store_state(const DevPtrs& p, const StepCfg& cfg, uint32_t i, const EnvState<double>& s,
                                            bool all) {
  const Planes64 a{p.s64, p.npad};
  const Planes32 b{p.s32, p.npad};
  const Planes32 c{p.ctr, p.npad};
  const uint32_t o8 = i * 8u, o4 = i * 4u;
#pragma unroll
  for (int j = 0; j < 3; ++j) a.st(0 + j, o8, s.pos[j]);
#pragma unroll
  for (int j = 0; j < 3; ++j) a.st(3 + j, o8, s.vel[j]);
#pragma unroll
  for (int j = 0; j < 9; ++j) a.st(6 + j, o8, s.rot[j]);
#pragma unroll
  for (int j = 0; j < 3; ++j) a.st(15 + j, o8, s.omega[j]);
  if (gaq::has_lag<F>(cfg) || all) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { a.st(18 + j, o8, s.rot_damp[j]); b.stf(4 + j, o4, s.cmds_damp[j]); }
  }
  if (gaq::noise_mode<F>(cfg) != gaq::NOISE_OFF || all) {
#pragma unroll
    for (int j = 0; j < 4; ++j) b.stf(0 + j, o4, s.ou[j]);
  }
  if (gaq::has_act_prev<F>(cfg) || all) {
#pragma unroll
    for (int j = 0; j < 4; ++j) b.stf(8 + j, o4, s.act_prev[j]);
  }
  if (gaq::has_env_goal<F>(cfg) || all) {
#pragma unroll
    for (int j = 0; j < 3; ++j) b.stf(12 + j, o4, (float)s.goal[j]);
  }
  c.stu(0, o4, (s.tick & 0xFFFFu) | (s.svd_ctr << 16));
}

template <uint32_t F>
__device__ __forceinline__ void load_model(const DevPtrs& p, const StepCfg& cfg, uint32_t i, const Model<double>& um,
                                           Model<double>& m) {
  if constexpr ((F & gaq::F_PER_ENV) == 0) { m = um; return; }
  const Planes64 q{p.par, p.npad};
  const uint32_t o8 = i * 8u;
  m.inv_mass = q.ld(PP_INV_MASS, o8);
#pragma unroll
  for (int j = 0; j < 3; ++j) { m.inertia[j] = q.ld(PP_INERTIA + j, o8); m.inv_inertia[j] = q.ld(PP_INV_INERTIA + j, o8); }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    m.thrust_max[j] = q.ld(PP_THRUST_MAX + j, o8);
    m.torque_max[j] = q.ld(PP_TORQUE_MAX + j, o8);
    m.prop_x[j] = q.ld(PP_PROP_X + j, o8);
    m.prop_y[j] = q.ld(PP_PROP_Y + j, o8);
  }
  m.linearity = q.ld(PP_LINEARITY, o8);
  m.arm = q.ld(PP_ARM, o8);
  m.vel_damp = q.ld(PP_VEL_DAMP, o8);
  m.damp_omega_q = q.ld(PP_DAMP_Q, o8);
  m.tau_up = 1.0; m.tau_down = 1.0;
  if (gaq::has_lag<F>(cfg)) { m.tau_up = q.ld(PP_TAU_UP, o8); m.tau_down = q.ld(PP_TAU_DOWN, o8); }
  m.ou_sigma = 0.0f;
  if (gaq::noise_mode<F>(cfg) != gaq::NOISE_OFF) m.ou_sigma = (float)q.ld(PP_OU_SIGMA, o8);
  m.mass = 0.0; m.c_drag = 0.0; m.c_roll = 0.0;
#pragma unroll
  for (int j = 0; j < 4; ++j) m.prop_z[j] = 0.0;
  if (((F & gaq::F_GENERIC) != 0) && cfg.drag) {
    m.mass = q.ld(PP_MASS, o8); m.c_drag = q.ld(PP_C_DRAG, o8); m.c_roll = q.ld(PP_C_ROLL, o8);
#pragma unroll
    for (int j = 0; j < 4; ++j) m.prop_z[j] = q.ld(PP_PROP_Z + j, o8);
  }
}

// Coalesced write-out of a workgroup's [rows, D] observation tile staged in LDS.
__device__ __forceinline__ void flush_obs_tile(const float* tile, float* obs, int64_t block_first, int64_t n, int D) {
  __syncthreads();
  const int64_t rows = (n - block_first) < kBlock ? (n - block_first) : kBlock;
  const int total = (int)rows * D;
  float* dst = obs + block_first * D;
  for (int k = threadIdx.x; k < total; k += kBlock) dst[k] = tile[k];
}

// ---- the fused step kernel: controller + step1 x sim_steps + crash + reward + done (+ reset) + obs ----
template <uint32_t F>
__global__ __launch_bounds__(kBlock) void step_kernel(DevPtrs p, StepCfg cfg, Model<double> um,
                                                       const float* __restrict__ actions, float* __restrict__ obs,
                                                       float* __restrict__ reward, uint8_t* __restrict__ done) {
  extern __shared__ float tile[];
  const int64_t block_first = (int64_t)blockIdx.x * kBlock;
  const int64_t i = block_first + threadIdx.x;
  const int D = cfg.obs_dim;
  bool is_done = false;
  if (i < p.n) {
    EnvState<double> s;
    Model<double> m;
    load_state<F>(p, cfg, (uint32_t)i, s);
    load_model<F>(p, cfg, (uint32_t)i, um, m);
    const float4 a4 = reinterpret_cast<const float4*>(actions)[i];
    const float act[4] = {a4.x, a4.y, a4.z, a4.w};
    gaq::StepOut out;
    float* row = tile + threadIdx.x * D;
    const float* nz = p.noise_in;
    const int64_t n = p.n;
    gaq::env_step<double, F>(
        s, m, cfg, act, cfg.env_offset + (uint64_t)i,
        [&](int k, int c) { return nz[((int64_t)k * 4 + c) * n + i]; }, out,
        [&](int k, float v) { row[k] = v; });
    store_state<F>(p, cfg, (uint32_t)i, s, false);
    reward[i] = out.reward;
    done[i] = out.done;
    is_done = out.done;
    if (!isfinite(out.reward)) atomicAdd(p.nan_count, 1u);
  }
  if (p.done_list) {   // wavefront compaction of the done env indices (host-side episode bookkeeping)
    uint32_t* cnt = p.done_count + (cfg.step_index & 1);
    if (i == 0) p.done_count[(cfg.step_index + 1) & 1] = 0;   // next step's counter
    const unsigned long long mask = __ballot(is_done);
    if (mask) {
      const int lane = threadIdx.x & 63;
      const int leader = __ffsll((long long)mask) - 1;
      uint32_t base = 0;
      if (lane == leader) base = atomicAdd(cnt, (uint32_t)__popcll(mask));
      base = __shfl(base, leader);
      if (is_done) p.done_list[base + __popcll(mask & ((1ull << lane) - 1ull))] = (uint32_t)i;
    }
  }
  flush_obs_tile(tile, obs, block_first, p.n, D);
}

// ---- reset kernel: QuadrotorEnv._reset for masked envs, and/or pack the observation of the current state ----
__global__ __launch_bounds__(kBlock) void reset_kernel(DevPtrs p, StepCfg cfg, const uint8_t* __restrict__ mask,
                                                        int do_reset, float* __restrict__ obs) {
  extern __shared__ float tile[];
  const int64_t block_first = (int64_t)blockIdx.x * kBlock;
  const int64_t i = block_first + threadIdx.x;
  const int D = cfg.obs_dim;
  if (i < p.n) {
    EnvState<double> s;
    StepCfg full = cfg;     // explicit reset / observe touch every plane irrespective of feature flags
    full.motor_lag = 1; full.noise = gaq::NOISE_PHILOX; full.need_act_prev = 1; full.per_env_goal = 1;
    load_state<gaq::F_GENERIC>(p, full, (uint32_t)i, s);
    float acc[3] = {0.0f, 0.0f, 9.81f};
    float hist[4] = {s.act_prev[0], s.act_prev[1], s.act_prev[2], s.act_prev[3]};
    if (do_reset && (mask == nullptr || mask[i])) {
      gaq::reset_env<double, gaq::F_GENERIC>(s, cfg, cfg.env_offset + (uint64_t)i, cfg.step_index);
      store_state<gaq::F_GENERIC>(p, full, (uint32_t)i, s, true);
      hist[0] = hist[1] = hist[2] = hist[3] = 0.0f;
    }
    if (obs) {
      float* row = tile + threadIdx.x * D;
      gaq::pack_obs<double, gaq::F_GENERIC>(s, cfg, acc, hist, [&](int k, float v) { row[k] = v; });
    }
  }
  if (obs) flush_obs_tile(tile, obs, block_first, p.n, D);
}

// ---- host side ---------------------------------------------------------------------------------------
thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(GAQ_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));      \
  } while (0)

int svd_period_of(double dt) {   // replay of `since_last_svd += dt; if since_last_svd > 0.5` (quadrotor.py:381-386)
  double t = 0.0; int k = 0;
  while (!(t > 0.5)) { t += dt; ++k; if (k > 1000000) break; }
  return k;
}

}  // namespace

struct gaq_env {
  gaq_config cfg;
  StepCfg sc;
  Model<double> um;
  DevPtrs d;
  int obs_dim = 18;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timing = false, timed = false;
  uint64_t reset_calls = 0;
  const float* noise_next = nullptr;
  std::vector<double> host_par;   // [kPar][npad] staging for per-env params
  bool any_lag = false, any_drag = false;
  bool force_generic = false;
  int variant = 0;   // gaq::Feature mask of the step kernel in use
};

namespace {

void derive_model(const gaq_model& g, double dt, Model<double>& m) {
  m.mass = g.mass; m.inv_mass = 1.0 / g.mass;
  for (int j = 0; j < 3; ++j) { m.inertia[j] = g.inertia[j]; m.inv_inertia[j] = 1.0 / g.inertia[j]; }
  for (int j = 0; j < 4; ++j) {
    m.thrust_max[j] = g.thrust_max[j]; m.torque_max[j] = g.torque_max[j];
    m.prop_x[j] = g.prop_pos[3 * j]; m.prop_y[j] = g.prop_pos[3 * j + 1]; m.prop_z[j] = g.prop_pos[3 * j + 2];
  }
  m.tau_up = 4 * dt / (g.damp_time_up + 1e-6);      // quadrotor.py:284-285
  m.tau_down = 4 * dt / (g.damp_time_down + 1e-6);
  m.linearity = g.linearity; m.arm = g.arm; m.vel_damp = g.vel_damp; m.damp_omega_q = g.damp_omega_quadratic;
  m.c_drag = g.c_drag; m.c_roll = g.c_roll; m.ou_sigma = (float)g.ou_sigma;
}

int check_model(const gaq_model& g) {
  if (!(g.mass > 0) || !(g.inertia[0] > 0) || !(g.inertia[1] > 0) || !(g.inertia[2] > 0))
    return fail(GAQ_ERR_INVALID, "model: mass and inertia must be positive");
  if (g.damp_time_up < 0 || g.damp_time_down < 0) return fail(GAQ_ERR_INVALID, "model: negative motor time constant");
  return GAQ_OK;
}

void refresh_feature_flags(gaq_env* e) {
  StepCfg& sc = e->sc;
  sc.motor_lag = e->any_lag ? 1 : 0;
  sc.drag = e->any_drag ? 1 : 0;
  // kernel variant: the specialised ("fast") instantiations cover RawControl, the 18-word observation,
  // the default reward terms and the yaw-only reset; anything else runs the generic instantiation.
  const gaq_config& c = e->cfg;
  const bool generic = e->force_generic || sc.drag || c.control == GAQ_CTRL_MELLINGER || c.noise == GAQ_NOISE_INPUT ||
                       c.reward_mode != GAQ_REW_QUADROTOR || c.obs_flags != 0 || sc.need_act_prev || sc.per_env_goal ||
                       sc.init_random_state || sc.use_acos;
  uint32_t f = c.per_env_params ? gaq::F_PER_ENV : 0u;
  if (generic) f |= gaq::F_GENERIC;
  else {
    if (sc.motor_lag) f |= gaq::F_LAG;
    if (c.noise == GAQ_NOISE_PHILOX) f |= gaq::F_NOISE;
  }
  e->variant = (int)f;
}

size_t lds_bytes(const gaq_env* e) { return (size_t)kBlock * e->obs_dim * sizeof(float); }

int launch_step(gaq_env* e, const float* actions, float* obs, float* reward, uint8_t* done, hipStream_t st) {
  if ((reinterpret_cast<uintptr_t>(actions) & 15) != 0) return fail(GAQ_ERR_INVALID, "actions must be 16-byte aligned");
  if (e->sc.noise == gaq::NOISE_INPUT) {
    if (!e->noise_next) return fail(GAQ_ERR_STATE, "GAQ_NOISE_INPUT: call gaq_set_noise_input_dev before each step");
    e->d.noise_in = e->noise_next;
    e->noise_next = nullptr;
  }
  const dim3 grid((unsigned)((e->d.n + kBlock - 1) / kBlock)), block(kBlock);
  const size_t lds = lds_bytes(e);
#define GAQ_LAUNCH(FEAT) \
  hipLaunchKernelGGL(step_kernel<(FEAT)>, grid, block, lds, st, e->d, e->sc, e->um, actions, obs, reward, done)
  switch (e->variant) {
    case 0: GAQ_LAUNCH(0u); break;
    case 1: GAQ_LAUNCH(1u); break;
    case 2: GAQ_LAUNCH(2u); break;
    case 3: GAQ_LAUNCH(3u); break;
    case 4: GAQ_LAUNCH(4u); break;
    case 5: GAQ_LAUNCH(5u); break;
    case 6: GAQ_LAUNCH(6u); break;
    case 7: GAQ_LAUNCH(7u); break;
    case 8: GAQ_LAUNCH(8u); break;
    default: GAQ_LAUNCH(9u); break;
  }
#undef GAQ_LAUNCH
  HIP_TRY(hipGetLastError());
  e->sc.step_index += 1;
  return GAQ_OK;
}

int launch_reset(gaq_env* e, const uint8_t* mask, int do_reset, float* obs, hipStream_t st) {
  StepCfg sc = e->sc;
  if (do_reset) { e->reset_calls += 1; sc.step_index = e->sc.step_index + (e->reset_calls << 44); }
  const dim3 grid((unsigned)((e->d.n + kBlock - 1) / kBlock)), block(kBlock);
  hipLaunchKernelGGL(reset_kernel, grid, block, lds_bytes(e), st, e->d, sc, mask, do_reset, obs);
  HIP_TRY(hipGetLastError());
  return GAQ_OK;
}

struct Scratch {   // device staging for the host-pointer entry points
  void* p = nullptr;
  ~Scratch() { if (p) (void)hipFree(p); }
  int alloc(size_t bytes) {
    hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
    return e == hipSuccess ? 0 : fail(GAQ_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
  }
};

}  // namespace

extern "C" {

int gaq_abi_version(void) { return GAQ_ABI_VERSION; }
const char* gaq_last_error(void) { return g_err.c_str(); }

int gaq_num_devices(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int gaq_create(const gaq_config* cfg, gaq_env** out) {
  if (!cfg || !out) return fail(GAQ_ERR_INVALID, "null argument");
  if (cfg->struct_size != sizeof(gaq_config) || cfg->abi_version != GAQ_ABI_VERSION)
    return fail(GAQ_ERR_INVALID, "gaq_config size/version mismatch (header vs library)");
  if (cfg->num_envs <= 0) return fail(GAQ_ERR_INVALID, "num_envs must be positive");
  if (cfg->num_envs > (int64_t)1 << 27) return fail(GAQ_ERR_INVALID, "num_envs above 2^27 per handle is not supported");
  if (!(cfg->sim_freq > 0) || cfg->sim_steps <= 0) return fail(GAQ_ERR_INVALID, "sim_freq and sim_steps must be positive");
  if (cfg->ep_len < 0 || cfg->ep_len >= 0xFFFF) return fail(GAQ_ERR_INVALID, "ep_len must be in [0, 65534]");
  if (cfg->control < 0 || cfg->control > 2) return fail(GAQ_ERR_INVALID, "unknown control mode");
  if (cfg->noise < 0 || cfg->noise > 2) return fail(GAQ_ERR_INVALID, "unknown noise mode");
  if (cfg->reward_mode < 0 || cfg->reward_mode > 1) return fail(GAQ_ERR_INVALID, "unknown reward mode");
  if (cfg->obs_flags & ~15) return fail(GAQ_ERR_INVALID, "unknown obs flags");
  if (cfg->control == GAQ_CTRL_MELLINGER && cfg->per_env_params)
    return fail(GAQ_ERR_INVALID, "Mellinger controller needs a uniform model (one inverse jacobian)");
  const double dt = 1.0 / cfg->sim_freq;
  const int period = svd_period_of(dt);
  if (period >= 0xFFFF) return fail(GAQ_ERR_INVALID, "sim_freq too high for the 16-bit SVD counter");
  if (cfg->sim_freq < 50.0) return fail(GAQ_ERR_INVALID, "sim_freq below 50 Hz is outside the rotation series' range");
  if (!cfg->per_env_params && check_model(cfg->model) != GAQ_OK) return GAQ_ERR_INVALID;
  int ndev = gaq_num_devices();
  if (ndev <= 0) return fail(GAQ_ERR_DEVICE, "no HIP device visible: libgaq has no CPU fallback");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(GAQ_ERR_INVALID, "device ordinal out of range");
  HIP_TRY(hipSetDevice(cfg->device));

  gaq_env* e = new (std::nothrow) gaq_env();
  if (!e) return fail(GAQ_ERR_DEVICE, "out of host memory");
  e->cfg = *cfg;
  int D = 18;
  if (cfg->obs_flags & GAQ_OBS_APPEND_H) D += 1;
  if (cfg->obs_flags & GAQ_OBS_APPEND_ACC) D += 3;
  if (cfg->obs_flags & GAQ_OBS_APPEND_ACT) D += 4;
  e->obs_dim = D;
  StepCfg& sc = e->sc;
  std::memset(&sc, 0, sizeof(sc));
  sc.dt = dt; sc.gravity = cfg->gravity;
  sc.room_lo[0] = -cfg->room_size; sc.room_lo[1] = -cfg->room_size; sc.room_lo[2] = 0.0;
  sc.room_hi[0] = cfg->room_size; sc.room_hi[1] = cfg->room_size; sc.room_hi[2] = cfg->room_size;
  sc.goal_default[0] = 0.0; sc.goal_default[1] = 0.0; sc.goal_default[2] = 2.0;   // quadrotor.py:1081
  sc.init_box = 2.0;                                                               // :728
  sc.sim_steps = cfg->sim_steps; sc.ep_len = cfg->ep_len; sc.svd_period = period;
  sc.control = cfg->control; sc.noise = cfg->noise; sc.reward_mode = cfg->reward_mode;
  sc.obs_flags = cfg->obs_flags; sc.obs_dim = D;
  static_assert(sizeof(gaq::RewCoeff) == sizeof(gaq_rew_coeff), "reward coefficient layout");
  std::memcpy(&sc.rew, &cfg->rew, sizeof(sc.rew));
  sc.need_act_prev = ((cfg->obs_flags & GAQ_OBS_APPEND_ACT) || cfg->rew.action_change != 0.0f) ? 1 : 0;
  sc.per_env_goal = cfg->resample_goal ? 1 : 0;
  sc.auto_reset = cfg->auto_reset ? 1 : 0;
  sc.init_random_state = cfg->init_random_state ? 1 : 0;
  sc.use_acos = (cfg->rew.rot != 0.0f || cfg->rew.attitude != 0.0f) ? 1 : 0;
  sc.seed = cfg->seed; sc.step_index = 0; sc.env_offset = (uint64_t)cfg->env_id_offset;

  if (!cfg->per_env_params) {
    derive_model(cfg->model, dt, e->um);
    e->any_lag = !(e->um.tau_up >= 1.0 && e->um.tau_down >= 1.0);
    e->any_drag = (cfg->model.c_drag != 0.0 || cfg->model.c_roll != 0.0);
    if (cfg->control == GAQ_CTRL_MELLINGER) {
      // quadrotor_jacobian (quadrotor_control.py:192-203) and its inverse (:290-291), Gauss-Jordan in fp64
      double J[4][8];
      const double ccw[4] = {-1, 1, -1, 1};
      for (int c = 0; c < 4; ++c) {
        J[0][c] = cfg->model.thrust_max[c] / cfg->model.mass;
        J[1][c] = (1.0 / cfg->model.inertia[0]) * (cfg->model.thrust_max[c] * cfg->model.prop_pos[3 * c + 1]);
        J[2][c] = (1.0 / cfg->model.inertia[1]) * (cfg->model.thrust_max[c] * -cfg->model.prop_pos[3 * c]);
        J[3][c] = (1.0 / cfg->model.inertia[2]) * (cfg->model.torque_max[c] * ccw[c]);
        for (int r = 0; r < 4; ++r) J[r][4 + c] = (r == c) ? 1.0 : 0.0;
      }
      for (int col = 0; col < 4; ++col) {
        int piv = col;
        for (int r = col + 1; r < 4; ++r) if (std::fabs(J[r][col]) > std::fabs(J[piv][col])) piv = r;
        if (std::fabs(J[piv][col]) < 1e-300) { delete e; return fail(GAQ_ERR_INVALID, "singular quadrotor jacobian"); }
        for (int c = 0; c < 8; ++c) std::swap(J[col][c], J[piv][c]);
        const double inv = 1.0 / J[col][col];
        for (int c = 0; c < 8; ++c) J[col][c] *= inv;
        for (int r = 0; r < 4; ++r) if (r != col) {
          const double f = J[r][col];
          for (int c = 0; c < 8; ++c) J[r][c] -= f * J[col][c];
        }
      }
      for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) sc.jinv[4 * r + c] = J[r][4 + c];
    }
  } else {
    std::memset(&e->um, 0, sizeof(e->um));
    e->any_lag = true;   // decided when parameters arrive
    e->any_drag = false;
  }
  refresh_feature_flags(e);

  DevPtrs& d = e->d;
  std::memset(&d, 0, sizeof(d));
  d.n = cfg->num_envs;
  d.npad = (cfg->num_envs + kBlock - 1) / kBlock * kBlock;
  hipError_t he = hipSuccess;
  auto alloc0 = [&](void** p, size_t bytes) {
    if (he != hipSuccess) return;
    he = hipMalloc(p, bytes);
    if (he == hipSuccess) he = hipMemset(*p, 0, bytes);
  };
  alloc0((void**)&d.s64, sizeof(double) * kP64 * d.npad);
  alloc0((void**)&d.s32, sizeof(float) * kP32 * d.npad);
  alloc0((void**)&d.ctr, sizeof(uint32_t) * d.npad);
  alloc0((void**)&d.done_count, sizeof(uint32_t) * 2);
  alloc0((void**)&d.nan_count, sizeof(uint32_t));
  if (cfg->compact_done) alloc0((void**)&d.done_list, sizeof(uint32_t) * d.npad);
  if (cfg->per_env_params) {
    double* par = nullptr;
    alloc0((void**)&par, sizeof(double) * kPar * d.npad);
    d.par = par;
    e->host_par.assign((size_t)kPar * d.npad, 0.0);
  }
  if (he == hipSuccess) he = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
  if (he == hipSuccess) he = hipEventCreate(&e->ev0);
  if (he == hipSuccess) he = hipEventCreate(&e->ev1);
  if (he != hipSuccess) {
    std::string msg = std::string("device allocation failed: ") + hipGetErrorString(he);
    gaq_destroy(e);
    return fail(GAQ_ERR_DEVICE, msg);
  }
  // rot planes start as identity so an un-reset env is a valid rigid body
  {
    std::vector<double> ones((size_t)d.npad, 1.0);
    for (int j : {6, 10, 14}) HIP_TRY(hipMemcpy(d.s64 + (size_t)j * d.npad, ones.data(), sizeof(double) * d.npad, hipMemcpyHostToDevice));
    std::vector<float> two((size_t)d.npad, 2.0f);
    HIP_TRY(hipMemcpy(d.s32 + (size_t)14 * d.npad, two.data(), sizeof(float) * d.npad, hipMemcpyHostToDevice));
  }
  *out = e;
  return GAQ_OK;
}

int gaq_destroy(gaq_env* e) {
  if (!e) return GAQ_OK;
  (void)hipSetDevice(e->cfg.device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  (void)hipFree(e->d.s64); (void)hipFree(e->d.s32); (void)hipFree(e->d.ctr);
  (void)hipFree(e->d.done_count); (void)hipFree(e->d.nan_count); (void)hipFree(e->d.done_list);
  (void)hipFree(const_cast<double*>(e->d.par));
  if (e->ev0) (void)hipEventDestroy(e->ev0);
  if (e->ev1) (void)hipEventDestroy(e->ev1);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
  return GAQ_OK;
}

int gaq_obs_dim(const gaq_env* e) { return e ? e->obs_dim : GAQ_ERR_INVALID; }
int64_t gaq_num_envs(const gaq_env* e) { return e ? e->d.n : GAQ_ERR_INVALID; }

int gaq_set_params(gaq_env* e, const gaq_model* models, int64_t first, int64_t count) {
  if (!e || !models) return fail(GAQ_ERR_INVALID, "null argument");
  if (!e->cfg.per_env_params) return fail(GAQ_ERR_STATE, "handle was created with per_env_params = 0");
  if (first < 0 || count < 0 || first + count > e->d.n) return fail(GAQ_ERR_INVALID, "env range out of bounds");
  HIP_TRY(hipSetDevice(e->cfg.device));
  const int64_t np = e->d.npad;
  double* hp = e->host_par.data();
  for (int64_t k = 0; k < count; ++k) {
    if (check_model(models[k]) != GAQ_OK) return GAQ_ERR_INVALID;
    Model<double> m;
    derive_model(models[k], e->sc.dt, m);
    const int64_t i = first + k;
    hp[PP_MASS * np + i] = m.mass; hp[PP_INV_MASS * np + i] = m.inv_mass;
    for (int j = 0; j < 3; ++j) { hp[(PP_INERTIA + j) * np + i] = m.inertia[j]; hp[(PP_INV_INERTIA + j) * np + i] = m.inv_inertia[j]; }
    for (int j = 0; j < 4; ++j) {
      hp[(PP_THRUST_MAX + j) * np + i] = m.thrust_max[j]; hp[(PP_TORQUE_MAX + j) * np + i] = m.torque_max[j];
      hp[(PP_PROP_X + j) * np + i] = m.prop_x[j]; hp[(PP_PROP_Y + j) * np + i] = m.prop_y[j]; hp[(PP_PROP_Z + j) * np + i] = m.prop_z[j];
    }
    hp[PP_TAU_UP * np + i] = m.tau_up; hp[PP_TAU_DOWN * np + i] = m.tau_down; hp[PP_LINEARITY * np + i] = m.linearity;
    hp[PP_ARM * np + i] = m.arm; hp[PP_VEL_DAMP * np + i] = m.vel_damp; hp[PP_DAMP_Q * np + i] = m.damp_omega_q;
    hp[PP_C_DRAG * np + i] = m.c_drag; hp[PP_C_ROLL * np + i] = m.c_roll; hp[PP_OU_SIGMA * np + i] = (double)models[k].ou_sigma;
  }
  HIP_TRY(hipStreamSynchronize(e->stream));
  for (int pl = 0; pl < kPar; ++pl)
    HIP_TRY(hipMemcpy(const_cast<double*>(e->d.par) + (size_t)pl * np + first, hp + (size_t)pl * np + first,
                      sizeof(double) * count, hipMemcpyHostToDevice));
  // a new QuadrotorDynamics starts with since_last_svd = 0 and a fresh OUNoise (quadrotor.py:104, :198)
  {
    std::vector<uint32_t> c((size_t)count);
    HIP_TRY(hipMemcpy(c.data(), e->d.ctr + first, sizeof(uint32_t) * count, hipMemcpyDeviceToHost));
    for (auto& v : c) v &= 0xFFFFu;
    HIP_TRY(hipMemcpy(e->d.ctr + first, c.data(), sizeof(uint32_t) * count, hipMemcpyHostToDevice));
    for (int j = 0; j < 4; ++j) HIP_TRY(hipMemset(e->d.s32 + (size_t)j * np + first, 0, sizeof(float) * count));
  }
  // feature flags over ALL envs of the handle
  bool lag = false, drag = false;
  for (int64_t i = 0; i < e->d.n; ++i) {
    if (!(hp[PP_TAU_UP * np + i] >= 1.0 && hp[PP_TAU_DOWN * np + i] >= 1.0)) lag = true;
    if (hp[PP_C_DRAG * np + i] != 0.0 || hp[PP_C_ROLL * np + i] != 0.0) drag = true;
  }
  e->any_lag = lag; e->any_drag = drag;
  refresh_feature_flags(e);
  return GAQ_OK;
}

int gaq_reset_dev(gaq_env* e, const uint8_t* mask_dev, float* obs_dev, void* stream) {
  if (!e) return fail(GAQ_ERR_INVALID, "null handle");
  HIP_TRY(hipSetDevice(e->cfg.device));
  return launch_reset(e, mask_dev, 1, obs_dev, (hipStream_t)stream);
}

int gaq_reset(gaq_env* e, const uint8_t* mask, float* obs_out) {
  if (!e) return fail(GAQ_ERR_INVALID, "null handle");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipDeviceSynchronize());   // earlier *_dev calls may still be running on the caller's stream
  const int64_t n = e->d.n;
  Scratch dm, dobs;
  if (mask) { if (dm.alloc(n)) return GAQ_ERR_DEVICE; HIP_TRY(hipMemcpyAsync(dm.p, mask, n, hipMemcpyHostToDevice, e->stream)); }
  if (obs_out && dobs.alloc(sizeof(float) * n * e->obs_dim)) return GAQ_ERR_DEVICE;
  int rc = launch_reset(e, mask ? (const uint8_t*)dm.p : nullptr, 1, obs_out ? (float*)dobs.p : nullptr, e->stream);
  if (rc) return rc;
  if (obs_out) HIP_TRY(hipMemcpyAsync(obs_out, dobs.p, sizeof(float) * n * e->obs_dim, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GAQ_OK;
}

int gaq_observe(gaq_env* e, float* obs_out) {
  if (!e || !obs_out) return fail(GAQ_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipDeviceSynchronize());
  const int64_t n = e->d.n;
  Scratch dobs;
  if (dobs.alloc(sizeof(float) * n * e->obs_dim)) return GAQ_ERR_DEVICE;
  int rc = launch_reset(e, nullptr, 0, (float*)dobs.p, e->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(obs_out, dobs.p, sizeof(float) * n * e->obs_dim, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GAQ_OK;
}

int gaq_step_dev(gaq_env* e, const float* actions, float* obs, float* reward, uint8_t* done, void* stream) {
  if (!e || !actions || !obs || !reward || !done) return fail(GAQ_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  hipStream_t st = (hipStream_t)stream;
  if (e->timing) HIP_TRY(hipEventRecord(e->ev0, st));
  int rc = launch_step(e, actions, obs, reward, done, st);
  if (rc) return rc;
  if (e->timing) { HIP_TRY(hipEventRecord(e->ev1, st)); e->timed = true; }
  return GAQ_OK;
}

int gaq_step_many_dev(gaq_env* e, int32_t T, const float* actions, float* obs, float* reward, uint8_t* done, void* stream) {
  if (!e || !actions || !obs || !reward || !done) return fail(GAQ_ERR_INVALID, "null argument");
  if (T <= 0) return fail(GAQ_ERR_INVALID, "T must be positive");
  if (e->sc.noise == gaq::NOISE_INPUT) return fail(GAQ_ERR_INVALID, "step_many does not support GAQ_NOISE_INPUT");
  HIP_TRY(hipSetDevice(e->cfg.device));
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = e->d.n;
  if (e->timing) HIP_TRY(hipEventRecord(e->ev0, st));
  for (int32_t t = 0; t < T; ++t) {
    int rc = launch_step(e, actions + (size_t)t * n * 4, obs + (size_t)t * n * e->obs_dim, reward + (size_t)t * n,
                         done + (size_t)t * n, st);
    if (rc) return rc;
  }
  if (e->timing) { HIP_TRY(hipEventRecord(e->ev1, st)); e->timed = true; }
  return GAQ_OK;
}

int gaq_step(gaq_env* e, const float* actions, float* obs, float* reward, uint8_t* done) {
  if (!e || !actions || !obs || !reward || !done) return fail(GAQ_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipDeviceSynchronize());
  const int64_t n = e->d.n;
  const int D = e->obs_dim;
  Scratch da, dobs, dr, dd;
  if (da.alloc(sizeof(float) * 4 * n) || dobs.alloc(sizeof(float) * D * n) || dr.alloc(sizeof(float) * n) || dd.alloc(n))
    return GAQ_ERR_DEVICE;
  HIP_TRY(hipMemcpyAsync(da.p, actions, sizeof(float) * 4 * n, hipMemcpyHostToDevice, e->stream));
  int rc = gaq_step_dev(e, (const float*)da.p, (float*)dobs.p, (float*)dr.p, (uint8_t*)dd.p, e->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(obs, dobs.p, sizeof(float) * D * n, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipMemcpyAsync(reward, dr.p, sizeof(float) * n, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipMemcpyAsync(done, dd.p, n, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GAQ_OK;
}

int gaq_set_noise_input_dev(gaq_env* e, const float* normals_dev) {
  if (!e || !normals_dev) return fail(GAQ_ERR_INVALID, "null argument");
  if (e->sc.noise != gaq::NOISE_INPUT) return fail(GAQ_ERR_STATE, "handle was not created with GAQ_NOISE_INPUT");
  e->noise_next = normals_dev;
  return GAQ_OK;
}

int gaq_get_state(gaq_env* e, double* hp) {
  if (!e || !hp) return fail(GAQ_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  HIP_TRY(hipDeviceSynchronize());
  const int64_t n = e->d.n, np = e->d.npad;
  std::vector<double> a((size_t)kP64 * np);
  std::vector<float> b((size_t)kP32 * np);
  std::vector<uint32_t> c((size_t)np);
  HIP_TRY(hipMemcpy(a.data(), e->d.s64, sizeof(double) * a.size(), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(b.data(), e->d.s32, sizeof(float) * b.size(), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(c.data(), e->d.ctr, sizeof(uint32_t) * c.size(), hipMemcpyDeviceToHost));
  for (int pl = 0; pl < kP64; ++pl) for (int64_t i = 0; i < n; ++i) hp[(size_t)pl * n + i] = a[(size_t)pl * np + i];
  // fp32 planes: device order ou, cmds_damp, act_prev, goal -> ABI order cmds_damp(22), ou(26), act_prev(30), goal(34)
  const int map32[kP32] = {26, 27, 28, 29, 22, 23, 24, 25, 30, 31, 32, 33, 34, 35, 36};
  for (int pl = 0; pl < kP32; ++pl) for (int64_t i = 0; i < n; ++i) hp[(size_t)map32[pl] * n + i] = (double)b[(size_t)pl * np + i];
  for (int64_t i = 0; i < n; ++i) { hp[(size_t)37 * n + i] = (double)(c[i] & 0xFFFFu); hp[(size_t)38 * n + i] = (double)(c[i] >> 16); }
  return GAQ_OK;
}

int gaq_set_state(gaq_env* e, const double* hp) {
  if (!e || !hp) return fail(GAQ_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  HIP_TRY(hipDeviceSynchronize());
  const int64_t n = e->d.n, np = e->d.npad;
  std::vector<double> a((size_t)kP64 * np, 0.0);
  std::vector<float> b((size_t)kP32 * np, 0.0f);
  std::vector<uint32_t> c((size_t)np, 0u);
  for (int pl = 0; pl < kP64; ++pl) for (int64_t i = 0; i < n; ++i) a[(size_t)pl * np + i] = hp[(size_t)pl * n + i];
  const int map32[kP32] = {26, 27, 28, 29, 22, 23, 24, 25, 30, 31, 32, 33, 34, 35, 36};
  for (int pl = 0; pl < kP32; ++pl) for (int64_t i = 0; i < n; ++i) b[(size_t)pl * np + i] = (float)hp[(size_t)map32[pl] * n + i];
  for (int64_t i = 0; i < n; ++i) {
    const double t = hp[(size_t)37 * n + i], s = hp[(size_t)38 * n + i];
    if (!(t >= 0 && t <= 65535 && s >= 0 && s <= 65535)) return fail(GAQ_ERR_INVALID, "tick / SVD counter out of range");
    c[i] = ((uint32_t)t & 0xFFFFu) | ((uint32_t)s << 16);
  }
  HIP_TRY(hipMemcpy(e->d.s64, a.data(), sizeof(double) * a.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->d.s32, b.data(), sizeof(float) * b.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->d.ctr, c.data(), sizeof(uint32_t) * c.size(), hipMemcpyHostToDevice));
  return GAQ_OK;
}

int gaq_done_list(gaq_env* e, uint32_t* idx_out, int64_t capacity, int64_t* count_out) {
  if (!e || !count_out) return fail(GAQ_ERR_INVALID, "null argument");
  if (!e->d.done_list) return fail(GAQ_ERR_STATE, "handle was created with compact_done = 0");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  HIP_TRY(hipDeviceSynchronize());
  if (e->sc.step_index == 0) { *count_out = 0; return GAQ_OK; }
  uint32_t cnt = 0;
  HIP_TRY(hipMemcpy(&cnt, e->d.done_count + ((e->sc.step_index - 1) & 1), sizeof(uint32_t), hipMemcpyDeviceToHost));
  *count_out = cnt;
  if (idx_out && cnt) {
    const int64_t m = cnt < capacity ? cnt : capacity;
    HIP_TRY(hipMemcpy(idx_out, e->d.done_list, sizeof(uint32_t) * m, hipMemcpyDeviceToHost));
  }
  return GAQ_OK;
}

int gaq_nan_count(gaq_env* e, int64_t* count_out) {
  if (!e || !count_out) return fail(GAQ_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  HIP_TRY(hipDeviceSynchronize());
  uint32_t c = 0;
  HIP_TRY(hipMemcpy(&c, e->d.nan_count, sizeof(uint32_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(e->d.nan_count, 0, sizeof(uint32_t)));
  *count_out = c;
  return GAQ_OK;
}

int gaq_set_timing(gaq_env* e, int32_t enabled) {
  if (!e) return fail(GAQ_ERR_INVALID, "null handle");
  e->timing = enabled != 0; e->timed = false;
  return GAQ_OK;
}

int gaq_last_kernel_ms(gaq_env* e, float* ms_out) {
  if (!e || !ms_out) return fail(GAQ_ERR_INVALID, "null argument");
  if (!e->timed) return fail(GAQ_ERR_STATE, "no timed launch recorded (gaq_set_timing)");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipEventSynchronize(e->ev1));
  HIP_TRY(hipEventElapsedTime(ms_out, e->ev0, e->ev1));
  return GAQ_OK;
}

void* gaq_stream(gaq_env* e) { return e ? (void*)e->stream : nullptr; }

int gaq_synchronize(gaq_env* e) {
  if (!e) return fail(GAQ_ERR_INVALID, "null handle");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GAQ_OK;
}

}  // extern "C"
