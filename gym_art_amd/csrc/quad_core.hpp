// quad_core.hpp -- per-environment quadrotor step arithmetic (one lane = one env).
//
// This is the body of the fused HIP kernel in gaq_kernels.hpp.  It is written as
// plain templated C++ with a host/device qualifier macro so that the very same
// arithmetic can be compiled by g++ into the numerics / sanitizer harness under
// tests/host_harness (test infrastructure; the product only ever runs it on the GPU).
//
// What it restates (reference = amolchanov86/gym_art, gym_art/quadrotor/):
//   step1()            quadrotor.py:273-436   QuadrotorDynamics.step1
//   raw_control()      quadrotor_control.py:72-92   RawControl
//   mellinger()        quadrotor_control.py:315-362 NonlinearPositionController.step
//   reward()           quadrotor.py:544-638 ; quadrotor_multi/quadrotor_multi.py:550-650
//   pack_obs()         get_state.py:5,134,147,219,236,249; sense_noise(): sensor_noise.py:100-168
//   reset_env()        quadrotor.py:1059-1144 QuadrotorEnv._reset
//
// Numerics: T is the arithmetic type of the integrator chain torque -> omega -> R ->
// vel -> pos.  The shipped kernels use T = double: four chained integrators amplify
// fp32 rounding to 1e-5..1e-4 over 500 steps (DESIGN.md "Numerics"),
// and at ~1 kflop per 350 B the path stays HBM-bound with fp64 VALU math.
// Reward, OU noise and the packed observation are fp32 (outputs, not fed back).
#pragma once

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define GAQ_HD __host__ __device__ __forceinline__
#else
#define GAQ_HD inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// hardware transcendental units (v_sin_f32 / v_cos_f32 / v_log_f32): only used for the fp32 noise draws
#define GAQ_SINF(x) __sinf(x)
#define GAQ_COSF(x) __cosf(x)
#define GAQ_LOGF(x) __logf(x)
// v_sqrt_f32 alone (1 ulp) instead of sqrtf's correctly rounded expansion (~12 instructions): the norms of the REWARD, an fp32 output
// nothing feeds back from -- 1.2e-7 relative on terms that are multiplied by dt before they are summed (tolerance of every reward check: 1e-6)
#ifndef GAQ_REWARD_SQRT_FAST
#define GAQ_REWARD_SQRT_FAST 1      // (0: A/B builds with sqrtf)
#endif
#if GAQ_REWARD_SQRT_FAST
#define GAQ_SQRTF_OUT(x) __builtin_amdgcn_sqrtf(x)
#else
#define GAQ_SQRTF_OUT(x) sqrtf(x)
#endif
#else
#define GAQ_SINF(x) sinf(x)
#define GAQ_COSF(x) cosf(x)
#define GAQ_LOGF(x) logf(x)
#define GAQ_SQRTF_OUT(x) sqrtf(x)
#endif

// tools/isa_hist.py --hot: a measurement-only build (never linked into the library) in which the cold code of an env step -- the
// in-kernel reset of a finished env -- is compiled out, so that the STATIC instruction histogram of the kernel approximates the
// instructions a wave executes on an ordinary step.
// Box-Muller on the bare hardware units (see box_muller below); 0 = through the libm-style wrappers (round 3's form, kept for A/B builds:
// profiles/r04_fast_box_muller_ab.txt -- 2^20 envs 49.9 -> 49.0 us, sensor noise 69.4 -> 67.8, CrazyFlie + sensor noise 93.5 -> 89.6,
// 65 536 envs 7.3 -> 6.9, 131 072 envs 8.67 -> 8.28)
#ifndef GAQ_FAST_BM
#define GAQ_FAST_BM 1
#endif
// The uniform model read FROM MEMORY at the point of use: instead of holding its 36 doubles in kernel-argument SGPRs --
// and, since they do not fit beside everything else, in spill lanes: 400-580 v_readlane / v_writelane per kernel -- for the whole launch, the
// kernels named by kModelMem below read each block's constants by scalar loads from the kernel-argument segment itself, in every block of
// every sub-step (s_load from the scalar cache; the values stay scalar operands, no VGPRs; model_fence() keeps the loads where they are
// used).  Measured at N = 2^20 (profiles/r04_model_from_memory_ab.txt; PMC: VALU instructions per wave 1875 -> 1578 in <1046>): sensor noise
// <1044> 69.4 -> 65.3 us, the 25-word observation 75.2 -> 70.5, the info dict's aux row <66580> 97.8 -> 85.9, the quaternion observation
// 78.5 -> 69.1 -- the packed-observation kernels WITHOUT motor lag.  The lag kernels lose 3-4 % (<1046> 90.7 -> 93.8, <150> 67.8 -> 70.4,
// <16406> 79.1 -> 82.2: their register allocation moves the wrong way, and at two waves per SIMD they are bound by latency, not by VALU issue:
// 16 % fewer VALU instructions left <1046>'s wave lifetime unchanged), the heads-are-the-observation kernels do not care (+-0.5 %).  A per-wave
// LDS copy of the model (broadcast ds_read_b64) instead: <1046> 92 -> 154 us -- dead.  GAQ_MODEL_MEM_OFF=1: A/B builds without it.
// ... and in the GENERIC kernels with a uniform model, whose spill-lane traffic was the largest of all (<520>: 4513 static v_readlane /
// v_writelane, <8> 1177, <584> 1171): the gyro-bias walk <8> 102 -> 90 us, Mellinger + info=True <520> 104 -> 93, info=True on fp64 planes <584>
// 92.7 -> 87, per-env goals <72> 83 -> 81 (profiles/r04_model_from_memory_ab.txt).  0: A/B builds without it.
#ifndef GAQ_MODEL_MEM_GENERIC
#define GAQ_MODEL_MEM_GENERIC 1
#endif
#ifndef GAQ_MODEL_MEM_OFF
#define GAQ_MODEL_MEM_OFF 0
#endif
#ifndef GAQ_PROBE_HOT
#define GAQ_PROBE_HOT 0
#endif

namespace gaq {

// Compile-time feature mask of a kernel instantiation.  Without F_GENERIC only the features named
// by the mask exist in the code (registers!); with F_GENERIC every runtime flag of StepCfg is honoured.
// F_ALIAS (specialised kernels only): the fp64 integrator state is stored split, value = hi + lo with
// hi = fp32 head of value kept IN the caller's observation tensor (the 18 observation words are exactly
// [pos-goal, vel, R, omega]), truncated toward zero, and the next 16 mantissa bits in a library-owned shadow
// array (39 significant bits in all; split_decode below).
enum Feature : uint32_t { F_PER_ENV = 1, F_LAG = 2, F_NOISE = 4, F_GENERIC = 8, F_ALIAS = 16,
                          F_FP32 = 32 /* with F_ALIAS: T = float and the fp32 observation rows ARE the whole state */,
                          F_LITE = 64 /* with F_GENERIC: without Mellinger, rotor drag, injected noise, gyro-bias walk */,
                          F_PREDRAW = 128 /* small batches (<= 2 waves per SIMD, where registers are free and every wave of the
                                             launch waits on its loads at the same time): the OU normals of the first two
                                             sub-steps are drawn by the kernel under the load latency and handed in */,
                          F_NT = 256 /* non-temporal cache policy on the streaming loads / stores of the state (gaq_kernels.hpp kLdAux) */,
                          F_DIAG = 512 /* with the full F_GENERIC: the rarely used extras that would otherwise cost the Mellinger /
                                          drag / bias-walk kernel a wave of occupancy -- aux outputs for the info dict, injected
                                          sensor-noise draws, the quaternion / t2w / t2t observation variants */ };
// F_PACK (with F_ALIAS): the state is stored split like in the alias layouts (fp32 heads + residual rows, library-owned) but the
// observation is NOT the heads -- body frame, appended height / accelerometer / previous action, sensor noise -- and is packed
// explicitly like in the plain-layout kernels.  108 + 108 B of state traffic instead of the fp64 planes' 144 + 144.
enum : uint32_t { F_PACK = 1024 };
// F_RZ (with F_PER_ENV): dynamics_randomize_every handled inside the step launch -- a finished, due env is promoted to the parameter
// planes staged for it (gaq.hip: par_next, refill pass).  A flag of its own so that every other instantiation stays exactly what it was.
enum : uint32_t { F_RZ = 2048 };
// Twins of the alias kernels whose observation is the state's heads (flags of their own so that the kernels everybody runs stay byte for
// byte what they were: the mere presence of either epilogue / prologue, even behind a wave-uniform branch that is never taken, cost the
// headline kernel 4-9 % -- hipcc reschedules the kernel-argument loads around it; profiles/r03_twin_ab.txt):
// F_ROWS: the launch also writes the packed [obs | reward | done] rows of the multi-GPU return path (gaq_set_packed_rows_dev);
// F_CTR:  graph-safe mode at small batches: the launch advances the device-resident step counter itself (gaq_kernels.hpp: step_counter_checkin)
enum : uint32_t { F_ROWS = 4096, F_CTR = 8192 };
// F_MELL: the Mellinger controller (NonlinearPositionController, quadrotor_control.py:315-362) in the SPECIALISED kernels -- uniform model
// (its inverse jacobian rides in the launch constants), the 18-word observation, any state layout.  Round 2 ran every Mellinger
// configuration in the full generic kernel on fp64 planes (237 VGPRs, 100 us per step at N = 2^20); per-env models (one jacobian per env)
// and the observation variants still do.
enum : uint32_t { F_MELL = 16384 };
// F_SWARM (with F_ALIAS | F_PACK): the swarm layer (own specification, SwarmCfg) on the SPLIT state -- library-owned fp32 heads (pos relative
// to the agent's own formation goal) + residual rows instead of the generic kernel's fp64 planes, the neighbour terms by wave shuffles as
// there, the observation rows (18 + 6 (agents - 1) words) packed straight into the wave's LDS buffer.  Uniform model, RawControl.
enum : uint32_t { F_SWARM = 32768 };
// F_AUXP (with F_ALIAS | F_PACK, RawControl; uniform or per-env models): the info dict's aux row (gaq_config.aux_outputs) and the quaternion / t2w / t2t
// observation variants ON THE SPLIT STATE -- they are outputs like the packed observation, nothing of them feeds back.  Round 3 ran them in
// the light generic kernel on fp64 planes (F_LITE | F_DIAG: 218 VGPRs, 94-112 us per step at N = 2^20); here they cost the packed-observation
// kernels their third wave per SIMD and nothing else.
enum : uint32_t { F_AUXP = 65536 };
// F_ENVX (with F_AUXP; F_BIAS: uniform model): the per-env planes that are STATE beside the 18 values, on the split state, by the same wave-uniform
// runtime flags as in the generic kernel -- the goal (resample_goal, excite: quadrotor.py:1078-1081, :957-963; the heads then hold pos - this
// env's goal, as in the swarm kernels) and, with F_BIAS on top, SensorNoise's gyro bias (the random walk of sensor_noise.py:160-168).
// Round 4's first half ran them on fp64 planes (per-env goals: the light generic kernel, 84 us per step at N = 2^20; the bias walk: the
// full one, 95 us).  Flags of their own: the goal code costs the F_AUXP kernels on the 168-VGPR line their third wave, the bias walk's
// (three more Philox blocks per observation) costs the goal kernels theirs (163-164 VGPRs without it, 175-195 with).
enum : uint32_t { F_ENVX = 131072, F_BIAS = 262144 };
template <uint32_t F> constexpr bool kEnvExtras = (F & F_GENERIC) != 0 || (F & F_ENVX) != 0;   // resample_goal / excite honoured by this instantiation
template <uint32_t F> constexpr bool kModelMem = !GAQ_MODEL_MEM_OFF && (((F & 1024u /*F_PACK*/) != 0 && (F & 16u /*F_ALIAS*/) != 0 &&
    (F & (1u /*F_PER_ENV*/ | 2u /*F_LAG*/ | 8u /*F_GENERIC*/ | 32u /*F_FP32*/ | 16384u /*F_MELL*/ | 32768u /*F_SWARM*/)) == 0) ||
    (GAQ_MODEL_MEM_GENERIC && (F & 8u) != 0 && (F & 1u) == 0));
// (model in memory: a compiler-level memory fence -- no instruction -- so that the loads of a block are issued in that block and their
//  registers die with it; hoisted out of the sub-step loop the model would be 70 registers again)
template <uint32_t F> GAQ_HD void model_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (kModelMem<F>) asm volatile("" ::: "memory");
#endif
}
template <uint32_t F> constexpr bool kSwarm = (F & F_GENERIC) != 0 || (F & F_SWARM) != 0;   // the neighbour terms exist in this instantiation
template <uint32_t F> constexpr bool kHeadsAreObs = (F & F_ALIAS) != 0 && (F & F_PACK) == 0;   // nothing to pack: the sink is dead code
template <uint32_t F> constexpr bool kDiag = (F & F_GENERIC) != 0 && (F & F_LITE) == 0 && (F & F_DIAG) != 0;
// the aux row of the info dict (last sub-step's accelerometer / omega_dot / torque, controller output, thrust_cmds_damp) and the quaternion /
// t2w / t2t observation variants: the diagnostics tier, and -- F_LITE | F_DIAG -- the LIGHT generic kernel with nothing but those added: what
// `info=True` or one of those observations on a RawControl batch needs, without the Mellinger / drag / bias-walk / injected-draw code
// that costs the full tier its second wave (255 VGPRs + spills)
template <uint32_t F> constexpr bool kAuxIsRow =      // the aux values go straight into the env's 17-word row in LDS (StepOut::aux_row): every
#if defined(__HIP_DEVICE_COMPILE__)                      // device kernel that has them; the host build keeps them in StepOut's fields
    (F & F_AUXP) != 0 || ((F & F_GENERIC) != 0 && (F & F_DIAG) != 0);
#else
    false;
#endif
template <uint32_t F> constexpr bool kAux = kDiag<F> || ((F & F_GENERIC) != 0 && (F & F_LITE) != 0 && (F & F_DIAG) != 0) || (F & F_AUXP) != 0;

// ---- enums shared with include/gaq.h (kept numerically identical there) ---------------
enum ControlMode { CTRL_RAW_ZERO_MIDDLE = 0, CTRL_RAW = 1, CTRL_MELLINGER = 2 };
enum NoiseMode { NOISE_OFF = 0, NOISE_PHILOX = 1, NOISE_INPUT = 2 };
enum RewardMode { REW_QUADROTOR = 0, REW_MULTI_LOG = 1 };
enum ObsFlags { OBS_BODY_FRAME = 1, OBS_APPEND_H = 2, OBS_APPEND_ACC = 4, OBS_APPEND_ACT = 8,
                // variants that raise NameError in the reference as shipped (missing imports in get_state.py; fixture G15):
                OBS_QUAT = 16 /* quaternion instead of R (:276-322) */, OBS_APPEND_T2W = 32, OBS_APPEND_T2T = 64 /* :325-384 */ };

// reward weights, in the order of the reference's sum (quadrotor.py:593-604)
struct RewCoeff {
  float pos, effort, crash, orient, yaw, rot, attitude, spin, action_change, vel;
  float pos_offset, pos_log_weight, pos_linear_weight;  // multi variant only
};

// SensorNoise parameters (sensor_noise.py:57-99); `enabled` = not bypassed
struct SenseNoise {
  int32_t enabled;
  float pos_norm_std, pos_unif_range, vel_norm_std, vel_unif_range, quat_norm_std, quat_unif_range;
  float gyro_noise_density, acc_static_noise_std, acc_dynamic_noise_ratio;
  // gyro_norm_std != 0 switches the gyro from white noise to the bias random walk of add_noise_to_omega (:160-168)
  float gyro_norm_std, gyro_random_walk, gyro_bias_correlation_time;
};

// Swarm layer (BASELINE config 5).  The reference snapshot holds no multi-agent code (SURVEY header note 2): this is
// the build's OWN specification, parity-unpinned -- DESIGN.md "Swarm layer".  A world = `agents` consecutive envs
// (agents is a power of two <= 16, so a world never straddles a 64-env tile and neighbours are wave shuffles).
struct SwarmCfg {
  int32_t agents;          // 0 / 1: off
  float goal_radius;       // agent a's goal = goal_default + goal_radius (cos, sin)(2 pi a / agents)
  float collision_dist;    // d_ij below this counts as a collision
  float prox_dist;         // proximity penalty falls off linearly to zero at this distance
  float w_collision, w_prox;
  int32_t response;        // 1: colliding agents exchange the normal component of their relative velocity (see swarm_interact)
};

// derived model constants of QuadrotorDynamics.update_model (quadrotor.py:142-208)
template <typename T>
struct Model {
  T mass, inv_mass;
  T inertia[3], inv_inertia[3];
  T thrust_max[4], torque_max[4];
  T prop_x[4], prop_y[4], prop_z[4];
  T tau_up, tau_down;  // 4*dt/(T_up+1e-6), 4*dt/(T_down+1e-6)  (NOT yet min'ed with 1)
  T linearity, arm, vel_damp, damp_omega_q, c_drag, c_roll;
  float ou_sigma;
  const double* jinv;  // Mellinger with per-env models: this env's 4x4 inverse jacobian (row-major), else nullptr -> StepCfg::jinv
};

// per-launch scalars (wave-uniform: every branch on them is a scalar branch)
struct StepCfg {
  double dt;
  double gravity;
  double room_lo[3], room_hi[3];
  double goal_default[3];
  double init_box;          // 2.0 (quadrotor.py:728)
  int32_t sim_steps, ep_len, svd_period;
  int32_t control, noise, reward_mode, obs_flags, obs_dim;
  int32_t motor_lag;        // 0: both taus >= 1 for every env (no motor state kept)
  int32_t drag;             // rotor drag / rolling moment branch present
  int32_t need_act_prev;    // obs has `act` or action_change weight != 0
  int32_t per_env_goal;     // goals differ between envs (resample_goal or excite): a goal plane is kept
  int32_t resample_goal;    // goal z ~ U(0.5, 2) at every reset (quadrotor.py:1078-1081)
  int32_t excite;           // new random goal whenever tick % 5 == 0 (:957-963)
  int32_t auto_reset, init_random_state;
  int32_t use_acos;         // rot / attitude weights != 0
  RewCoeff rew;
  SenseNoise sense;         // observation noise (generic kernel)
  SwarmCfg swarm;           // neighbour reward / observation terms (generic kernel)
  // gyro-bias random walk b <- pi b + sigma n (sensor_noise.py:163-167), per add_noise call and folded over the
  // three calls the reference makes per env step (quadrotor.py:946, :970, :988): pi^3, sigma sqrt(1 + pi^2 + pi^4)
  int32_t compact_params;   // per-env parameters: every env's torque_max / prop_pos follow the reference's construction
                            // (t2t * thrust_max, +-motor_xy - com), so the kernel rebuilds them from 5 planes instead of loading 12
  int32_t zero_damp;        // per-env parameters: vel_damp and damp_omega_quadratic are zero for EVERY env of the handle (all
                            // shipped models and their perturbations): the two planes are not loaded
  int32_t action_f32;       // RawControl called with float32 action ARRAYS: the reference then computes 0.5*(a+1) and the clip
                            // in float32 (quadrotor_control.py:88-92); 0 = float64 arrays holding the same values
  int32_t sense_input;      // sensor-noise draws come from the caller (gaq_set_sense_input_dev) instead of Philox
  int32_t aux;              // keep the last sub-step's accelerometer / omega_dot / torque, the controller output and
                            // thrust_cmds_damp for the info dict (quadrotor.py:994-1006); generic kernel only
  int32_t ablate;           // diagnostics (GAQ_ABLATE=1): skip the arithmetic, the state passes through -- times the kernel's data path
  int32_t gyro_bias;        // the bias model is on (sense.enabled && sense.gyro_norm_std != 0)
  float gyro_pi, gyro_sigma, gyro_pi_step, gyro_sigma_step;
  // t2w / t2t observation components (quadrotor.py:706-712; get_state.py:335-338): relative noise std, clip = scaling range
  float t2w_std, t2w_min, t2w_max, t2t_std, t2t_min, t2t_max;
  double jinv[16];          // Mellinger: inverse jacobian (quadrotor_control.py:290-291)
  uint64_t seed, step_index, env_offset;
};

template <uint32_t F> GAQ_HD bool has_lag(const StepCfg& c) { if constexpr ((F & F_GENERIC) != 0) return c.motor_lag != 0; else return (F & F_LAG) != 0; }
template <uint32_t F> GAQ_HD int noise_mode(const StepCfg& c) {
  if constexpr ((F & F_GENERIC) != 0 && (F & F_LITE) != 0) return c.noise == NOISE_PHILOX ? NOISE_PHILOX : NOISE_OFF;
  else if constexpr ((F & F_GENERIC) != 0) return c.noise;
  else return (F & F_NOISE) ? NOISE_PHILOX : NOISE_OFF;
}
// previous-action plane (`_act` observations, action-change reward term): generic and specialised plain-layout kernels
template <uint32_t F> GAQ_HD bool has_act_prev(const StepCfg& c) { if constexpr (kHeadsAreObs<F>) return false; else return c.need_act_prev != 0; }
template <uint32_t F> GAQ_HD bool has_env_goal(const StepCfg& c) {
  if constexpr ((F & F_GENERIC) != 0 || (F & F_ENVX) != 0) return c.per_env_goal != 0; else return (F & F_SWARM) != 0;   // (formation goals: one per agent)
}
template <uint32_t F> GAQ_HD int swarm_agents(const StepCfg& c) { if constexpr (kSwarm<F>) return c.swarm.agents; else return 0; }
template <uint32_t F> constexpr bool kGyroBias = ((F & F_GENERIC) != 0 && (F & F_LITE) == 0) || (F & F_BIAS) != 0;   // the bias plane exists here
template <uint32_t F> GAQ_HD bool has_gyro_bias(const StepCfg& c) { if constexpr (kGyroBias<F>) return c.gyro_bias != 0; else return false; }

template <typename T>
struct EnvState {
  T pos[3], vel[3], rot[9], omega[3];
  T rot_damp[4];      // thrust_rot_damp (noise-free motor filter state)
  float cmds_damp[4]; // thrust_cmds_damp (noised, clipped): only feeds the up/down tau choice
  float ou[4];        // OUNoise.state
  float act_prev[4];  // env.actions[0] of the previous step
  T goal[3];
  float gyro_bias[3]; // SensorNoise.gyro_bias (sensor_noise.py:98): survives resets, like the object holding it
  uint32_t tick, svd_ctr;
};

struct StepOut {
  float reward;
  uint8_t done, crashed;
  float acc_meter[3];
  // info-dict extras of the LAST sub-step (cfg.aux, generic kernel): dynamics.omega_dot, dynamics.torque, controller.action,
  // dynamics.thrust_cmds_damp (quadrotor.py:994-1006)
  float omega_dot[3], torque[3], ctrl[4], cmds[4];
  // device kernels with the aux row (kAuxIsRow): this env's 17-word aux row (AUX_* order) in the wave's LDS buffer -- the values are stored where they are formed
  // instead of living in 17 registers from the last sub-step to the end of the env step (they decide that kernel's occupancy)
  float* aux_row = nullptr;
};
enum { AUX_ACC = 0, AUX_OMEGA_DOT = 3, AUX_TORQUE = 6, AUX_CTRL = 9, AUX_CMDS = 13, AUX_WORDS = 17 };

// ---- small math --------------------------------------------------------------------
// clamp by min/max (v_max_f64 + v_min_f64: 2 instructions instead of 2 compares + 4 selects).  Unlike
// np.clip this maps NaN to `lo`; env_step therefore poisons the reward when its inputs are not finite,
// which is where the reference's NaN check (quadrotor.py:633-636) looks.
GAQ_HD double clampv(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }
GAQ_HD float clampv(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
GAQ_HD double sqrt_t(double x) { return sqrt(x); }
GAQ_HD float sqrt_t(float x) { return sqrtf(x); }

// ---- split state: the observation word carries the top 24 bits of a state value, 16 more ride beside it ----
// hi = value truncated toward zero to fp32 (the observation word, within one fp32 ulp of the value), q = the
// next 16 bits of the value's fp64 mantissa.  (double)hi has those bits zero, so decoding is one OR into the low
// word of the converted double and encoding is a rounding fix-up plus a bit-field extract -- ~10 instructions per
// value for both directions (a scaled-residual format cost ~22).  39 significant bits: relative error <= 2^-39 =
// 1.8e-12 per store, against which the fp32-state drift of DESIGN.md "Numerics" shrinks to ~1e-8 over 500 steps.
GAQ_HD double split_decode(float hi, uint32_t q) {
  const uint64_t b = __builtin_bit_cast(uint64_t, (double)hi) | ((uint64_t)(q & 0xFFFFu) << 13);
  return __builtin_bit_cast(double, b);
}
GAQ_HD float split_hi(double v) {
  const float h = (float)v;                                 // round to nearest ...
  uint32_t hb = __builtin_bit_cast(uint32_t, h);
  if (fabs((double)h) > fabs(v)) hb -= 1u;                  // ... then one ulp back toward zero if it rounded away
  return __builtin_bit_cast(float, hb);                     // (NaN compares false and stays NaN)
}
// 32-bit residual: all 29 mantissa bits the fp32 head does not hold -> exact
GAQ_HD double split_decode32(float hi, uint32_t q) {
  return __builtin_bit_cast(double, __builtin_bit_cast(uint64_t, (double)hi) | (uint64_t)(q & 0x1FFFFFFFu));
}
GAQ_HD uint32_t split_lo32(double v) {
  return (uint32_t)__builtin_bit_cast(uint64_t, v) & 0x1FFFFFFFu;
}
GAQ_HD uint32_t split_lo(double v) {
  return (uint32_t)(__builtin_bit_cast(uint64_t, v) >> 13) & 0xFFFFu;
}

// sin(t)/t and (1-cos t)/t^2 as series in q = t^2, 10 terms (2^10/21! = 2e-17 at q = 2, i.e. |omega| dt <= 1.41 rad
// per sub-step; omega is clipped to 40 rad/s per axis (quadrotor.py:91,405), so every sim_freq >= 50 Hz is in range,
// checked in gaq_create).  No sqrt, division or trig in the Rodrigues update, and exact at omega == 0 where the
// reference skips it (quadrotor.py:373).
// Terms k >= 4 are summed in fp32: they enter as q^4 * tail with q^4 a_4 <= 4e-5 (50 Hz) / 6e-10 (200 Hz), so the
// fp32 rounding of the tail is <= 2e-12 / 4e-17 of A -- and fp32 constants are instruction literals, whereas every
// fp64 constant occupies a register pair for the whole sub-step loop (it cost 36 VGPRs and a wave of occupancy).
template <typename T>
GAQ_HD void sinc_cosc(T q, T& A, T& B) {
  // a_k = (-1)^k/(2k+1)!, b_k = (-1)^k/(2k+2)!
  const float qf = (float)q;
  float ta = -8.2206352466243297e-18f;             // a9
  ta = fmaf(ta, qf, 2.8114572543455206e-15f);      // a8
  ta = fmaf(ta, qf, -7.6471637318198164e-13f);     // a7
  ta = fmaf(ta, qf, 1.6059043836821613e-10f);      // a6
  ta = fmaf(ta, qf, -2.5052108385441720e-08f);     // a5
  ta = fmaf(ta, qf, 2.7557319223985893e-06f);      // a4
  float tb = -4.1103176233121648e-19f;             // b9
  tb = fmaf(tb, qf, 1.5619206968586225e-16f);      // b8
  tb = fmaf(tb, qf, -4.7794773323873853e-14f);     // b7
  tb = fmaf(tb, qf, 1.1470745597729725e-11f);      // b6
  tb = fmaf(tb, qf, -2.0876756987868100e-09f);     // b5
  tb = fmaf(tb, qf, 2.7557319223985888e-07f);      // b4
  A = T(1.0) + q * (T(-1.0 / 6) + q * (T(1.0 / 120) + q * (T(-1.0 / 5040) + q * T(ta))));
  B = T(0.5) + q * (T(-1.0 / 24) + q * (T(1.0 / 720) + q * (T(-1.0 / 40320) + q * T(tb))));
}

// Philox4x32-10 (Salmon et al. 2011), counter-based: stateless per (env, step, stream).
struct Philox {
  uint32_t c[4];
  GAQ_HD static void round1(uint32_t* c, uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
  }
  GAQ_HD Philox(uint64_t seed, uint64_t env, uint64_t step, uint32_t stream) {
    c[0] = (uint32_t)env; c[1] = (uint32_t)(env >> 32) ^ (stream << 24);
    c[2] = (uint32_t)step; c[3] = (uint32_t)(step >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      round1(c, k0, k1);
      k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
  }
  GAQ_HD double u01(int i) const { return ((double)c[i] + 0.5) * (1.0 / 4294967296.0); }  // (0,1)
};
enum RngStream { RNG_OU0 = 0 /* + substep */, RNG_RESET_A = 64, RNG_RESET_B = 65, RNG_RESET_C = 66, RNG_RESET_D = 67,
                 RNG_SENSE0 = 100 /* .. 108 */, RNG_EXCITE = 120 };

// 4 standard normals from one Philox block (Box-Muller, fp32: they only drive the OU noise)
GAQ_HD void box_muller(uint32_t b1, uint32_t b2, float& n0, float& n1) {   // two 24-bit integers -> two standard normals
  const float u1 = ((float)b1 + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = ((float)b2 + 0.5f) * (1.0f / 16777216.0f);
#if defined(__HIP_DEVICE_COMPILE__) && GAQ_FAST_BM
  // the hardware units directly: v_log_f32 is log2 (-2 ln u = -2 ln2 log2 u), v_sqrt_f32 without the IEEE fix-up sequence (1 ulp; u1 >= 2^-25,
  // so the argument is a normal number in [6e-8, 34.7]), v_sin_f32 / v_cos_f32 take their argument in REVOLUTIONS (u2 itself): 11 instructions
  // per pair instead of 27
  const float r = __builtin_amdgcn_sqrtf(-1.38629436111989061883f * __builtin_amdgcn_logf(u1));
  n0 = r * __builtin_amdgcn_cosf(u2);
  n1 = r * __builtin_amdgcn_sinf(u2);
#else
  const float r = sqrtf(-2.0f * GAQ_LOGF(u1));
  const float a = 6.28318530717958647692f * u2;
  n0 = r * GAQ_COSF(a);
  n1 = r * GAQ_SINF(a);
#endif
}
GAQ_HD void normals4(const Philox& p, float n[4]) {
#pragma unroll
  for (int h = 0; h < 2; ++h) box_muller(p.c[2 * h] >> 8, p.c[2 * h + 1] >> 8, n[2 * h], n[2 * h + 1]);
}

// 10 standard normals from TWO Philox blocks: ten 24-bit uniforms out of the 256 bits -- the top 24 bits of the eight words, and two more
// from the low bytes of six of them -- into five Box-Muller pairs.  (The sensor-noise model of SensorNoise() needs nine normals per call;
// one block per four of them was three blocks, and a Philox block is ~5 % of a whole env step's instructions.)
GAQ_HD void normals10(const Philox& a, const Philox& b, float n[10]) {
  const uint32_t w[8] = {a.c[0], a.c[1], a.c[2], a.c[3], b.c[0], b.c[1], b.c[2], b.c[3]};
  uint32_t u[10];
#pragma unroll
  for (int k = 0; k < 8; ++k) u[k] = w[k] >> 8;
  u[8] = (w[0] & 0xFFu) | ((w[1] & 0xFFu) << 8) | ((w[2] & 0xFFu) << 16);
  u[9] = (w[3] & 0xFFu) | ((w[4] & 0xFFu) << 8) | ((w[5] & 0xFFu) << 16);
#pragma unroll
  for (int h = 0; h < 5; ++h) box_muller(u[2 * h], u[2 * h + 1], n[2 * h], n[2 * h + 1]);
}

// ---- controllers --------------------------------------------------------------------
// RawControl.step (quadrotor_control.py:88-92).  NB the zero-middle variant clips to
// [-1, 1] (low = -ones, :82), the dynamics re-clip to [0, 1] (quadrotor.py:279).
// `f32` = the reference was handed a float32 ARRAY: `self.scale * (action + self.bias)` is then float32 arithmetic
// (0.5f * (a + 1.0f), the sum rounded to 24 bits) and the clip against the float64 bounds widens the result; with a
// float64 array holding the same values the sum is exact.  The two differ by up to 6e-8 in the command.
template <typename T>
GAQ_HD void raw_control(const float a[4], int mode, T cmd[4], bool f32 = false) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (mode == CTRL_RAW_ZERO_MIDDLE) {
      const T t = f32 ? T(0.5f * (a[i] + 1.0f)) : T(0.5) * (T(a[i]) + T(1));
      cmd[i] = clampv(t, T(-1), T(1));
    } else {
      cmd[i] = clampv(T(a[i]), T(0), T(1));
    }
  }
}

template <typename T> GAQ_HD void cross3(const T a[3], const T b[3], T o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
// the same with every product rounded before the subtraction, like NumPy's cross (used in the rotational subsystem)
template <typename T> GAQ_HD void cross3_nofma(const T a[3], const T b[3], T o[3]) {
#pragma clang fp contract(off)
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
// quad_utils.py:35-41: returns the vector unchanged when its norm is < 1e-5
template <typename T> GAQ_HD void normalize3(T v[3]) {
  const T n = sqrt_t(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  // (one division and three products instead of NumPy's three divisions: an fp64 division is eleven instructions around a 12-cycle
  //  v_rcp_f64; the quotients differ from v / n in the last fp64 bit at most, nine orders below the fp32 observation they end in)
  if (!(n < T(0.00001))) { const T inv = T(1) / n; v[0] = v[0] * inv; v[1] = v[1] * inv; v[2] = v[2] * inv; }
}

// NonlinearPositionController.step (quadrotor_control.py:315-362); gains :299-300
template <typename T>
GAQ_HD void mellinger(const EnvState<T>& s, const StepCfg& cfg, const double* jinv_env, T cmd[4], bool first_after_reset = false) {
  const double* jinv = jinv_env ? jinv_env : cfg.jinv;
  T tg[3] = {s.goal[0] - s.pos[0], s.goal[1] - s.pos[1], s.goal[2] - s.pos[2]};
  const T n = sqrt_t(tg[0] * tg[0] + tg[1] * tg[1] + tg[2] * tg[2]);
  const T sc = (n <= T(4)) ? T(1) : T(4) / n;   // clamp_norm (quad_utils.py:63-67)
  T acc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) acc[i] = T(-4.5) * (-(sc * tg[i])) - T(3.5) * s.vel[i];
  acc[2] += T(9.81);
  T zb[3] = {acc[0], acc[1], acc[2]};
  normalize3(zb);
  const T xc[3] = {T(1), T(0), T(0)};
  T yb[3]; cross3(zb, xc, yb); normalize3(yb);
  T xb[3]; cross3(yb, zb, xb);
  // R_des = [xb yb zb] (columns); e_R = 0.5 vee(R_des^T R - R^T R_des)
  const T* R = s.rot;
  const T* col[3] = {xb, yb, zb};
  T M[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      // (R_des^T R)[i][j] = sum_k col_i[k] R[k][j] ; (R^T R_des)[i][j] = sum_k R[k][i] col_j[k]
      M[3 * i + j] = (col[i][0] * R[0 + j] + col[i][1] * R[3 + j] + col[i][2] * R[6 + j]) -
                     (R[0 + i] * col[j][0] + R[3 + i] * col[j][1] + R[6 + i] * col[j][2]);
    }
  T eR[3] = {T(0.5) * M[7], T(0.5) * M[2], T(0.5) * M[3]};
  eR[2] *= T(0.2);
  T des[4];
  des[0] = acc[0] * R[2] + acc[1] * R[5] + acc[2] * R[8];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    // dynamics.omega is a float32 array until the first step1 after set_state (quadrotor.py:223): `kd_a * e_w` (:350) is
    // then a float32 product
    const T kdw = first_after_reset ? T(50.0f * (float)s.omega[i]) : T(50) * s.omega[i];
    des[1 + i] = T(-200) * eR[i] - kdw;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const T t = T(jinv[4 * i]) * des[0] + T(jinv[4 * i + 1]) * des[1] + T(jinv[4 * i + 2]) * des[2] + T(jinv[4 * i + 3]) * des[3];
    cmd[i] = clampv(t, T(0), T(1));
  }
}

// ---- polar factor of a near-orthogonal 3x3 (what U @ Vt of np.linalg.svd returns, quadrotor.py:384-385)
template <typename T>
GAQ_HD void polar3(T R[9]) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {   // Newton X <- (X + X^-T)/2, quadratic: 1e-7 -> 1e-14 -> exact
    T c[9];
    cross3(R + 3, R + 6, c);       // cofactor rows
    cross3(R + 6, R + 0, c + 3);
    cross3(R + 0, R + 3, c + 6);
    const T det = R[0] * c[0] + R[1] * c[1] + R[2] * c[2];
    const T h = T(0.5) / det;
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = T(0.5) * R[i] + h * c[i];
  }
}

// ---- the rotational subsystem's two arithmetic blocks, in two roundings -------------------------------------
// Motor filter -> thrusts -> torque -> Euler's equations is the part of the dynamics that can be chaotic -- with motor
// lag a tumbling quad amplifies ONE ulp by up to ~5e9 over 500 steps (tools/chaos_baseline.py) -- and it is closed: R
// and the translation only integrate its output.  EXACT = no fused multiply-add contraction, operations in the
// reference's order: bit-identical to NumPy (which has no FMA), so no episode can part from the reference, however
// chaotic (tools/oracle_drift.py: all 4096 randomised-CrazyFlie episodes within 6e-8, where the contracted build left
// 0.07 % of them 1e-6..1e-4 away).  Kernels that can see motor lag use it.  Without lag the amplification is ~1e4 and
// the contracted form (2.4 % faster on the whole kernel) holds 6e-8 on every episode: the uniform no-lag kernels keep it.
#define GAQ_TORQUE_LOOP                                                                                              \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                    \
    /* linearity == 1 (every shipped model): (1-1) c^2 + 1 c = c exactly, so the short form is bit-identical */      \
    const T th = (m.linearity == T(1)) ? m.thrust_max[i] * c[i]                                                      \
                                       : m.thrust_max[i] * ((T(1) - m.linearity) * (c[i] * c[i]) + m.linearity * c[i]); \
    tq[0] += m.prop_y[i] * th;                                                                                       \
    tq[1] += (-m.prop_x[i]) * th;                                                                                    \
    tq[2] += m.torque_max[i] * ccw[i] * c[i];                                                                        \
    fz += th;                                                                                                        \
  }
// rotor thrusts -> body force / torque (:302-310, :140, :182, :88)
template <typename T, bool EXACT>
GAQ_HD void thrust_torque(const Model<T>& m, const T c[4], T tq[3], T& fz) {
  const T ccw[4] = {T(-1), T(1), T(-1), T(1)};
  if constexpr (EXACT) {
#pragma clang fp contract(off)
    GAQ_TORQUE_LOOP
  } else {
    GAQ_TORQUE_LOOP
  }
}
#undef GAQ_TORQUE_LOOP

#define GAQ_OMEGA_BODY(CROSS)                                                                                        \
  const T iw[3] = {m.inertia[0] * s.omega[0], m.inertia[1] * s.omega[1], m.inertia[2] * s.omega[2]};                 \
  const T nw[3] = {-s.omega[0], -s.omega[1], -s.omega[2]};                                                           \
  T cr[3]; CROSS(nw, iw, cr);                                                                                        \
  if (m.damp_omega_q != T(0)) {                                                                                      \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                                  \
      const T wd = m.inv_inertia[j] * (cr[j] + tq[j]);                                                               \
      wd_out[j] = wd;                                                                                                \
      T w2 = s.omega[j] * s.omega[j];                                                                                \
      /* the reference holds omega as a float32 array right after set_state (:223), so the very first */            \
      /* `omega ** 2` (:403) is a float32 product */                                                                 \
      if (first_after_reset) { const float wf = (float)s.omega[j]; w2 = T(wf * wf); }                                \
      const T damp = clampv(m.damp_omega_q * w2, T(0), T(1));                                                        \
      cr[j] = s.omega[j] + (T(1) - damp) * dt * wd;                                                                  \
    }                                                                                                                \
  } else { /* no quadratic damping (every shipped model): omega + (1 - 0) dt wd */                                   \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                                  \
      const T wd = m.inv_inertia[j] * (cr[j] + tq[j]);                                                               \
      wd_out[j] = wd;                                                                                                \
      cr[j] = s.omega[j] + dt * wd;                                                                                  \
    }                                                                                                                \
  }                                                                                                                  \
  _Pragma("unroll") for (int j = 0; j < 3; ++j) s.omega[j] = clampv(cr[j], T(-40), T(40)); /* omega_max (:91) */
// angular velocity: Euler's equations, diagonal inertia (:398-405)
template <typename T, bool EXACT>
// (wd_out: omega_dot of this sub-step, ALWAYS written -- a plain array of the caller's that nobody reads is dead code, whereas a pointer
//  that may be null made it a stack object: three scratch stores per sub-step in the kernels that keep the info dict's aux row)
GAQ_HD void euler_omega(EnvState<T>& s, const Model<T>& m, T dt, const T tq[3], bool first_after_reset, T wd_out[3]) {
  if constexpr (EXACT) {
#pragma clang fp contract(off)
    GAQ_OMEGA_BODY(cross3_nofma)
  } else {
    GAQ_OMEGA_BODY(cross3)
  }
}
#undef GAQ_OMEGA_BODY

// ---- one simulation sub-step: QuadrotorDynamics.step1 (quadrotor.py:273-436) ----------------
// u[] = clip(cmd, 0, 1), w[] = sqrt(u) (hoisted: same for every sub-step of an env step; w is only
// read when a motor lag exists).
template <typename T, uint32_t F>
GAQ_HD void step1(EnvState<T>& s, const Model<T>& m, const StepCfg& cfg, const T u[4], const T w[4],
                  const float nrm[4], bool first_after_reset, float* acc_meter, StepOut* aux = nullptr) {
  constexpr bool G = (F & F_GENERIC) != 0;
  const T dt = T(cfg.dt);
  T c[4];
  model_fence<F>();
  // motor lag (:284-296); part of the rotational subsystem: no FMA contraction (see thrust_torque above)
  if (has_lag<F>(cfg)) {
#pragma clang fp contract(off)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // thrust_cmds_damp of the previous sub-step: without noise it is exactly thrust_rot_damp^2 (:296), recomputed in
      // fp64 rather than read from its fp32 plane, so that the up/down choice never differs from the reference's
      const T prev = (noise_mode<F>(cfg) == NOISE_OFF) ? s.rot_damp[i] * s.rot_damp[i] : T(s.cmds_damp[i]);
      T tau = (u[i] < prev) ? m.tau_down : m.tau_up;
      tau = tau > T(1) ? T(1) : tau;
      s.rot_damp[i] = tau * (w[i] - s.rot_damp[i]) + s.rot_damp[i];
      c[i] = s.rot_damp[i] * s.rot_damp[i];
    }
  } else {
    // tau == 1: thrust_rot_damp = 1*(sqrt(u) - x) + x, thrust_cmds_damp = sqrt(u)^2 = u to 1 ulp
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = u[i];
  }
  // OU thrust noise (:299-300 ; quad_utils.py:197-201, theta 0.15, mu 0)
  if (noise_mode<F>(cfg) != NOISE_OFF) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s.ou[i] = s.ou[i] + (0.15f * (0.0f - s.ou[i]) + m.ou_sigma * nrm[i]);
      c[i] = clampv(c[i] + u[i] * T(s.ou[i]), T(0), T(1));
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = clampv(c[i], T(0), T(1));
  }
  if (has_lag<F>(cfg)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) s.cmds_damp[i] = (float)c[i];
  }
  // kernels that can see motor lag run the rotational subsystem bit-identically to NumPy (see thrust_torque above)
  // (and the Mellinger kernels, like the generic kernel they come from: the controller closes a loop around the rotational subsystem)
  constexpr bool EXACT = (F & (F_LAG | F_PER_ENV | F_GENERIC | F_MELL)) != 0;
  T tq[3] = {T(0), T(0), T(0)};
  T fz = T(0);
  model_fence<F>();
  thrust_torque<T, EXACT>(m, c, tq, fz);
  model_fence<F>();
  T drag_f[3] = {T(0), T(0), T(0)};
  if constexpr (G && (F & F_LITE) == 0) {
    if (cfg.drag && (m.c_drag != T(0) || m.c_roll != T(0))) {   // rotor drag and rolling moment (:318-356)
      const T* R = s.rot;
      T vb[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) vb[j] = R[j] * s.vel[0] + R[3 + j] * s.vel[1] + R[6 + j] * s.vel[2];
      T df[3] = {T(0), T(0), T(0)}, dtq[3] = {T(0), T(0), T(0)}, rtq[3] = {T(0), T(0), T(0)};
      const T ccw[4] = {T(-1), T(1), T(-1), T(1)};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const T pp[3] = {m.prop_x[i], m.prop_y[i], m.prop_z[i]};
        T wxp[3]; cross3(s.omega, pp, wxp);
        const T vr[3] = {vb[0] + wxp[0], vb[1] + wxp[1], T(0)};
        const T sq = sqrt_t(c[i]);
        const T fi[3] = {-m.c_drag * sq * vr[0], -m.c_drag * sq * vr[1], -m.c_drag * sq * vr[2]};
        T ti[3]; cross3(fi, pp, ti);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          df[j] += fi[j]; dtq[j] += ti[j];
          rtq[j] += -m.c_roll * ccw[i] * sq * vr[j];
        }
      }
      T visc[3] = {dtq[0] + rtq[0], dtq[1] + rtq[1], dtq[2] + rtq[2]};
      const T vn = sqrt_t(vb[0] * vb[0] + vb[1] * vb[1] + vb[2] * vb[2]);
      const T fn = sqrt_t(df[0] * df[0] + df[1] * df[1] + df[2] * df[2]);
      const T fclip = clampv(fn, T(0), vn * m.mass / (T(2) * dt));
      if (fn > T(1e-6)) {
#pragma unroll
        for (int j = 0; j < 3; ++j) df[j] = (df[j] / fn) * fclip;
      }
      const T iw[3] = {s.omega[0] * m.inertia[0], s.omega[1] * m.inertia[1], s.omega[2] * m.inertia[2]};
      const T tn = sqrt_t(visc[0] * visc[0] + visc[1] * visc[1] + visc[2] * visc[2]);
      const T tclip = clampv(tn, T(0), sqrt_t(iw[0] * iw[0] + iw[1] * iw[1] + iw[2] * iw[2]) / (T(2) * dt));
      if (tn > T(1e-6)) {
#pragma unroll
        for (int j = 0; j < 3; ++j) visc[j] = (visc[j] / tn) * tclip;
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) { tq[j] += visc[j]; drag_f[j] = df[j]; }
    }
  }
  // rotation: Rodrigues with omega in the world frame (:370-378)
  {
    T* R = s.rot;
    const T wx = R[0] * s.omega[0] + R[1] * s.omega[1] + R[2] * s.omega[2];
    const T wy = R[3] * s.omega[0] + R[4] * s.omega[1] + R[5] * s.omega[2];
    const T wz = R[6] * s.omega[0] + R[7] * s.omega[1] + R[8] * s.omega[2];
    const T w2 = wx * wx + wy * wy + wz * wz;
    T A, B;
    sinc_cosc(w2 * dt * dt, A, B);
    const T a = A * dt, b = B * (dt * dt);
    const T d0 = T(1) - b * w2;
    // D = d0 I + a [w]x + b w w^T with the shared products formed once
    const T awx = a * wx, awy = a * wy, awz = a * wz;
    const T bwx = b * wx, bwy = b * wy, bwz = b * wz;
    const T sxy = bwx * wy, sxz = bwx * wz, syz = bwy * wz;
    const T D[9] = {d0 + bwx * wx, sxy - awz, sxz + awy,
                    sxy + awz, d0 + bwy * wy, syz - awx,
                    sxz - awy, syz + awx, d0 + bwz * wz};
    // R <- D R, one column of R at a time (3 live temporaries instead of 9)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const T r0 = R[j], r1 = R[3 + j], r2 = R[6 + j];
      R[j] = D[0] * r0 + D[1] * r1 + D[2] * r2;
      R[3 + j] = D[3] * r0 + D[4] * r1 + D[5] * r2;
      R[6 + j] = D[6] * r0 + D[7] * r1 + D[8] * r2;
    }
    // mandatory re-orthonormalisation every 0.5 s of simulated time (:381-386); the period in
    // sub-steps is replayed on the host from the reference's fp64 accumulation of dt.
    s.svd_ctr += 1;
    if (s.svd_ctr >= (uint32_t)cfg.svd_period) { polar3(R); s.svd_ctr = 0; }
  }
  T wd[3] = {T(0), T(0), T(0)};
  model_fence<F>();
  euler_omega<T, EXACT>(s, m, dt, tq, first_after_reset, wd);
  model_fence<F>();
  if constexpr (kAux<F>) {
    if (aux) {
      if constexpr (kAuxIsRow<F>) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { aux->aux_row[AUX_OMEGA_DOT + j] = (float)wd[j]; aux->aux_row[AUX_TORQUE + j] = (float)tq[j]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) aux->aux_row[AUX_CMDS + j] = (float)c[j];
      } else {
#pragma unroll
        for (int j = 0; j < 3; ++j) { aux->omega_dot[j] = (float)wd[j]; aux->torque[j] = (float)tq[j]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) aux->cmds[j] = (float)c[j];
      }
    }
  }
  // translation (:418-436): pos uses the old vel, acc uses the new R
  {
    const T* R = s.rot;
    T acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      s.pos[j] = clampv(s.pos[j] + dt * s.vel[j], T(cfg.room_lo[j]), T(cfg.room_hi[j]));
      T f = R[3 * j + 2] * (fz + drag_f[2]);
      if constexpr (G && (F & F_LITE) == 0) f = R[3 * j] * drag_f[0] + R[3 * j + 1] * drag_f[1] + f;
      acc[j] = m.inv_mass * f;
    }
    acc[2] += T(-9.81);
    if (m.vel_damp != T(0)) {
#pragma unroll
      for (int j = 0; j < 3; ++j) s.vel[j] = (T(1) - m.vel_damp) * s.vel[j] + dt * acc[j];
    } else {   // (1 - 0) vel + dt acc
#pragma unroll
      for (int j = 0; j < 3; ++j) s.vel[j] = s.vel[j] + dt * acc[j];
    }
    if constexpr (!kHeadsAreObs<F>) {
      if (acc_meter) {   // accelerometer = R^T (acc + (0,0,g)) (:436)
        const T g2 = acc[2] + T(cfg.gravity);
#pragma unroll
        for (int j = 0; j < 3; ++j) acc_meter[j] = (float)(R[j] * acc[0] + R[3 + j] * acc[1] + R[6 + j] * g2);
      }
    }
  }
}

// ---- reward: compute_reward_weighted (quadrotor.py:544-638; quadrotor_multi.py:550-650) -------------
template <typename T, uint32_t F>
GAQ_HD float reward(const EnvState<T>& s, const StepCfg& cfg, const float a[4], const float ap[4], bool crashed) {
  const RewCoeff& w = cfg.rew;
  const T dx = s.goal[0] - s.pos[0], dy = s.goal[1] - s.pos[1], dz = s.goal[2] - s.pos[2];
  const float dist = GAQ_SQRTF_OUT((float)(dx * dx + dy * dy + dz * dz));
  float cost = w.pos * dist;
  if (cfg.reward_mode != REW_QUADROTOR)   // quadrotor_multi.py:554 (wave-uniform branch, also in the specialised kernels)
    cost = w.pos * (w.pos_log_weight * logf(dist + w.pos_offset) + w.pos_linear_weight * dist);
  cost += w.effort * GAQ_SQRTF_OUT(a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3]);
  cost += w.crash * (crashed ? 1.0f : 0.0f);
  cost += w.orient * (float)(-s.rot[8]);
  cost += w.yaw * (float)(-s.rot[0]);
  if (cfg.use_acos) {   // rot / attitude weights (:575-581): wave-uniform, also in the specialised kernels
    // The reference takes arccos((tr R - 1) / 2) and arccos(R22) in fp64.  In fp32 the arccos is useless exactly where a trained policy
    // lives -- near a hover the argument is 1 - theta^2/2 and its rounding alone (6e-8) moves the angle by up to 3.5e-4 rad, 1.4e-5 of
    // reward at dt = 0.04 with a weight of 1 -- and an fp64 arccos inside the fused rollout loop costs 25 VGPRs.  The half-angle form
    // arccos(c) = 2 atan2(sqrt(1 - c), sqrt(1 + c)) needs 1 - c and 1 + c in fp64 and nothing else: 3e-7 relative at every angle, and it is
    // a function of the reference's own argument c alone (clip included) -- a form that also used the sine of the angle would agree with
    // the reference only as far as R is orthonormal (1e-10 between re-orthonormalisations), which near theta = 0 is not far enough.
    const double cs = clampv((((double)s.rot[0] + (double)s.rot[4] + (double)s.rot[8]) - 1.0) / 2.0, -1.0, 1.0);
    cost += w.rot * (2.0f * atan2f(sqrtf((float)(1.0 - cs)), sqrtf((float)(1.0 + cs))));
    const double ct = clampv((double)s.rot[8], -1.0, 1.0);
    cost += w.attitude * (2.0f * atan2f(sqrtf((float)(1.0 - ct)), sqrtf((float)(1.0 + ct))));
  }
  if (has_act_prev<F>(cfg) && w.action_change != 0.0f) {
    const float d0 = a[0] - ap[0], d1 = a[1] - ap[1], d2 = a[2] - ap[2], d3 = a[3] - ap[3];
    cost += w.action_change * GAQ_SQRTF_OUT(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3);
  }
  cost += w.spin * GAQ_SQRTF_OUT((float)(s.omega[0] * s.omega[0] + s.omega[1] * s.omega[1] + s.omega[2] * s.omega[2]));
  cost += w.vel * GAQ_SQRTF_OUT((float)(s.vel[0] * s.vel[0] + s.vel[1] * s.vel[1] + s.vel[2] * s.vel[2]));
  return -(float)cfg.dt * cost;
}

// ---- swarm terms (own specification, see SwarmCfg) ----------------------------------------------------------
// `Swarm` supplies neighbour j (= agent (a + j) mod agents of the same world): neighbour(j, mine, theirs) exchanges
// the six floats (pos, vel).  In the kernel this is a wave shuffle; NoSwarm stands in where the layer is off.
struct NoSwarm {
  GAQ_HD void neighbour(int, const float*, float* o) const { for (int k = 0; k < 6; ++k) o[k] = 0.0f; }
  GAQ_HD bool any(bool b) const { return b; }      // "does any lane of my wave want this?" (one env per call on the host)
};

// cost_i = sum_{j != i} ( w_collision [d_ij < collision_dist] + w_prox max(0, 1 - d_ij / prox_dist) ), and -- cfg.swarm.response --
// the collision RESPONSE of agent i: every neighbour j closer than collision_dist that is still approaching (vrel = (v_i - v_j) . n < 0
// with n = (p_i - p_j) / d_ij) takes the normal component of the relative velocity out of v_i: dv_i = - sum_j vrel_ij n_ij.  Both agents
// of a pair compute it from the same pre-response velocities, so the pair exchanges exactly that component -- the perfectly elastic
// collision of two equal masses, momentum conserved; contacts with several neighbours add up.  One pass over the neighbours serves both.
template <typename T, typename Swarm>
GAQ_HD float swarm_interact(const EnvState<T>& s, const StepCfg& cfg, Swarm&& sw, float dv[3]) {
  const float me[6] = {(float)s.pos[0], (float)s.pos[1], (float)s.pos[2], (float)s.vel[0], (float)s.vel[1], (float)s.vel[2]};
  float cost = 0.0f;
  dv[0] = 0.0f; dv[1] = 0.0f; dv[2] = 0.0f;
  const float inv_prox = 1.0f / cfg.swarm.prox_dist;
#pragma unroll 1
  for (int j = 1; j < cfg.swarm.agents; ++j) {
    float o[6];
    sw.neighbour(j, me, o);
    const float dx = o[0] - me[0], dy = o[1] - me[1], dz = o[2] - me[2];
    const float d = sqrtf(dx * dx + dy * dy + dz * dz);
    if (d < cfg.swarm.collision_dist) {
      cost += cfg.swarm.w_collision;
      const float inv = 1.0f / fmaxf(d, 1e-6f);
      const float nx = -dx * inv, ny = -dy * inv, nz = -dz * inv;                       // from j to i
      const float vrel = (me[3] - o[3]) * nx + (me[4] - o[4]) * ny + (me[5] - o[5]) * nz;
      if (vrel < 0.0f) { dv[0] -= vrel * nx; dv[1] -= vrel * ny; dv[2] -= vrel * nz; }
    }
    cost += cfg.swarm.w_prox * fmaxf(0.0f, 1.0f - d * inv_prox);
  }
  return cost;
}

// ---- observation: get_state.state_<obs_repr>, with or without SensorNoise.add_noise ------------------------
GAQ_HD float uni_pm(uint32_t bits, float range) { return (((float)(bits >> 8) + 0.5f) * (2.0f / 16777216.0f) - 1.0f) * range; }

// SensorNoise.add_noise (sensor_noise.py:100-158): Gaussian (+ optional uniform) noise on pos and vel, gyro noise
// (white, or -- gyro_norm_std != 0 -- the bias random walk of add_noise_to_omega :160-168 plus white noise), a
// small-angle quaternion perturbation of the attitude (quat_from_small_angle :9-21; rot2quat -> quatXquat -> quat2R
// == R(q_theta) * R for orthonormal R) and static + proportional accelerometer noise.  Observation-only: nothing
// here feeds back into the dynamics; the only state is the gyro bias, advanced by (b_pi, b_sigma) per call
// (`gyro_bias` may be nullptr when the bias model is off).
// The reference draws from numpy's global MT19937; here the draws are Philox streams keyed by (env, key).
// Injected draws (cfg.sense_input, parity tests): `src(call, slot, j)` returns the standard draw the reference made in
// add_noise call `call` (0, 1: the two discarded calls of a step; 2: the call behind the returned observation), slot =
// 0 pos n, 1 pos u, 2 vel n, 3 vel u, 4 gyro n (bias model: the bias increment), 5 gyro white n (bias model), 6 quat n,
// 7 quat u, 8 acc static n, 9 acc proportional n; normal slots N(0,1), uniform slots U(0,1) (numpy: low + (high-low) u).
struct NoSense {
  GAQ_HD float operator()(int, int, int) const { return 0.0f; }
};
template <typename T, bool INPUT = false, typename SenseSrc = NoSense>
GAQ_HD void sense_noise(const StepCfg& cfg, uint64_t env_global, uint64_t key, T pos[3], T vel[3], T rot[9], T omega[3],
                        float acc[3], float* gyro_bias, int calls, SenseSrc&& src, bool want_qtheta, float qtheta_out[4]) {
  // (want_qtheta + an array that is always there, not a pointer that may be null: a `cond ? array : nullptr` argument turns the caller's
  //  array into a stack object -- two scratch stores per pack_obs call in every kernel that has the quaternion observations)
  const SenseNoise& sn = cfg.sense;
  // 21 normals (0-2 pos, 3-5 vel, 6-8 gyro white: TWO Philox blocks for the nine, normals10; 9-11 attitude: a third; 12-17 accelerometer,
  // 18-20 gyro-bias increment: three more) and three blocks of uniforms; only the blocks a configuration uses are drawn (all branches
  // wave-uniform): SensorNoise() on the 18-word observation needs two of the nine.
  float n[24];
#pragma unroll
  for (int j = 0; j < 24; ++j) n[j] = 0.0f;
  const bool want_acc = (cfg.obs_flags & OBS_APPEND_ACC) != 0;
  const bool want_bias = cfg.gyro_bias && gyro_bias;
  float up[3] = {0.0f, 0.0f, 0.0f}, uv[3] = {0.0f, 0.0f, 0.0f}, uq[3] = {0.0f, 0.0f, 0.0f};
  float b_pi = calls == 3 ? cfg.gyro_pi_step : cfg.gyro_pi, b_sigma = calls == 3 ? cfg.gyro_sigma_step : cfg.gyro_sigma;
  bool injected = false;
  if constexpr (INPUT) injected = cfg.sense_input != 0;
  if (injected) {
    // the recorded draws of the observation's own call; the bias walk takes the earlier calls' increments one by one
    const int c = 2;
    if (want_bias && calls == 3) {
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int j = 0; j < 3; ++j) gyro_bias[j] = cfg.gyro_pi * gyro_bias[j] + cfg.gyro_sigma * src(e, 4, j);
    }
    b_pi = cfg.gyro_pi; b_sigma = cfg.gyro_sigma;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      n[j] = src(c, 0, j); n[3 + j] = src(c, 2, j);
      n[6 + j] = want_bias ? src(c, 5, j) : src(c, 4, j);
      n[18 + j] = src(c, 4, j);
      n[9 + j] = src(c, 6, j); n[12 + j] = src(c, 8, j); n[15 + j] = src(c, 9, j);
      up[j] = (2.0f * src(c, 1, j) - 1.0f) * sn.pos_unif_range;
      uv[j] = (2.0f * src(c, 3, j) - 1.0f) * sn.vel_unif_range;
      uq[j] = (2.0f * src(c, 7, j) - 1.0f) * sn.quat_unif_range;
    }
  } else {
    {   // position, velocity and gyro normals (n[0..8]): two blocks; the attitude perturbation's (n[9..11]) only when it is on
      const Philox ra(cfg.seed, env_global, key, RNG_SENSE0), rb(cfg.seed, env_global, key, RNG_SENSE0 + 1u);
      float t[10];
      normals10(ra, rb, t);
#pragma unroll
      for (int j = 0; j < 9; ++j) n[j] = t[j];
      if (sn.quat_norm_std != 0.0f) { const Philox rq(cfg.seed, env_global, key, RNG_SENSE0 + 2u); normals4(rq, n + 8); n[8] = t[8]; }
    }
    if (want_acc || want_bias) {
#pragma unroll
      for (int j = 3; j < 6; ++j) { const Philox r(cfg.seed, env_global, key, RNG_SENSE0 + (uint32_t)j); normals4(r, n + 4 * j); }
    }
    if (sn.pos_unif_range != 0.0f) {
      const Philox u0(cfg.seed, env_global, key, RNG_SENSE0 + 6u);
#pragma unroll
      for (int j = 0; j < 3; ++j) up[j] = uni_pm(u0.c[j], sn.pos_unif_range);
    }
    if (sn.vel_unif_range != 0.0f) {
      const Philox u1(cfg.seed, env_global, key, RNG_SENSE0 + 7u);
#pragma unroll
      for (int j = 0; j < 3; ++j) uv[j] = uni_pm(u1.c[j], sn.vel_unif_range);
    }
    if (sn.quat_unif_range != 0.0f) {
      const Philox u2(cfg.seed, env_global, key, RNG_SENSE0 + 8u);
#pragma unroll
      for (int j = 0; j < 3; ++j) uq[j] = uni_pm(u2.c[j], sn.quat_unif_range);
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    pos[j] += T(sn.pos_norm_std * n[j] + up[j]);
    vel[j] += T(sn.vel_norm_std * n[3 + j] + uv[j]);
    if (want_bias) {
      gyro_bias[j] = b_pi * gyro_bias[j] + b_sigma * n[18 + j];
      omega[j] += T(gyro_bias[j] + sn.gyro_random_walk * n[6 + j]);
    } else {
      omega[j] += T(sn.gyro_noise_density * n[6 + j]);
    }
  }
  if (sn.quat_norm_std != 0.0f || sn.quat_unif_range != 0.0f) {   // otherwise q_theta = (1,0,0,0): R goes through untouched
    // quat_from_small_angle (sensor_noise.py:9-21), then rot2quat -> quatXquat -> quat2R (:144-147).  The reference's
    // quatXquat(quat, quat_theta) (quad_utils.py:92-99) has the cross terms of the Hamilton product quat_theta * quat, so for
    // an orthonormal R the result is R(q_theta) * R: the perturbation acts in the WORLD frame (round 1 had R * R(q_theta),
    // which only the value-for-value test against the reference's recorded draws could tell apart).  fp64 throughout.
    double th[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) th[j] = (double)(sn.quat_norm_std * n[9 + j]) + (double)uq[j];
    const double q2 = (th[0] * th[0] + th[1] * th[1] + th[2] * th[2]) * 0.25;
    double qw, f;
    if (q2 < 1.0) { qw = sqrt(1.0 - q2); f = 0.5; } else { qw = 1.0 / sqrt(1.0 + q2); f = 0.5 * qw; }
    double qx = th[0] * f, qy = th[1] * f, qz = th[2] * f;
    const double inv = 1.0 / sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
    qw *= inv; qx *= inv; qy *= inv; qz *= inv;
    if (want_qtheta) {   // the quaternion observations perturb the quaternion itself (sensor_noise.py:148-151): hand q_theta back
      // (handed back as fp32: a unit quaternion to 6e-8, against the 1e-6 the observation is held to; four registers instead of eight)
      qtheta_out[0] = (float)qw; qtheta_out[1] = (float)qx; qtheta_out[2] = (float)qy; qtheta_out[3] = (float)qz;
    } else {
      // quat2R (quad_utils.py:82-87)
      const T Q[9] = {T(1.0 - 2 * qy * qy - 2 * qz * qz), T(2 * qx * qy - 2 * qz * qw), T(2 * qx * qz + 2 * qy * qw),
                      T(2 * qx * qy + 2 * qz * qw), T(1.0 - 2 * qx * qx - 2 * qz * qz), T(2 * qy * qz - 2 * qx * qw),
                      T(2 * qx * qz - 2 * qy * qw), T(2 * qy * qz + 2 * qx * qw), T(1.0 - 2 * qx * qx - 2 * qy * qy)};
      T N[9];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) N[3 * i + j] = Q[3 * i] * rot[j] + Q[3 * i + 1] * rot[3 + j] + Q[3 * i + 2] * rot[6 + j];
#pragma unroll
      for (int i = 0; i < 9; ++i) rot[i] = N[i];
    }
  }
  if (want_acc) {
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[j] = acc[j] + sn.acc_static_noise_std * n[12 + j] + acc[j] * (sn.acc_dynamic_noise_ratio * n[15 + j]);
  }
}

// `act_hist` = env.actions[1] at packing time.  Writes cfg.obs_dim floats through put(k, value).
// `noise_key` selects the sensor-noise draws of this observation (step index, or the reset's episode key).
// `calls` = add_noise calls the reference makes up to and including this observation: 3 for the observation of a
// step, 1 for reset / state_vector (only the gyro bias, which those calls advance, can tell the difference).
// Swarm: the self block is followed by (pos_j - pos_i, vel_j - vel_i) of the agents-1 neighbours, world frame, true state.
template <typename T, uint32_t F, typename Sink, typename Swarm = NoSwarm, typename SenseSrc = NoSense>
GAQ_HD void pack_obs(EnvState<T>& s, const StepCfg& cfg, const float acc_meter[3], const float act_hist[4],
                     Sink&& put, uint64_t env_global = 0, uint64_t noise_key = 0, int calls = 1, Swarm&& sw = NoSwarm(),
                     SenseSrc&& get_sense = NoSense(), T t2w = T(0), T t2t = T(0)) {
  constexpr bool HEAVY = kAux<F>;       // the quaternion / t2w / t2t variants: the F_DIAG tiers of the generic kernel (full, and light = F_LITE | F_DIAG)
  constexpr bool INJECT = kDiag<F>;     // injected sensor draws (parity tests): the full diagnostics tier only
  bool quat = false;
  if constexpr (HEAVY) quat = (cfg.obs_flags & OBS_QUAT) != 0;
  float qth[4] = {1.0f, 0.0f, 0.0f, 0.0f};
  T pos[3] = {s.pos[0], s.pos[1], s.pos[2]};
  T v[3] = {s.vel[0], s.vel[1], s.vel[2]};
  T rot[9], om[3] = {s.omega[0], s.omega[1], s.omega[2]};
  float acc[3] = {acc_meter[0], acc_meter[1], acc_meter[2]};
#pragma unroll
  for (int j = 0; j < 9; ++j) rot[j] = s.rot[j];
  // (wave-uniform; the specialised plain-layout kernels take it too -- white-noise gyro only, the bias random walk
  //  needs the generic kernel's bias plane -- and in the alias kernels, whose sink discards everything, it is dead code)
  if (cfg.sense.enabled)
    sense_noise<T, INJECT>(cfg, env_global, noise_key, pos, v, rot, om, acc, kGyroBias<F> ? s.gyro_bias : nullptr, calls, get_sense,
                          quat, qth);
  T rel[3] = {pos[0] - s.goal[0], pos[1] - s.goal[1], pos[2] - s.goal[2]};
  {
    if (cfg.obs_flags & OBS_BODY_FRAME) {   // with the TRUE attitude (get_state.py:159-160 uses self.dynamics.rot)
      const T* R = s.rot;
      const T r0 = R[0] * rel[0] + R[3] * rel[1] + R[6] * rel[2], r1 = R[1] * rel[0] + R[4] * rel[1] + R[7] * rel[2],
              r2 = R[2] * rel[0] + R[5] * rel[1] + R[8] * rel[2];
      const T v0 = R[0] * v[0] + R[3] * v[1] + R[6] * v[2], v1 = R[1] * v[0] + R[4] * v[1] + R[7] * v[2],
              v2 = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
      rel[0] = r0; rel[1] = r1; rel[2] = r2; v[0] = v0; v[1] = v1; v[2] = v2;
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) put(j, (float)rel[j], -1);
#pragma unroll
  for (int j = 0; j < 3; ++j) put(3 + j, (float)v[j], -1);
  int k = 18;
  if (quat) {
    if constexpr (HEAVY) {
      // self.quat = R2quat(self.dynamics.rot) (get_state.py:277; quad_utils.py:101-108: w from the trace, no branch) of the
      // TRUE attitude; with sensor noise the quaternion itself is perturbed, quatXquat(quat, quat_theta) (sensor_noise.py:148-151)
      const T* R = s.rot;
      const double w = sqrt(1.0 + (double)R[0] + (double)R[4] + (double)R[8]) / 2.0, w4 = 4.0 * w;
      double q[4] = {w, ((double)R[7] - (double)R[5]) / w4, ((double)R[2] - (double)R[6]) / w4, ((double)R[3] - (double)R[1]) / w4};
      if (cfg.sense.enabled) {
        const double* a = q;                             // quatXquat (quad_utils.py:92-99), term for term
        const double b[4] = {(double)qth[0], (double)qth[1], (double)qth[2], (double)qth[3]};
        const double nq[4] = {a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] - a[2] * b[3] + a[3] * b[2],
                              a[0] * b[2] + a[1] * b[3] + a[2] * b[0] - a[3] * b[1], a[0] * b[3] - a[1] * b[2] + a[2] * b[1] + a[3] * b[0]};
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = nq[j];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) put(6 + j, (float)q[j], -1);
#pragma unroll
      for (int j = 0; j < 3; ++j) put(10 + j, (float)om[j], -1);
      k = 13;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 9; ++j) put(6 + j, (float)rot[j], -1);
#pragma unroll
    for (int j = 0; j < 3; ++j) put(15 + j, (float)om[j], -1);
  }
  // appended words: `k` is the position in the row (depends on which appendices are on), `slot` a fixed id -- 0 the
  // height, 1-3 the accelerometer, 4-7 the previous action -- for sinks that keep the observation in registers
  if (cfg.obs_flags & OBS_APPEND_H) put(k++, (float)pos[2], 0);
  if constexpr (!kHeadsAreObs<F>) {
    if (cfg.obs_flags & OBS_APPEND_ACC) {
#pragma unroll
      for (int j = 0; j < 3; ++j) put(k++, acc[j], 1 + j);
    }
    if (cfg.obs_flags & OBS_APPEND_ACT) {
#pragma unroll
      for (int j = 0; j < 4; ++j) put(k++, act_hist[j], 4 + j);
    }
  }
  if constexpr (HEAVY) {
    if (cfg.obs_flags & (OBS_APPEND_T2W | OBS_APPEND_T2T)) {
      // get_state.py:335-338: the model's thrust-to-weight (torque-to-thrust) ratio with noise relative to its value,
      // clipped to [min, max] and mapped to [0, 1]; one normal each per state_vector call
      float nz[4];
      bool injected = false;
      if constexpr (INJECT) injected = cfg.sense_input != 0;
      if (injected) { nz[0] = get_sense(2, 10, 0); nz[1] = get_sense(2, 11, 0); }
      else { const Philox r(cfg.seed, env_global, noise_key, RNG_SENSE0 + 9u); normals4(r, nz); }
      if (cfg.obs_flags & OBS_APPEND_T2W) {   // (slots 8, 9: fixed ids like the other appended words', for sinks that keep the row in registers)
        const double x = clampv((double)t2w + fabs(((double)cfg.t2w_std / 2) * (double)t2w) * (double)nz[0], (double)cfg.t2w_min, (double)cfg.t2w_max);
        // (the noisy ratio is formed in fp64 like the reference's; the map to [0, 1] is an fp32 division of the fp32-rounded offset: within
        //  one fp32 ulp of the reference's fp64 quotient, and a fifth of an fp64 division's instructions)
        put(k++, (float)(x - (double)cfg.t2w_min) / (cfg.t2w_max - cfg.t2w_min), 8);
      }
      if (cfg.obs_flags & OBS_APPEND_T2T) {
        const double x = clampv((double)t2t + fabs(((double)cfg.t2t_std / 2) * (double)t2t) * (double)nz[1], (double)cfg.t2t_min, (double)cfg.t2t_max);
        put(k++, (float)(x - (double)cfg.t2t_min) / (cfg.t2t_max - cfg.t2t_min), 9);
      }
    }
  }
  if constexpr (kSwarm<F>) {
    if (cfg.swarm.agents > 1) {
      const float me[6] = {(float)s.pos[0], (float)s.pos[1], (float)s.pos[2], (float)s.vel[0], (float)s.vel[1], (float)s.vel[2]};
#pragma unroll 1
      for (int j = 1; j < cfg.swarm.agents; ++j) {
        float o[6];
        sw.neighbour(j, me, o);
#pragma unroll
        for (int c = 0; c < 6; ++c) put(k++, o[c] - me[c], -1);
      }
    }
  }
}

// ---- reset: QuadrotorEnv._reset (quadrotor.py:1059-1144) with a counter-based RNG ---------------------
// The reference draws from MT19937 streams; on device only the DISTRIBUTION is reproduced
// (tests compare against 4000 reference resets, fixture G8).
template <typename T, uint32_t F>
GAQ_HD void reset_env(EnvState<T>& s, const StepCfg& cfg, uint64_t env_global, uint64_t episode_key) {
  const Philox r(cfg.seed, env_global, episode_key, RNG_RESET_A);
  T goal[3] = {T(cfg.goal_default[0]), T(cfg.goal_default[1]), T(cfg.goal_default[2])};
  if constexpr (kSwarm<F>) {
    if (cfg.swarm.agents > 1) {   // formation: the world's agents sit on a circle around the default goal
      const float ang = 6.2831853071795864769f * (float)(env_global % (uint64_t)cfg.swarm.agents) / (float)cfg.swarm.agents;
      goal[0] = T((float)cfg.goal_default[0] + cfg.swarm.goal_radius * cosf(ang));
      goal[1] = T((float)cfg.goal_default[1] + cfg.swarm.goal_radius * sinf(ang));
    }
  }
  if constexpr (kEnvExtras<F>) {
    if (cfg.resample_goal) {   // goal z ~ U(0.5, 2) (:1079)
      const Philox g(cfg.seed, env_global, episode_key, RNG_RESET_B);
      goal[2] = T((float)(0.5 + 1.5 * g.u01(0)));
    }
  }
  const double box = cfg.init_box;
  double p[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) p[j] = (2.0 * r.u01(j) - 1.0) * box + (double)goal[j];   // :1087
  if (p[2] < 0.25) p[2] = 0.25;                                                         // :1094
#pragma unroll
  for (int j = 0; j < 3; ++j) { s.pos[j] = T(p[j]); s.goal[j] = goal[j]; }
  const bool random_state = cfg.init_random_state != 0;   // wave-uniform; also honoured by the specialised kernels
  if (!random_state) {
#pragma unroll
    for (int j = 0; j < 3; ++j) { s.vel[j] = T(0); s.omega[j] = T(0); }
    // randyaw() rejected until dot(R[:,0], to_xyhat(-pos)) >= 0.5 (:1124-1126): the accepted yaw is
    // uniform on psi0 +- pi/3, psi0 = direction to the origin in the xy plane.
    double dxn = -p[0], dyn = -p[1];
    const double nn = sqrt(dxn * dxn + dyn * dyn);
    if (nn < 0.00001) { dxn = 1.0; dyn = 0.0; } else { dxn /= nn; dyn /= nn; }  // (the reference loops forever here)
    // sin in fp32 (the angle is random anyway), cos = sqrt(1 - sin^2) in fp64 so that R is
    // orthonormal to fp64 round-off like the reference's rotZ (quad_utils.py:116-119); |delta| <= pi/3 -> cos > 0
    const float delta = (float)((2.0 * r.u01(3) - 1.0) * 1.04719755119659774615);
    const double sd = (double)GAQ_SINF(delta);
    const double cd = sqrt(1.0 - sd * sd);
    const double cpsi = dxn * cd - dyn * sd, spsi = dyn * cd + dxn * sd;
    s.rot[0] = T(cpsi); s.rot[1] = T(-spsi); s.rot[2] = T(0);
    s.rot[3] = T(spsi); s.rot[4] = T(cpsi); s.rot[5] = T(0);
    s.rot[6] = T(0); s.rot[7] = T(0); s.rot[8] = T(1);
  }
  {
    if (random_state) {
      // random_state (:227-239) with vel_max = 1, omega_max = 2 pi (:717-718, :1113-1115)
      const Philox a(cfg.seed, env_global, episode_key, RNG_RESET_C);
      const Philox b(cfg.seed, env_global, episode_key, RNG_RESET_D);
      const Philox g(cfg.seed, env_global, episode_key, RNG_RESET_B);
      double v[3], w[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) { v[j] = 2.0 * a.u01(j) - 1.0; w[j] = (2.0 * b.u01(j) - 1.0) * 6.283185307179586; }
      const double vm = a.u01(3) * 1.0 / (sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) + 1e-6);
      const double wm = b.u01(3) * 6.283185307179586 / (sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]) + 1e-6);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        s.vel[j] = T(vm * v[j]);
        s.omega[j] = T((float)(wm * w[j]));   // set_state casts omega to float32 (:223)
      }
      // rand_uniform_rot3d (quad_utils.py:47-58): two isotropic unit vectors (rejecting near-parallel
      // pairs) -> orthonormal frame [fwd left up].  Isotropic directions from (z, phi) instead of
      // normalised Gaussians: same distribution on the sphere.  Unit vectors are re-normalised in fp64.
      double up[3], fw[3] = {1.0, 0.0, 0.0};
      {
        const double z = 2.0 * g.u01(1) - 1.0, rr = sqrt(1.0 - z * z);
        const float ph = (float)(6.283185307179586 * g.u01(2));
        up[0] = rr * (double)GAQ_COSF(ph); up[1] = rr * (double)GAQ_SINF(ph); up[2] = z;
      }
      bool ok = false;
#pragma unroll 1
      for (uint32_t t = 0; t < 16 && !ok; ++t) {
        const Philox h(cfg.seed, env_global, episode_key, 68 + t);
        const double z = 2.0 * h.u01(0) - 1.0, rr = sqrt(1.0 - z * z);
        const float ph = (float)(6.283185307179586 * h.u01(1));
        fw[0] = rr * (double)GAQ_COSF(ph); fw[1] = rr * (double)GAQ_SINF(ph); fw[2] = z;
        ok = !(fw[0] * up[0] + fw[1] * up[1] + fw[2] * up[2] > 0.95);
      }
      const double fn = 1.0 / sqrt(fw[0] * fw[0] + fw[1] * fw[1] + fw[2] * fw[2]);
      fw[0] *= fn; fw[1] *= fn; fw[2] *= fn;
      double lf[3] = {up[1] * fw[2] - up[2] * fw[1], up[2] * fw[0] - up[0] * fw[2], up[0] * fw[1] - up[1] * fw[0]};
      const double ln = 1.0 / sqrt(lf[0] * lf[0] + lf[1] * lf[1] + lf[2] * lf[2]);
      lf[0] *= ln; lf[1] *= ln; lf[2] *= ln;
      const double u2[3] = {fw[1] * lf[2] - fw[2] * lf[1], fw[2] * lf[0] - fw[0] * lf[2], fw[0] * lf[1] - fw[1] * lf[0]};
#pragma unroll
      for (int j = 0; j < 3; ++j) { s.rot[3 * j] = T(fw[j]); s.rot[3 * j + 1] = T(lf[j]); s.rot[3 * j + 2] = T(u2[j]); }
    }
  }
  // dynamics.reset() (:438-440) and env counters (:1139-1141).  svd_ctr and the OU state survive.
#pragma unroll
  for (int i = 0; i < 4; ++i) { s.rot_damp[i] = T(0); s.cmds_damp[i] = 0.0f; s.act_prev[i] = 0.0f; }
  s.tick = 0;
}

// ---- one env step: QuadrotorEnv._step (quadrotor.py:942-1028) ----------------------------------------
// get_normal(k, i): for NOISE_INPUT, normal i of sub-step k.  put_obs(k, v): observation sink.
// term_row: where to write the terminal observation of an env that is auto-reset in this step (or nullptr).
template <typename T, uint32_t F, typename NormalSrc, typename Sink, typename Swarm = NoSwarm, typename SenseSrc = NoSense>
GAQ_HD void env_step(EnvState<T>& s, const Model<T>& m, const StepCfg& cfg, const float action[4], uint64_t env_global,
                     NormalSrc&& get_normal, StepOut& out, Sink&& put_obs, float* term_row = nullptr, Swarm&& sw = NoSwarm(),
                     SenseSrc&& get_sense = NoSense()) {
  constexpr bool G = (F & F_GENERIC) != 0;
  float hist1[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (has_act_prev<F>(cfg)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) hist1[i] = s.act_prev[i];          // actions[1] <- actions[0] (:943)
  }
  if constexpr (kEnvExtras<F>) {
    if (cfg.excite && (s.tick % 5u) == 0u) {   // :957-963: goal ~ U(-0.5, 0.5)^2 x U(1.5, 2.5), before controller and reward
      const Philox e(cfg.seed, env_global, cfg.step_index, RNG_EXCITE);
      s.goal[0] = T((float)(e.u01(0) - 0.5)); s.goal[1] = T((float)(e.u01(1) - 0.5)); s.goal[2] = T((float)(1.5 + e.u01(2)));
    }
  }
  // thrust_to_weight / torque_to_thrust of this env's model, for the t2w / t2t observation components: sum(thrust_max) =
  // g m t2w (the motor asymmetry is normalised to sum 4, quadrotor.py:174-175), torque_max = t2t thrust_max (:176)
  T t2w = T(0), t2t = T(0);
  if constexpr (kAux<F>) {
    if (cfg.obs_flags & (OBS_APPEND_T2W | OBS_APPEND_T2T)) {
      t2w = (((m.thrust_max[0] + m.thrust_max[1]) + m.thrust_max[2]) + m.thrust_max[3]) * m.inv_mass * T(1.0 / 9.81);
      t2t = m.torque_max[0] / m.thrust_max[0];
    }
  }
  T cmd[4];
  bool mell = false;
  if constexpr ((F & F_MELL) != 0) {
    mell = true;
    mellinger(s, cfg, m.jinv, cmd, s.tick == 0);
  } else {
    if constexpr (G && (F & F_LITE) == 0) mell = cfg.control == CTRL_MELLINGER;
    if constexpr (G && (F & F_LITE) == 0) { if (mell) mellinger(s, cfg, m.jinv, cmd, s.tick == 0); }
    if (!mell) raw_control(action, cfg.control, cmd, cfg.action_f32 != 0);
  }
  bool want_aux = false;
  if constexpr (kAux<F>) {
    want_aux = cfg.aux != 0;
    if (want_aux) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {                                 // controller.action (quadrotor_control.py:91, :362)
        if constexpr (kAuxIsRow<F>) out.aux_row[AUX_CTRL + i] = (float)cmd[i]; else out.ctrl[i] = (float)cmd[i];
      }
    }
  }
  T u[4], w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    u[i] = clampv(cmd[i], T(0), T(1));                              // :279
    w[i] = has_lag<F>(cfg) ? sqrt_t(u[i]) : T(0);                   // :294
  }
  const bool fresh = (s.tick == 0);
  // NaN / Inf canary over everything a clamp could launder (pos, omega, the action) or that the default
  // reward does not look at (vel): x * 0 is NaN exactly when x is not finite
  const float poison = (float)(((s.pos[0] + s.pos[1] + s.pos[2]) + (s.vel[0] + s.vel[1] + s.vel[2]) +
                                (s.omega[0] + s.omega[1] + s.omega[2])) * T(0)) +
                       ((action[0] + action[1]) + (action[2] + action[3])) * 0.0f;
  out.acc_meter[0] = 0.0f; out.acc_meter[1] = 0.0f; out.acc_meter[2] = (float)cfg.gravity;
  for (int k = 0; k < cfg.sim_steps; ++k) {                         // dynamics.step (:261-262)
    float nrm[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int nm = noise_mode<F>(cfg);
    if (nm == NOISE_PHILOX) {
      bool have = false;
      if constexpr ((F & F_PREDRAW) != 0) {
        if (k < 2) {
#pragma unroll
          for (int i = 0; i < 4; ++i) nrm[i] = get_normal(k, i);     // the same draws, made earlier
          have = true;
        }
      }
      if (!have) {
        const Philox r(cfg.seed, env_global, cfg.step_index, RNG_OU0 + (uint32_t)k);
        normals4(r, nrm);
      }
    } else if (nm == NOISE_INPUT) {
#pragma unroll
      for (int i = 0; i < 4; ++i) nrm[i] = get_normal(k, i);
    }
    float* am = nullptr;
    if constexpr (!kHeadsAreObs<F>) am = (((cfg.obs_flags & OBS_APPEND_ACC) || want_aux) && k == cfg.sim_steps - 1) ? out.acc_meter : nullptr;
    StepOut* auxp = nullptr;      // (only where the aux row exists: `&out` escaping costs the other kernels a 68-byte stack object)
    if constexpr (kAux<F>) auxp = (want_aux && k == cfg.sim_steps - 1) ? &out : nullptr;
    step1<T, F>(s, m, cfg, u, w, nrm, fresh && k == 0, am, auxp);
  }
  float swarm_penalty = 0.0f;
  if constexpr (kSwarm<F>) {
    if (cfg.swarm.agents > 1) {     // every agent of the world is here together (a world never straddles a wave tile)
      float dv[3];
      swarm_penalty = swarm_interact(s, cfg, sw, dv);
      if (cfg.swarm.response) {     // the collision response acts on the integrated state; reward and observation see the result
#pragma unroll
        for (int j = 0; j < 3; ++j) s.vel[j] += T(dv[j]);
      }
    }
  }
  const bool crashed = s.pos[2] <= m.arm;                           // :977 (:978-981 is always False)
  out.crashed = crashed;
  out.reward = reward<T, F>(s, cfg, action, hist1, crashed) + poison; // :984
  if constexpr (kSwarm<F>) out.reward -= (float)cfg.dt * swarm_penalty;
  if (s.tick < 0xFFFFu) s.tick += 1;                                // :986
  const bool done = s.tick > (uint32_t)cfg.ep_len;                  // :987
  out.done = done;
  if (has_act_prev<F>(cfg)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) s.act_prev[i] = action[i];
  }
  if constexpr (kSwarm<F>) {
    // swarm: the terminal row holds neighbour terms, i.e. wave shuffles -- they must not sit in a branch that only the
    // finishing lanes take (a masked reset or set_state can desynchronise the ticks inside a world): every lane packs,
    // only the finishing ones store
    if (cfg.swarm.agents > 1 && sw.any(cfg.auto_reset && done && term_row != nullptr)) {
      const bool wr = cfg.auto_reset && done && term_row != nullptr;
      pack_obs<T, F>(s, cfg, out.acc_meter, hist1, [&](int k, float v, int) { if (wr) term_row[k] = v; }, env_global,
                     cfg.step_index ^ (1ull << 62), 3, sw, get_sense, t2w, t2t);
    }
  }
  if (!GAQ_PROBE_HOT && cfg.auto_reset && done) {
    // vector-env convention: the observation returned with done=1 is the first one of the new episode; the last
    // one of the finished episode (what the reference returns with done=True, needed to bootstrap a value at this
    // time-limit truncation) goes to the caller's terminal-observation row when one was registered
    // (its sensor-noise draws are keyed apart from those of the new episode's first observation below; the three
    // add_noise calls of the finished step advance the gyro bias whether or not the row is wanted)
    bool packed = false;
    if constexpr (kSwarm<F>) packed = cfg.swarm.agents > 1;      // swarm rows were packed above, with the whole wave taking part
    if (!packed) {
      if (term_row) {
        pack_obs<T, F>(s, cfg, out.acc_meter, hist1, [&](int k, float v, int) { term_row[k] = v; }, env_global,
                       cfg.step_index ^ (1ull << 62), 3, sw, get_sense, t2w, t2t);
      } else if (has_gyro_bias<F>(cfg)) {
        pack_obs<T, F>(s, cfg, out.acc_meter, hist1, [&](int, float, int) {}, env_global, cfg.step_index ^ (1ull << 62), 3, NoSwarm(), get_sense);
      }
    }
    reset_env<T, F>(s, cfg, env_global, cfg.step_index + 1);
    out.acc_meter[0] = 0.0f; out.acc_meter[1] = 0.0f; out.acc_meter[2] = 9.81f;   // set_state (:221)
#pragma unroll
    for (int i = 0; i < 4; ++i) hist1[i] = 0.0f;
  }
  const bool after_reset = cfg.auto_reset && done;                                     // :1143 (one add_noise call)
  if constexpr (kAuxIsRow<F>) {
    if (want_aux) {
#pragma unroll
      for (int j = 0; j < 3; ++j) out.aux_row[AUX_ACC + j] = out.acc_meter[j];
    }
  }
  pack_obs<T, F>(s, cfg, out.acc_meter, hist1, put_obs, env_global, cfg.step_index, after_reset ? 1 : 3, sw, get_sense, t2w, t2t);   // :988
}

}  // namespace gaq
