/* gaq.h -- C ABI of libgaq.so: the MI355X-native batched quadrotor simulator.
 *
 * Drop-in boundary for ONE path of amolchanov86/gym_art: what `QuadrotorEnv.step()` /
 * `reset()` do per call (gym_art/quadrotor/quadrotor.py:942-1028, :1059-1144), i.e.
 * controller -> QuadrotorDynamics.step (step1 x sim_steps, :261-436) -> crash test ->
 * compute_reward_weighted (:544-638) -> tick/done -> state_<obs_repr> (get_state.py),
 * for a batch of N independent environments held in device memory (struct of arrays).
 *
 * Plain C: opaque handle, plain pointers and sizes, int status codes.  No torch types.
 * The reference has no FFI of its own (it is pure Python); each entry point below names
 * the reference call it stands in for.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every function returns 0 on success, a negative gaq_status on error; the message
 *     is available (thread-local) from gaq_last_error().
 *   - host-pointer variants (gaq_step, gaq_reset, ...) synchronise before returning;
 *     *_dev variants take device pointers and are asynchronous on `stream`, a hipStream_t passed
 *     as void* with HIP's own meaning of NULL (the legacy default stream, which is also what
 *     torch.cuda.current_stream().cuda_stream is unless the caller switched streams).  The
 *     host-pointer variants run on the handle's private stream, gaq_stream().
 *   - the caller owns every in/out buffer; the library owns the handle and its device
 *     state and allocates nothing per step.
 *   - a handle is not thread-safe; distinct handles are independent.
 *   - actions are [N,4] float32 row-major, 16-byte aligned; obs is [N,obs_dim] float32
 *     row-major; reward [N] float32; done [N] uint8.
 */
#ifndef GAQ_H
#define GAQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GAQ_ABI_VERSION 5

typedef struct gaq_env gaq_env;

typedef enum gaq_status {
  GAQ_OK = 0,
  GAQ_ERR_INVALID = -1,   /* bad argument / unsupported configuration (reference: ValueError / assert) */
  GAQ_ERR_DEVICE = -2,    /* HIP runtime error, no device */
  GAQ_ERR_NAN = -3,       /* non-finite reward seen (reference: ValueError, quadrotor.py:633-636) */
  GAQ_ERR_STATE = -4      /* call sequence error */
} gaq_status;

/* RawControl zero-middle / RawControl [0,1] / Mellinger (quadrotor_control.py:72-92, :315-362) */
enum { GAQ_CTRL_RAW_ZERO_MIDDLE = 0, GAQ_CTRL_RAW = 1, GAQ_CTRL_MELLINGER = 2 };
/* thrust (OU) noise source: off / on-device Philox4x32-10 / caller-supplied normals */
enum { GAQ_NOISE_OFF = 0, GAQ_NOISE_PHILOX = 1, GAQ_NOISE_INPUT = 2 };
/* reward of quadrotor.py:544-638 / log-distance variant of quadrotor_multi.py:550-650 */
enum { GAQ_REW_QUADROTOR = 0, GAQ_REW_MULTI_LOG = 1 };
/* observation layout flags (get_state.py): base is [pos-goal, vel, R row-major, omega] = 18 */
enum { GAQ_OBS_BODY_FRAME = 1, GAQ_OBS_APPEND_H = 2, GAQ_OBS_APPEND_ACC = 4, GAQ_OBS_APPEND_ACT = 8,
       /* variants whose code is complete in the reference but raises NameError as shipped (get_state.py lacks the imports of
        * `normal` / `R2quat`); pinned by the patched-import fixture G15: the quaternion R2quat(rot) in place of the 9 words of R
        * (get_state.py:276-322; 13-word base block), the noisy normalised thrust-to-weight / torque-to-thrust ratio appended
        * (:325-384) */
       GAQ_OBS_QUAT = 16, GAQ_OBS_APPEND_T2W = 32, GAQ_OBS_APPEND_T2T = 64 };

/* Derived model constants: what QuadrotorDynamics.update_model computes (quadrotor.py:142-208). */
typedef struct gaq_model {
  double mass;
  double inertia[3];        /* diagonal of I_com (:152) */
  double thrust_max[4];     /* g*m*t2w*asym/4 (:175) */
  double torque_max[4];     /* t2t*thrust_max (:176) */
  double prop_pos[12];      /* [4][3], COM-corrected (:179) */
  double damp_time_up;      /* motor time constants, seconds (:159-160) */
  double damp_time_down;
  double linearity;         /* (:156) */
  double arm;               /* |motor_xy| (:200): crash height */
  double ou_sigma;          /* 0.2 * thrust_noise_ratio (:198) */
  double vel_damp;          /* (:163) */
  double damp_omega_quadratic; /* (:164) */
  double c_drag, c_roll;    /* (:157-158) */
} gaq_model;

#define GAQ_MODEL_NUM_DOUBLES 33  /* sizeof(gaq_model)/8: row length of gaq_set_params */

/* Reward weights (quadrotor.py:799-806; multi: quadrotor_multi.py:811-818). */
typedef struct gaq_rew_coeff {
  float pos, effort, crash, orient, yaw, rot, attitude, spin, action_change, vel;
  float pos_offset, pos_log_weight, pos_linear_weight;
} gaq_rew_coeff;

/* SensorNoise (sensor_noise.py:57-99); enabled = 0 is the reference's `sense_noise=None` (bypass).
 * gyro_norm_std == 0: white gyro noise of std gyro_noise_density (:130-131); != 0: the per-env gyro-bias random
 * walk of add_noise_to_omega (:160-168) with gyro_random_walk / gyro_bias_correlation_time. */
typedef struct gaq_sense_noise {
  int32_t enabled;
  float pos_norm_std, pos_unif_range, vel_norm_std, vel_unif_range, quat_norm_std, quat_unif_range;
  float gyro_noise_density, acc_static_noise_std, acc_dynamic_noise_ratio;
  float gyro_norm_std, gyro_random_walk, gyro_bias_correlation_time;
} gaq_sense_noise;

/* Swarm layer (BASELINE config 5: "8-agent swarm x 131072 worlds with neighbour-distance reward").  The reference
 * snapshot contains no multi-agent code (its quadrotor_multi fork is a single-agent env with a log-distance reward),
 * so this is the library's own specification -- parity-unpinned, see DESIGN.md "Swarm layer":
 *   world w = envs [w*agents, (w+1)*agents); agents is a power of two <= 16 and divides num_envs and env_id_offset;
 *   goal of agent a = (0,0,2) + goal_radius * (cos, sin, 0)(2 pi a / agents);
 *   reward_i -= dt * sum_{j != i} ( w_collision * [d_ij < collision_dist] + w_prox * max(0, 1 - d_ij / prox_dist) );
 *   optional collision response (gaq_swarm.response): colliding, approaching pairs exchange their normal relative velocity;
 *   observation = the configured self block + (pos_j - pos_i, vel_j - vel_i) for j = a+1 .. a+agents-1 (mod agents), as fp32
 *   differences of the fp32-rounded positions / velocities (exact to an ulp of the position, not of the difference). */
typedef struct gaq_swarm {
  int32_t agents;           /* 0 or 1: off */
  float goal_radius;
  float collision_dist;
  float prox_dist;
  float w_collision, w_prox;
  int32_t response;         /* 1: collision RESPONSE -- a pair closer than collision_dist that is still approaching exchanges the normal
                               component of its relative velocity (elastic collision of equal masses: v_i -= ((v_i - v_j) . n) n with
                               n = (p_i - p_j) / d_ij, summed over the colliding neighbours), applied to the integrated state before
                               reward and observation; 0: collisions are a reward term only (agents pass through each other) */
} gaq_swarm;

/* Everything QuadrotorEnv.__init__ fixes for the life of the env (quadrotor.py:653-827). */
typedef struct gaq_config {
  uint32_t struct_size;     /* = sizeof(gaq_config), ABI check */
  uint32_t abi_version;     /* = GAQ_ABI_VERSION */
  int64_t num_envs;         /* N envs held by this handle (this GPU's shard) */
  int64_t env_id_offset;    /* global index of env 0: RNG streams are keyed by global id, so results
                               do not depend on how a batch is sharded over GPUs */
  int32_t device;           /* HIP device ordinal */
  uint64_t seed;
  double sim_freq;          /* dt = 1/sim_freq (:790) */
  int32_t sim_steps;        /* step1 calls per env step (:261-262) */
  int32_t ep_len;           /* int(ep_time/(dt*sim_steps)) (:792); done = tick > ep_len (:987) */
  double room_size;         /* room box [[-s,-s,0],[s,s,s]] (:723) */
  double gravity;           /* used by the accelerometer only (:436) */
  double t2w_std, t2t_std;  /* relative noise of the t2w / t2t observation components (ctor arguments, :658; clip and scaling
                               ranges [1.5, 10] and [0.005, 1] are the reference's constants, :707-712) */
  int32_t control;          /* GAQ_CTRL_* */
  int32_t noise;            /* GAQ_NOISE_* */
  int32_t reward_mode;      /* GAQ_REW_* */
  int32_t obs_flags;        /* GAQ_OBS_* */
  int32_t auto_reset;       /* 1: an env that reports done is re-initialised inside the same launch and
                               its returned obs is the first of the new episode; 0: reference behaviour
                               (caller resets) */
  int32_t init_random_state;/* (:1100-1115) */
  int32_t resample_goal;    /* (:1078-1081) */
  int32_t per_env_params;   /* 1: model constants come from gaq_set_params, one row per env */
  int32_t compact_done;     /* 1: keep a per-step compacted list of done env indices */
  int32_t obs_state_alias;  /* How the 18 integrator words are stored (18-word world-frame observation with RawControl and the
                               default reward terms only -- see gaq_obs_is_state; otherwise 0 is used whatever is asked):
                               0: fp64 planes, the observation tensor is write-only (the caller owns it, like the reference).
                               1: split state whose fp32 head IS the caller's observation tensor (value = obs word + residual
                                  bits held by the library).  Least traffic (277 B/env-step).  CONTRACT: the buffer written by
                                  step k (or reset) is step k+1's INPUT -- it must still hold those bytes and stay allocated
                                  when step k+1 runs; the same buffer may be passed again (in place) or another one (rollout
                                  storage [T,N,D]).  Do not modify or free it in between.  GAQ_CHECK_ALIAS=1 in the environment
                                  makes every call verify a checksum of those rows and fail with GAQ_ERR_STATE if they changed.
                               2: the same split state with LIBRARY-owned heads; the caller's tensor receives a copy (+72 B/env-
                                  step of writes).  No contract: the caller may do anything with the observation.  This is what
                                  gym_art_amd.QuadrotorEnv uses unless told otherwise. */
  int32_t fp32_state;       /* 1: fp32 arithmetic and state -- the 18-word observation tensor IS the whole state (implies the
                               obs_state_alias contract).  Throughput-first: trajectories drift 1e-5..3e-4 (relative) from
                               the reference over 500 steps, i.e. OUTSIDE the 1e-5 parity bar that the default fp64 path
                               meets.  Refused (GAQ_ERR_INVALID) for configurations that need the generic kernel. */
  int32_t excite;           /* 1: a new goal ~ U(-0.5,0.5)^2 x U(1.5,2.5) whenever tick % 5 == 0 (:957-963) */
  int32_t aux_outputs;      /* 1: keep what the info dict's obs_comp needs beyond the state (quadrotor.py:994-1006): the last
                               sub-step's accelerometer, omega_dot and torque, the controller output and thrust_cmds_damp,
                               GAQ_AUX_WORDS floats per env, read with gaq_get_aux.  Runs the generic kernel. */
  int32_t action_f32;       /* RawControl arithmetic when the CALLER's action array is float32 (what action_space.sample() and most
                               policies produce): the reference then forms 0.5*(a+1) in float32 (quadrotor_control.py:88-92);
                               0 = the caller's array is float64 holding float32-representable values (sum exact).  The two differ
                               by <= 6e-8 in the command.  Can be switched per call with gaq_set_action_dtype. */
  int32_t sense_input;      /* 1: sensor-noise (and t2w / t2t observation) draws come from gaq_set_sense_input_dev (parity tests)
                               instead of the device RNG */
  gaq_swarm swarm;
  gaq_rew_coeff rew;
  gaq_sense_noise sense;    /* observation noise; forces the generic kernel and the plain state layout */
  gaq_model model;          /* used when per_env_params == 0 */
} gaq_config;

/* number of visible HIP devices (0 when none / no driver) */
int gaq_num_devices(void);
const char* gaq_last_error(void);
int gaq_abi_version(void);

/* 1 if this library is a MEASUREMENT build (-DGAQ_DIAG_BUILD): only such a build honours the timing-only ablations of GAQ_ABLATE
 * (which give wrong physics by construction); the product library refuses a non-zero GAQ_ABLATE at gaq_create. */
int gaq_is_diag_build(void);

/* Which kernel would gaq_create pick?  Pure host logic (no device needed): the configuration -> feature mask -> instantiation map
 * of the library, so that every reachable combination can be enumerated on a GPU-less host (tests/test_plan_cpu.py) -- a mask
 * without an instantiation is a test failure there, not a runtime GAQ_ERR_STATE.  `motor_lag` / `rotor_drag`: what the per-env
 * parameters (gaq_set_params / the randomizer) bring, -1 = derive from cfg->model (uniform model) or gaq_create's assumption before
 * parameters arrive; `randomize_every`: gaq_randomizer.every; `num_cus`: compute units of the device (the small-batch size rule counts
 * waves per SIMD; hipDeviceProp_t::multiProcessorCount, 256 on a whole MI355X).  No reference counterpart. */
typedef struct gaq_plan_info {
  int32_t obs_dim;
  int32_t state_layout;         /* as gaq_state_layout: 0 fp64 planes, 1 heads in the caller's tensor, 2 library-owned heads */
  int32_t fp32;                 /* fp32_state in effect */
  int32_t step_variant;         /* feature mask of the step kernel (csrc/quad_core.hpp: enum Feature) */
  int32_t step_instantiated;    /* 1 if the library holds that instantiation */
  int32_t launchable;           /* 1 if a step would launch (0 also for rotor drag arriving on a split-state handle: refused loudly) */
  int32_t rollout_variant;      /* mask of the fused kernel gaq_step_many_dev would launch, -1 = one launch per step */
  int32_t rollout_instantiated;
  int32_t lds_per_wave;         /* bytes of LDS per wave of the step launch */
  int32_t rows_variant;         /* the instantiation a step launches while packed rows are registered (gaq_set_packed_rows_dev: the rows are
                                   written by the step launch itself), -1 = the pack launch follows the step */
  int32_t ctr_variant;          /* the instantiation a step launches in graph-safe mode when it advances the step counter itself, -1 = a
                                   one-thread launch follows the step */
  /* graph-safe step counter of a handle of cfg->num_envs envs (gaq_set_graph_safe): the device words sum to step_index << ctr_shift; a
   * self-counting step launch has ctr_waves waves, every one adds 1 except the first, which adds ctr_inc0: 2^ctr_shift per launch */
  int32_t ctr_waves, ctr_shift, ctr_inc0;
} gaq_plan_info;
int gaq_plan(const gaq_config* cfg, int32_t motor_lag, int32_t rotor_drag, int32_t randomize_every, int32_t num_cus,
             gaq_plan_info* out);

/* QuadrotorEnv.__init__ (quadrotor.py:653-827): allocate device state for cfg->num_envs envs. */
int gaq_create(const gaq_config* cfg, gaq_env** out);
int gaq_destroy(gaq_env* env);
int gaq_obs_dim(const gaq_env* env);
/* 1 if this handle runs with obs_state_alias == 1 in effect (the caller's observation tensor is the state head) */
int gaq_obs_is_state(const gaq_env* env);
/* the obs_state_alias value in effect: 0 fp64 planes, 1 heads in the caller's tensor, 2 library-owned heads */
int gaq_state_layout(const gaq_env* env);
int64_t gaq_num_envs(const gaq_env* env);
/* feature mask of the step kernel this handle currently launches (gaq_plan_info.step_variant; changes when parameters arrive) */
int gaq_kernel_variant(const gaq_env* env);
/* ... and of the instantiation the NEXT gaq_step_dev launches: that kernel, or its twin that also writes the registered packed rows
 * (gaq_plan_info.rows_variant) / advances the graph-safe step counter itself (gaq_plan_info.ctr_variant) */
int gaq_launch_variant(const gaq_env* env);
/* test infrastructure: the distinct feature masks of the step (kind 0) / fused rollout (kind 1) instantiations THIS PROCESS has launched so
 * far, ascending; writes min(count, capacity) of them and returns the count (tools/kernel_coverage.py: which kernels a test run reached) */
int gaq_launched_variants(int kind, uint32_t* out, int capacity);

/* update_dynamics / resample_dynamics (quadrotor.py:852-894, :1030-1056) for per-env models:
 * `models` = `count` rows of gaq_model for envs [first, first+count).  Clears the SVD counter
 * and OU state of those envs like constructing a new QuadrotorDynamics does (:104, :198). */
int gaq_set_params(gaq_env* env, const gaq_model* models, int64_t first, int64_t count);
/* The same for a scattered set: models[k] goes to env env_idx[k] (per-episode re-randomisation of the envs that just
 * finished, dynamics_randomize_every, quadrotor.py:1063-1066) -- one call, one upload. */
int gaq_set_params_indexed(gaq_env* env, const gaq_model* models, const int64_t* env_idx, int64_t count);

/* ---- parameter pipeline on the device (SURVEY 8f.3 "or move it on device") -------------------------------------------
 * One quadrotor's parameter tree: the reference's nested dict (quad_models.py:1-176: geom / damp / noise / motor), flat,
 * in the dict's own order.  Shape of the shipped models (every link has a mass, the arms a length); RandomQuad's
 * density-based trees and dynamics_simplification stay on the host path (gaq_set_params). */
#define GAQ_TREE_DOUBLES 40
typedef struct gaq_quad_params {
  double body[4];        /* geom.body      l, w, h, m */
  double payload[4];     /* geom.payload   l, w, h, m */
  double arms[4];        /* geom.arms      l, w, h, m */
  double motors[3];      /* geom.motors    h, r, m */
  double propellers[3];  /* geom.propellers h, r, m */
  double motor_pos[3];   /* geom.motor_pos.xyz */
  double arms_pos[2];    /* geom.arms_pos  angle (deg), z */
  double payload_pos[3]; /* geom.payload_pos xy[2], z_sign */
  double damp[2];        /* damp.vel, damp.omega_quadratic */
  double noise[1];       /* noise.thrust_noise_ratio */
  double motor[11];      /* motor.thrust_to_weight, assymetry[4], torque_to_thrust, linearity, C_drag, C_roll, damp_time_up,
                            damp_time_down */
} gaq_quad_params;

/* RelativeSampler (quadrotor_randomization.py:345-358 -> perturb_dyn_parameters :70-104 -> check_quad_param_limits :16-46)
 * around `base`, per env, ON THE DEVICE, followed by QuadLink (inertia.py:182-310) and update_model (quadrotor.py:142-208).
 * every > 0: an env that reports done and whose finished-episode count k satisfies (k + 1) % every == 0 gets new
 * parameters INSIDE that step launch (dynamics_randomize_every, quadrotor.py:1063-1066), its SVD counter and OU state
 * cleared like a new QuadrotorDynamics (:104, :198): every env's next draw is derived ahead of time into a second set of
 * parameter planes (+360 B per env) and the step kernel only switches the env over.  Draws are Philox streams keyed by
 * (seed, global env index, resample count): the distribution of the reference's numpy draws, not its stream. */
typedef struct gaq_randomizer {
  int32_t sampler;                 /* 0: normal(loc = v, scale = |ratio/2 v|), 1: uniform(v - v ratio, v + v ratio);
                                      2: RandomQuad -- randomquad_parameters (quadrotor_randomization.py:142-243): a random
                                      quadrotor per draw; `ratio` and `base` are not used */
  int32_t every;                   /* dynamics_randomize_every (needs auto_reset = 1); 0 = only when gaq_randomize_dev is called */
  double ratio[GAQ_TREE_DOUBLES];  /* noise ratio per leaf (RelativeSampler noise_ratio / noise_ratio_custom), gaq_quad_params order */
  gaq_quad_params base;            /* the nominal model, dynamics_change already applied; C_drag = C_roll = 0 */
} gaq_randomizer;

/* Install the sampler on a per_env_params handle (RawControl, or Mellinger: the per-env inverse jacobians, quadrotor_control.py:290-291, are
 * rebuilt on the device whenever parameter planes change).  From here on the handle's parameters live on the device:
 * gaq_set_params is refused, gaq_get_params / gaq_get_param_trees read them back. */
int gaq_set_randomizer(gaq_env* env, const gaq_randomizer* rz);
/* resample_dynamics() now for the envs whose mask byte is non-zero (NULL = all): one launch, asynchronous on `stream`. */
int gaq_randomize_dev(gaq_env* env, const uint8_t* mask_dev_or_null, void* stream);
/* Caller-chosen trees for envs [first, first+count), derived on the device (QuadLink + update_model; the limits are NOT
 * applied): the device-side counterpart of gaq_set_params.  links_by_density != 0: RandomQuad's form of the tree -- the five `m`
 * leaves hold densities (mass = density x volume, inertia.py:96-97, :155-156) and arms.l is derived (:223-224). */
int gaq_set_param_trees(gaq_env* env, const gaq_quad_params* trees, int32_t links_by_density, int64_t first, int64_t count);
/* Read back: the derived constants (what update_model computed) / the sampled trees of envs [first, first+count). */
int gaq_get_params(gaq_env* env, gaq_model* models_out, int64_t first, int64_t count);
int gaq_get_param_trees(gaq_env* env, gaq_quad_params* trees_out, int64_t first, int64_t count);

/* ---- checkpoint / resume ----------------------------------------------------------------------------------------------
 * What a bit-exact continuation needs besides the state planes (gaq_get_state / gaq_set_state) and the parameters: the counters
 * behind the RNG keys (steps launched, reset calls) and, per env, the finished-episode and resample counts of the device
 * randomizer (NULL where not wanted / not a per_env_params handle).  gaq_set_counters on a handle with a randomizer installed
 * rebuilds every env's parameters from its resample count: they are a function of (seed, global env index, count).
 * No reference counterpart: the reference pickles its constructor arguments only (quadrotor.py:688). */
typedef struct gaq_counters {
  uint64_t step_index;    /* step launches so far: third word of the Philox keys of thrust noise and in-kernel resets */
  uint64_t reset_calls;   /* gaq_reset / gaq_reset_dev calls so far: keys the reset draws apart from the steps */
} gaq_counters;
int gaq_get_counters(gaq_env* env, gaq_counters* out, uint32_t* episodes_out_or_null, uint32_t* resamples_out_or_null);
int gaq_set_counters(gaq_env* env, const gaq_counters* in, const uint32_t* episodes_or_null, const uint32_t* resamples_or_null);

/* QuadrotorEnv.reset (quadrotor.py:1149 -> :1059-1144) for the envs whose mask byte is non-zero
 * (NULL = all).  Writes the [N,obs_dim] observation (rows of un-reset envs = current obs).  A masked reset leaves every bit of the
 * unmasked envs alone: with the gyro-bias random walk on, their rows are a peek (one add_noise call of sensor_noise.py:166 applied to the
 * row, not kept), whereas gaq_observe -- the reference's state_vector() -- advances the walk of every env like the reference does. */
int gaq_reset(gaq_env* env, const uint8_t* mask_or_null, float* obs_out);
int gaq_reset_dev(gaq_env* env, const uint8_t* mask_dev_or_null, float* obs_dev, void* stream);

/* QuadrotorEnv.step (quadrotor.py:1155 -> :942-1028). */
int gaq_step(gaq_env* env, const float* actions, float* obs, float* reward, uint8_t* done);
/* NB obs_state_alias == 1: obs_dev is ALSO the state head that the next gaq_step_dev / gaq_step_many_dev / gaq_reset_dev
 * reads -- keep it allocated and unmodified until then (see gaq_config.obs_state_alias). */
int gaq_step_dev(gaq_env* env, const float* actions_dev, float* obs_dev, float* reward_dev, uint8_t* done_dev,
                 void* stream);

/* T consecutive env steps with actions [T,N,4] and outputs [T,N,obs_dim], [T,N], [T,N]
 * resident on the device (the rollout loops of quadrotor.py:1278-1305, :1424-1428). */
int gaq_step_many_dev(gaq_env* env, int32_t T, const float* actions_dev, float* obs_dev, float* reward_dev,
                      uint8_t* done_dev, void* stream);

/* GAQ_NOISE_INPUT: normals for the NEXT step, layout [sim_steps][4][N] float32 (device pointer,
 * must stay valid until that step has run).  Stands in for numpy.random.randn inside OUNoise.noise
 * (quad_utils.py:197-201) so that noisy trajectories can be compared bit-for-bit in structure. */
int gaq_set_noise_input_dev(gaq_env* env, const float* normals_dev);

/* gaq_config.sense_input: the standard draws of the NEXT step's three SensorNoise.add_noise calls (quadrotor.py:946, :970,
 * :988; a reset or gaq_observe makes one call and reads call index 2), layout [3 calls][12 slots][3][N] float32, device
 * pointer valid until that step has run.  Slots in the order the reference draws them (sensor_noise.py:116-157): 0 pos normal,
 * 1 pos uniform, 2 vel normal, 3 vel uniform, 4 gyro normal (bias model: the bias increment), 5 gyro white normal (bias model
 * only), 6 quat normal, 7 quat uniform, 8 acc static normal, 9 acc proportional normal; then the state function's own
 * draws, first column only: 10 the t2w normal, 11 the t2t normal (get_state.py:335, :375).  Normals are N(0,1), uniforms U(0,1). */
int gaq_set_sense_input_dev(gaq_env* env, const float* draws_dev);

/* see gaq_config.action_f32 */
int gaq_set_action_dtype(gaq_env* env, int32_t is_float32);

/* gaq_config.aux_outputs: per env [accelerometer 3 | omega_dot 3 | torque 3 | controller.action 4 | thrust_cmds_damp 4] of the
 * most recent step (info["obs_comp"] entries acc, omega_dot, torque, act_clipped, act_filtered; quadrotor.py:994-1006). */
#define GAQ_AUX_WORDS 17
int gaq_get_aux(gaq_env* env, float* host_out /* [N, GAQ_AUX_WORDS] */);

/* Full state exchange (teacher forcing, checkpoint/resume, tests).  Host buffer of
 * GAQ_STATE_PLANES planes of N doubles, plane-major:
 *   0-2 pos, 3-5 vel, 6-14 rot (row-major), 15-17 omega, 18-21 thrust_rot_damp,
 *   22-25 thrust_cmds_damp, 26-29 OU state, 30-33 previous action, 34-36 goal,
 *   37 tick, 38 SVD counter (sub-steps since the last re-orthonormalisation),
 *   39-41 gyro bias of the sensor-noise model (SensorNoise.gyro_bias, sensor_noise.py:98). */
#define GAQ_STATE_PLANES 42
int gaq_get_state(gaq_env* env, double* host_planes);
int gaq_set_state(gaq_env* env, const double* host_planes);

/* Observation of the current state without stepping (state_vector(self), quadrotor.py:1143). */
int gaq_observe(gaq_env* env, float* obs_out);

/* Rollout bookkeeping around step() (the loops of quadrotor.py:1278-1305, :1424-1428; `traj_count` :990).
 * - terminal observations: with auto_reset the row returned with done=1 belongs to the NEW episode; when a buffer
 *   [N,obs_dim] is registered here, the last observation of the finished episode (what the reference returns with
 *   done=True; every termination on this path is a time-limit truncation) is written to that env's row in the same
 *   launch.  Rows of envs that did not finish are left untouched.  NULL unregisters.
 * - episode tracking: per-env running return and length on the device; gaq_episode_stats returns (and optionally
 *   clears) the totals over the episodes finished so far. */
int gaq_set_terminal_obs_dev(gaq_env* env, float* term_obs_dev);
int gaq_track_episodes(gaq_env* env, int32_t enabled);
int gaq_episode_stats(gaq_env* env, int64_t* episodes, double* return_sum, double* length_sum, double* return_sqsum,
                      int32_t clear);

/* compact_done: indices (local) of the envs that reported done in the last step. */
int gaq_done_list(gaq_env* env, uint32_t* idx_out, int64_t capacity, int64_t* count_out);

/* Multi-GPU return path ("pack [obs, reward, done] into the same buffer as a 20-word row to keep it a single
 * collective"): rows_dev[i] = [obs[i, 0..D-1], reward[i], (float)done[i]], i.e. [N, obs_dim + 2] float32 row-major.  One
 * small launch on `stream`; the caller then issues ONE gather of the packed rows (gym_art_amd/sharding.py). */
int gaq_pack_rows_dev(gaq_env* env, const float* obs_dev, const float* reward_dev, const uint8_t* done_dev, float* rows_dev,
                      void* stream);

/* The same rows WITHOUT the extra launch: once a buffer [N, obs_dim + 2] is registered here, every gaq_step_dev call also leaves the
 * packed rows of its outputs in it -- for the split-state kernels whose observation is the state's heads (obs_state_alias 1 / 2 with the
 * 18-word observation: BASELINE config 4's shards) assembled in the step kernel's LDS buffer and stored with the launch's other 16-byte
 * pieces, for every other configuration by the pack launch, enqueued by gaq_step_dev itself: bit for bit what gaq_pack_rows_dev makes of
 * obs / reward / done either way.  The step's ordinary outputs are still written.  NULL unregisters.  Not honoured by the fused
 * rollouts of gaq_step_many_dev (it then steps one launch at a time). */
int gaq_set_packed_rows_dev(gaq_env* env, float* rows_dev_or_null);

/* number of envs whose reward was non-finite since the last call (clears the counter) */
int gaq_nan_count(gaq_env* env, int64_t* count_out);

/* HIP-graph capture.  The *_dev entry points only enqueue kernels, so they can be captured (e.g. inside
 * torch.cuda.graph together with the policy).  By default the step index that keys the noise / reset random streams
 * is a host counter passed by value -- a captured launch would replay the same draws.  With graph-safe mode on, the
 * index lives in device memory, so every replay is a new step.  At small batches (where a launch is latency: up to two waves per
 * SIMD) the split-state kernels advance it THEMSELVES -- every wave checks in with one non-returning atomic after reading it: ONE graph
 * node per step and nothing waits; larger batches and every other kernel are followed by a one-thread launch.
 * Alias layout: capture with the observation buffer used in place (same tensor in and out of every captured step). */
int gaq_set_graph_safe(gaq_env* env, int32_t enabled);

/* Device time (ms, HIP events on the launch stream) of the most recent gaq_step*_dev /
 * gaq_step call's kernel(s); used by bench.py for the roofline figure. */
int gaq_last_kernel_ms(gaq_env* env, float* ms_out);
int gaq_set_timing(gaq_env* env, int32_t enabled);

/* Measurement aid (bench.py roofline.peak_measured; SURVEY 8d "also measure an on-box copy kernel"): copy `bytes` (a multiple of 16) from
 * src to dst on the current device with the access shape of the step kernels' streaming traffic -- one 16-byte load and one 16-byte store
 * per lane -- asynchronously on `stream`.  2 x bytes / time is the bandwidth the step kernels' layout can reach on THIS box. */
int gaq_hbm_copy_dev(void* dst_dev, const void* src_dev, size_t bytes, void* stream);

int gaq_synchronize(gaq_env* env);
/* the handle's private hipStream_t (used by the host-pointer entry points) */
void* gaq_stream(gaq_env* env);

/* ---- one batch over several devices, ONE process (BASELINE config 4 for a plain-C caller; SURVEY 8b `device_ids`, 8e "single process,
 * one stream per device").  The reference loop `reset(); while not done: step()` (quadrotor.py:1278-1305, :1424-1428) stays one
 * call per step: the batch of cfg->num_envs envs is cut into contiguous shards of whole 64-env tiles (and whole swarm worlds), shard k
 * lives on device_ids[k] as an ordinary gaq_env with env_id_offset = cfg->env_id_offset + first_k -- the random streams are keyed by the
 * GLOBAL env index, so results do not depend on the sharding, bit for bit -- and every call fans out over the shards' own streams:
 *   actions [N,4] on device_ids[0] --(peer copy of each remote shard's slice)--> step launch per shard --(peer copy of the shard's
 *   obs / reward / done slices)--> the caller's [N, ...] tensors on device_ids[0].
 * Shards that live on device_ids[0] itself read and write the caller's tensors in place (no copy).  A device may be listed more than
 * once (its shards then share it: how a one-GPU box tests this).  `stream` is a stream of device_ids[0]; the call is asynchronous on
 * it like gaq_step_dev: work of the other devices is ordered after what `stream` held at the call and `stream` continues after it.
 * cfg->device is ignored; every other field means what it means for gaq_create.  Per-shard settings (parameters, randomizer, state
 * exchange, graph-safe mode ...) go through the shard handles: gaq_sharded_shard(). */
typedef struct gaq_sharded gaq_sharded;
int gaq_create_sharded(const gaq_config* cfg, const int32_t* device_ids, int32_t num_devices, gaq_sharded** out);
/* the same over handles the caller made (shard k = envs[k], consecutive global ranges: env_id_offset of shard k+1 = that of shard k + its
 * num_envs; same observation width; every shard but the last a multiple of 64 envs).  The handles stay the caller's: destroy them
 * AFTER the sharded handle.  (gym_art_amd.QuadrotorEnv(device_ids=[...]) builds its shards as Python envs and steps them through this.) */
int gaq_sharded_from_handles(gaq_env* const* envs, int32_t num_shards, gaq_sharded** out);
int gaq_destroy_sharded(gaq_sharded* s);
int gaq_sharded_num_shards(const gaq_sharded* s);
gaq_env* gaq_sharded_shard(gaq_sharded* s, int32_t k);                 /* borrowed */
int gaq_sharded_range(const gaq_sharded* s, int32_t k, int64_t* first, int64_t* count, int32_t* device);
int64_t gaq_sharded_num_envs(const gaq_sharded* s);
/* the split gaq_create_sharded makes of n envs over `num_shards` shards (pure host arithmetic: no device needed): whole 64-env tiles,
 * rounded up to `align` envs (swarm agents per world, 1 otherwise), the first shards one unit larger */
int gaq_shard_range(int64_t n, int32_t num_shards, int32_t k, int32_t align, int64_t* first, int64_t* count);
/* QuadrotorEnv.reset() / step() of the whole batch; device pointers on device_ids[0], asynchronous on `stream` */
int gaq_reset_sharded_dev(gaq_sharded* s, const uint8_t* mask_dev_or_null, float* obs_dev, void* stream);
int gaq_step_sharded_dev(gaq_sharded* s, const float* actions_dev, float* obs_dev, float* reward_dev, uint8_t* done_dev, void* stream);
/* ... and with host pointers (synchronous; staged through device_ids[0]) */
int gaq_reset_sharded(gaq_sharded* s, const uint8_t* mask_or_null, float* obs_out);
int gaq_step_sharded(gaq_sharded* s, const float* actions, float* obs, float* reward, uint8_t* done);
int gaq_synchronize_sharded(gaq_sharded* s);

#ifdef __cplusplus
}
#endif
#endif /* GAQ_H */
