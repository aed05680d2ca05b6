// gaq.hip -- the C ABI of include/gaq.h, the launch logic and the small kernels (reset, export, parameter pipeline, bookkeeping) around the
// fused step / rollout kernels of gaq_kernels.hpp (instantiated in gaq_inst.hip).
//
// One lane = one environment; one wavefront = one TILE of 64 environments.  Device state is kept
// tile-major ("array of struct of arrays"): for every tile each state component is a run of 64 values
// and the components of a tile are contiguous, so a wavefront's whole working set is a handful of
// contiguous 1-KiB pieces.  The step kernel moves those pieces with 16-byte-per-lane transfers --
// HBM -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, no VGPR staging), LDS -> HBM by ds_read_b128 +
// buffer_store_dwordx4 -- and each lane picks its own environment's values out of the LDS image with
// conflict-free ds_read_b64.  (Measured on MI355X: a copy runs at 6.2 TB/s with 16 B/lane, 5.8 with
// 8 B/lane and 3.8 with 4 B/lane; the first version of this kernel used 8- and 4-byte plane accesses
// and ran exactly at the rate those widths allow.)  The caller-facing observation tensor [N,D] is
// transposed through the same LDS buffer and written with 16 B/lane stores.  No MFMA: the largest
// contraction is 3x3.3x3.  See DESIGN.md for the byte accounting and quad_core.hpp for the arithmetic.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <set>
#include <string>
#include <vector>

#include "gaq_kernels.hpp"

// every step / rollout instantiation is compiled in gaq_inst.hip (eight translation units, in parallel); here they are only declared
#define GAQ_X(FEAT) extern template __global__ GAQ_STEP_SIG(FEAT)
GAQ_STEP_ALL(GAQ_X)
#undef GAQ_X
#define GAQ_X(FEAT) extern template __global__ GAQ_ROLL_SIG(FEAT)
GAQ_ROLL_ALL(GAQ_X)
#undef GAQ_X

namespace {

using namespace gaqk;

// ---- optional episode bookkeeping: running return / length per env, totals of finished episodes ----
__global__ __launch_bounds__(kBlock) void episode_kernel(int64_t n, const float* __restrict__ reward,
                                                          const uint8_t* __restrict__ done, float* __restrict__ ep_ret,
                                                          uint32_t* __restrict__ ep_len, double* __restrict__ acc) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  float ret = 0.0f; uint32_t len = 0; bool fin = false;
  if (i < n) {
    ret = ep_ret[i] + reward[i];
    len = ep_len[i] + 1u;
    fin = done[i] != 0;
    ep_ret[i] = fin ? 0.0f : ret;
    ep_len[i] = fin ? 0u : len;
  }
  // wave-level reduction of the finished episodes, then one atomic per wave and quantity
  double c = fin ? 1.0 : 0.0, sr = fin ? (double)ret : 0.0, sl = fin ? (double)len : 0.0, sq = fin ? (double)ret * ret : 0.0;
  if (__ballot(fin)) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { c += __shfl_down(c, o); sr += __shfl_down(sr, o); sl += __shfl_down(sl, o); sq += __shfl_down(sq, o); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(acc + 0, c); atomicAdd(acc + 1, sr); atomicAdd(acc + 2, sl); atomicAdd(acc + 3, sq); }
  }
}

// ---- multi-GPU return path: [obs | reward | done] packed into one [N, D+2] fp32 row array, so that the
// stacked result travels in ONE collective.  4 B per lane, fully coalesced on the store side; 20 MB at 131 072 envs.
__global__ __launch_bounds__(kBlock) void pack_rows_kernel(int64_t n, int D, const float* __restrict__ obs,
                                                            const float* __restrict__ reward, const uint8_t* __restrict__ done,
                                                            float* __restrict__ rows) {
  const int W = D + 2;
  const int64_t total = n * W;
  for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < total; k += (int64_t)gridDim.x * kBlock) {
    const int64_t i = k / W;
    const int c = (int)(k - i * W);
    rows[k] = c < D ? obs[i * D + c] : c == D ? reward[i] : (float)done[i];
  }
}

// update_dynamics builds a NEW QuadrotorDynamics (quadrotor.py:857): since_last_svd = 0 (:104) and a fresh OUNoise (:198)
// for the envs whose parameters were replaced: env idx[k], or first + k when idx is null
__global__ __launch_bounds__(kBlock) void clear_dynamics_kernel(DevPtrs p, const int64_t* __restrict__ idx, int64_t first, int64_t count) {
  const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= count) return;
  const int64_t i = idx ? idx[k] : first + k;
  p.ctr[i] &= 0xFFFFu;
  float* ou = p.ou + (i / kTile) * (4 * kTile) + (i % kTile);
#pragma unroll
  for (int j = 0; j < 4; ++j) ou[j * kTile] = 0.0f;
}

// order-independent 64-bit checksum of a word array: sum over i of mix(word_i, i) (GAQ_CHECK_ALIAS)
__global__ __launch_bounds__(kBlock) void checksum_kernel(const uint32_t* __restrict__ w, int64_t n, uint64_t* __restrict__ out) {
  uint64_t acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    uint64_t x = ((uint64_t)w[i] << 32) ^ (uint64_t)i * 0x9E3779B97F4A7C15ull;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    acc += x;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(reinterpret_cast<unsigned long long*>(out), (unsigned long long)acc);
}

// ---- parameter pipeline on the device (quad_params_dev.hpp): sampler + QuadLink + update_model per env ---------------
struct Randomizer {           // gaq_randomizer, by value in the launch arguments (660 B)
  int32_t sampler, every;
  double ratio[gaq::TL_COUNT];
  gaq::ParamTree base;
};

// [ntiles*64] the resample count at which ALL 45 planes of an env were last written: the fourth quarter of the traj | rcount | rz_flag
// allocation (the step kernels' buffer resource covers the first three)
__device__ __forceinline__ uint32_t* pfull_of(const DevPtrs& p) { return p.traj + 3 * p.ntiles * kTile; }

// env i's planes of the tile-major parameter array, exactly what set_params_impl writes on the host path
// (`staged`: into the env's row of par_next -- [45] doubles, plane order -- while the env keeps flying its current planes; the step
//  kernel moves the row into the planes when it promotes the env, and clears the counters then)
__device__ __forceinline__ void write_model_planes(const DevPtrs& p, double dt, int64_t i, const gaq::DerivedModel& dm, bool staged = false) {
  double* tp = staged ? p.par_next + i * (int64_t)kPar
                      : const_cast<double*>(p.par) + (i / kTile) * (int64_t)(kPar * kTile) + (i % kTile);
  const int stride = staged ? 1 : kTile;
  auto P = [&](int plane) -> double& { return tp[plane * stride]; };
  P(PP_MASS) = dm.mass; P(PP_INV_MASS) = 1.0 / dm.mass;
#pragma unroll
  for (int j = 0; j < 3; ++j) { P(PP_INERTIA + j) = dm.inertia[j]; P(PP_INV_INERTIA + j) = 1.0 / dm.inertia[j]; }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    P(PP_THRUST_MAX + j) = dm.thrust_max[j]; P(PP_TORQUE_MAX + j) = dm.torque_max[j];
    P(PP_PROP_X + j) = dm.prop_pos[3 * j]; P(PP_PROP_Y + j) = dm.prop_pos[3 * j + 1]; P(PP_PROP_Z + j) = dm.prop_pos[3 * j + 2];
  }
  P(PP_TAU_UP) = 4 * dt / (dm.damp_time_up + 1e-6); P(PP_TAU_DOWN) = 4 * dt / (dm.damp_time_down + 1e-6);   // quadrotor.py:284-285
  P(PP_T_UP) = dm.damp_time_up; P(PP_T_DOWN) = dm.damp_time_down;
  P(PP_LINEARITY) = dm.linearity; P(PP_ARM) = dm.arm; P(PP_VEL_DAMP) = dm.vel_damp; P(PP_DAMP_Q) = dm.damp_omega_quadratic;
  P(PP_C_DRAG) = dm.c_drag; P(PP_C_ROLL) = dm.c_roll;
  if (staged) P(PP_OU_SIGMA) = (double)(float)dm.ou_sigma;                                               // (a double in the row)
  else reinterpret_cast<float*>(tp - (i % kTile) + PP_OU_SIGMA * kTile)[i % kTile] = (float)dm.ou_sigma;  // fp32 plane
  // construction hints of the compact path: derive_tree formed torque_max and prop_pos.xy with these very operations
  P(PP_T2T) = dm.t2t; P(PP_MX) = dm.motor_x; P(PP_MY) = dm.motor_y; P(PP_COMX) = dm.com[0]; P(PP_COMY) = dm.com[1];
  P(PP_COMPACT_OK) = 1.0;
  if (staged) return;
  // every plane of env i now belongs to its resample count (a hot-planes-only promotion in the step kernel moves 19 of the 45 and leaves
  // this word alone: count != pfull then says "the other 26 are a draw behind", gaq_get_params)
  pfull_of(p)[i] = p.rcount[i];
  // a new QuadrotorDynamics: since_last_svd = 0 (quadrotor.py:104) and a fresh OUNoise (:198)
  p.ctr[i] &= 0xFFFFu;
  float* ou = p.ou + (i / kTile) * (4 * kTile) + (i % kTile);
#pragma unroll
  for (int j = 0; j < 4; ++j) ou[j * kTile] = 0.0f;
}

// mode 0: the refill pass of dynamics_randomize_every (quadrotor.py:1063-1066 per env).  The step kernel PROMOTES a finished, due
//         env to the planes staged for it in par_next and flags it; this pass derives the following draw (index = the env's resample
//         count) into par_next for every flagged env.  Nothing waits for it: an env needs its staged planes only when its next
//         episode ends, so the pass runs every min(64, ep_len + 1) steps (launch_step) instead of between every two step launches,
//         where one lane's ~6000-instruction derivation was 27 us of pure latency (122 -> ~95 us per step with every episode of
//         2^20 staggered envs re-randomised);
// mode 1: now, for the envs of `sel` (null = all): current planes = the next draw, and the env is flagged for mode 0.
// mode 2: gaq_set_counters -- every env's planes rebuilt from its resample count; mode 4: the same where the planes are behind the count
//         (envs promoted with the hot planes only since their last full write); mode 3: gaq_get_params, see there.
// trees_out != nullptr (gaq_get_param_trees): no state is touched, the tree of env first + k's LAST resample is written out.
__global__ __launch_bounds__(kBlock) void rerandomize_kernel(DevPtrs p, StepCfg cfg, Randomizer rz, const uint8_t* __restrict__ sel,
                                                              int mode, double* __restrict__ trees_out, int64_t first, int64_t count) {
  const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (mode == 3) {   // gaq_get_params: a READ.  trees_out[k][kPar] = the full plane row of env first + k's last draw where the planes in
    if (k >= count) return;                     // memory are behind it (hot-planes-only promotions); nothing of the handle is touched
    const int64_t i = first + k;
    const uint32_t rc = p.rcount[i];
    double* row = trees_out + k * (int64_t)kPar;
    if (rc == pfull_of(p)[i]) { row[PP_COMPACT_OK] = -1.0; return; }      // every plane in memory is current (never drawn, or written whole)
    gaq::ParamTree t;
    if (rz.sampler == 2) gaq::random_quad_tree(cfg.seed, cfg.env_offset + (uint64_t)i, rc - 1, t);
    else gaq::perturb_tree(rz.base, rz.ratio, rz.sampler, cfg.seed, cfg.env_offset + (uint64_t)i, rc - 1, t);
    gaq::DerivedModel dm;
    gaq::derive_tree(t, dm, rz.sampler == 2);
    DevPtrs q = p;
    q.par_next = trees_out - first * (int64_t)kPar;                       // (write_model_planes' staged form writes row i of par_next)
    write_model_planes(q, cfg.dt, i, dm, true);
    return;
  }
  if (trees_out) {
    if (k >= count) return;
    const int64_t i = first + k;
    gaq::ParamTree t;
    const uint32_t rc = p.rcount[i];
    if (rc == 0) { t = rz.base; }
    else if (rz.sampler == 2) { gaq::random_quad_tree(cfg.seed, cfg.env_offset + (uint64_t)i, rc - 1, t); }
    else { gaq::perturb_tree(rz.base, rz.ratio, rz.sampler, cfg.seed, cfg.env_offset + (uint64_t)i, rc - 1, t); }
    for (int j = 0; j < gaq::TL_COUNT; ++j) trees_out[k * gaq::TL_COUNT + j] = t.v[j];
    return;
  }
  if (mode == 0) {
    const int64_t i = k;
    if (i >= p.n) return;
    const uint32_t promoted = p.rz_flag[i];
    if (!promoted) return;
    if (promoted > 1u) atomicAdd(p.rz_overrun, promoted - 1u);      // consumed planes that were one draw old: must not happen
    gaq::ParamTree t;
    const uint32_t rc = p.rcount[i];
    if (rz.sampler == 2) gaq::random_quad_tree(cfg.seed, cfg.env_offset + (uint64_t)i, rc, t);
    else gaq::perturb_tree(rz.base, rz.ratio, rz.sampler, cfg.seed, cfg.env_offset + (uint64_t)i, rc, t);
    gaq::DerivedModel dm;
    gaq::derive_tree(t, dm, rz.sampler == 2);
    write_model_planes(p, cfg.dt, i, dm, true);
    p.rz_flag[i] = 0;
    return;
  }
  const int64_t i = k;
  if (i >= p.n) return;
  if (mode == 2 || mode == 4) {   // gaq_set_counters: the current planes are those of the env's LAST draw (count - 1); nothing else is touched
    const uint32_t rc = p.rcount[i];                                  // (mode 4: only where the planes in memory are behind the count)
    if (mode == 4 && rc == pfull_of(p)[i]) return;
    gaq::ParamTree t;
    if (rc == 0) { t = rz.base; if (rz.sampler == 2) { pfull_of(p)[i] = 0u; return; } }      // never drawn: the planes the handle was given stay
    else if (rz.sampler == 2) gaq::random_quad_tree(cfg.seed, cfg.env_offset + (uint64_t)i, rc - 1, t);
    else gaq::perturb_tree(rz.base, rz.ratio, rz.sampler, cfg.seed, cfg.env_offset + (uint64_t)i, rc - 1, t);
    gaq::DerivedModel dm;
    gaq::derive_tree(t, dm, rz.sampler == 2);
    const uint32_t keep = p.ctr[i];
    float ou[4];
    float* op = p.ou + (i / kTile) * (4 * kTile) + (i % kTile);
    for (int j = 0; j < 4; ++j) ou[j] = op[j * kTile];
    write_model_planes(p, cfg.dt, i, dm);                            // (clears the SVD counter and the OU state: put them back)
    p.ctr[i] = keep;
    for (int j = 0; j < 4; ++j) op[j * kTile] = ou[j];
    if (p.rz_every > 0) p.rz_flag[i] = 1u;
    return;
  }
  if (!(sel == nullptr || sel[i] != 0)) return;
  const uint32_t rc = p.rcount[i];
  p.rcount[i] = rc + 1u;
  gaq::ParamTree t;
  if (rz.sampler == 2) gaq::random_quad_tree(cfg.seed, cfg.env_offset + (uint64_t)i, rc, t);
  else gaq::perturb_tree(rz.base, rz.ratio, rz.sampler, cfg.seed, cfg.env_offset + (uint64_t)i, rc, t);
  gaq::DerivedModel dm;
  gaq::derive_tree(t, dm, rz.sampler == 2);
  write_model_planes(p, cfg.dt, i, dm);
  if (p.rz_every > 0) p.rz_flag[i] = 1u;  // its staged planes are one draw behind now
}

// Mellinger on per-env models whose parameters the DEVICE samples: the inverse jacobian of env i (quadrotor_control.py:192-203, :290-291)
// from the parameter planes the step kernels fly with (load_model: the compact construction included), for every env or for those that
// finished in the step launch just before (`done`: the only ones a launch can have promoted to new planes).  The same Gauss-Jordan
// elimination as the host's inverse_jacobian; thrust_max / mass is taken as thrust_max * (1 / mass) -- the plane the kernels read.
__global__ __launch_bounds__(kBlock) void jinv_kernel(DevPtrs p, StepCfg cfg, Model<double> um, const uint8_t* __restrict__ done) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= p.n || !p.jinv) return;
  if (done && !done[i]) return;
  Model<double> m;
  load_model<gaq::F_PER_ENV>(p, cfg, i / kTile, (uint32_t)(i % kTile), um, m);
  double J[4][8];
  const double ccw[4] = {-1, 1, -1, 1};
  for (int c = 0; c < 4; ++c) {
    J[0][c] = m.thrust_max[c] * m.inv_mass;
    J[1][c] = m.inv_inertia[0] * (m.thrust_max[c] * m.prop_y[c]);
    J[2][c] = m.inv_inertia[1] * (m.thrust_max[c] * -m.prop_x[c]);
    J[3][c] = m.inv_inertia[2] * (m.torque_max[c] * ccw[c]);
    for (int r = 0; r < 4; ++r) J[r][4 + c] = (r == c) ? 1.0 : 0.0;
  }
  for (int col = 0; col < 4; ++col) {
    int piv = col;
    for (int r = col + 1; r < 4; ++r) if (fabs(J[r][col]) > fabs(J[piv][col])) piv = r;
    for (int c = 0; c < 8; ++c) { const double t = J[col][c]; J[col][c] = J[piv][c]; J[piv][c] = t; }
    const double inv = 1.0 / J[col][col];      // (a singular jacobian gives non-finite controls: the NaN guard of the step reports it)
    for (int c = 0; c < 8; ++c) J[col][c] *= inv;
    for (int r = 0; r < 4; ++r) if (r != col) {
      const double f = J[r][col];
      for (int c = 0; c < 8; ++c) J[r][c] -= f * J[col][c];
    }
  }
  double* out = const_cast<double*>(p.jinv) + i * 16;
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) out[4 * r + c] = J[r][4 + c];
}

// caller-chosen trees [count][40] for envs first .. first+count-1: QuadLink + update_model on the device (no sampling)
__global__ __launch_bounds__(kBlock) void derive_trees_kernel(DevPtrs p, StepCfg cfg, const double* __restrict__ trees, int by_density,
                                                               int64_t first, int64_t count) {
  const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= count) return;
  gaq::ParamTree t;
  for (int j = 0; j < gaq::TL_COUNT; ++j) t.v[j] = trees[k * gaq::TL_COUNT + j];
  gaq::DerivedModel dm;
  gaq::derive_tree(t, dm, by_density != 0);
  write_model_planes(p, cfg.dt, first + k, dm);
}

// On-box HBM calibration (gaq_hbm_copy_dev; bench.py's roofline.peak_measured, SURVEY 8d "also measure an on-box copy kernel"): the access
// shape of the step kernels' streaming traffic -- one 16-byte buffer load and one 16-byte buffer store per lane, every byte once
__global__ __launch_bounds__(kBlock) void hbm_copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (; i < n16; i += stride) dst[i] = src[i];
}

// graph-safe mode: the step index lives in device memory so that a captured graph draws fresh noise / reset keys on every replay (a
// host-side counter would be baked in).  Single-step launches advance it themselves (gaq_kernels.hpp: step_counter_checkin); the fused
// T-step rollout is followed by this one-thread launch (inc = T << ctr_shift, onto the first of the counter's words)
__global__ void bump_kernel(uint64_t* ctr, uint64_t inc) { *ctr += inc; }
// ... and before a launch that reads the first word alone (every step kernel without F_CTR) the check-ins of earlier F_CTR launches,
// spread over the counter's other words, are folded into it
__global__ void fold_counter_kernel(uint64_t* ctr) {
  uint64_t sum = 0;
  for (int k = 0; k < kCtrSlots; ++k) { sum += ctr[k * kCtrStride]; ctr[k * kCtrStride] = 0; }
  ctr[0] = sum;
}

// ---- reset / observe kernel (not on the per-step path: plain 8- and 4-byte tile accesses) -----------------
struct TileDirect {
  const DevPtrs& p; int64_t tile; uint32_t lane;
  __device__ __forceinline__ double ld64(const double* arr, int planes, int plane) const {
    auto r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(arr + tile * planes * kTile), 0, planes * kTile * 8, 0x00020000);
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, lane * 8u, plane * (kTile * 8), 0));
  }
  __device__ __forceinline__ void st64(double* arr, int planes, int plane, double v) const {
    auto r = __builtin_amdgcn_make_buffer_rsrc(arr + tile * planes * kTile, 0, planes * kTile * 8, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, lane * 8u, plane * (kTile * 8), 0);
  }
  __device__ __forceinline__ float ld32(const float* arr, int plane) const {
    auto r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(arr + tile * 4 * kTile), 0, kGrpBytes, 0x00020000);
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane * 4u, plane * (kTile * 4), 0));
  }
  __device__ __forceinline__ void st32(float* arr, int plane, float v) const {
    auto r = __builtin_amdgcn_make_buffer_rsrc(arr + tile * 4 * kTile, 0, kGrpBytes, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), r, lane * 4u, plane * (kTile * 4), 0);
  }
};

// `obs`: where the observation goes (or null).  Split-state layouts whose observation IS the heads: the same rows are the
// state (obs = the head rows).  `hi_out` != null (F_PACK handles): the heads go to the library's rows `hi_out` with plain
// stores and the observation is packed like in the plain layout.
__global__ __launch_bounds__(kBlock) void reset_kernel(DevPtrs p, StepCfg cfg, Model<double> um, const uint8_t* __restrict__ mask,
                                                        int do_reset, float* obs, int alias, uint64_t key_offset, float* hi_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63u;
  if (p.step_ctr) cfg.step_index = step_counter_peek(p, lane);
  cfg.step_index += key_offset;                                            // reset calls are keyed apart from steps
  const int64_t tile = (int64_t)blockIdx.x * (kBlock / kTile) + wave;
  if (tile >= p.ntiles) return;
  const int D = cfg.obs_dim;
  char* rows = smem + wave * (kTile * D * 4);
  const int64_t i = tile * kTile + lane;
  const TileDirect t{p, tile, lane};
  if (i < p.n) {
    EnvState<double> s;
#pragma unroll
    for (int j = 0; j < 3; ++j) { s.goal[j] = (double)t.ld32(p.goal, j); s.gyro_bias[j] = t.ld32(p.gyro, j); }
    if (alias) {   // value = observation word + residual (quad_core.hpp F_ALIAS); the goal is the default one
      double v[18];   // (alias == 2, fp32 mode: the observation word is the whole value)
#pragma unroll
      for (int k = 0; k < 18; ++k) v[k] = lo_decode(alias, p.obs_in, p.lo, i, k);
#pragma unroll
      for (int j = 0; j < 3; ++j) { s.pos[j] = v[j] + s.goal[j]; s.vel[j] = v[3 + j]; s.omega[j] = v[15 + j]; }
#pragma unroll
      for (int j = 0; j < 9; ++j) s.rot[j] = v[6 + j];
    } else {
#pragma unroll
      for (int j = 0; j < 3; ++j) { s.pos[j] = t.ld64(p.core, kCorePlanes, j); s.vel[j] = t.ld64(p.core, kCorePlanes, 3 + j);
                                    s.omega[j] = t.ld64(p.core, kCorePlanes, 15 + j); }
#pragma unroll
      for (int j = 0; j < 9; ++j) s.rot[j] = t.ld64(p.core, kCorePlanes, 6 + j);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { s.rot_damp[j] = t.ld64(p.lag, kLagPlanes, j); s.cmds_damp[j] = t.ld32(p.cmds, j);
                                  s.ou[j] = t.ld32(p.ou, j); s.act_prev[j] = t.ld32(p.actp, j); }
    const uint32_t cw = p.ctr[i];
    s.tick = cw & 0xFFFFu; s.svd_ctr = cw >> 16;
    float acc[3] = {0.0f, 0.0f, 9.81f};
    float hist[4] = {s.act_prev[0], s.act_prev[1], s.act_prev[2], s.act_prev[3]};
    if (do_reset && (mask == nullptr || mask[i])) {
      gaq::reset_env<double, gaq::F_GENERIC>(s, cfg, cfg.env_offset + (uint64_t)i, cfg.step_index);
#pragma unroll
      for (int j = 0; j < 3; ++j) t.st32(p.goal, j, (float)s.goal[j]);
      if (!alias) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { t.st64(p.core, kCorePlanes, j, s.pos[j]); t.st64(p.core, kCorePlanes, 3 + j, s.vel[j]);
                                      t.st64(p.core, kCorePlanes, 15 + j, s.omega[j]); }
#pragma unroll
        for (int j = 0; j < 9; ++j) t.st64(p.core, kCorePlanes, 6 + j, s.rot[j]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { t.st64(p.lag, kLagPlanes, j, s.rot_damp[j]); t.st32(p.cmds, j, s.cmds_damp[j]);
                                    t.st32(p.actp, j, s.act_prev[j]); }
      p.ctr[i] = (s.tick & 0xFFFFu) | (s.svd_ctr << 16);
      hist[0] = hist[1] = hist[2] = hist[3] = 0.0f;
    }
    float hi18[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) hi18[k] = 0.0f;
    if (alias) {   // extra mantissa bits of the (possibly new) state; its head goes out as the observation below
      double v[18];
#pragma unroll
      for (int j = 0; j < 3; ++j) { v[j] = s.pos[j] - s.goal[j]; v[3 + j] = s.vel[j]; v[15 + j] = s.omega[j]; }
#pragma unroll
      for (int j = 0; j < 9; ++j) v[6 + j] = s.rot[j];
#pragma unroll
      for (int k = 0; k < 18; ++k) {
        if (alias == 2) { hi18[k] = (float)v[k]; }
        else { lo_encode(alias, p.lo, i, k, v[k]); hi18[k] = split_hi(v[k]); }
      }
      if (hi_out) {
#pragma unroll
        for (int k = 0; k < 18; ++k) hi_out[i * 18 + k] = hi18[k];
      }
    }
    if (obs) {
      float* row = reinterpret_cast<float*>(rows) + lane * D;
      if (alias && !hi_out) {          // the observation words ARE the (truncated) state heads
#pragma unroll
        for (int k = 0; k < 18; ++k) row[k] = hi18[k];
      } else {
        const float* sz = p.sense_in;
        const int64_t n = p.n;
        double t2w = 0.0, t2t = 0.0;
        if (cfg.obs_flags & (gaq::OBS_APPEND_T2W | gaq::OBS_APPEND_T2T)) {   // as in env_step: sum(thrust_max) = g m t2w
          double th[4], tq0 = um.torque_max[0], im = um.inv_mass;
#pragma unroll
          for (int j = 0; j < 4; ++j) th[j] = um.thrust_max[j];
          if (p.par) {
#pragma unroll
            for (int j = 0; j < 4; ++j) th[j] = t.ld64(p.par, kPar, PP_THRUST_MAX + j);
            tq0 = t.ld64(p.par, kPar, PP_TORQUE_MAX); im = t.ld64(p.par, kPar, PP_INV_MASS);
          }
          t2w = (((th[0] + th[1]) + th[2]) + th[3]) * im / 9.81; t2t = tq0 / th[0];
        }
        gaq::pack_obs<double, gaq::F_GENERIC | gaq::F_DIAG>(s, cfg, acc, hist, [&](int k, float v, int) { row[k] = v; },
                                            cfg.env_offset + (uint64_t)i, cfg.step_index, 1, WaveSwarm{lane, cfg.swarm.agents},
                                            [&](int c, int slot, int j) { return sz ? sz[((int64_t)(c * 12 + slot) * 3 + j) * n + i] : 0.0f; }, t2w, t2t);
        // state_vector() advanced the bias random walk (sensor_noise.py:166): kept for the envs this call is about -- all of them for
        // gaq_observe and an unmasked reset, the masked ones for a masked reset.  The others get an observation too (the call returns
        // the whole batch), but as a peek: a masked reset leaves every bit of an unmasked env alone
        // (tests/test_gpu_api_matrix.py found the bias of the unmasked envs one add_noise call ahead of an undisturbed twin's).
        if (cfg.gyro_bias && (!do_reset || mask == nullptr || mask[i])) {
#pragma unroll
          for (int j = 0; j < 3; ++j) t.st32(p.gyro, j, s.gyro_bias[j]);
        }
      }
    }
  }
  if (obs) {
    wave_lds_fence();
    flush_obs(obs, p.n, D, tile, rows, lane);
  }
}

// ---- gaq_get_state, device half: tile-major arrays (or the split alias rows) -> GAQ_STATE_PLANES x N doubles ----
// plane-major so that the host needs ONE copy; not on the per-step path (plain 8-/4-byte accesses).
__global__ __launch_bounds__(kBlock) void export_kernel(DevPtrs p, int alias, double* __restrict__ out, float* __restrict__ aux_out = nullptr) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= p.n) return;
  if (aux_out) {   // (gaq_step without copies: the info dict's aux rows ride along)
    for (int k = 0; k < gaq::AUX_WORDS; ++k) aux_out[i * gaq::AUX_WORDS + k] = p.aux[i * gaq::AUX_WORDS + k];
  }
  const int64_t n = p.n;
  const int64_t tile = i / kTile;
  const int lane = (int)(i % kTile);
  auto grp = [&](const float* a, int plane) { return (double)a[tile * (4 * kTile) + plane * kTile + lane]; };
  for (int k = 0; k < 3; ++k) out[(34 + k) * n + i] = grp(p.goal, k);
  if (alias) {
    for (int k = 0; k < 18; ++k)
      out[(int64_t)k * n + i] = lo_decode(alias, p.obs_in, p.lo, i, k) + (k < 3 ? grp(p.goal, k) : 0.0);
  } else {
    for (int k = 0; k < kCorePlanes; ++k) out[(int64_t)k * n + i] = p.core[tile * (kCorePlanes * kTile) + k * kTile + lane];
  }
  for (int j = 0; j < 4; ++j) {
    out[(18 + j) * n + i] = p.lag[tile * (kLagPlanes * kTile) + j * kTile + lane];
    out[(22 + j) * n + i] = grp(p.cmds, j);
    out[(26 + j) * n + i] = grp(p.ou, j);
    out[(30 + j) * n + i] = grp(p.actp, j);
  }
  const uint32_t c = p.ctr[i];
  out[37 * n + i] = (double)(c & 0xFFFFu);
  out[38 * n + i] = (double)(c >> 16);
  for (int k = 0; k < 3; ++k) out[(39 + k) * n + i] = grp(p.gyro, k);
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
  uint32_t noted_step = 0xFFFFFFFFu, noted_roll = 0xFFFFFFFFu;   // last masks handed to launch_record()
  int lds_raised_for = -1; bool reset_lds_raised = false;   // hipFuncAttributeMaxDynamicSharedMemorySize already raised
  hipStream_t user_stream = nullptr;   // the stream of the most recent *_dev call (NULL = HIP's legacy default stream)
  bool user_stream_used = false;
  const float* sense_next = nullptr;   // gaq_set_sense_input_dev: draws of the next step / reset
  std::vector<double> host_par;   // [ntiles][kPar][64] staging for per-env params
  bool dev_params = false;        // the parameters are managed on the device (randomizer / gaq_set_param_trees): host_par is stale
  bool rz_on = false;             // gaq_set_randomizer installed
  bool cold_stale = false;        // an F_RZ launch has promoted envs with the hot planes only: the other planes of those envs are behind
                                  // their resample count (device word per env: pfull_of) until somebody writes them whole again
  int rz_since_refill = 0;        // step launches since the last refill pass of the staged parameter planes
  bool rz_refill_now = false;     // run the refill pass before the next step launch (ticks may have been set by the caller)
  Randomizer rz;
  std::vector<uint8_t> pflags;    // per env: 1 motor lag, 2 rotor drag, 4 not compact-constructible, 8 vel / omega damping
  int64_t cnt_lag = 0, cnt_drag = 0, cnt_noncompact = 0, cnt_damp = 0;   // envs with each flag set
  bool any_lag = false, any_drag = false;
  bool force_generic = false;
  bool ctr_spread = false;  // graph-safe mode: F_CTR launches have left check-ins in the counter's words beyond the first
  int num_cus = 256;      // compute units of the device (hipDeviceProp_t::multiProcessorCount): the small-batch size rule counts waves per SIMD
  int variant = 0;        // gaq::Feature mask of the step kernel in use
  int lds_per_wave = 0;   // bytes of LDS each wave of the step kernel uses
  bool needs_generic = false;
  bool fused_rollout = true;     // gaq_step_many_dev uses the fused T-step kernel when it can (GAQ_NO_FUSED=1 disables)
  bool alias = false;     // obs_state_alias in effect: state head lives in the observation tensor `last_obs`
  bool pack = false;      // split state (alias) whose observation is NOT the heads: packed explicitly (F_PACK); implies shadow
  bool shadow = false;    // obs_state_alias == 2: split state with LIBRARY-owned heads (own_obs); the caller's tensor gets a copy
  bool check_alias = false;       // GAQ_CHECK_ALIAS=1 (debug): checksum the aliased observation rows after every launch and
  uint64_t* alias_sum_dev = nullptr;   // verify them before the next one (the caller must not have modified them)
  uint64_t alias_sum = 0; bool alias_sum_valid = false;
  bool lomix = false;     // alias layout with the mixed residual rows (omega exact): per-env parameters or a model with motor lag
  bool fp32 = false;      // fp32_state in effect (implies alias): fp32 arithmetic, the observation rows are the whole state
  float* own_obs = nullptr;      // [n][18] library-owned observation buffer (host-pointer entry points, set_state)
  const float* last_obs = nullptr;  // where the previous step / reset wrote the observation
  uint64_t* step_ctr_mem = nullptr; // device word behind DevPtrs::step_ctr (allocated at create, used in graph-safe mode)
  // staging of the host-pointer entry points (gaq_step, gaq_get_state), allocated on first use and kept:
  // device [actions 16n | reward 4n | done n | pad | obs 4 D n] with a pinned host mirror; device [42][n] doubles
  char* stage_dev = nullptr; char* stage_pin = nullptr; size_t stage_bytes = 0;
  char* stage_map = nullptr;     // the device's address of stage_pin when the small-batch host path runs without copies (gaq_step)
  char* info_map = nullptr;      // ... and of info_pin
  // info-dict handles (aux_outputs) with a pinned mirror: gaq_step also brings the exported state planes and the aux rows home in
  // its one synchronisation, so that the gaq_get_state + gaq_get_aux that build the info dict (quadrotor.py:993-1028) cost no
  // further round trip.  Valid until the next launch / upload that changes the state.
  char* info_pin = nullptr; bool info_valid = false;
  size_t off_rew = 0, off_done = 0, off_obs = 0;
  double* export_dev = nullptr;
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
  m.jinv = nullptr;
}

// quadrotor_jacobian (quadrotor_control.py:192-203) and its inverse (:290-291), Gauss-Jordan with partial pivoting in fp64
bool inverse_jacobian(const gaq_model& g, double out[16]) {
  double J[4][8];
  const double ccw[4] = {-1, 1, -1, 1};
  for (int c = 0; c < 4; ++c) {
    J[0][c] = g.thrust_max[c] / g.mass;
    J[1][c] = (1.0 / g.inertia[0]) * (g.thrust_max[c] * g.prop_pos[3 * c + 1]);
    J[2][c] = (1.0 / g.inertia[1]) * (g.thrust_max[c] * -g.prop_pos[3 * c]);
    J[3][c] = (1.0 / g.inertia[2]) * (g.torque_max[c] * ccw[c]);
    for (int r = 0; r < 4; ++r) J[r][4 + c] = (r == c) ? 1.0 : 0.0;
  }
  for (int col = 0; col < 4; ++col) {
    int piv = col;
    for (int r = col + 1; r < 4; ++r) if (std::fabs(J[r][col]) > std::fabs(J[piv][col])) piv = r;
    if (std::fabs(J[piv][col]) < 1e-300) return false;
    for (int c = 0; c < 8; ++c) std::swap(J[col][c], J[piv][c]);
    const double inv = 1.0 / J[col][col];
    for (int c = 0; c < 8; ++c) J[col][c] *= inv;
    for (int r = 0; r < 4; ++r) if (r != col) {
      const double f = J[r][col];
      for (int c = 0; c < 8; ++c) J[r][c] -= f * J[col][c];
    }
  }
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) out[4 * r + c] = J[r][4 + c];
  return true;
}

int check_model(const gaq_model& g) {
  if (!(g.mass > 0) || !(g.inertia[0] > 0) || !(g.inertia[1] > 0) || !(g.inertia[2] > 0))
    return fail(GAQ_ERR_INVALID, "model: mass and inertia must be positive");
  if (g.damp_time_up < 0 || g.damp_time_down < 0) return fail(GAQ_ERR_INVALID, "model: negative motor time constant");
  return GAQ_OK;
}

// Does this model follow the reference's construction?  torque_max = t2t * thrust_max (quadrotor.py:176) for one t2t,
// and prop_pos.xy = (sx mx - comx, sy my - comy) with the sign pattern of inertia.py:238-240.  The hints are searched
// within an ulp of the obvious candidates and accepted only if they give back the model's numbers bit for bit.
bool find_construction(const Model<double>& m, double hint[5]) {
  auto same = [](double a, double b) { return std::memcmp(&a, &b, sizeof(double)) == 0 || (a == 0.0 && b == 0.0); };
  auto around = [](double v, double out[3]) { out[0] = v; out[1] = std::nextafter(v, -INFINITY); out[2] = std::nextafter(v, INFINITY); };
  bool ok = false;
  if (m.thrust_max[0] != 0.0) {
    double cand[3]; around(m.torque_max[0] / m.thrust_max[0], cand);
    for (double t : cand) {
      bool all = true;
      for (int j = 0; j < 4; ++j) all = all && same(t * m.thrust_max[j], m.torque_max[j]);
      if (all) { hint[0] = t; ok = true; break; }
    }
  }
  if (!ok) return false;
  const double sx[4] = {1.0, -1.0, -1.0, 1.0}, sy[4] = {-1.0, -1.0, 1.0, 1.0};
  auto axis = [&](const double p[4], const double sgn[4], double& mo, double& co) {
    // p[j] = sgn[j] * mo - co:  with a = value at sgn = +1, b = value at sgn = -1:  mo ~ (a - b) / 2, co ~ -(a + b) / 2
    double a = 0, b = 0;
    for (int j = 0; j < 4; ++j) (sgn[j] > 0 ? a : b) = p[j];
    double mc[3], cc[3]; around((a - b) * 0.5, mc); around(-(a + b) * 0.5, cc);
    for (double mm : mc) for (double c : cc) {
      bool all = true;
      for (int j = 0; j < 4; ++j) all = all && same(sgn[j] * mm - c, p[j]);
      if (all) { mo = mm; co = c; return true; }
    }
    return false;
  };
  return axis(m.prop_x, sx, hint[1], hint[3]) && axis(m.prop_y, sy, hint[2], hint[4]);
}

// ---- kernel selection: PURE host logic (no HIP call, no handle) shared by gaq_create, the parameter entry points and gaq_plan ----------
// so that the whole (configuration -> feature mask -> instantiation) map can be enumerated on a GPU-less host
// (tests/test_plan_cpu.py): a reachable mask without an instantiation is a test failure there, not a runtime GAQ_ERR_STATE.
struct Layout { bool alias, pack, shadow, fp32; };       // how the 18 integrator words are stored (gaq_config.obs_state_alias)
struct Selection { uint32_t variant; bool generic; int lds_per_wave; };

bool step_instantiated(uint32_t f) {
  switch (f) {
#define GAQ_X(FEAT) case (FEAT):
    GAQ_STEP_ALL(GAQ_X)
#undef GAQ_X
      return true;
    default: return false;
  }
}
bool roll_instantiated(uint32_t f) {
  switch (f) {
#define GAQ_X(FEAT) case (FEAT):
    GAQ_ROLL_ALL(GAQ_X)
#undef GAQ_X
      return true;
    default: return false;
  }
}
// Which instantiations this PROCESS has launched (gaq_launched_variants): the test suite's kernel coverage report is built from it
// (tools/kernel_coverage.py).  A handle remembers the last mask it recorded, so the steady state costs one compare per launch.
struct LaunchRecord {
  std::mutex mu;
  std::set<uint32_t> seen[2];      // 0: step_kernel<F>, 1: rollout_kernel<F>
  void note(int kind, uint32_t f) { std::lock_guard<std::mutex> g(mu); seen[kind].insert(f); }
};
LaunchRecord& launch_record() { static LaunchRecord r; return r; }

const void* step_kernel_ptr(uint32_t f) {
  switch (f) {
#define GAQ_X(FEAT) case (FEAT): return (const void*)&step_kernel<(FEAT)>;
    GAQ_STEP_ALL(GAQ_X)
#undef GAQ_X
    default: return nullptr;
  }
}
// the instantiation gaq_step_many_dev's fused path launches for a step variant (0xFFFFFFFF: no fused rollout for this variant)
uint32_t rollout_variant_of(uint32_t variant, const Layout& L, bool generic) {
  const uint32_t base = variant & ~(gaq::F_PREDRAW | gaq::F_NT | gaq::F_ROWS | gaq::F_CTR);
  if (!L.alias || L.pack || generic || (variant & gaq::F_MELL)) return 0xFFFFFFFFu;
  if ((base >= 16u && base <= 23u) || (variant >= 48u && variant <= 55u)) return base;
  return 0xFFFFFFFFu;
}

// Twins of the alias kernels (quad_core.hpp F_ROWS / F_CTR): what a step launch runs when the packed rows are registered / in graph-safe
// mode.  Rows win when both are on (the counter is then advanced by bump_kernel as for every kernel without a F_CTR twin).
uint32_t rows_twin_of(uint32_t variant) { return step_instantiated(variant | gaq::F_ROWS) ? (variant | gaq::F_ROWS) : 0xFFFFFFFFu; }
uint32_t ctr_twin_of(uint32_t variant) { return step_instantiated(variant | gaq::F_CTR) ? (variant | gaq::F_CTR) : 0xFFFFFFFFu; }

int observation_dim(const gaq_config* cfg) {
  int D = (cfg->obs_flags & GAQ_OBS_QUAT) ? 13 : 18;
  if (cfg->obs_flags & GAQ_OBS_APPEND_T2W) D += 1;
  if (cfg->obs_flags & GAQ_OBS_APPEND_T2T) D += 1;
  if (cfg->obs_flags & GAQ_OBS_APPEND_H) D += 1;
  if (cfg->obs_flags & GAQ_OBS_APPEND_ACC) D += 3;
  if (cfg->obs_flags & GAQ_OBS_APPEND_ACT) D += 4;
  if (cfg->swarm.agents > 1) D += 6 * (cfg->swarm.agents - 1);
  return D;
}

// gaq_config -> what is fixed for the life of the handle, validated.  Everything that depends on the PARAMETERS (motor lag, rotor
// drag, compact / zero-damping planes) is filled in by refresh_feature_flags / flags_from_counts.
int fill_step_cfg(const gaq_config* cfg, StepCfg& sc, int& obs_dim) {
  if (cfg->struct_size != sizeof(gaq_config) || cfg->abi_version != GAQ_ABI_VERSION)
    return fail(GAQ_ERR_INVALID, "gaq_config size/version mismatch (header vs library)");
  if (cfg->num_envs <= 0) return fail(GAQ_ERR_INVALID, "num_envs must be positive");
  if (cfg->num_envs > (int64_t)1 << 27) return fail(GAQ_ERR_INVALID, "num_envs above 2^27 per handle is not supported");
  if (!(cfg->sim_freq > 0) || cfg->sim_steps <= 0) return fail(GAQ_ERR_INVALID, "sim_freq and sim_steps must be positive");
  if (cfg->sim_steps > 64)   // the OU noise streams of the sub-steps are ids 0 .. sim_steps-1; 64+ belong to resets and sensors
    return fail(GAQ_ERR_INVALID, "sim_steps above 64 is not supported (random-stream ids of the sub-steps)");
  if (cfg->ep_len < 0 || cfg->ep_len >= 0xFFFF) return fail(GAQ_ERR_INVALID, "ep_len must be in [0, 65534]");
  if (cfg->control < 0 || cfg->control > 2) return fail(GAQ_ERR_INVALID, "unknown control mode");
  if (cfg->noise < 0 || cfg->noise > 2) return fail(GAQ_ERR_INVALID, "unknown noise mode");
  if (cfg->reward_mode < 0 || cfg->reward_mode > 1) return fail(GAQ_ERR_INVALID, "unknown reward mode");
  if (cfg->obs_flags & ~127) return fail(GAQ_ERR_INVALID, "unknown obs flags");
  if ((cfg->obs_flags & GAQ_OBS_QUAT) && cfg->swarm.agents > 1) return fail(GAQ_ERR_INVALID, "the quaternion observation is not available for swarms");
  if (cfg->swarm.agents > 1) {
    const int a = cfg->swarm.agents;
    // the observation rows of a wave's 64 agents (18 + 6 (agents - 1) words each) are staged in LDS, four waves per workgroup:
    // 16 agents need 110 KB of the CU's 160 KB, 32 would need 209 KB
    if (a > 16 || (a & (a - 1)) != 0) return fail(GAQ_ERR_INVALID, "swarm.agents must be a power of two <= 16");
    if (cfg->num_envs % a != 0 || cfg->env_id_offset % a != 0)
      return fail(GAQ_ERR_INVALID, "num_envs and env_id_offset must be multiples of swarm.agents (whole worlds per handle)");
    if (!(cfg->swarm.prox_dist > 0.0f) || !(cfg->swarm.collision_dist >= 0.0f) || !(cfg->swarm.goal_radius >= 0.0f))
      return fail(GAQ_ERR_INVALID, "swarm distances must be positive");
  }
  const double dt = 1.0 / cfg->sim_freq;
  const int period = svd_period_of(dt);
  if (period >= 0xFFFF) return fail(GAQ_ERR_INVALID, "sim_freq too high for the 16-bit SVD counter");
  if (cfg->sim_freq < 50.0) return fail(GAQ_ERR_INVALID, "sim_freq below 50 Hz is outside the rotation series' range");
  const int D = observation_dim(cfg);
  obs_dim = D;
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
  static_assert(sizeof(gaq::SwarmCfg) == sizeof(gaq_swarm), "swarm layout");
  std::memcpy(&sc.swarm, &cfg->swarm, sizeof(sc.swarm));
  if (sc.swarm.agents <= 1) std::memset(&sc.swarm, 0, sizeof(sc.swarm));
  static_assert(sizeof(gaq::SenseNoise) == sizeof(gaq_sense_noise), "sensor noise layout");
  static_assert(gaq::AUX_WORDS == GAQ_AUX_WORDS, "aux row layout");
  std::memcpy(&sc.sense, &cfg->sense, sizeof(sc.sense));
  if (cfg->sense.enabled && cfg->sense.gyro_norm_std != 0.0f) {
    // add_noise_to_omega (sensor_noise.py:160-168) with dt = env.dt = 1/sim_freq (quadrotor.py:790)
    const double tau = cfg->sense.gyro_bias_correlation_time;
    if (!(tau > 0.0)) return fail(GAQ_ERR_INVALID, "gyro_bias_correlation_time must be positive");
    const double sg = (double)cfg->sense.gyro_noise_density / std::sqrt(dt);
    const double sb = std::sqrt(-(sg * sg) * (tau / 2) * (std::exp(-2 * dt / tau) - 1));
    const double pi = std::exp(-dt / tau);
    sc.gyro_bias = 1;
    sc.gyro_pi = (float)pi; sc.gyro_sigma = (float)sb;
    sc.gyro_pi_step = (float)(pi * pi * pi); sc.gyro_sigma_step = (float)(sb * std::sqrt(1.0 + pi * pi + pi * pi * pi * pi));
  }
  sc.need_act_prev = ((cfg->obs_flags & GAQ_OBS_APPEND_ACT) || cfg->rew.action_change != 0.0f) ? 1 : 0;
  sc.resample_goal = cfg->resample_goal ? 1 : 0;
  sc.excite = cfg->excite ? 1 : 0;
  sc.aux = cfg->aux_outputs ? 1 : 0;
  sc.action_f32 = cfg->action_f32 ? 1 : 0;
  sc.sense_input = (cfg->sense_input && (cfg->sense.enabled || (cfg->obs_flags & (GAQ_OBS_APPEND_T2W | GAQ_OBS_APPEND_T2T)))) ? 1 : 0;
  sc.t2w_std = (float)cfg->t2w_std; sc.t2w_min = 1.5f; sc.t2w_max = 10.0f;       // quadrotor.py:706-712
  sc.t2t_std = (float)cfg->t2t_std; sc.t2t_min = 0.005f; sc.t2t_max = 1.0f;
  sc.per_env_goal = (sc.resample_goal || sc.excite || sc.swarm.agents > 1) ? 1 : 0;
  sc.auto_reset = cfg->auto_reset ? 1 : 0;
  sc.init_random_state = cfg->init_random_state ? 1 : 0;
  sc.use_acos = (cfg->rew.rot != 0.0f || cfg->rew.attitude != 0.0f) ? 1 : 0;
  sc.seed = cfg->seed; sc.step_index = 0; sc.env_offset = (uint64_t)cfg->env_id_offset;
  return GAQ_OK;
}

// aux row / quaternion / t2w / t2t observation on the split state (quad_core.hpp F_AUXP)?  Only what those kernels hold: RawControl (uniform
// or per-env models) or Mellinger on a uniform model, fp64 arithmetic, a split layout asked for, no swarm -- and a reason to be there at all
int env_override(const char* name);
// ... and the per-env planes that are state beside the 18 values: goals (resample_goal, excite; quad_core.hpp F_ENVX) and, for a uniform
// model, the gyro bias of SensorNoise's random walk (F_BIAS)
bool envx_wanted(const gaq_config& c, const StepCfg& sc) {
  const bool bias_walk = sc.sense.enabled && sc.gyro_bias;      // (the bias walk: uniform RawControl models only; per-env batches and Mellinger
  return (sc.resample_goal || sc.excite || bias_walk) &&        //  keep the generic kernel for it)
         !((c.per_env_params || c.control == GAQ_CTRL_MELLINGER) && bias_walk);
}
bool auxp_capable(const gaq_config& c, const StepCfg& sc) {
  const bool obs_diag = (c.obs_flags & (GAQ_OBS_QUAT | GAQ_OBS_APPEND_T2W | GAQ_OBS_APPEND_T2T)) != 0;
  return (sc.aux || obs_diag || envx_wanted(c, sc)) && !(c.control == GAQ_CTRL_MELLINGER && c.per_env_params) && c.obs_state_alias != 0 && !c.fp32_state &&
         sc.swarm.agents <= 1 && env_override("GAQ_NO_AUXP") != 1;
}

// Does this configuration need the generic instantiation (which honours every runtime flag and keeps fp64 state planes)?
// `heavy`: one of the register-hungry rarities is on (the full tier); `diag`: the diagnostics tier on top of it.
void generic_tiers(const gaq_config& c, const StepCfg& sc, bool force_generic, bool& generic, bool& heavy, bool& diag) {
  const bool obs_diag = (c.obs_flags & (GAQ_OBS_QUAT | GAQ_OBS_APPEND_T2W | GAQ_OBS_APPEND_T2T)) != 0;
  const bool bias_walk = sc.sense.enabled && sc.gyro_bias;
  // Mellinger runs in the specialised kernels (F_MELL), uniform or per-env models (one inverse jacobian per env, read where it is used): the
  // 18-word observation in any layout, the packed observations (body frame, appended height / accelerometer / action, sensor noise) on the
  // split state; the quaternion / t2w / t2t variants, fp32 state and fp64 planes with a packed observation keep the generic kernel
  const bool mell_packable = (c.obs_flags & ~(GAQ_OBS_BODY_FRAME | GAQ_OBS_APPEND_H | GAQ_OBS_APPEND_ACC | GAQ_OBS_APPEND_ACT)) == 0 &&
                             c.obs_state_alias != 0;      // (the packed observations exist on the split state: F_PACK)
  const bool mell_heads = c.obs_flags == 0 && !c.sense.enabled && !sc.need_act_prev;
  const bool mell_generic = c.control == GAQ_CTRL_MELLINGER && (c.fp32_state || !(mell_heads || mell_packable || auxp_capable(c, sc)));
  // the swarm layer runs on the split state (F_SWARM) for a uniform model under RawControl with one of the packable observations, when
  // a split layout was asked for (obs_state_alias != 0: the class default); anything else about it keeps the light generic kernel
  const bool swarm_generic = sc.swarm.agents > 1 &&
                             (c.per_env_params || c.control == GAQ_CTRL_MELLINGER || c.obs_state_alias == 0 || c.fp32_state || c.sense.enabled ||
                              (c.obs_flags & ~(GAQ_OBS_BODY_FRAME | GAQ_OBS_APPEND_H | GAQ_OBS_APPEND_ACC | GAQ_OBS_APPEND_ACT)) != 0);
  // the info dict's aux row and the quaternion / t2w / t2t observations ride on the SPLIT state (F_AUXP) for a uniform RawControl model
  // when a split layout was asked for (the class default); per-env models, Mellinger, swarms, fp32 state and fp64 planes keep the generic tiers
  const bool auxp = auxp_capable(c, sc);
  const bool envx = auxp && envx_wanted(c, sc);   // per-env goals / the gyro-bias walk ride there too (F_ENVX, F_BIAS)
  generic = force_generic || sc.drag || mell_generic || c.noise == GAQ_NOISE_INPUT || ((sc.resample_goal || sc.excite) && !envx) || (sc.aux && !auxp) ||
            sc.sense_input || (obs_diag && !auxp) || (bias_walk && !envx) || swarm_generic;
  // the lighter generic instantiation: everything generic except the register-hungry rarities
  heavy = force_generic || sc.drag || c.control == GAQ_CTRL_MELLINGER || c.noise == GAQ_NOISE_INPUT || sc.sense_input ||
          bias_walk || ((sc.aux || obs_diag) && c.per_env_params);
  // the diagnostics tier of the full generic kernel (aux outputs, injected sensor draws, quaternion / t2w / t2t observations); the aux row
  // and those observation variants ALONE on a uniform model (info=True, obs_repr="xyz_vxyz_quat_omega" ... on a RawControl batch) ride
  // on the light kernel: F_LITE | F_DIAG (per-env models: the light kernel's 247 VGPRs leave no room for them)
  diag = sc.aux || obs_diag || (heavy && sc.sense_input);
}

// split state: when the observation is exactly the 18 heads (world frame, no noise, nothing appended) they can be one and the same
// rows; otherwise (body frame, appended height / accelerometer / action, sensor noise) the state is still stored split,
// library-owned, and the observation is packed beside it (F_PACK) -- unless the generic kernel is needed: then fp64 planes
Layout decide_layout(const gaq_config& c, const StepCfg& sc, int D, bool generic) {
  Layout L;
  const bool auxp = auxp_capable(c, sc);      // (the aux row is packed beside the observation: never the heads-are-the-observation kernels)
  const bool heads_are_obs = D == 18 && !c.sense.enabled && c.obs_flags == 0 && !sc.need_act_prev && !(auxp && (sc.aux || envx_wanted(c, sc)));
  const bool packable = !c.fp32_state &&
                        (c.obs_flags & ~(GAQ_OBS_BODY_FRAME | GAQ_OBS_APPEND_H | GAQ_OBS_APPEND_ACC | GAQ_OBS_APPEND_ACT |
                                         (auxp ? (GAQ_OBS_QUAT | GAQ_OBS_APPEND_T2W | GAQ_OBS_APPEND_T2T) : 0))) == 0;
  L.alias = (c.obs_state_alias != 0 || c.fp32_state != 0) && (heads_are_obs || packable) && !generic;
  L.pack = L.alias && !heads_are_obs;
  L.fp32 = c.fp32_state != 0;
  L.shadow = L.alias && (c.obs_state_alias == 2 || L.pack) && !c.fp32_state;
  return L;
}

// `num_cus`: compute units of the device (hipDeviceProp_t::multiProcessorCount; 256 on a whole MI355X, fewer on a partitioned one).
// `predraw_env` / `nt_env`: the GAQ_PREDRAW / GAQ_NT measurement overrides (-1: the size rule decides).
Selection select_kernel(const gaq_config& c, const StepCfg& sc, const Layout& L, int obs_dim, bool force_generic, int rz_every,
                        int num_cus, int predraw_env, int nt_env) {
  Selection out;
  bool generic, heavy, diag;
  generic_tiers(c, sc, force_generic, generic, heavy, diag);
  // kernel variant: the specialised instantiations cover RawControl, the 18-word observation, the default
  // reward terms and the yaw-only reset; anything else runs the generic instantiation.
  uint32_t f = c.per_env_params ? gaq::F_PER_ENV : 0u;
  if (generic) {
    f |= gaq::F_GENERIC;
    if (!heavy) f |= gaq::F_LITE;
    if (diag) f |= gaq::F_DIAG;          // (with F_LITE: the aux row and nothing else of that tier)
  } else {
    if (sc.motor_lag) f |= gaq::F_LAG;
    if (c.noise == GAQ_NOISE_PHILOX) f |= gaq::F_NOISE;
    if (c.control == GAQ_CTRL_MELLINGER) f |= gaq::F_MELL;
    if (sc.swarm.agents > 1) f |= gaq::F_SWARM;
  }
  if (L.alias && !generic) f |= gaq::F_ALIAS;
  if (L.pack && L.alias && !generic) f |= gaq::F_PACK;
  if (L.pack && L.alias && !generic && auxp_capable(c, sc)) f |= gaq::F_AUXP;
  if ((f & gaq::F_AUXP) && envx_wanted(c, sc)) f |= gaq::F_ENVX | ((sc.sense.enabled && sc.gyro_bias) ? gaq::F_BIAS : 0u);
  if (L.fp32 && L.alias && !generic) f |= gaq::F_FP32;
  // per-episode re-randomisation on the device: the instantiation that promotes finished envs to their staged planes (one per
  // feature set, no batch-size-specific variants: big and small handles -- shards -- run the very same code)
  if (c.per_env_params && rz_every > 0) f |= gaq::F_RZ;
  // small batches: noise drawn under the load latency / non-temporal streaming, by waves per SIMD (4 SIMDs per CU)
  if (f == 20u || f == 22u || f == 23u) {
    const int64_t tiles = (c.num_envs + kTile - 1) / kTile;
    const int64_t simds = (int64_t)(num_cus > 0 ? num_cus : 256) * 4;
    // defaults by batch size, from the 2 x 2 measurement profiles/r02_v4_small_batch_policy_2x2.txt:
    // non-temporal streaming up to two waves per SIMD (-3 % at 65 536 envs, -9 ... -15 % at 131 072 on 256 CUs; +6 % at 2^20);
    // noise drawn under the load latency from two waves per SIMD up (-1.4 ... -4 %; at ONE wave per SIMD it costs 6-7 %)
    bool nt = tiles <= 2 * simds, predraw = tiles > simds;
    if (predraw_env >= 0) predraw = predraw_env != 0;
    if (nt_env >= 0) nt = nt_env != 0;
    if (predraw && sc.sim_steps <= 2) f |= gaq::F_PREDRAW;
    if (nt) f |= gaq::F_NT;
  }
  out.variant = f;
  out.generic = generic;
  const int obs_rows = kTile * obs_dim * 4;
  int lpw;
  if (generic) {
    const int img = tile_image<gaq::F_GENERIC>(sc).total;
    lpw = img > obs_rows ? img : obs_rows;                     // obs rows reuse the image buffer
    if (f & gaq::F_DIAG) {                                     // ... with the info dict's aux rows behind them (gaq_kernels.hpp kAuxRowsInLds)
      const int both = ((obs_rows + 15) & ~15) + kTile * gaq::AUX_WORDS * 4;
      lpw = lpw > both ? lpw : both;
    }
  } else {
    int img = (L.fp32 ? kRowsLds : L.alias ? kRowsLds + kLoRowsLds : kCoreBytes) +
              (sc.motor_lag ? kLagBytes + kGrpBytes : 0) +
              (c.noise == GAQ_NOISE_PHILOX ? kGrpBytes : 0) +
              ((sc.need_act_prev && (!L.alias || L.pack)) ? kGrpBytes : 0) +   // previous-action plane (not when the heads are the obs)
              (sc.swarm.agents > 1 ? kGrpBytes : 0) +                         // formation-goal plane (F_SWARM)
              ((f & gaq::F_ENVX) && sc.per_env_goal ? kGrpBytes : 0) +         // goal plane / gyro-bias plane (F_ENVX)
              ((f & gaq::F_BIAS) && sc.gyro_bias ? kGrpBytes : 0);
    lpw = img > obs_rows ? img : obs_rows;                     // obs rows reuse the image buffer
    if (f & gaq::F_AUXP) {                                     // ... and the info dict's aux rows sit behind them (gaq_kernels.hpp kAuxRowsInLds)
      const int both = ((obs_rows + 15) & ~15) + kTile * gaq::AUX_WORDS * 4;
      lpw = lpw > both ? lpw : both;
    }
    // (the F_ROWS twins stage their 20-word packed rows in the same buffer: 5120 B, below the alias image's 8192+)
  }
  out.lds_per_wave = (lpw + 15) & ~15;
  return out;
}

// Graph-safe step counter of a handle of `ntiles` tiles: a self-counting step launch has `waves` waves (whole workgroups: the waves past
// the last tile check in too), every one adds 1 except the launch's first, which adds inc0 = 2^shift - (waves - 1): 2^shift per launch,
// and fewer than 2^shift have landed whenever a wave of the launch reads (its own is missing): gaq_kernels.hpp step_counter_checkin.
// Pure host arithmetic (gaq_plan reports it; tests/test_plan_cpu.py models the check-ins against it for every launch size).
void counter_plan(int64_t ntiles, uint64_t& waves, uint32_t& shift, uint32_t& inc0) {
  const int wpb = kBlock / kTile;
  waves = (uint64_t)((ntiles + wpb - 1) / wpb) * wpb;
  shift = 0;
  while (((uint64_t)1 << shift) < waves) ++shift;
  inc0 = (uint32_t)(((uint64_t)1 << shift) - (waves - 1));
}

int env_override(const char* name) { const char* v = getenv(name); return v ? (v[0] == '1' ? 1 : 0) : -1; }

void refresh_feature_flags(gaq_env* e) {
  StepCfg& sc = e->sc;
  sc.motor_lag = e->any_lag ? 1 : 0;
  sc.drag = e->any_drag ? 1 : 0;
  const Layout L{e->alias, e->pack, e->shadow, e->fp32};
  const Selection sel = select_kernel(e->cfg, sc, L, e->obs_dim, e->force_generic, e->d.rz_every, e->num_cus,
                                      env_override("GAQ_PREDRAW"), env_override("GAQ_NT"));
  e->variant = (int)sel.variant;
  e->needs_generic = sel.generic;
  e->lds_per_wave = sel.lds_per_wave;
}

// state-encoding mode of the reset / export kernels: 0 fp64 planes, 1 split with 16-bit residuals, 2 fp32 rows, 3 split
// with the mixed residual rows (pos / vel / R 16 bits, omega exact)
int alias_mode(const gaq_env* e) { return e->fp32 ? 2 : !e->alias ? 0 : e->lomix ? 3 : 1; }

// GAQ_CHECK_ALIAS=1 (debug): in alias mode 1 the observation tensor returned by step k is step k+1's input.  A caller that
// edits it in place (running-mean normalisation, clamp_, buffer reuse) corrupts the physics silently -- unless the rows are
// checksummed after every launch and verified before the next one.  Costs a reduction kernel and a stream sync per step.
int record_alias_rows(gaq_env* e, hipStream_t st) {
  if (!e->check_alias || e->shadow) return GAQ_OK;
  HIP_TRY(hipMemsetAsync(e->alias_sum_dev, 0, sizeof(uint64_t), st));
  const int64_t words = e->d.n * 18;
  hipLaunchKernelGGL(checksum_kernel, dim3(1024), dim3(kBlock), 0, st, reinterpret_cast<const uint32_t*>(e->last_obs), words, e->alias_sum_dev);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(&e->alias_sum, e->alias_sum_dev, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  e->alias_sum_valid = true;
  return GAQ_OK;
}
int verify_alias_rows(gaq_env* e, hipStream_t st) {
  if (!e->check_alias || e->shadow || !e->alias_sum_valid) return GAQ_OK;
  uint64_t now = 0;
  HIP_TRY(hipMemsetAsync(e->alias_sum_dev, 0, sizeof(uint64_t), st));
  const int64_t words = e->d.n * 18;
  hipLaunchKernelGGL(checksum_kernel, dim3(1024), dim3(kBlock), 0, st, reinterpret_cast<const uint32_t*>(e->last_obs), words, e->alias_sum_dev);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(&now, e->alias_sum_dev, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (now != e->alias_sum)
    return fail(GAQ_ERR_STATE, "obs_state_alias: the observation tensor returned by the previous step / reset was modified before "
                               "this call -- it is the integrator state's fp32 head (include/gaq.h gaq_config.obs_state_alias); "
                               "copy it before editing, or create the handle with obs_state_alias = 2 (library-owned heads)");
  return GAQ_OK;
}

// the refill pass of the staged parameter planes (rerandomize_kernel mode 0), on the stream of the step launches
int launch_refill(gaq_env* e, hipStream_t st) {
  const dim3 grid((unsigned)((e->d.n + kBlock - 1) / kBlock)), block(kBlock);
  hipLaunchKernelGGL(rerandomize_kernel, grid, block, 0, st, e->d, e->sc, e->rz, (const uint8_t*)nullptr, 0, (double*)nullptr,
                     (int64_t)0, (int64_t)0);
  HIP_TRY(hipGetLastError());
  e->rz_since_refill = 0; e->rz_refill_now = false;
  return GAQ_OK;
}

// Mellinger with device-sampled per-env models: bring the inverse jacobians up to the parameter planes (jinv_kernel), on the stream that
// changed them
int launch_jinv(gaq_env* e, hipStream_t st, const uint8_t* done) {
  if (!e->d.jinv || !e->dev_params) return GAQ_OK;
  const dim3 grid((unsigned)((e->d.n + kBlock - 1) / kBlock)), block(kBlock);
  hipLaunchKernelGGL(jinv_kernel, grid, block, 0, st, e->d, e->sc, e->um, done);
  HIP_TRY(hipGetLastError());
  return GAQ_OK;
}

// the instantiation the next step launch runs: the handle's kernel, or its F_ROWS / F_CTR twin (packed rows registered / graph-safe mode)
uint32_t launch_variant_of(const gaq_env* e) {
  const uint32_t v = (uint32_t)e->variant;
  if (e->d.rows_out && rows_twin_of(v) != 0xFFFFFFFFu) return rows_twin_of(v);
  if (e->d.step_ctr && ctr_twin_of(v) != 0xFFFFFFFFu) return ctr_twin_of(v);
  return v;
}

int launch_step(gaq_env* e, const float* actions, float* obs, float* reward, uint8_t* done, hipStream_t st) {
  e->info_valid = false;
  if ((reinterpret_cast<uintptr_t>(actions) & 15) != 0) return fail(GAQ_ERR_INVALID, "actions must be 16-byte aligned");
  if ((reinterpret_cast<uintptr_t>(obs) & 15) != 0) return fail(GAQ_ERR_INVALID, "obs must be 16-byte aligned");
  if (e->sc.noise == gaq::NOISE_INPUT) {
    if (!e->noise_next) return fail(GAQ_ERR_STATE, "GAQ_NOISE_INPUT: call gaq_set_noise_input_dev before each step");
    e->d.noise_in = e->noise_next;
    e->noise_next = nullptr;
  }
  if (e->sc.sense_input) {
    if (!e->sense_next) return fail(GAQ_ERR_STATE, "sense_input: call gaq_set_sense_input_dev before each step");
    e->d.sense_in = e->sense_next;
    e->sense_next = nullptr;
  }
  const int tiles_per_block = kBlock / kTile;
  const dim3 grid((unsigned)((e->d.ntiles + tiles_per_block - 1) / tiles_per_block)), block(kBlock);
  const size_t lds = (size_t)e->lds_per_wave * tiles_per_block;
  const int lpw = e->lds_per_wave;
  const float* caller_obs = obs;              // (the shadow layouts step on the library's own rows below)
  if (e->alias) {
    if (e->needs_generic) return fail(GAQ_ERR_STATE, "obs_state_alias: parameters now need the generic kernel (rotor drag); "
                                                     "create the handle without obs_state_alias");
    e->d.obs_in = e->last_obs;
    if (int rc = verify_alias_rows(e, st)) return rc;
    e->d.obs_copy = nullptr;
    if (e->shadow) {              // heads stay in the library's own rows (updated in place); the caller's tensor gets a copy
      if (obs != e->own_obs) e->d.obs_copy = obs;
      obs = e->own_obs;
    }
  }
  if (lds > 65536 && e->lds_raised_for != e->variant) {
    // large swarm observation rows: more dynamic LDS than the 64 KB a launch may use by default (the CU has 160 KB)
    const void* fn = step_kernel_ptr((uint32_t)e->variant);
    if (!fn || lds > 160 * 1024) return fail(GAQ_ERR_INVALID, "observation rows too large for the CU's LDS");
    HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    e->lds_raised_for = e->variant;
  }
  const uint32_t launch_variant = launch_variant_of(e);
  if (e->d.rz_every > 0 && (launch_variant & gaq::F_RZ) != 0) {
    // (the condition of the step kernel's epilogue: gaq_kernels.hpp `hot_only`)
    const bool hot_only = e->sc.compact_params != 0 && e->sc.zero_damp != 0 && (launch_variant & gaq::F_FP32) == 0;
    if (hot_only) e->cold_stale = true;
    else if (e->cold_stale) {
      // this launch's promotions move all 45 planes and say nothing about the env's earlier ones: bring the envs that hot-only promotions
      // left behind up to date first (rare: the parameter flags changed under a live randomizer)
      const dim3 g1((unsigned)((e->d.n + kBlock - 1) / kBlock));
      hipLaunchKernelGGL(rerandomize_kernel, g1, block, 0, st, e->d, e->sc, e->rz, (const uint8_t*)nullptr, 4, (double*)nullptr, (int64_t)0, (int64_t)0);
      HIP_TRY(hipGetLastError());
      e->rz_refill_now = true;
      e->cold_stale = false;
    }
  }
  if (e->d.rz_every > 0 && e->rz_refill_now) { if (int rc = launch_refill(e, st)) return rc; }
  const bool self_counting = (launch_variant & gaq::F_CTR) != 0;
  if (e->d.step_ctr && !self_counting) {   // this kernel reads the counter's first word alone: fold the others into it
    // Eagerly the host knows whether F_CTR launches have left check-ins in the other words (ctr_spread).  A launch that is being CAPTURED
    // runs later, any number of times, after who knows which other launches (an F_CTR graph, eager small-batch steps): its graph always
    // carries the fold node, so that a replay never reads a stale first word.
    bool capturing = false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess) capturing = cs == hipStreamCaptureStatusActive; else (void)hipGetLastError();
    if (e->ctr_spread || capturing) {
      hipLaunchKernelGGL(fold_counter_kernel, dim3(1), dim3(1), 0, st, e->d.step_ctr);
      HIP_TRY(hipGetLastError());
      if (!capturing) e->ctr_spread = false;
    }
  }
  if (launch_variant != e->noted_step) { launch_record().note(0, launch_variant); e->noted_step = launch_variant; }
#define GAQ_LAUNCH(FEAT) \
  hipLaunchKernelGGL(step_kernel<(FEAT)>, grid, block, lds, st, e->d, e->sc, e->um, actions, obs, reward, done, lpw)
  switch (launch_variant) {
#define GAQ_X(FEAT) case (int)(FEAT): GAQ_LAUNCH(FEAT); break;
    GAQ_STEP_ALL(GAQ_X)
#undef GAQ_X
    default: return fail(GAQ_ERR_STATE, "internal: no kernel instantiation for this feature mask");
  }
#undef GAQ_LAUNCH
  HIP_TRY(hipGetLastError());
  if (e->d.ep_ret) {
    const dim3 g2((unsigned)((e->d.n + kBlock - 1) / kBlock));
    hipLaunchKernelGGL(episode_kernel, g2, block, 0, st, e->d.n, reward, done, e->d.ep_ret, e->d.ep_len, e->d.ep_acc);
    HIP_TRY(hipGetLastError());
  }
  if (e->d.rz_every > 0 && e->d.jinv) { if (int rc = launch_jinv(e, st, done)) return rc; }     // (the promoted envs' inverse jacobians)
  if (e->d.rz_every > 0) {
    // dynamics_randomize_every on the device: the step kernel promoted the finished, due envs to their staged planes; the
    // refill pass (the next draw of every promoted env -> par_next) is due before any of them can finish again, i.e. within
    // ep_len + 1 steps.  Under graph capture (device-resident step index) the host cannot count replays: every step.
    e->rz_since_refill += 1;
    const int period = e->sc.ep_len + 1 < 64 ? e->sc.ep_len + 1 : 64;
    if (e->d.step_ctr || e->rz_since_refill >= period) {
      if (int rc = launch_refill(e, st)) return rc;
    }
  }
  if (e->d.rows_out && (launch_variant & gaq::F_ROWS) == 0) {   // packed rows of the multi-GPU return path: no fused twin for this kernel
    const int64_t total = e->d.n * (e->obs_dim + 2);
    int64_t blocks = (total + kBlock - 1) / kBlock;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, e->d.n, e->obs_dim, caller_obs, (const float*)reward,
                       (const uint8_t*)done, e->d.rows_out);
    HIP_TRY(hipGetLastError());
  }
  if (e->d.step_ctr) {            // graph-safe mode: an F_CTR kernel advanced the device-resident counter itself
    if (self_counting) e->ctr_spread = true;
    else { hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(1), 0, st, e->d.step_ctr, (uint64_t)1 << e->d.ctr_shift); HIP_TRY(hipGetLastError()); }
  }
  e->sc.step_index += 1;
  if (e->alias) { e->last_obs = obs; if (int rc = record_alias_rows(e, st)) return rc; }
  return GAQ_OK;
}

int launch_reset(gaq_env* e, const uint8_t* mask, int do_reset, float* obs, hipStream_t st) {
  e->info_valid = false;      // (an observing pass advances the gyro-bias walk: state, too)
  if (obs && (reinterpret_cast<uintptr_t>(obs) & 15) != 0) return fail(GAQ_ERR_INVALID, "obs must be 16-byte aligned");
  StepCfg sc = e->sc;
  uint64_t key_offset = 0;
  if (do_reset) { e->reset_calls += 1; key_offset = e->reset_calls << 44; }
  if (sc.sense_input && obs) {
    if (!e->sense_next) return fail(GAQ_ERR_STATE, "sense_input: call gaq_set_sense_input_dev before an observing reset / gaq_observe");
    e->d.sense_in = e->sense_next;
    e->sense_next = nullptr;
  }
  if (e->alias) {
    // the observation written here becomes the state head: without a caller buffer use the library's own
    if (!obs && !e->pack) obs = e->own_obs;
    e->d.obs_in = e->last_obs;
    if (int rc = verify_alias_rows(e, st)) return rc;
  }
  float* caller_obs = obs;
  float* hi_out = nullptr;
  if (e->alias && e->pack) hi_out = e->own_obs;                 // heads to the library's rows, the packed observation to the caller
  else if (e->alias && e->shadow) obs = e->own_obs;
  const int tiles_per_block = kBlock / kTile;
  const dim3 grid((unsigned)((e->d.ntiles + tiles_per_block - 1) / tiles_per_block)), block(kBlock);
  const size_t lds = (size_t)kTile * e->obs_dim * 4 * tiles_per_block;
  if (lds > 65536 && !e->reset_lds_raised) {
    if (lds > 160 * 1024) return fail(GAQ_ERR_INVALID, "observation rows too large for the CU's LDS");
    HIP_TRY(hipFuncSetAttribute((const void*)&reset_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    e->reset_lds_raised = true;
  }
  hipLaunchKernelGGL(reset_kernel, grid, block, lds, st, e->d, sc, e->um, mask, do_reset, obs, alias_mode(e), key_offset, hi_out);
  HIP_TRY(hipGetLastError());
  if (e->alias) {
    e->last_obs = e->pack ? e->own_obs : obs;
    if (e->shadow && !e->pack && caller_obs != obs)
      HIP_TRY(hipMemcpyAsync(caller_obs, obs, sizeof(float) * (size_t)e->d.n * 18, hipMemcpyDeviceToDevice, st));
    if (int rc = record_alias_rows(e, st)) return rc;
  }
  return GAQ_OK;
}

// Wait for this HANDLE's work only: its private stream and the stream the caller last handed to a *_dev entry point
// (a device-wide synchronise would stall every other handle and every other library on the GPU).
int sync_handle(gaq_env* e) {
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (e->user_stream_used) HIP_TRY(hipStreamSynchronize(e->user_stream));
  return GAQ_OK;
}

// graph-safe mode: the device-resident step counter (kCtrSlots words, sum = step_index << ctr_shift; gaq_kernels.hpp)
int read_step_counter(gaq_env* e, uint64_t* step) {
  std::vector<uint64_t> w((size_t)kCtrSlots * kCtrStride);
  HIP_TRY(hipMemcpy(w.data(), e->step_ctr_mem, w.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
  uint64_t sum = 0;
  for (int k = 0; k < kCtrSlots; ++k) sum += w[k * kCtrStride];
  *step = sum >> e->d.ctr_shift;
  return GAQ_OK;
}
int write_step_counter(gaq_env* e, uint64_t step) {
  std::vector<uint64_t> w((size_t)kCtrSlots * kCtrStride, 0);
  w[0] = step << e->d.ctr_shift;
  HIP_TRY(hipMemcpy(e->step_ctr_mem, w.data(), w.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
  e->ctr_spread = false;
  return GAQ_OK;
}

// Per-episode re-randomisation keeps every env's NEXT draw staged (par_next) and refills it off the critical path; an env promoted twice
// between two refill passes flew on a stale draw.  The refill schedule makes that impossible (launch_step), the device counts it anyway,
// and every synchronous entry point that hands results to the caller looks at the count: wrong parameters must not pass silently.
int check_overrun(gaq_env* e) {
  if (!e->d.rz_overrun) return GAQ_OK;
  uint32_t o = 0;
  HIP_TRY(hipMemcpy(&o, e->d.rz_overrun, sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (o) return fail(GAQ_ERR_STATE, "internal: an env was re-randomised before its staged parameter planes were refilled");
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

// index of env i's value of `plane` inside a tile-major array with `planes` planes per tile
inline size_t tidx(int64_t i, int planes, int plane) {
  return (size_t)(i / kTile) * planes * kTile + (size_t)plane * kTile + (size_t)(i % kTile);
}

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
  StepCfg sc0;
  int D = 18;
  if (int rc = fill_step_cfg(cfg, sc0, D)) return rc;
  if (!cfg->per_env_params && check_model(cfg->model) != GAQ_OK) return GAQ_ERR_INVALID;
  int ndev = gaq_num_devices();
  if (ndev <= 0) return fail(GAQ_ERR_DEVICE, "no HIP device visible: libgaq has no CPU fallback");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(GAQ_ERR_INVALID, "device ordinal out of range");
  HIP_TRY(hipSetDevice(cfg->device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));

  gaq_env* e = new (std::nothrow) gaq_env();
  if (!e) return fail(GAQ_ERR_DEVICE, "out of host memory");
  e->cfg = *cfg;
  e->obs_dim = D;
  e->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;   // the small-batch size rule counts waves per SIMD
  StepCfg& sc = e->sc;
  sc = sc0;
  const double dt = sc.dt;

  if (!cfg->per_env_params) {
    derive_model(cfg->model, dt, e->um);
    e->um.jinv = nullptr;
    e->any_lag = !(e->um.tau_up >= 1.0 && e->um.tau_down >= 1.0);
    e->any_drag = (cfg->model.c_drag != 0.0 || cfg->model.c_roll != 0.0);
    if (cfg->control == GAQ_CTRL_MELLINGER && !inverse_jacobian(cfg->model, sc.jinv)) {
      delete e;
      return fail(GAQ_ERR_INVALID, "singular quadrotor jacobian");
    }
  } else {
    std::memset(&e->um, 0, sizeof(e->um));
    e->any_lag = true;   // decided when parameters arrive
    e->any_drag = false;
  }
  { const char* nf = getenv("GAQ_NO_FUSED"); if (nf && nf[0] == '1') e->fused_rollout = false; }
  { const char* fg = getenv("GAQ_FORCE_GENERIC"); if (fg && fg[0] == '1') e->force_generic = true; }   // tests: generic vs specialised
  e->lomix = cfg->per_env_params != 0 || e->any_lag;    // fixed for the life of the handle (the residual rows' format)
  {
    sc.motor_lag = e->any_lag ? 1 : 0; sc.drag = e->any_drag ? 1 : 0;
    bool generic, heavy, diag;
    generic_tiers(*cfg, sc, e->force_generic, generic, heavy, diag);
    const Layout L = decide_layout(*cfg, sc, D, generic);      // (the generic kernel keeps fp64 planes: plain layout whatever was asked)
    e->alias = L.alias; e->pack = L.pack; e->fp32 = L.fp32; e->shadow = L.shadow;
  }
  { const char* ca = getenv("GAQ_CHECK_ALIAS"); e->check_alias = ca && ca[0] == '1'; }
  {   // timing-only ablations (tools/latency_breakdown.py, tools/rz_ablate.sh): wrong physics by construction, so they exist
      // only in a measurement build (-DGAQ_DIAG_BUILD) -- the product library refuses the variable instead of ignoring it
    const char* ab = getenv("GAQ_ABLATE");
    sc.ablate = ab ? atoi(ab) : 0;
    if (sc.ablate != 0 && !kDiagBuild) {
      delete e;
      return fail(GAQ_ERR_INVALID, "GAQ_ABLATE is set but this libgaq is not a measurement build (make EXTRA=-DGAQ_DIAG_BUILD OUT=...): "
                                   "the ablations skip parts of the step and give wrong physics; unset the variable");
    }
    if (sc.ablate != 0) fprintf(stderr, "libgaq: GAQ_ABLATE=%d in a measurement build -- results are WRONG by construction, only timings mean anything\n", sc.ablate);
  }
  refresh_feature_flags(e);
  if (e->fp32 && !e->alias) {   // an explicit request for reduced precision is never dropped silently
    delete e;
    return fail(GAQ_ERR_INVALID, "fp32_state needs the specialised kernels (18-word world-frame obs, RawControl, default reward terms)");
  }
  if (!step_instantiated((uint32_t)e->variant)) {   // (tests/test_plan_cpu.py enumerates the reachable masks: this cannot happen)
    delete e;
    return fail(GAQ_ERR_STATE, "internal: no kernel instantiation for this feature mask");
  }

  DevPtrs& d = e->d;
  std::memset(&d, 0, sizeof(d));
  d.n = cfg->num_envs;
  d.ntiles = (cfg->num_envs + kTile - 1) / kTile;
  {   // graph-safe step counter: one step launch adds 2^ctr_shift in all (gaq_kernels.hpp: step_counter_checkin)
    uint64_t waves;
    counter_plan(d.ntiles, waves, d.ctr_shift, d.ctr_inc0);
  }
  const size_t nt = (size_t)d.ntiles;
  hipError_t he = hipSuccess;
  auto alloc0 = [&](void** p, size_t bytes) {
    if (he != hipSuccess) return;
    he = hipMalloc(p, bytes);
    if (he == hipSuccess) he = hipMemset(*p, 0, bytes);
  };
  if (e->alias) {
    if (!e->fp32) alloc0((void**)&d.lo, nt * (e->lomix ? (size_t)kMixRowsBytes : (size_t)kLoRowsBytes));
    alloc0((void**)&e->own_obs, nt * kRowsBytes);
  } else {
    alloc0((void**)&d.core, nt * kCoreBytes);
  }
  alloc0((void**)&d.lag, nt * kLagBytes);
  alloc0((void**)&d.ou, nt * kGrpBytes);
  alloc0((void**)&d.cmds, nt * kGrpBytes);
  alloc0((void**)&d.actp, nt * kGrpBytes);
  alloc0((void**)&d.goal, nt * kGrpBytes);
  alloc0((void**)&d.gyro, nt * kGrpBytes);
  alloc0((void**)&e->step_ctr_mem, sizeof(uint64_t) * kCtrSlots * kCtrStride);
  alloc0((void**)&e->alias_sum_dev, sizeof(uint64_t));
  alloc0((void**)&d.ctr, nt * kTile * sizeof(uint32_t));
  alloc0((void**)&d.done_count, sizeof(uint32_t) * 2);
  alloc0((void**)&d.nan_count, sizeof(uint32_t));
  if (cfg->compact_done) alloc0((void**)&d.done_list, nt * kTile * sizeof(uint32_t));
  if (cfg->aux_outputs) alloc0((void**)&d.aux, nt * kTile * gaq::AUX_WORDS * sizeof(float));
  if (cfg->per_env_params) {
    double* par = nullptr;
    double* jinv_dev = nullptr;
    alloc0((void**)&par, nt * kParBytes);
    if (cfg->control == GAQ_CTRL_MELLINGER) alloc0((void**)&jinv_dev, nt * kTile * 16 * sizeof(double));
    d.par = par;
    d.jinv = jinv_dev;
    alloc0((void**)&d.traj, 4 * nt * kTile * sizeof(uint32_t));          // traj | rcount | rz_flag | pfull (small kernels only: pfull_of)
    d.rcount = d.traj ? d.traj + nt * kTile : nullptr;
    d.rz_flag = d.traj ? d.traj + 2 * nt * kTile : nullptr;
    // padding envs (and envs whose parameters have not arrived yet) get a harmless unit model so their lanes stay
    // finite: every plane 1 except drag / damping (0) and the construction hints (t2t 1, motor_xy 1, com 0 -> +-1)
    e->host_par.assign(nt * kPar * kTile, 1.0);
    for (size_t t = 0; t < nt; ++t)
      for (int pl : {(int)PP_VEL_DAMP, (int)PP_DAMP_Q, (int)PP_C_DRAG, (int)PP_C_ROLL, (int)PP_COMX, (int)PP_COMY, (int)PP_OU_SIGMA})
        std::fill_n(e->host_par.begin() + (t * kPar + pl) * kTile, kTile, 0.0);
    e->pflags.assign((size_t)cfg->num_envs, 0);
  }
  if (he == hipSuccess) he = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
  if (he == hipSuccess) he = hipEventCreate(&e->ev0);
  if (he == hipSuccess) he = hipEventCreate(&e->ev1);
  if (he != hipSuccess) {
    std::string msg = std::string("device allocation failed: ") + hipGetErrorString(he);
    gaq_destroy(e);
    return fail(GAQ_ERR_DEVICE, msg);
  }
  // identity rotation and the default goal so that an un-reset env is a valid rigid body
  auto init_state = [&]() -> int {
    std::vector<float> goal(nt * 4 * kTile, 0.0f);
    for (int64_t i = 0; i < d.ntiles * kTile; ++i) {
      goal[tidx(i, 4, 2)] = 2.0f;
      if (sc.swarm.agents > 1) {   // formation goals (the first reset recomputes them on the device)
        const float ang = 6.2831853071795864769f * (float)(((uint64_t)cfg->env_id_offset + (uint64_t)i) % (uint64_t)sc.swarm.agents) /
                          (float)sc.swarm.agents;
        goal[tidx(i, 4, 0)] = sc.swarm.goal_radius * cosf(ang);
        goal[tidx(i, 4, 1)] = sc.swarm.goal_radius * sinf(ang);
      }
    }
    HIP_TRY(hipMemcpy(d.goal, goal.data(), goal.size() * sizeof(float), hipMemcpyHostToDevice));
    if (e->alias) {
      std::vector<float> rows(nt * kTile * 18, 0.0f);
      for (int64_t i = 0; i < d.ntiles * kTile; ++i) {
        rows[i * 18 + 0] = -goal[tidx(i, 4, 0)]; rows[i * 18 + 1] = -goal[tidx(i, 4, 1)];   // pos - goal with pos = 0 (formation goals: swarm)
        rows[i * 18 + 2] = -2.0f;
        for (int j : {6, 10, 14}) rows[i * 18 + j] = 1.0f;
      }
      HIP_TRY(hipMemcpy(e->own_obs, rows.data(), rows.size() * sizeof(float), hipMemcpyHostToDevice));
      e->last_obs = e->own_obs;
    } else {
      std::vector<double> core(nt * kCorePlanes * kTile, 0.0);
      for (int64_t i = 0; i < d.ntiles * kTile; ++i)
        for (int j : {6, 10, 14}) core[tidx(i, kCorePlanes, j)] = 1.0;
      HIP_TRY(hipMemcpy(d.core, core.data(), core.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if (cfg->per_env_params)
      HIP_TRY(hipMemcpy(const_cast<double*>(d.par), e->host_par.data(), e->host_par.size() * sizeof(double), hipMemcpyHostToDevice));
    return GAQ_OK;
  };
  if (int rc = init_state()) { gaq_destroy(e); return rc; }
  *out = e;
  return GAQ_OK;
}

int gaq_destroy(gaq_env* e) {
  if (!e) return GAQ_OK;
  (void)hipSetDevice(e->cfg.device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  (void)hipDeviceSynchronize();
  (void)hipFree(e->d.core); (void)hipFree(e->d.lo); (void)hipFree(e->own_obs); (void)hipFree(e->d.lag); (void)hipFree(e->d.ou); (void)hipFree(e->d.cmds);
  (void)hipFree(e->d.actp); (void)hipFree(e->d.goal); (void)hipFree(e->d.gyro); (void)hipFree(e->step_ctr_mem); if (e->info_pin) (void)hipHostFree(e->info_pin); (void)hipFree(e->alias_sum_dev); (void)hipFree(e->d.ctr);
  (void)hipFree(e->d.done_count); (void)hipFree(e->d.nan_count); (void)hipFree(e->d.done_list);
  (void)hipFree(e->d.ep_ret); (void)hipFree(e->d.ep_len); (void)hipFree(e->d.ep_acc); (void)hipFree(e->d.aux);
  (void)hipFree(const_cast<double*>(e->d.par)); (void)hipFree(const_cast<double*>(e->d.jinv));
  (void)hipFree(e->d.traj); (void)hipFree(e->d.rz_overrun);
  (void)hipFree(e->stage_dev); (void)hipFree(e->export_dev);
  if (e->stage_pin) (void)hipHostFree(e->stage_pin);
  if (e->ev0) (void)hipEventDestroy(e->ev0);
  if (e->ev1) (void)hipEventDestroy(e->ev1);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
  return GAQ_OK;
}

int gaq_is_diag_build(void) { return kDiagBuild ? 1 : 0; }

// The kernel selection of gaq_create (+ what parameters would bring) without a device: pure host logic, see select_kernel above.
int gaq_plan(const gaq_config* cfg, int32_t motor_lag, int32_t rotor_drag, int32_t randomize_every, int32_t num_cus, gaq_plan_info* out) {
  if (!cfg || !out) return fail(GAQ_ERR_INVALID, "null argument");
  StepCfg sc;
  int D = 18;
  if (int rc = fill_step_cfg(cfg, sc, D)) return rc;
  bool lag = motor_lag > 0, drag = rotor_drag > 0;
  if (!cfg->per_env_params) {
    if (check_model(cfg->model) != GAQ_OK) return GAQ_ERR_INVALID;
    Model<double> um;
    derive_model(cfg->model, sc.dt, um);
    if (motor_lag < 0) lag = !(um.tau_up >= 1.0 && um.tau_down >= 1.0);
    if (rotor_drag < 0) drag = (cfg->model.c_drag != 0.0 || cfg->model.c_roll != 0.0);
  } else {
    if (motor_lag < 0) lag = true;      // what gaq_create assumes until parameters arrive
    if (rotor_drag < 0) drag = false;
  }
  sc.motor_lag = lag ? 1 : 0; sc.drag = drag ? 1 : 0;
  const bool force_generic = env_override("GAQ_FORCE_GENERIC") == 1;
  bool generic, heavy, diag;
  generic_tiers(*cfg, sc, force_generic, generic, heavy, diag);
  // the layout is fixed at gaq_create, i.e. from the model of the configuration (per-env handles: no drag yet)
  StepCfg sc_create = sc;
  if (cfg->per_env_params) { sc_create.drag = 0; }
  bool g0, h0, d0;
  generic_tiers(*cfg, sc_create, force_generic, g0, h0, d0);
  const Layout L = decide_layout(*cfg, sc_create, D, g0);
  const int rz = (cfg->per_env_params && randomize_every > 0) ? randomize_every : 0;
  const Selection sel = select_kernel(*cfg, sc, L, D, force_generic, rz, num_cus, env_override("GAQ_PREDRAW"), env_override("GAQ_NT"));
  out->obs_dim = D;
  out->state_layout = !L.alias ? 0 : L.shadow ? 2 : 1;
  out->fp32 = (L.fp32 && L.alias) ? 1 : 0;
  out->step_variant = (int32_t)sel.variant;
  out->step_instantiated = step_instantiated(sel.variant) ? 1 : 0;
  // (rotor drag arriving on a split-state handle is refused at the launch: obs_state_alias needs the specialised kernels)
  out->launchable = (out->step_instantiated && !(L.alias && sel.generic) && !(L.fp32 && !L.alias)) ? 1 : 0;
  const uint32_t rv = (rz > 0) ? 0xFFFFFFFFu : rollout_variant_of(sel.variant, L, sel.generic);
  out->rollout_variant = rv == 0xFFFFFFFFu ? -1 : (int32_t)rv;
  out->rollout_instantiated = rv == 0xFFFFFFFFu ? 0 : (roll_instantiated(rv) ? 1 : 0);
  out->lds_per_wave = sel.lds_per_wave;
  const uint32_t rt = rows_twin_of(sel.variant), ct = ctr_twin_of(sel.variant);
  out->rows_variant = rt == 0xFFFFFFFFu ? -1 : (int32_t)rt;
  out->ctr_variant = ct == 0xFFFFFFFFu ? -1 : (int32_t)ct;
  {
    uint64_t waves; uint32_t sh, inc0;
    counter_plan((cfg->num_envs + kTile - 1) / kTile, waves, sh, inc0);
    out->ctr_waves = (int32_t)waves; out->ctr_shift = (int32_t)sh; out->ctr_inc0 = (int32_t)inc0;
  }
  return GAQ_OK;
}

int gaq_kernel_variant(const gaq_env* e) { return e ? e->variant : GAQ_ERR_INVALID; }
int gaq_launch_variant(const gaq_env* e) { return e ? (int)launch_variant_of(e) : GAQ_ERR_INVALID; }
int gaq_launched_variants(int kind, uint32_t* out, int capacity) {
  if (kind < 0 || kind > 1 || capacity < 0 || (capacity > 0 && !out)) return GAQ_ERR_INVALID;
  LaunchRecord& r = launch_record();
  std::lock_guard<std::mutex> g(r.mu);
  int k = 0;
  for (uint32_t f : r.seen[kind]) { if (k < capacity) out[k] = f; ++k; }
  return k;
}
int gaq_obs_dim(const gaq_env* e) { return e ? e->obs_dim : GAQ_ERR_INVALID; }
int gaq_obs_is_state(const gaq_env* e) { return (e && e->alias && !e->shadow) ? 1 : 0; }
int gaq_state_layout(const gaq_env* e) { return !e ? GAQ_ERR_INVALID : !e->alias ? 0 : e->shadow ? 2 : 1; }
int64_t gaq_num_envs(const gaq_env* e) { return e ? e->d.n : GAQ_ERR_INVALID; }

// flag byte of env i (1 motor lag, 2 rotor drag, 4 not compact-constructible, 8 vel / omega damping); the handle-wide
// counts move by the difference, so nothing ever scans all envs
static void set_env_flags(gaq_env* e, int64_t i, uint8_t nf) {
  const uint8_t of = e->pflags[(size_t)i];
  e->cnt_lag += (nf & 1) - (of & 1); e->cnt_drag += ((nf >> 1) & 1) - ((of >> 1) & 1);
  e->cnt_noncompact += ((nf >> 2) & 1) - ((of >> 2) & 1); e->cnt_damp += ((nf >> 3) & 1) - ((of >> 3) & 1);
  e->pflags[(size_t)i] = nf;
}
// kernel selection of the handle from the running counts
static void flags_from_counts(gaq_env* e) {
  e->any_lag = e->cnt_lag > 0; e->any_drag = e->cnt_drag > 0;
  e->sc.compact_params = (e->cnt_noncompact == 0 && !getenv("GAQ_NO_COMPACT")) ? 1 : 0;
  e->sc.zero_damp = (e->cnt_damp == 0 && !getenv("GAQ_NO_COMPACT")) ? 1 : 0;
  refresh_feature_flags(e);
}
static uint8_t tree_flags(const gaq_quad_params& t, double dt) {
  const double tu = 4 * dt / (t.motor[9] + 1e-6), td = 4 * dt / (t.motor[10] + 1e-6);
  return (uint8_t)((!(tu >= 1.0 && td >= 1.0) ? 1 : 0) | ((t.motor[7] != 0.0 || t.motor[8] != 0.0) ? 2 : 0) |
                   ((t.damp[0] != 0.0 || t.damp[1] != 0.0) ? 8 : 0));
}

// shared by gaq_set_params / gaq_set_params_indexed: `idx` == nullptr means envs first .. first+count-1
static int set_params_impl(gaq_env* e, const gaq_model* models, const int64_t* idx, int64_t first, int64_t count) {
  if (!e || !models) return fail(GAQ_ERR_INVALID, "null argument");
  e->info_valid = false;
  if (!e->cfg.per_env_params) return fail(GAQ_ERR_STATE, "handle was created with per_env_params = 0");
  if (e->dev_params) return fail(GAQ_ERR_STATE, "this handle's parameters are managed on the device (gaq_set_randomizer / "
                                                "gaq_set_param_trees): gaq_set_params is not available");
  if (count < 0) return fail(GAQ_ERR_INVALID, "negative count");
  if (count == 0) return GAQ_OK;
  auto env_of = [&](int64_t k) { return idx ? idx[k] : first + k; };
  int64_t lo = e->d.n, hi = -1;
  for (int64_t k = 0; k < count; ++k) {
    const int64_t i = env_of(k);
    if (i < 0 || i >= e->d.n) return fail(GAQ_ERR_INVALID, "env index out of bounds");
    lo = i < lo ? i : lo; hi = i > hi ? i : hi;
  }
  HIP_TRY(hipSetDevice(e->cfg.device));
  double* hp = e->host_par.data();
  std::vector<double> ji(e->d.jinv ? (size_t)count * 16 : 0);
  for (int64_t k = 0; k < count; ++k) {
    if (check_model(models[k]) != GAQ_OK) return GAQ_ERR_INVALID;
    if (e->d.jinv && !inverse_jacobian(models[k], ji.data() + (size_t)k * 16)) return fail(GAQ_ERR_INVALID, "singular quadrotor jacobian");
  }
  for (int64_t k = 0; k < count; ++k) {
    Model<double> m;
    derive_model(models[k], e->sc.dt, m);
    const int64_t i = env_of(k);
    auto P = [&](int plane) -> double& { return hp[tidx(i, kPar, plane)]; };
    P(PP_MASS) = m.mass; P(PP_INV_MASS) = m.inv_mass;
    for (int j = 0; j < 3; ++j) { P(PP_INERTIA + j) = m.inertia[j]; P(PP_INV_INERTIA + j) = m.inv_inertia[j]; }
    for (int j = 0; j < 4; ++j) {
      P(PP_THRUST_MAX + j) = m.thrust_max[j]; P(PP_TORQUE_MAX + j) = m.torque_max[j];
      P(PP_PROP_X + j) = m.prop_x[j]; P(PP_PROP_Y + j) = m.prop_y[j]; P(PP_PROP_Z + j) = m.prop_z[j];
    }
    P(PP_TAU_UP) = m.tau_up; P(PP_TAU_DOWN) = m.tau_down; P(PP_LINEARITY) = m.linearity;
    P(PP_T_UP) = models[k].damp_time_up; P(PP_T_DOWN) = models[k].damp_time_down;
    P(PP_ARM) = m.arm; P(PP_VEL_DAMP) = m.vel_damp; P(PP_DAMP_Q) = m.damp_omega_q;
    P(PP_C_DRAG) = m.c_drag; P(PP_C_ROLL) = m.c_roll;
    reinterpret_cast<float*>(hp + tidx(i - i % kTile, kPar, PP_OU_SIGMA))[i % kTile] = (float)models[k].ou_sigma;   // fp32 plane
    // construction hints: accepted only when they reproduce the given numbers bit for bit
    double hint[5] = {0, 0, 0, 0, 0};
    const bool compact_ok = find_construction(m, hint);
    P(PP_COMPACT_OK) = compact_ok ? 1.0 : 0.0;
    for (int j = 0; j < 5; ++j) P(PP_T2T + j) = hint[j];
    // flag byte of this env; the handle-wide counts move by the difference (no scan over all envs)
    set_env_flags(e, i, (uint8_t)((!(m.tau_up >= 1.0 && m.tau_down >= 1.0) ? 1 : 0) | ((m.c_drag != 0.0 || m.c_roll != 0.0) ? 2 : 0) |
                                  (!compact_ok ? 4 : 0) | ((m.vel_damp != 0.0 || m.damp_omega_q != 0.0) ? 8 : 0)));
  }
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (int rc_ = sync_handle(e)) return rc_;
  if (e->d.jinv) {   // Mellinger: one inverse jacobian per env (quadrotor_control.py:290-291)
    if (!idx) {
      HIP_TRY(hipMemcpy(const_cast<double*>(e->d.jinv) + (size_t)first * 16, ji.data(), ji.size() * sizeof(double), hipMemcpyHostToDevice));
    } else {
      for (int64_t k = 0; k < count; ++k)
        HIP_TRY(hipMemcpy(const_cast<double*>(e->d.jinv) + (size_t)idx[k] * 16, ji.data() + (size_t)k * 16, 16 * sizeof(double), hipMemcpyHostToDevice));
    }
  }
  // upload the touched tiles only: runs of adjacent touched tiles go in one copy each (a contiguous range is one run)
  {
    std::vector<int64_t> tiles((size_t)count);
    for (int64_t k = 0; k < count; ++k) tiles[(size_t)k] = env_of(k) / kTile;
    std::sort(tiles.begin(), tiles.end());
    tiles.erase(std::unique(tiles.begin(), tiles.end()), tiles.end());
    for (size_t a = 0; a < tiles.size();) {
      size_t b = a + 1;
      while (b < tiles.size() && tiles[b] == tiles[b - 1] + 1) ++b;
      HIP_TRY(hipMemcpy(const_cast<double*>(e->d.par) + (size_t)tiles[a] * kPar * kTile, hp + (size_t)tiles[a] * kPar * kTile,
                        (b - a) * (size_t)kParBytes, hipMemcpyHostToDevice));
      a = b;
    }
  }
  // a new QuadrotorDynamics starts with since_last_svd = 0 and a fresh OUNoise (quadrotor.py:104, :198)
  if (!idx) {
    const dim3 grid((unsigned)((count + kBlock - 1) / kBlock)), block(kBlock);
    hipLaunchKernelGGL(clear_dynamics_kernel, grid, block, 0, e->stream, e->d, (const int64_t*)nullptr, first, count);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->stream));
  } else {
    Scratch di;
    if (di.alloc(sizeof(int64_t) * (size_t)count)) return GAQ_ERR_DEVICE;
    HIP_TRY(hipMemcpy(di.p, idx, sizeof(int64_t) * (size_t)count, hipMemcpyHostToDevice));
    const dim3 grid((unsigned)((count + kBlock - 1) / kBlock)), block(kBlock);
    hipLaunchKernelGGL(clear_dynamics_kernel, grid, block, 0, e->stream, e->d, (const int64_t*)di.p, (int64_t)0, count);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  flags_from_counts(e);
  return GAQ_OK;
}

int gaq_set_params(gaq_env* e, const gaq_model* models, int64_t first, int64_t count) {
  if (e && (first < 0 || count < 0 || first + count > e->d.n)) return fail(GAQ_ERR_INVALID, "env range out of bounds");
  return set_params_impl(e, models, nullptr, first, count);
}

int gaq_set_params_indexed(gaq_env* e, const gaq_model* models, const int64_t* env_idx, int64_t count) {
  if (!env_idx) return fail(GAQ_ERR_INVALID, "null argument");
  return set_params_impl(e, models, env_idx, 0, count);
}

static int need_device_params(gaq_env* e) {
  if (!e) return fail(GAQ_ERR_INVALID, "null handle");
  if (!e->cfg.per_env_params) return fail(GAQ_ERR_STATE, "handle was created with per_env_params = 0");
  return GAQ_OK;
}
static int check_tree(const gaq_quad_params& t, bool by_density = false) {
  const double* v = reinterpret_cast<const double*>(&t);
  for (int k = 0; k < GAQ_TREE_DOUBLES; ++k) if (!std::isfinite(v[k])) return fail(GAQ_ERR_INVALID, "parameter tree: non-finite leaf");
  if (!by_density && !(t.body[3] + t.payload[3] + 4 * (t.arms[3] + t.motors[2] + t.propellers[2]) > 0))
    return fail(GAQ_ERR_INVALID, "parameter tree: total mass must be positive");
  if (t.motor[7] != 0.0 || t.motor[8] != 0.0)
    return fail(GAQ_ERR_INVALID, "parameter tree: rotor drag / rolling moment (C_drag, C_roll != 0) needs the generic kernel and the host path (gaq_set_params)");
  return GAQ_OK;
}

int gaq_set_randomizer(gaq_env* e, const gaq_randomizer* rz) {
  if (int rc = need_device_params(e)) return rc;
  e->info_valid = false;
  if (!rz) return fail(GAQ_ERR_INVALID, "null argument");
  if (rz->sampler < 0 || rz->sampler > 2 || rz->every < 0) return fail(GAQ_ERR_INVALID, "randomizer: unknown sampler / negative period");
  if (rz->every > 0 && !e->cfg.auto_reset)
    return fail(GAQ_ERR_INVALID, "randomizer: every > 0 (dynamics_randomize_every inside the step launch) needs auto_reset = 1 -- without it a "
                                 "finished env reports done on every step until the caller resets it; call gaq_randomize_dev(mask) then");
  if (rz->sampler != 2) { if (int rc = check_tree(rz->base)) return rc; }
  for (int k = 0; k < GAQ_TREE_DOUBLES; ++k) if (!std::isfinite(rz->ratio[k])) return fail(GAQ_ERR_INVALID, "randomizer: non-finite noise ratio");
  static_assert(sizeof(gaq::ParamTree) == sizeof(gaq_quad_params) && gaq::TL_COUNT == GAQ_TREE_DOUBLES, "parameter tree layout");
  HIP_TRY(hipSetDevice(e->cfg.device));
  if (int rc_ = sync_handle(e)) return rc_;
  e->rz.sampler = rz->sampler; e->rz.every = rz->every;
  std::memcpy(e->rz.ratio, rz->ratio, sizeof(e->rz.ratio));
  std::memcpy(&e->rz.base, &rz->base, sizeof(e->rz.base));
  e->rz_on = true; e->dev_params = true;
  if (rz->every > 0 && !e->d.par_next) {       // per-episode re-randomisation: staged planes of every env's NEXT draw + flags
    const size_t nt = (size_t)e->d.ntiles;
    // everything is allocated and filled BEFORE the handle's pointers change: an error on the way leaves the handle as it was
    Scratch both_, over_;                      // [par planes | skew | par_next rows] in one allocation; the overrun counter
    if (both_.alloc(2 * nt * kParBytes + kParNextSkew * sizeof(double)) || over_.alloc(sizeof(uint32_t))) return GAQ_ERR_DEVICE;
    double* both = (double*)both_.p;
    HIP_TRY(hipMemcpy(both, e->d.par, nt * kParBytes, hipMemcpyDeviceToDevice));
    HIP_TRY(hipMemset(both + nt * kPar * kTile, 0, nt * kParBytes + kParNextSkew * sizeof(double)));      // rows: filled by the first refill pass
    HIP_TRY(hipMemset(over_.p, 0, sizeof(uint32_t)));
    {   // nothing staged yet: the first refill pass derives every env's next draw
      std::vector<uint32_t> ones(nt * kTile, 1u);
      HIP_TRY(hipMemcpy(e->d.rz_flag, ones.data(), ones.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    (void)hipFree(const_cast<double*>(e->d.par));
    e->d.par = both;
    e->d.par_next = both + nt * kPar * kTile + kParNextSkew;
    e->d.rz_overrun = (uint32_t*)over_.p;
    both_.p = nullptr; over_.p = nullptr;      // owned by the handle now
  }
  e->d.rz_every = e->d.par_next ? rz->every : 0;
  e->rz_refill_now = true;
  // what the sampler can produce is known from the nominal model: a leaf that is zero stays zero (scale = |ratio/2 v|),
  // so lag / damping exist iff the base has them; the derived planes always follow the compact construction
  // (RandomQuad: motor time constants U(0.15, 0.2) s -> lag; no drag, no damping: quadrotor_randomization.py:211-229)
  const uint8_t nf = rz->sampler == 2 ? (uint8_t)1 : tree_flags(rz->base, e->sc.dt);
  for (int64_t i = 0; i < e->d.n; ++i) set_env_flags(e, i, nf);
  flags_from_counts(e);
  return GAQ_OK;
}

int gaq_randomize_dev(gaq_env* e, const uint8_t* mask_dev, void* stream) {
  if (int rc = need_device_params(e)) return rc;
  e->info_valid = false;
  if (!e->rz_on) return fail(GAQ_ERR_STATE, "no randomizer installed (gaq_set_randomizer)");
  HIP_TRY(hipSetDevice(e->cfg.device));
  e->user_stream = (hipStream_t)stream; e->user_stream_used = true;
  const dim3 grid((unsigned)((e->d.n + kBlock - 1) / kBlock)), block(kBlock);
  hipLaunchKernelGGL(rerandomize_kernel, grid, block, 0, (hipStream_t)stream, e->d, e->sc, e->rz, mask_dev, 1, (double*)nullptr, (int64_t)0, (int64_t)0);
  HIP_TRY(hipGetLastError());
  if (int rc = launch_jinv(e, (hipStream_t)stream, nullptr)) return rc;
  if (e->d.rz_every > 0) return launch_refill(e, (hipStream_t)stream);      // the redrawn envs' staged planes: one draw further
  return GAQ_OK;
}

int gaq_set_param_trees(gaq_env* e, const gaq_quad_params* trees, int32_t links_by_density, int64_t first, int64_t count) {
  if (int rc = need_device_params(e)) return rc;
  e->info_valid = false;
  if (!trees) return fail(GAQ_ERR_INVALID, "null argument");
  if (first < 0 || count < 0 || first + count > e->d.n) return fail(GAQ_ERR_INVALID, "env range out of bounds");
  if (count == 0) return GAQ_OK;
  if (!e->dev_params && (e->cnt_lag | e->cnt_drag | e->cnt_noncompact | e->cnt_damp) != 0)
    return fail(GAQ_ERR_STATE, "this handle already holds host-supplied parameters (gaq_set_params): do not mix the two paths");
  for (int64_t k = 0; k < count; ++k) if (int rc = check_tree(trees[k], links_by_density != 0)) return rc;
  HIP_TRY(hipSetDevice(e->cfg.device));
  if (int rc_ = sync_handle(e)) return rc_;
  Scratch dt_;
  if (dt_.alloc(sizeof(gaq_quad_params) * (size_t)count)) return GAQ_ERR_DEVICE;
  HIP_TRY(hipMemcpy(dt_.p, trees, sizeof(gaq_quad_params) * (size_t)count, hipMemcpyHostToDevice));
  const dim3 grid((unsigned)((count + kBlock - 1) / kBlock)), block(kBlock);
  hipLaunchKernelGGL(derive_trees_kernel, grid, block, 0, e->stream, e->d, e->sc, (const double*)dt_.p, (int)links_by_density, first, count);
  HIP_TRY(hipGetLastError());
  e->dev_params = true;
  if (int rc = launch_jinv(e, e->stream, nullptr)) return rc;
  HIP_TRY(hipStreamSynchronize(e->stream));
  for (int64_t k = 0; k < count; ++k) set_env_flags(e, first + k, tree_flags(trees[k], e->sc.dt));
  flags_from_counts(e);
  return GAQ_OK;
}

int gaq_get_params(gaq_env* e, gaq_model* out, int64_t first, int64_t count) {
  if (!e || !out) return fail(GAQ_ERR_INVALID, "null argument");
  if (!e->cfg.per_env_params) return fail(GAQ_ERR_STATE, "handle was created with per_env_params = 0");
  if (first < 0 || count < 0 || first + count > e->d.n) return fail(GAQ_ERR_INVALID, "env range out of bounds");
  if (count == 0) return GAQ_OK;
  HIP_TRY(hipSetDevice(e->cfg.device));
  if (int rc_ = sync_handle(e)) return rc_;
  if (int rc_ = check_overrun(e)) return rc_;
  // A READ: nothing of the handle changes.  Per-episode re-randomisation moves only the planes the step kernels read when it promotes an
  // env (gaq_kernels.hpp: kHotPlanes); for exactly those envs (resample count != the count of their last full write) the whole row is
  // derived afresh from (seed, global env index, count) into a scratch buffer -- only the envs asked for, whatever the randomizer's
  // period is NOW (gaq_set_randomizer(every = 0) after a period of promotions leaves the stale planes stale)
  std::vector<double> rows;
  if (e->rz_on && e->cold_stale) {
    Scratch rs_;
    if (rs_.alloc(sizeof(double) * (size_t)count * kPar)) return GAQ_ERR_DEVICE;
    const dim3 grid((unsigned)((count + kBlock - 1) / kBlock)), block(kBlock);
    hipLaunchKernelGGL(rerandomize_kernel, grid, block, 0, e->stream, e->d, e->sc, e->rz, (const uint8_t*)nullptr, 3, (double*)rs_.p, first, count);
    HIP_TRY(hipGetLastError());
    rows.resize((size_t)count * kPar);
    HIP_TRY(hipMemcpyAsync(rows.data(), rs_.p, rows.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  const int64_t t0 = first / kTile, t1 = (first + count - 1) / kTile + 1;
  std::vector<double> buf((size_t)(t1 - t0) * kPar * kTile);
  HIP_TRY(hipMemcpy(buf.data(), e->d.par + (size_t)t0 * kPar * kTile, buf.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (int64_t k = 0; k < count; ++k) {
    const int64_t i = first + k - t0 * kTile;
    const double* row = (!rows.empty() && rows[(size_t)k * kPar + PP_COMPACT_OK] > 0.0) ? &rows[(size_t)k * kPar] : nullptr;
    auto P = [&](int plane) { return row ? row[plane] : buf[tidx(i, kPar, plane)]; };
    gaq_model& m = out[k];
    m.mass = P(PP_MASS);
    for (int j = 0; j < 3; ++j) m.inertia[j] = P(PP_INERTIA + j);
    for (int j = 0; j < 4; ++j) {
      m.thrust_max[j] = P(PP_THRUST_MAX + j); m.torque_max[j] = P(PP_TORQUE_MAX + j);
      m.prop_pos[3 * j] = P(PP_PROP_X + j); m.prop_pos[3 * j + 1] = P(PP_PROP_Y + j); m.prop_pos[3 * j + 2] = P(PP_PROP_Z + j);
    }
    m.damp_time_up = P(PP_T_UP); m.damp_time_down = P(PP_T_DOWN); m.linearity = P(PP_LINEARITY); m.arm = P(PP_ARM);
    m.ou_sigma = row ? row[PP_OU_SIGMA] : (double)reinterpret_cast<const float*>(&buf[tidx(i - i % kTile, kPar, PP_OU_SIGMA)])[i % kTile];
    m.vel_damp = P(PP_VEL_DAMP); m.damp_omega_quadratic = P(PP_DAMP_Q); m.c_drag = P(PP_C_DRAG); m.c_roll = P(PP_C_ROLL);
  }
  return GAQ_OK;
}

int gaq_get_param_trees(gaq_env* e, gaq_quad_params* out, int64_t first, int64_t count) {
  if (int rc = need_device_params(e)) return rc;
  if (!out) return fail(GAQ_ERR_INVALID, "null argument");
  if (!e->rz_on) return fail(GAQ_ERR_STATE, "no randomizer installed (gaq_set_randomizer): the sampled trees are a function of its settings");
  if (first < 0 || count < 0 || first + count > e->d.n) return fail(GAQ_ERR_INVALID, "env range out of bounds");
  if (count == 0) return GAQ_OK;
  HIP_TRY(hipSetDevice(e->cfg.device));
  if (int rc_ = sync_handle(e)) return rc_;
  Scratch dt_;
  if (dt_.alloc(sizeof(gaq_quad_params) * (size_t)count)) return GAQ_ERR_DEVICE;
  const dim3 grid((unsigned)((count + kBlock - 1) / kBlock)), block(kBlock);
  hipLaunchKernelGGL(rerandomize_kernel, grid, block, 0, e->stream, e->d, e->sc, e->rz, (const uint8_t*)nullptr, 1, (double*)dt_.p, first, count);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, dt_.p, sizeof(gaq_quad_params) * (size_t)count, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GAQ_OK;
}

int gaq_reset_dev(gaq_env* e, const uint8_t* mask_dev, float* obs_dev, void* stream) {
  if (!e) return fail(GAQ_ERR_INVALID, "null handle");
  HIP_TRY(hipSetDevice(e->cfg.device));
  e->user_stream = (hipStream_t)stream; e->user_stream_used = true;
  return launch_reset(e, mask_dev, 1, obs_dev, (hipStream_t)stream);
}

int gaq_reset(gaq_env* e, const uint8_t* mask, float* obs_out) {
  if (!e) return fail(GAQ_ERR_INVALID, "null handle");
  HIP_TRY(hipSetDevice(e->cfg.device));
  if (int rc_ = sync_handle(e)) return rc_;   // earlier *_dev calls may still be running on the caller's stream
  const int64_t n = e->d.n;
  Scratch dm, dobs;
  if (mask) { if (dm.alloc(n)) return GAQ_ERR_DEVICE; HIP_TRY(hipMemcpyAsync(dm.p, mask, n, hipMemcpyHostToDevice, e->stream)); }
  float* dev_obs = nullptr;
  if (e->alias && !e->pack) dev_obs = e->own_obs;     // the device copy must outlive the call: it is state
  else if (obs_out) { if (dobs.alloc(sizeof(float) * n * e->obs_dim)) return GAQ_ERR_DEVICE; dev_obs = (float*)dobs.p; }
  int rc = launch_reset(e, mask ? (const uint8_t*)dm.p : nullptr, 1, dev_obs, e->stream);
  if (rc) return rc;
  if (obs_out) HIP_TRY(hipMemcpyAsync(obs_out, dev_obs, sizeof(float) * n * e->obs_dim, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GAQ_OK;
}

int gaq_observe(gaq_env* e, float* obs_out) {
  if (!e || !obs_out) return fail(GAQ_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  if (int rc_ = sync_handle(e)) return rc_;
  const int64_t n = e->d.n;
  if (e->alias && !e->pack) {   // the current observation is the state head itself
    HIP_TRY(hipMemcpy(obs_out, e->last_obs, sizeof(float) * n * 18, hipMemcpyDeviceToHost));
    return GAQ_OK;
  }
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
  e->user_stream = (hipStream_t)stream; e->user_stream_used = true;
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
  e->info_valid = false;
  if (e->sc.noise == gaq::NOISE_INPUT) return fail(GAQ_ERR_INVALID, "step_many does not support GAQ_NOISE_INPUT");
  const int64_t n = e->d.n;
  if (T > 1 && (((size_t)n * e->obs_dim * 4) & 15)) return fail(GAQ_ERR_INVALID, "step_many needs N*obs_dim*4 to be a multiple of 16");
  HIP_TRY(hipSetDevice(e->cfg.device));
  e->user_stream = (hipStream_t)stream; e->user_stream_used = true;
  hipStream_t st = (hipStream_t)stream;
  if (e->timing) HIP_TRY(hipEventRecord(e->ev0, st));
  const Layout L_{e->alias, e->pack, e->shadow, e->fp32};
  const uint32_t roll_variant = rollout_variant_of((uint32_t)e->variant, L_, e->needs_generic);
  const bool fused = T > 1 && roll_variant != 0xFFFFFFFFu && e->fused_rollout && !e->d.ep_ret && !e->d.done_list && !e->d.rows_out &&
                     !(e->rz_on && e->rz.every > 0);
  if (fused) {
    if ((reinterpret_cast<uintptr_t>(actions) & 15) || (reinterpret_cast<uintptr_t>(obs) & 15))
      return fail(GAQ_ERR_INVALID, "actions and obs must be 16-byte aligned");
    e->d.obs_in = e->last_obs;
    if (int rc = verify_alias_rows(e, st)) return rc;
    e->d.obs_copy = nullptr;
    e->d.hi_final = e->shadow ? e->own_obs : nullptr;     // shadow mode: the final heads return to the library's own rows
    const int tiles_per_block = kBlock / kTile;
    const dim3 grid((unsigned)((e->d.ntiles + tiles_per_block - 1) / tiles_per_block)), block(kBlock);
    const size_t lds = (size_t)e->lds_per_wave * tiles_per_block;
    const int lpw = e->lds_per_wave;
    if (roll_variant != e->noted_roll) { launch_record().note(1, roll_variant); e->noted_roll = roll_variant; }
#define GAQ_ROLL(FEAT) \
  hipLaunchKernelGGL(rollout_kernel<(FEAT)>, grid, block, lds, st, e->d, e->sc, e->um, (int)T, actions, obs, reward, done, lpw)
    switch (roll_variant) {
#define GAQ_X(FEAT) case (FEAT): GAQ_ROLL(FEAT); break;
      GAQ_ROLL_ALL(GAQ_X)
#undef GAQ_X
      default: return fail(GAQ_ERR_STATE, "internal: no rollout instantiation for this feature mask");
    }
#undef GAQ_ROLL
    HIP_TRY(hipGetLastError());
    if (e->d.step_ctr) { hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(1), 0, st, e->d.step_ctr, (uint64_t)T << e->d.ctr_shift); HIP_TRY(hipGetLastError()); }
    e->sc.step_index += (uint64_t)T;
    e->last_obs = e->shadow ? e->own_obs : obs + (size_t)(T - 1) * n * 18;
    e->d.hi_final = nullptr;
    if (int rc = record_alias_rows(e, st)) return rc;
  } else {
    for (int32_t t = 0; t < T; ++t) {
      int rc = launch_step(e, actions + (size_t)t * n * 4, obs + (size_t)t * n * e->obs_dim, reward + (size_t)t * n,
                           done + (size_t)t * n, st);
      if (rc) return rc;
    }
  }
  if (e->timing) { HIP_TRY(hipEventRecord(e->ev1, st)); e->timed = true; }
  return GAQ_OK;
}

int gaq_step(gaq_env* e, const float* actions, float* obs, float* reward, uint8_t* done) {
  if (!e || !actions || !obs || !reward || !done) return fail(GAQ_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  if (int rc_ = sync_handle(e)) return rc_;
  const size_t n = (size_t)e->d.n;
  const size_t D = (size_t)e->obs_dim;
  if (!e->stage_dev) {   // persistent device block; small batches also get a pinned host mirror
    e->off_rew = (16 * n + 15) & ~(size_t)15;
    e->off_done = e->off_rew + 4 * n;
    e->off_obs = (e->off_done + n + 15) & ~(size_t)15;
    e->stage_bytes = e->off_obs + 4 * D * n;
    HIP_TRY(hipMalloc((void**)&e->stage_dev, e->stage_bytes));
    // below ~1 MiB the call is latency-bound: one pinned H2D + one pinned D2H beat four pageable copies (2.5x at
    // N = 1); above it the extra host memcpy costs more than the pageable DMA path loses
    if (e->stage_bytes <= ((size_t)1 << 20)) {
      // ... and the kernels reach that mirror themselves (mapped, host-coherent memory): the actions are read and observation / reward / done
      // written over the bus by the step launch, no copy-engine operation at all -- a single-env step() is then one launch + one
      // synchronisation instead of a copy in, a launch and a copy out (GAQ_ZERO_COPY=0: the copies, for the A/B)
      const bool zc = env_override("GAQ_ZERO_COPY") != 0;
      HIP_TRY(hipHostMalloc((void**)&e->stage_pin, e->stage_bytes, zc ? hipHostMallocMapped : hipHostMallocDefault));
      if (zc) {
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, e->stage_pin, 0) == hipSuccess && dp) e->stage_map = static_cast<char*>(dp);
        else (void)hipGetLastError();
      }
    }
  }
  char* dv = e->stage_dev;
  char* pin = e->stage_pin;
  if (pin && e->stage_map) {
    char* mp = e->stage_map;
    // alias layout proper (the caller's tensor IS the state's heads): that tensor has to live on the device -> the library's rows + one copy
    const bool heads_in_obs = e->alias && !e->pack && !e->shadow;
    float* obs_arg = heads_in_obs ? e->own_obs : reinterpret_cast<float*>(mp + e->off_obs);
    std::memcpy(pin, actions, 16 * n);
    int rc = gaq_step_dev(e, reinterpret_cast<const float*>(mp), obs_arg, reinterpret_cast<float*>(mp + e->off_rew),
                          reinterpret_cast<uint8_t*>(mp + e->off_done), e->stream);
    if (rc) return rc;
    if (heads_in_obs) HIP_TRY(hipMemcpyAsync(pin + e->off_obs, e->own_obs, 4 * D * n, hipMemcpyDeviceToHost, e->stream));
    if (e->d.aux) {   // the info dict's inputs ride along: state planes and aux rows written into the mapped mirror by one small kernel
      const size_t sbytes = sizeof(double) * GAQ_STATE_PLANES * n, abytes = sizeof(float) * gaq::AUX_WORDS * n;
      if (!e->info_pin) {
        HIP_TRY(hipHostMalloc((void**)&e->info_pin, sbytes + abytes, hipHostMallocMapped));
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, e->info_pin, 0) == hipSuccess && dp) e->info_map = static_cast<char*>(dp);
        else (void)hipGetLastError();
      }
      if (e->alias) e->d.obs_in = e->last_obs;
      const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
      if (e->info_map) {
        hipLaunchKernelGGL(export_kernel, grid, block, 0, e->stream, e->d, alias_mode(e), reinterpret_cast<double*>(e->info_map),
                           reinterpret_cast<float*>(e->info_map + sbytes));
        HIP_TRY(hipGetLastError());
      } else {
        if (!e->export_dev) HIP_TRY(hipMalloc((void**)&e->export_dev, sbytes));
        hipLaunchKernelGGL(export_kernel, grid, block, 0, e->stream, e->d, alias_mode(e), e->export_dev, (float*)nullptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(e->info_pin, e->export_dev, sbytes, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipMemcpyAsync(e->info_pin + sbytes, e->d.aux, abytes, hipMemcpyDeviceToHost, e->stream));
      }
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->info_valid = e->d.aux != nullptr;
    std::memcpy(reward, pin + e->off_rew, 4 * n);
    std::memcpy(done, pin + e->off_done, n);
    std::memcpy(obs, pin + e->off_obs, 4 * D * n);
    if (int rc_ = check_overrun(e)) return rc_;
    for (size_t i = 0; i < n; ++i) {
      if (!std::isfinite(reward[i])) {
        HIP_TRY(hipMemset(e->d.nan_count, 0, sizeof(uint32_t)));
        return fail(GAQ_ERR_NAN, "QuadEnv: reward is Nan");
      }
    }
    return GAQ_OK;
  }
  // alias mode: the observation on the device is state and must persist -> the library's own buffer
  float* dev_obs = (e->alias && !e->pack) ? e->own_obs : reinterpret_cast<float*>(dv + e->off_obs);
  if (pin) std::memcpy(pin, actions, 16 * n);
  HIP_TRY(hipMemcpyAsync(dv, pin ? (const void*)pin : (const void*)actions, 16 * n, hipMemcpyHostToDevice, e->stream));
  int rc = gaq_step_dev(e, reinterpret_cast<const float*>(dv), dev_obs, reinterpret_cast<float*>(dv + e->off_rew),
                        reinterpret_cast<uint8_t*>(dv + e->off_done), e->stream);
  if (rc) return rc;
  if (pin) {
    if (e->alias && !e->pack) {
      HIP_TRY(hipMemcpyAsync(pin + e->off_rew, dv + e->off_rew, e->off_obs - e->off_rew, hipMemcpyDeviceToHost, e->stream));
      HIP_TRY(hipMemcpyAsync(pin + e->off_obs, dev_obs, 4 * D * n, hipMemcpyDeviceToHost, e->stream));
    } else {
      HIP_TRY(hipMemcpyAsync(pin + e->off_rew, dv + e->off_rew, e->stage_bytes - e->off_rew, hipMemcpyDeviceToHost, e->stream));
    }
    if (e->d.aux) {   // the info dict's inputs ride along (see gaq_env::info_pin)
      const size_t sbytes = sizeof(double) * GAQ_STATE_PLANES * n, abytes = sizeof(float) * gaq::AUX_WORDS * n;
      if (!e->info_pin) HIP_TRY(hipHostMalloc((void**)&e->info_pin, sbytes + abytes, hipHostMallocDefault));
      if (!e->export_dev) HIP_TRY(hipMalloc((void**)&e->export_dev, sbytes));
      if (e->alias) e->d.obs_in = e->last_obs;
      const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
      hipLaunchKernelGGL(export_kernel, grid, block, 0, e->stream, e->d, alias_mode(e), e->export_dev, (float*)nullptr);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(e->info_pin, e->export_dev, sbytes, hipMemcpyDeviceToHost, e->stream));
      HIP_TRY(hipMemcpyAsync(e->info_pin + sbytes, e->d.aux, abytes, hipMemcpyDeviceToHost, e->stream));
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->info_valid = e->d.aux != nullptr;
    std::memcpy(reward, pin + e->off_rew, 4 * n);
    std::memcpy(done, pin + e->off_done, n);
    std::memcpy(obs, pin + e->off_obs, 4 * D * n);
  } else {
    HIP_TRY(hipMemcpyAsync(obs, dev_obs, 4 * D * n, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(reward, dv + e->off_rew, 4 * n, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(done, dv + e->off_done, n, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  if (int rc_ = check_overrun(e)) return rc_;
  // the reference raises on a non-finite reward inside step() (quadrotor.py:633-636): same here, on the host copy
  for (size_t i = 0; i < n; ++i) {
    if (!std::isfinite(reward[i])) {
      HIP_TRY(hipMemset(e->d.nan_count, 0, sizeof(uint32_t)));
      return fail(GAQ_ERR_NAN, "QuadEnv: reward is Nan");
    }
  }
  return GAQ_OK;
}

int gaq_set_noise_input_dev(gaq_env* e, const float* normals_dev) {
  if (!e || !normals_dev) return fail(GAQ_ERR_INVALID, "null argument");
  if (e->sc.noise != gaq::NOISE_INPUT) return fail(GAQ_ERR_STATE, "handle was not created with GAQ_NOISE_INPUT");
  e->noise_next = normals_dev;
  return GAQ_OK;
}

int gaq_set_sense_input_dev(gaq_env* e, const float* draws_dev) {
  if (!e || !draws_dev) return fail(GAQ_ERR_INVALID, "null argument");
  if (!e->sc.sense_input) return fail(GAQ_ERR_STATE, "handle was not created with sense_input (and sensor noise or a t2w / t2t observation)");
  e->sense_next = draws_dev;
  return GAQ_OK;
}

int gaq_set_action_dtype(gaq_env* e, int32_t is_float32) {
  if (!e) return fail(GAQ_ERR_INVALID, "null handle");
  e->sc.action_f32 = is_float32 ? 1 : 0;       // a launch constant: takes effect with the next step
  return GAQ_OK;
}

int gaq_get_aux(gaq_env* e, float* host_out) {
  if (!e || !host_out) return fail(GAQ_ERR_INVALID, "null argument");
  if (!e->d.aux) return fail(GAQ_ERR_STATE, "handle was created with aux_outputs = 0");
  if (e->info_valid) {
    std::memcpy(host_out, e->info_pin + sizeof(double) * GAQ_STATE_PLANES * (size_t)e->d.n, sizeof(float) * (size_t)e->d.n * gaq::AUX_WORDS);
    return GAQ_OK;
  }
  HIP_TRY(hipSetDevice(e->cfg.device));
  if (int rc_ = sync_handle(e)) return rc_;
  HIP_TRY(hipMemcpy(host_out, e->d.aux, sizeof(float) * (size_t)e->d.n * gaq::AUX_WORDS, hipMemcpyDeviceToHost));
  return GAQ_OK;
}

// ABI state planes (include/gaq.h) <-> tile-major device arrays
int gaq_get_state(gaq_env* e, double* hp) {
  if (!e || !hp) return fail(GAQ_ERR_INVALID, "null argument");
  if (e->info_valid) { std::memcpy(hp, e->info_pin, sizeof(double) * GAQ_STATE_PLANES * (size_t)e->d.n); return GAQ_OK; }
  HIP_TRY(hipSetDevice(e->cfg.device));
  if (int rc_ = sync_handle(e)) return rc_;
  const size_t n = (size_t)e->d.n;
  const size_t bytes = sizeof(double) * GAQ_STATE_PLANES * n;
  if (!e->export_dev) HIP_TRY(hipMalloc((void**)&e->export_dev, bytes));
  if (e->alias) e->d.obs_in = e->last_obs;
  const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
  hipLaunchKernelGGL(export_kernel, grid, block, 0, e->stream, e->d, alias_mode(e), e->export_dev, (float*)nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(hp, e->export_dev, bytes, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GAQ_OK;
}

int gaq_set_state(gaq_env* e, const double* hp) {
  if (!e || !hp) return fail(GAQ_ERR_INVALID, "null argument");
  e->info_valid = false;
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (int rc_ = sync_handle(e)) return rc_;
  const int64_t n = e->d.n;
  const size_t nt = (size_t)e->d.ntiles;
  std::vector<double> core(nt * kCorePlanes * kTile, 0.0), lag(nt * kLagPlanes * kTile, 0.0);
  std::vector<float> ou(nt * 4 * kTile, 0.f), cmds(nt * 4 * kTile, 0.f), actp(nt * 4 * kTile, 0.f), goal(nt * 4 * kTile, 0.f),
      gyro(nt * 4 * kTile, 0.f);
  std::vector<uint32_t> c(nt * kTile, 0u);
  for (int64_t i = n; i < (int64_t)(nt * kTile); ++i) { for (int j : {6, 10, 14}) core[tidx(i, kCorePlanes, j)] = 1.0; }
  for (int64_t i = 0; i < n; ++i) {
    for (int pl = 0; pl < kCorePlanes; ++pl) core[tidx(i, kCorePlanes, pl)] = hp[(size_t)pl * n + i];
    for (int j = 0; j < 4; ++j) {
      lag[tidx(i, kLagPlanes, j)] = hp[(size_t)(18 + j) * n + i];
      cmds[tidx(i, 4, j)] = (float)hp[(size_t)(22 + j) * n + i];
      ou[tidx(i, 4, j)] = (float)hp[(size_t)(26 + j) * n + i];
      actp[tidx(i, 4, j)] = (float)hp[(size_t)(30 + j) * n + i];
    }
    for (int j = 0; j < 3; ++j) goal[tidx(i, 4, j)] = (float)hp[(size_t)(34 + j) * n + i];
    for (int j = 0; j < 3; ++j) gyro[tidx(i, 4, j)] = (float)hp[(size_t)(39 + j) * n + i];
    const double t = hp[(size_t)37 * n + i], s = hp[(size_t)38 * n + i];
    if (!(t >= 0 && t <= 65535 && s >= 0 && s <= 65535)) return fail(GAQ_ERR_INVALID, "tick / SVD counter out of range");
    c[i] = ((uint32_t)t & 0xFFFFu) | ((uint32_t)s << 16);
  }
  if (e->alias) {
    std::vector<float> hi(nt * kTile * 18, 0.0f);
    const int mode = alias_mode(e);
    std::vector<unsigned char> lo(nt * (size_t)(e->lomix ? kMixRowsBytes : kLoRowsBytes), 0);
    for (int64_t i = 0; i < n; ++i)
      for (int k = 0; k < 18; ++k) {
        const double v = core[tidx(i, kCorePlanes, k)] - (k < 3 ? (double)goal[tidx(i, 4, k)] : 0.0);
        hi[i * 18 + k] = e->fp32 ? (float)v : split_hi(v);
        lo_encode(mode, lo.data(), i, k, v);
      }
    HIP_TRY(hipMemcpy(e->own_obs, hi.data(), hi.size() * 4, hipMemcpyHostToDevice));
    if (!e->fp32) HIP_TRY(hipMemcpy(e->d.lo, lo.data(), lo.size(), hipMemcpyHostToDevice));
    e->last_obs = e->own_obs;
  } else {
    HIP_TRY(hipMemcpy(e->d.core, core.data(), core.size() * 8, hipMemcpyHostToDevice));
  }
  HIP_TRY(hipMemcpy(e->d.lag, lag.data(), lag.size() * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->d.ou, ou.data(), ou.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->d.cmds, cmds.data(), cmds.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->d.actp, actp.data(), actp.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->d.goal, goal.data(), goal.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->d.gyro, gyro.data(), gyro.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->d.ctr, c.data(), c.size() * 4, hipMemcpyHostToDevice));
  e->rz_refill_now = true;      // the caller may have moved episode clocks: refill the staged parameter planes before the next step
  return GAQ_OK;
}

int gaq_done_list(gaq_env* e, uint32_t* idx_out, int64_t capacity, int64_t* count_out) {
  if (!e || !count_out) return fail(GAQ_ERR_INVALID, "null argument");
  if (!e->d.done_list) return fail(GAQ_ERR_STATE, "handle was created with compact_done = 0");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (int rc_ = sync_handle(e)) return rc_;
  uint64_t step = e->sc.step_index;
  if (e->d.step_ctr) { if (int rc_ = read_step_counter(e, &step)) return rc_; }   // replays advance only this one
  if (step == 0) { *count_out = 0; return GAQ_OK; }
  uint32_t cnt = 0;
  HIP_TRY(hipMemcpy(&cnt, e->d.done_count + ((step - 1) & 1), sizeof(uint32_t), hipMemcpyDeviceToHost));
  *count_out = cnt;
  if (idx_out && cnt) {
    const int64_t m = cnt < capacity ? cnt : capacity;
    HIP_TRY(hipMemcpy(idx_out, e->d.done_list, sizeof(uint32_t) * m, hipMemcpyDeviceToHost));
  }
  return GAQ_OK;
}

int gaq_pack_rows_dev(gaq_env* e, const float* obs, const float* reward, const uint8_t* done, float* rows, void* stream) {
  if (!e || !obs || !reward || !done || !rows) return fail(GAQ_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  e->user_stream = (hipStream_t)stream; e->user_stream_used = true;
  const int64_t total = e->d.n * (e->obs_dim + 2);
  int64_t blocks = (total + kBlock - 1) / kBlock;
  if (blocks > 16384) blocks = 16384;                                      // grid-stride above 4 M words
  hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, e->d.n, e->obs_dim, obs, reward,
                     done, rows);
  HIP_TRY(hipGetLastError());
  return GAQ_OK;
}

int gaq_set_packed_rows_dev(gaq_env* e, float* rows_dev) {
  if (!e) return fail(GAQ_ERR_INVALID, "null handle");
  if (rows_dev && (reinterpret_cast<uintptr_t>(rows_dev) & 15) != 0) return fail(GAQ_ERR_INVALID, "rows must be 16-byte aligned");
  e->d.rows_out = rows_dev;
  return GAQ_OK;
}

int gaq_nan_count(gaq_env* e, int64_t* count_out) {
  if (!e || !count_out) return fail(GAQ_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (int rc_ = sync_handle(e)) return rc_;
  uint32_t c = 0;
  HIP_TRY(hipMemcpy(&c, e->d.nan_count, sizeof(uint32_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(e->d.nan_count, 0, sizeof(uint32_t)));
  *count_out = c;
  return check_overrun(e);
}

int gaq_set_terminal_obs_dev(gaq_env* e, float* term_obs_dev) {
  if (!e) return fail(GAQ_ERR_INVALID, "null handle");
  e->d.term_obs = term_obs_dev;
  return GAQ_OK;
}

int gaq_track_episodes(gaq_env* e, int32_t enabled) {
  if (!e) return fail(GAQ_ERR_INVALID, "null handle");
  HIP_TRY(hipSetDevice(e->cfg.device));
  if (int rc_ = sync_handle(e)) return rc_;
  if (enabled && !e->d.ep_ret) {
    const size_t np = (size_t)e->d.ntiles * kTile;
    HIP_TRY(hipMalloc((void**)&e->d.ep_ret, np * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&e->d.ep_len, np * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void**)&e->d.ep_acc, 4 * sizeof(double)));
    HIP_TRY(hipMemset(e->d.ep_ret, 0, np * sizeof(float)));
    HIP_TRY(hipMemset(e->d.ep_len, 0, np * sizeof(uint32_t)));
    HIP_TRY(hipMemset(e->d.ep_acc, 0, 4 * sizeof(double)));
  } else if (!enabled && e->d.ep_ret) {
    (void)hipFree(e->d.ep_ret); (void)hipFree(e->d.ep_len); (void)hipFree(e->d.ep_acc);
    e->d.ep_ret = nullptr; e->d.ep_len = nullptr; e->d.ep_acc = nullptr;
  }
  return GAQ_OK;
}

int gaq_episode_stats(gaq_env* e, int64_t* episodes, double* return_sum, double* length_sum, double* return_sqsum, int32_t clear) {
  if (!e || !episodes || !return_sum || !length_sum || !return_sqsum) return fail(GAQ_ERR_INVALID, "null argument");
  if (!e->d.ep_acc) return fail(GAQ_ERR_STATE, "episode tracking is off (gaq_track_episodes)");
  HIP_TRY(hipSetDevice(e->cfg.device));
  if (int rc_ = sync_handle(e)) return rc_;
  double a[4];
  HIP_TRY(hipMemcpy(a, e->d.ep_acc, sizeof(a), hipMemcpyDeviceToHost));
  if (clear) HIP_TRY(hipMemset(e->d.ep_acc, 0, sizeof(a)));
  *episodes = (int64_t)(a[0] + 0.5); *return_sum = a[1]; *length_sum = a[2]; *return_sqsum = a[3];
  return GAQ_OK;
}

int gaq_get_counters(gaq_env* e, gaq_counters* out, uint32_t* episodes_out, uint32_t* resamples_out) {
  if (!e || !out) return fail(GAQ_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  if (int rc_ = sync_handle(e)) return rc_;
  if (int rc_ = check_overrun(e)) return rc_;
  out->step_index = e->sc.step_index;
  if (e->d.step_ctr) { if (int rc_ = read_step_counter(e, &out->step_index)) return rc_; }   // graph replays advance only this one
  out->reset_calls = e->reset_calls;
  if ((episodes_out || resamples_out) && !e->d.traj) return fail(GAQ_ERR_STATE, "handle was created with per_env_params = 0: no per-env counts");
  const size_t bytes = sizeof(uint32_t) * (size_t)e->d.n;
  if (episodes_out) HIP_TRY(hipMemcpy(episodes_out, e->d.traj, bytes, hipMemcpyDeviceToHost));
  if (resamples_out) HIP_TRY(hipMemcpy(resamples_out, e->d.rcount, bytes, hipMemcpyDeviceToHost));
  return GAQ_OK;
}

int gaq_set_counters(gaq_env* e, const gaq_counters* in, const uint32_t* episodes, const uint32_t* resamples) {
  if (!e || !in) return fail(GAQ_ERR_INVALID, "null argument");
  if ((episodes || resamples) && !e->d.traj) return fail(GAQ_ERR_STATE, "handle was created with per_env_params = 0: no per-env counts");
  HIP_TRY(hipSetDevice(e->cfg.device));
  if (int rc_ = sync_handle(e)) return rc_;
  e->info_valid = false;
  e->sc.step_index = in->step_index;
  e->reset_calls = in->reset_calls;
  if (e->d.step_ctr) { if (int rc_ = write_step_counter(e, in->step_index)) return rc_; }
  HIP_TRY(hipMemset(e->d.done_count, 0, sizeof(uint32_t) * 2));
  const size_t bytes = sizeof(uint32_t) * (size_t)e->d.n;
  if (episodes) HIP_TRY(hipMemcpy(e->d.traj, episodes, bytes, hipMemcpyHostToDevice));
  if (resamples) {
    HIP_TRY(hipMemcpy(e->d.rcount, resamples, bytes, hipMemcpyHostToDevice));
    if (e->rz_on) {   // the parameters are a function of (seed, global env index, resample count): rebuild them, then the staged ones
      const dim3 grid((unsigned)((e->d.n + kBlock - 1) / kBlock)), block(kBlock);
      hipLaunchKernelGGL(rerandomize_kernel, grid, block, 0, e->stream, e->d, e->sc, e->rz, (const uint8_t*)nullptr, 2, (double*)nullptr,
                         (int64_t)0, (int64_t)0);
      HIP_TRY(hipGetLastError());
      if (int rc = launch_jinv(e, e->stream, nullptr)) return rc;
      if (e->d.rz_every > 0) { if (int rc = launch_refill(e, e->stream)) return rc; }
      HIP_TRY(hipStreamSynchronize(e->stream));
      e->cold_stale = false;      // (mode 2 wrote every env's planes whole)
    }
  }
  e->rz_refill_now = true;
  return GAQ_OK;
}

int gaq_set_graph_safe(gaq_env* e, int32_t enabled) {
  if (!e) return fail(GAQ_ERR_INVALID, "null handle");
  HIP_TRY(hipSetDevice(e->cfg.device));
  if (int rc_ = sync_handle(e)) return rc_;
  if (enabled && !e->d.step_ctr) {
    if (int rc_ = write_step_counter(e, e->sc.step_index)) return rc_;
    e->d.step_ctr = e->step_ctr_mem;
  } else if (!enabled && e->d.step_ctr) {
    uint64_t v = 0;
    if (int rc_ = read_step_counter(e, &v)) return rc_;
    e->sc.step_index = v;
    e->d.step_ctr = nullptr;
  }
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

// ---- one batch over several devices, one process (include/gaq.h: gaq_sharded) --------------------------------------------------
// Shard k is an ordinary gaq_env on device dev[k] holding the global envs [first[k], first[k] + count[k]).  A call records an event on
// the caller's stream (device dev[0]), every REMOTE shard's own stream waits for it, pulls its slice of the inputs over with a peer
// copy, launches its step and pushes its slices of obs / reward / done back; shards that live on dev[0] run on the caller's stream and
// read / write the caller's tensors in place; finally the caller's stream waits for the remote shards' events.  Nothing blocks the host.
struct gaq_sharded {
  struct Shard {
    gaq_env* env = nullptr;
    int64_t first = 0, count = 0;
    int dev = 0;
    bool direct = false;             // lives on the root device: steps on the caller's stream, straight on the caller's tensors
    hipStream_t st = nullptr;        // remote shards: their own stream ...
    hipEvent_t ev = nullptr;         // ... and the event the caller's stream waits for
    float* act = nullptr; float* obs = nullptr; float* rew = nullptr; uint8_t* done = nullptr; uint8_t* mask = nullptr;   // on dev
  };
  std::vector<Shard> sh;
  bool owns = false;
  int root = 0;
  int64_t n = 0;
  int D = 18;
  hipEvent_t start = nullptr;        // on the root device: "the caller's inputs are ready"
  hipStream_t root_stream = nullptr; // host-pointer forms
  char* stage = nullptr;             // [actions 16n | reward 4n | done n | mask n | obs 4 D n] on the root device
  size_t off_rew = 0, off_done = 0, off_mask = 0, off_obs = 0;
};

namespace {

struct DeviceGuard {                 // the caller's current device is the caller's business (torch keeps its own idea of it)
  int prev = -1;
  DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int shard_range_impl(int64_t n, int32_t K, int32_t k, int32_t align, int64_t* first, int64_t* count) {
  if (n <= 0 || K <= 0 || k < 0 || k >= K || align <= 0) return fail(GAQ_ERR_INVALID, "gaq_shard_range: bad argument");
  int64_t a = kTile, b = align;
  while (b) { const int64_t t = a % b; a = b; b = t; }              // gcd
  const int64_t unit = (int64_t)kTile / a * align;                   // whole tiles AND whole worlds
  const int64_t units = (n + unit - 1) / unit;
  const int64_t base = units / K, extra = units % K;
  int64_t f = ((int64_t)k * base + (k < extra ? k : extra)) * unit;
  int64_t l = f + (base + (k < extra ? 1 : 0)) * unit;
  if (f > n) f = n;
  if (l > n) l = n;
  *first = f; *count = l - f;
  return GAQ_OK;
}

void free_sharded(gaq_sharded* s) {
  if (!s) return;
  for (auto& x : s->sh) {
    if (hipSetDevice(x.dev) != hipSuccess) continue;
    if (x.st) { (void)hipStreamSynchronize(x.st); (void)hipStreamDestroy(x.st); }
    if (x.ev) (void)hipEventDestroy(x.ev);
    (void)hipFree(x.act); (void)hipFree(x.obs); (void)hipFree(x.rew); (void)hipFree(x.done); (void)hipFree(x.mask);
    if (s->owns && x.env) (void)gaq_destroy(x.env);
  }
  if (hipSetDevice(s->root) == hipSuccess) {
    if (s->root_stream) { (void)hipStreamSynchronize(s->root_stream); (void)hipStreamDestroy(s->root_stream); }
    if (s->start) (void)hipEventDestroy(s->start);
    (void)hipFree(s->stage);
  }
  delete s;
}

// streams, events and the remote shards' local buffers; s->sh[k].{env, first, count, dev} are filled in
int finish_sharded(gaq_sharded* s) {
  s->root = s->sh[0].dev;
  s->D = s->sh[0].env->obs_dim;
  s->n = 0;
  const bool force_copy = env_override("GAQ_SHARDED_FORCE_COPY") == 1;    // tests on a one-GPU box: every shard takes the remote path
  for (auto& x : s->sh) {
    if (x.env->obs_dim != s->D) return fail(GAQ_ERR_INVALID, "sharded: the shards' observation widths differ");
    if (x.first != s->n) return fail(GAQ_ERR_INVALID, "sharded: shard ranges must be consecutive");
    if ((int64_t)x.env->cfg.env_id_offset != s->sh[0].env->cfg.env_id_offset + x.first)
      return fail(GAQ_ERR_INVALID, "sharded: env_id_offset of every shard must continue the previous shard's global range");
    if (&x != &s->sh.back() && (x.count % kTile) != 0)
      return fail(GAQ_ERR_INVALID, "sharded: every shard but the last must hold a multiple of 64 envs (16-byte aligned slices)");
    s->n += x.count;
    x.direct = (x.dev == s->root) && !force_copy;
  }
  HIP_TRY(hipSetDevice(s->root));
  HIP_TRY(hipEventCreateWithFlags(&s->start, hipEventDisableTiming));
  HIP_TRY(hipStreamCreateWithFlags(&s->root_stream, hipStreamNonBlocking));
  for (auto& x : s->sh) {
    if (x.direct) continue;
    HIP_TRY(hipSetDevice(x.dev));
    if (x.dev != s->root) {            // best effort: with peer access the copies are direct xGMI DMA, without it the runtime stages them
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, x.dev, s->root) == hipSuccess && can) { if (hipDeviceEnablePeerAccess(s->root, 0) != hipSuccess) (void)hipGetLastError(); }
      HIP_TRY(hipSetDevice(s->root));
      if (hipDeviceCanAccessPeer(&can, s->root, x.dev) == hipSuccess && can) { if (hipDeviceEnablePeerAccess(x.dev, 0) != hipSuccess) (void)hipGetLastError(); }
      HIP_TRY(hipSetDevice(x.dev));
    }
    HIP_TRY(hipStreamCreateWithFlags(&x.st, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&x.ev, hipEventDisableTiming));
    const size_t c = (size_t)x.count;
    HIP_TRY(hipMalloc((void**)&x.act, 16 * c));
    HIP_TRY(hipMalloc((void**)&x.obs, 4 * (size_t)s->D * c));
    HIP_TRY(hipMalloc((void**)&x.rew, 4 * c));
    HIP_TRY(hipMalloc((void**)&x.done, c));
    HIP_TRY(hipMalloc((void**)&x.mask, c));
  }
  return GAQ_OK;
}

hipError_t copy_between(void* dst, int dst_dev, const void* src, int src_dev, size_t bytes, hipStream_t st) {
  if (dst_dev == src_dev) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);
  return hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, st);
}

// reset (actions == nullptr) or step of every shard; device pointers on the root device
int fan_out(gaq_sharded* s, const float* actions, const uint8_t* mask, float* obs, float* reward, uint8_t* done, hipStream_t ust) {
  const size_t D = (size_t)s->D;
  bool any_remote = false;
  for (auto& x : s->sh) any_remote = any_remote || !x.direct;
  if (any_remote) {
    HIP_TRY(hipSetDevice(s->root));
    HIP_TRY(hipEventRecord(s->start, ust));
    for (auto& x : s->sh) {
      if (x.direct) continue;
      const size_t f = (size_t)x.first, c = (size_t)x.count;
      HIP_TRY(hipSetDevice(x.dev));
      HIP_TRY(hipStreamWaitEvent(x.st, s->start, 0));
      int rc;
      if (actions) {
        HIP_TRY(copy_between(x.act, x.dev, actions + 4 * f, s->root, 16 * c, x.st));
        rc = gaq_step_dev(x.env, x.act, x.obs, x.rew, x.done, x.st);
      } else {
        if (mask) HIP_TRY(copy_between(x.mask, x.dev, mask + f, s->root, c, x.st));
        rc = gaq_reset_dev(x.env, mask ? x.mask : nullptr, x.obs, x.st);
      }
      if (rc) return rc;
      HIP_TRY(copy_between(obs + D * f, s->root, x.obs, x.dev, 4 * D * c, x.st));
      if (actions) {
        HIP_TRY(copy_between(reward + f, s->root, x.rew, x.dev, 4 * c, x.st));
        HIP_TRY(copy_between(done + f, s->root, x.done, x.dev, c, x.st));
      }
      HIP_TRY(hipEventRecord(x.ev, x.st));
    }
  }
  for (auto& x : s->sh) {              // the root device's own shards: on the caller's stream, in the caller's tensors
    if (!x.direct) continue;
    const size_t f = (size_t)x.first;
    const int rc = actions ? gaq_step_dev(x.env, actions + 4 * f, obs + D * f, reward + f, done + f, ust)
                           : gaq_reset_dev(x.env, mask ? mask + f : nullptr, obs + D * f, ust);
    if (rc) return rc;
  }
  if (any_remote) {
    HIP_TRY(hipSetDevice(s->root));
    for (auto& x : s->sh) if (!x.direct) HIP_TRY(hipStreamWaitEvent(ust, x.ev, 0));
  }
  return GAQ_OK;
}

int need_stage(gaq_sharded* s) {
  if (s->stage) return GAQ_OK;
  const size_t n = (size_t)s->n, D = (size_t)s->D;
  s->off_rew = (16 * n + 15) & ~(size_t)15;
  s->off_done = s->off_rew + 4 * n;
  s->off_mask = (s->off_done + n + 15) & ~(size_t)15;
  s->off_obs = (s->off_mask + n + 15) & ~(size_t)15;
  HIP_TRY(hipSetDevice(s->root));
  HIP_TRY(hipMalloc((void**)&s->stage, s->off_obs + 4 * D * n));
  return GAQ_OK;
}

}  // namespace

extern "C" {

int gaq_hbm_copy_dev(void* dst, const void* src, size_t bytes, void* stream) {
  if (!dst || !src) return fail(GAQ_ERR_INVALID, "null argument");
  if ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src) | bytes) & 15) return fail(GAQ_ERR_INVALID, "gaq_hbm_copy_dev: 16-byte alignment");
  const size_t n16 = bytes / 16;
  if (n16 == 0) return GAQ_OK;
  size_t blocks = (n16 + kBlock - 1) / kBlock;
  if (blocks > ((size_t)1 << 20)) blocks = (size_t)1 << 20;          // (grid-stride beyond 4 GiB)
  hipLaunchKernelGGL(hbm_copy_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, reinterpret_cast<const u32x4*>(src),
                     reinterpret_cast<u32x4*>(dst), n16);
  HIP_TRY(hipGetLastError());
  return GAQ_OK;
}

int gaq_shard_range(int64_t n, int32_t num_shards, int32_t k, int32_t align, int64_t* first, int64_t* count) {
  if (!first || !count) return fail(GAQ_ERR_INVALID, "null argument");
  return shard_range_impl(n, num_shards, k, align, first, count);
}

int gaq_create_sharded(const gaq_config* cfg, const int32_t* device_ids, int32_t num_devices, gaq_sharded** out) {
  if (!cfg || !device_ids || !out) return fail(GAQ_ERR_INVALID, "null argument");
  if (num_devices <= 0 || num_devices > 64) return fail(GAQ_ERR_INVALID, "sharded: num_devices must be in [1, 64]");
  if (cfg->struct_size != sizeof(gaq_config) || cfg->abi_version != GAQ_ABI_VERSION)
    return fail(GAQ_ERR_INVALID, "gaq_config size/version mismatch (header vs library)");
  if (cfg->num_envs <= 0) return fail(GAQ_ERR_INVALID, "num_envs must be positive");
  const int nd = gaq_num_devices();
  if (nd <= 0) return fail(GAQ_ERR_DEVICE, "no HIP device visible: libgaq has no CPU fallback");
  for (int k = 0; k < num_devices; ++k)
    if (device_ids[k] < 0 || device_ids[k] >= nd) return fail(GAQ_ERR_INVALID, "sharded: device id out of range");
  DeviceGuard guard;
  gaq_sharded* s = new (std::nothrow) gaq_sharded();
  if (!s) return fail(GAQ_ERR_DEVICE, "out of host memory");
  s->owns = true;
  const int align = cfg->swarm.agents > 1 ? cfg->swarm.agents : 1;
  for (int k = 0; k < num_devices; ++k) {
    int64_t f = 0, c = 0;
    if (int rc = shard_range_impl(cfg->num_envs, num_devices, k, align, &f, &c)) { free_sharded(s); return rc; }
    if (c == 0) continue;                 // fewer tiles than devices: the tail devices stay idle
    gaq_config sc = *cfg;
    sc.num_envs = c; sc.env_id_offset = cfg->env_id_offset + f; sc.device = device_ids[k];
    gaq_env* e = nullptr;
    if (int rc = gaq_create(&sc, &e)) { free_sharded(s); return rc; }
    gaq_sharded::Shard x;
    x.env = e; x.first = f; x.count = c; x.dev = device_ids[k];
    s->sh.push_back(x);
  }
  if (int rc = finish_sharded(s)) { free_sharded(s); return rc; }
  *out = s;
  return GAQ_OK;
}

int gaq_sharded_from_handles(gaq_env* const* envs, int32_t num_shards, gaq_sharded** out) {
  if (!envs || !out) return fail(GAQ_ERR_INVALID, "null argument");
  if (num_shards <= 0 || num_shards > 64) return fail(GAQ_ERR_INVALID, "sharded: num_shards must be in [1, 64]");
  DeviceGuard guard;
  gaq_sharded* s = new (std::nothrow) gaq_sharded();
  if (!s) return fail(GAQ_ERR_DEVICE, "out of host memory");
  s->owns = false;
  int64_t f = 0;
  for (int k = 0; k < num_shards; ++k) {
    if (!envs[k]) { free_sharded(s); return fail(GAQ_ERR_INVALID, "null shard handle"); }
    gaq_sharded::Shard x;
    x.env = envs[k]; x.first = f; x.count = envs[k]->d.n; x.dev = envs[k]->cfg.device;
    f += x.count;
    s->sh.push_back(x);
  }
  if (int rc = finish_sharded(s)) { free_sharded(s); return rc; }
  *out = s;
  return GAQ_OK;
}

int gaq_destroy_sharded(gaq_sharded* s) {
  if (!s) return GAQ_OK;
  DeviceGuard guard;
  free_sharded(s);
  return GAQ_OK;
}

int gaq_sharded_num_shards(const gaq_sharded* s) { return s ? (int)s->sh.size() : GAQ_ERR_INVALID; }
int64_t gaq_sharded_num_envs(const gaq_sharded* s) { return s ? s->n : 0; }
gaq_env* gaq_sharded_shard(gaq_sharded* s, int32_t k) { return (s && k >= 0 && k < (int32_t)s->sh.size()) ? s->sh[k].env : nullptr; }
int gaq_sharded_range(const gaq_sharded* s, int32_t k, int64_t* first, int64_t* count, int32_t* device) {
  if (!s || k < 0 || k >= (int32_t)s->sh.size()) return fail(GAQ_ERR_INVALID, "sharded: no such shard");
  if (first) *first = s->sh[k].first;
  if (count) *count = s->sh[k].count;
  if (device) *device = s->sh[k].dev;
  return GAQ_OK;
}

int gaq_step_sharded_dev(gaq_sharded* s, const float* actions, float* obs, float* reward, uint8_t* done, void* stream) {
  if (!s || !actions || !obs || !reward || !done) return fail(GAQ_ERR_INVALID, "null argument");
  DeviceGuard guard;
  return fan_out(s, actions, nullptr, obs, reward, done, (hipStream_t)stream);
}

int gaq_reset_sharded_dev(gaq_sharded* s, const uint8_t* mask_dev, float* obs, void* stream) {
  if (!s || !obs) return fail(GAQ_ERR_INVALID, "null argument");
  DeviceGuard guard;
  return fan_out(s, nullptr, mask_dev, obs, nullptr, nullptr, (hipStream_t)stream);
}

int gaq_synchronize_sharded(gaq_sharded* s) {
  if (!s) return fail(GAQ_ERR_INVALID, "null handle");
  DeviceGuard guard;
  for (auto& x : s->sh) {
    HIP_TRY(hipSetDevice(x.dev));
    if (x.st) HIP_TRY(hipStreamSynchronize(x.st));
    if (int rc = sync_handle(x.env)) return rc;
  }
  HIP_TRY(hipSetDevice(s->root));
  HIP_TRY(hipStreamSynchronize(s->root_stream));
  return GAQ_OK;
}

int gaq_reset_sharded(gaq_sharded* s, const uint8_t* mask, float* obs_out) {
  if (!s || !obs_out) return fail(GAQ_ERR_INVALID, "null argument");
  DeviceGuard guard;
  if (int rc = gaq_synchronize_sharded(s)) return rc;
  if (int rc = need_stage(s)) return rc;
  const size_t n = (size_t)s->n, D = (size_t)s->D;
  HIP_TRY(hipSetDevice(s->root));
  if (mask) HIP_TRY(hipMemcpyAsync(s->stage + s->off_mask, mask, n, hipMemcpyHostToDevice, s->root_stream));
  if (int rc = fan_out(s, nullptr, mask ? reinterpret_cast<const uint8_t*>(s->stage + s->off_mask) : nullptr,
                       reinterpret_cast<float*>(s->stage + s->off_obs), nullptr, nullptr, s->root_stream)) return rc;
  HIP_TRY(hipSetDevice(s->root));
  HIP_TRY(hipMemcpyAsync(obs_out, s->stage + s->off_obs, 4 * D * n, hipMemcpyDeviceToHost, s->root_stream));
  HIP_TRY(hipStreamSynchronize(s->root_stream));
  return GAQ_OK;
}

int gaq_step_sharded(gaq_sharded* s, const float* actions, float* obs, float* reward, uint8_t* done) {
  if (!s || !actions || !obs || !reward || !done) return fail(GAQ_ERR_INVALID, "null argument");
  DeviceGuard guard;
  if (int rc = gaq_synchronize_sharded(s)) return rc;
  if (int rc = need_stage(s)) return rc;
  const size_t n = (size_t)s->n, D = (size_t)s->D;
  HIP_TRY(hipSetDevice(s->root));
  HIP_TRY(hipMemcpyAsync(s->stage, actions, 16 * n, hipMemcpyHostToDevice, s->root_stream));
  if (int rc = fan_out(s, reinterpret_cast<const float*>(s->stage), nullptr, reinterpret_cast<float*>(s->stage + s->off_obs),
                       reinterpret_cast<float*>(s->stage + s->off_rew), reinterpret_cast<uint8_t*>(s->stage + s->off_done), s->root_stream)) return rc;
  HIP_TRY(hipSetDevice(s->root));
  HIP_TRY(hipMemcpyAsync(obs, s->stage + s->off_obs, 4 * D * n, hipMemcpyDeviceToHost, s->root_stream));
  HIP_TRY(hipMemcpyAsync(reward, s->stage + s->off_rew, 4 * n, hipMemcpyDeviceToHost, s->root_stream));
  HIP_TRY(hipMemcpyAsync(done, s->stage + s->off_done, n, hipMemcpyDeviceToHost, s->root_stream));
  HIP_TRY(hipStreamSynchronize(s->root_stream));
  for (auto& x : s->sh) if (int rc = check_overrun(x.env)) return rc;
  // the reference raises on a non-finite reward inside step() (quadrotor.py:633-636)
  for (size_t i = 0; i < n; ++i) {
    if (!std::isfinite(reward[i])) {
      for (auto& x : s->sh) { HIP_TRY(hipSetDevice(x.dev)); HIP_TRY(hipMemset(x.env->d.nan_count, 0, sizeof(uint32_t))); }
      return fail(GAQ_ERR_NAN, "QuadEnv: reward is Nan");
    }
  }
  return GAQ_OK;
}

}  // extern "C"
