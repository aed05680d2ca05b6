// gaq_kernels.hpp -- the templated device code of libgaq: layout constants, the per-wave LDS image, the fused step kernel and the fused
// T-step rollout kernel.  Header-only (templates and force-inlined helpers) so that the kernel instantiations can be compiled in
// several translation units side by side (gaq_inst.hip, -DGAQ_PART=k) while gaq.hip holds the C ABI, the launch logic and the small
// non-template kernels.  The lists of instantiations are at the end of this file.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/gaq.h"
#include "quad_core.hpp"
#include "quad_params_dev.hpp"

namespace gaqk {

// Cache-policy bits (the `aux` immediate of the buffer builtins: 1 = sc0, 2 = nt, 16 = sc1) of the step kernels' streaming
// traffic -- every state byte is loaded once and stored once per launch.  Build-time knobs so that the policies can be
// A/B-measured (tools/aux_variants.sh); the defaults are what measured best.
#ifndef GAQ_LD_AUX
#define GAQ_LD_AUX 0      // HBM -> LDS DMA loads of the state image
#endif
#ifndef GAQ_ST_AUX
#define GAQ_ST_AUX 0      // LDS -> HBM stores of the new state / observation rows
#endif
#ifndef GAQ_ACT_AUX
#define GAQ_ACT_AUX 0     // the action tile and the counter word (read once)
#endif
// ... and what measured best (tools/aux_variants.sh, profiles/r02_v4_aux_policy_ab.txt): at N = 2^20 sc1 stores are worth
// 1.5 % on one kernel and cost 0.5-1 % on the others, nt costs 2-6 %: the large-batch kernels keep the default policy.  At
// one or two waves per SIMD non-temporal loads AND stores (the F_NT instantiations) take 3 % (65 536 envs) to 9-15 % (131 072)
// off the step -- the launch does not leave its whole output as dirty lines for the kernel boundary to write back.
template <uint32_t F> constexpr int kLdAux = (F & gaq::F_NT) ? 2 : GAQ_LD_AUX;
template <uint32_t F> constexpr int kStAux = (F & gaq::F_NT) ? 2 : GAQ_ST_AUX;
// ... and of the caller's COPY of the observation rows in the library-owned-heads layout: written once, never read by the library
// (non-temporal: measured -- the per-env CrazyFlie with per-episode re-randomisation 121 -> 107 us per step, the default configuration 60.8 -> 58.9,
//  Mellinger 65 -> 62-66; sc1 changes nothing: profiles/r04_obs_copy_policy_ab.txt)
#ifndef GAQ_COPY_AUX
#define GAQ_COPY_AUX 2
#endif
template <uint32_t F> constexpr int kCopyAux = (F & gaq::F_NT) ? 2 : GAQ_COPY_AUX;
// Timing-only ablations (GAQ_ABLATE bits: 1 skip the arithmetic, 2 skip the promotion's plane copy, 4 skip the whole promotion;
// tools/latency_breakdown.py, tools/rz_ablate.sh) give WRONG physics by construction.  They exist only in a measurement build
// (make EXTRA=-DGAQ_DIAG_BUILD OUT=...): in the product library the tests below fold to `false` at compile time and gaq_create
// refuses a non-zero GAQ_ABLATE, so a stray environment variable can never silently change the results.
#ifdef GAQ_DIAG_BUILD
constexpr bool kDiagBuild = true;
#else
constexpr bool kDiagBuild = false;
#endif
__host__ __device__ __forceinline__ bool ablated(const gaq::StepCfg& cfg, int bit) { return kDiagBuild && (cfg.ablate & bit) != 0; }
constexpr int kBlock = 256;                 // 4 wavefronts = 4 tiles per workgroup (no block-level sync anywhere)
constexpr int kTile = 64;
constexpr int kCorePlanes = 18;             // pos3 vel3 rot9 omega3 (fp64)
constexpr int kLagPlanes = 4;               // thrust_rot_damp (fp64)
constexpr int kCoreBytes = kCorePlanes * kTile * 8;   // 9216
constexpr int kLagBytes = kLagPlanes * kTile * 8;     // 2048
constexpr int kGrpBytes = 4 * kTile * 4;              // 1024: one group of four fp32 planes
constexpr int kRowBytes = 18 * 4;                     // 72: one env's 18-word observation / residual row
constexpr int kRowsBytes = kTile * kRowBytes;         // 4608: a tile's rows (4.5 KiB)
constexpr int kRowsLds = 5 * 1024;                    // LDS reserved for the hi rows: the 5th 1-KiB piece is half used
constexpr int kLoRowBytes = 18 * 2;                   // 36: 16 extra mantissa bits per value (see split_decode)
constexpr int kLoRowsBytes = kTile * kLoRowBytes;     // 2304 (2.25 KiB)
constexpr int kLoRowsLds = 3 * 1024;
// Width of the residual field.  16 bits (39 significant bits per value) is enough for models without motor lag:
// 4096 full-scale random Hummingbird episodes stay within 2.4e-7 of the fp64 planes (tools/alias_drift.py).  With
// motor lag the up/down time-constant choice (quadrotor.py:287-293) is a comparison of nearly equal numbers, a
// 2^-39 perturbation of what feeds it flips it now and then and the trajectories part macroscopically (0.15 % of
// CrazyFlie episodes off by > 1e-5 with 16-bit residuals everywhere).  What feeds it is the CLOSED rotational subsystem
// {motor filter, omega} (thrusts -> torque -> Euler's equations -> omega; R, vel and pos only integrate its output and
// never feed back without rotor drag, which needs the generic kernel).  So every kernel that can see lag -- F_LAG, and
// F_PER_ENV whose parameters may bring it -- keeps omega EXACT (32 residual bits: fp32 head + 29 bits = the whole fp64
// mantissa; thrust_rot_damp is an fp64 plane anyway) and pos / vel / R with 16 residual bits: the "mixed" row of 11
// words = [15 x int16 + pad | 3 x u32] = 44 B instead of 72 B for 18 x u32 (round 1), -56 B/env-step of traffic.
constexpr int kMixRowWords = 11;
constexpr int kMixRowBytes = kMixRowWords * 4;        // 44
constexpr int kMixRowsBytes = kTile * kMixRowBytes;   // 2816 (2.75 KiB: three 1-KiB pieces, like the 16-bit rows)
template <uint32_t F> constexpr bool kLoMix = (F & (gaq::F_PER_ENV | gaq::F_LAG)) != 0;
constexpr int kPar = 45;                    // fp64 per-env parameter planes (37 model planes + 5 construction hints + 1 flag + 2 raw time constants)
constexpr int kParBytes = kPar * kTile * 8;
constexpr int kHotPlanes = 19;              // what a promotion moves on the compact path: planes 1-4, 8-12, 28-31, 36-41 (see the step kernel's epilogue)
constexpr int kParNextSkew = 544;           // doubles between the end of par and par_next (4352 B): a promoted env's source and destination
                                            // words do not sit a round multiple of the channel interleave apart
enum ParPlane { PP_MASS = 0, PP_INV_MASS = 1, PP_INERTIA = 2, PP_INV_INERTIA = 5, PP_THRUST_MAX = 8, PP_TORQUE_MAX = 12,
                PP_PROP_X = 16, PP_PROP_Y = 20, PP_PROP_Z = 24, PP_TAU_UP = 28, PP_TAU_DOWN = 29, PP_LINEARITY = 30,
                PP_ARM = 31, PP_VEL_DAMP = 32, PP_DAMP_Q = 33, PP_C_DRAG = 34, PP_C_ROLL = 35, PP_OU_SIGMA = 36,
                // construction hints found by gaq_set_params (bit-exact or absent): torque_max = t2t * thrust_max (quadrotor.py:176),
                // prop_pos.xy = (+-mx - comx, +-my - comy) (inertia.py:240,307); PP_COMPACT_OK is host-only
                PP_T2T = 37, PP_MX = 38, PP_MY = 39, PP_COMX = 40, PP_COMY = 41, PP_COMPACT_OK = 42,
                // the motor time constants as given (the kernels read tau = 4 dt / (T + 1e-6)); read back by gaq_get_params
                PP_T_UP = 43, PP_T_DOWN = 44 };

struct DevPtrs {
  double* core;      // [ntiles][18][64]   (not allocated in alias mode)
  void* lo;          // alias mode: residual rows, [ntiles*64][18] int16 or [ntiles*64][11] words (mixed, see kLoMix); value = obs word + decode(lo)
  const float* obs_in;  // alias mode: the observation tensor written by the previous step / reset
  float* obs_copy;      // alias mode 2 ("shadow": the library owns the state heads): the caller's observation tensor, which
                        // receives a copy of the new heads; nullptr otherwise
  float* hi_final;      // fused rollout in alias mode 2: where the final state heads go (the library's own rows)
  double* lag;       // [ntiles][4][64]   thrust_rot_damp
  float* ou;         // [ntiles][4][64]   OU noise state
  float* cmds;       // [ntiles][4][64]   thrust_cmds_damp
  float* actp;       // [ntiles][4][64]   previous action
  float* goal;       // [ntiles][4][64]   goal xyz (+1 unused plane)
  float* gyro;       // [ntiles][4][64]   SensorNoise.gyro_bias xyz (+1 unused plane)
  uint32_t* ctr;     // [ntiles*64]       tick | svd_ctr << 16
  const double* par; // [ntiles][43][64] or nullptr
  const double* jinv;     // [n][16] per-env inverse jacobians (Mellinger with per-env models) or nullptr
  const float* noise_in;  // [sim_steps][4][n] or nullptr
  const float* sense_in;  // [3][12][3][n] recorded sensor-noise draws of the next step (gaq_config.sense_input) or nullptr
  float* aux;             // [n][GAQ_AUX_WORDS] info-dict extras of the last step (gaq_config.aux_outputs) or nullptr
  uint32_t* done_list;    // [ntiles*64] or nullptr
  uint32_t* done_count;   // [2] (ping-pong by step parity)
  uint32_t* nan_count;    // [1]
  float* term_obs;        // [n][obs_dim] or nullptr: terminal observations of auto-reset envs
  float* ep_ret;          // [ntiles*64] running episode return (episode tracking) or nullptr
  uint32_t* ep_len;       // [ntiles*64] running episode length
  double* ep_acc;         // [4]: finished episodes, sum of returns, sum of lengths, sum of squared returns
  uint64_t* step_ctr;     // device-resident step counter ([kCtrSlots] words, one per cache line) or nullptr (gaq_set_graph_safe): step index =
                          // (sum of the words) >> ctr_shift; the low bits count the waves of a RUNNING F_CTR step launch that have checked in.
                          // Kernels without F_CTR read the first word alone: the host folds the others into it before it launches one
  uint32_t ctr_shift;     // log2 of the counter units per step (the step launch's wave count rounded up to a power of two)
  uint32_t ctr_inc0;      // what the launch's first wave adds (the others add 1): 2^ctr_shift - (waves - 1): one launch adds 2^ctr_shift
  float* rows_out;        // [n][obs_dim + 2] packed [obs | reward | (float) done] rows of the multi-GPU return path (gaq_set_packed_rows_dev) or
                          // nullptr: written by the step launch itself in the F_ROWS instantiations, by pack_rows_kernel otherwise
  uint32_t* rcount;       // [ntiles*64] per-env resample count (key of the device-side parameter sampler) or nullptr
  uint32_t* traj;         // [ntiles*64] per-env finished-episode count (dynamics_randomize_every) or nullptr; traj, rcount and rz_flag
                          // are consecutive thirds of ONE allocation: the step kernels reach all three through one buffer resource
  double* par_next;       // per-episode re-randomisation: the NEXT draw of every env as one row [45] per env, derived off the critical path
                          // (= par + ntiles * kPar * 64: the second half of one allocation, so that the step kernels need no pointer for it)
  uint32_t* rz_flag;      // [ntiles*64] promotions since the last refill: != 0 = env consumed its staged planes, par_next has to be refilled
  uint32_t* rz_overrun;   // [1] envs promoted twice between two refill passes (must stay 0; checked by gaq_nan_count)
  int32_t rz_every;       // dynamics_randomize_every handled by the step launch (0 = off)
  int64_t n, ntiles;
};

using gaq::EnvState;
using gaq::Model;
using gaq::StepCfg;

// arithmetic / state type of a kernel instantiation: fp64 (the parity path) or, with F_FP32, fp32 throughout
template <uint32_t F> struct RealOf { using type = double; };
#define GAQ_REAL_FP32(FEAT) template <> struct RealOf<(FEAT)> { using type = float; };
GAQ_REAL_FP32(48u) GAQ_REAL_FP32(49u) GAQ_REAL_FP32(50u) GAQ_REAL_FP32(51u) GAQ_REAL_FP32(52u) GAQ_REAL_FP32(53u) GAQ_REAL_FP32(54u) GAQ_REAL_FP32(55u)
#undef GAQ_REAL_FP32
template <uint32_t F> using Real = typename RealOf<F>::type;

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

// ---- per-wave LDS image of a tile -----------------------------------------------------------------------
// core @0; then, when present, lag | ou | cmds | actp | goal | gyro.  For the specialised kernels the offsets
// are compile-time constants; the generic kernel computes them from its (wave-uniform) flags.
struct TileImage { int lo, lag, ou, cmds, actp, goal, gyro, total; };

template <uint32_t F>
__host__ __device__ __forceinline__ TileImage tile_image(const StepCfg& cfg) {
  TileImage t;
  t.lo = kRowsLds;
  int o = (F & gaq::F_FP32) ? kRowsLds : (F & gaq::F_ALIAS) ? kRowsLds + kLoRowsLds : kCoreBytes;   // alias: hi rows @0, lo rows @kRowsLds
  t.lag = o;  if (gaq::has_lag<F>(cfg)) o += kLagBytes;
  t.ou = o;   if (gaq::noise_mode<F>(cfg) != gaq::NOISE_OFF) o += kGrpBytes;
  t.cmds = o; if (gaq::has_lag<F>(cfg)) o += kGrpBytes;
  t.actp = o; if (gaq::has_act_prev<F>(cfg)) o += kGrpBytes;
  t.goal = o; if (gaq::has_env_goal<F>(cfg)) o += kGrpBytes;
  t.gyro = o; if (gaq::has_gyro_bias<F>(cfg)) o += kGrpBytes;
  t.total = o;
  return t;
}

// HBM -> LDS: `pieces` contiguous KiB of a tile section, 16 B per lane per piece, no VGPR staging.
// `g` and `l` are wave-uniform; every lane of the wave must be active.
template <int PIECES, int AUX = GAQ_LD_AUX>
__device__ __forceinline__ void dma_in(const void* g, char* l, uint32_t lane) {
  auto r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g), 0, PIECES * 1024, 0x00020000);
#pragma unroll
  for (int k = 0; k < PIECES; ++k)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(l + k * 1024), 16, lane * 16u, k * 1024, 0, AUX);
}
// LDS -> HBM, the mirror image.
template <int PIECES, int AUX = GAQ_ST_AUX>
__device__ __forceinline__ void copy_out(void* g, const char* l, uint32_t lane) {
  auto r = __builtin_amdgcn_make_buffer_rsrc(g, 0, PIECES * 1024, 0x00020000);
#pragma unroll
  for (int k = 0; k < PIECES; ++k) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(l + k * 1024 + lane * 16u);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, lane * 16u, k * 1024, AUX);
  }
}

// The 18 fp32 heads of a new state = the fp64 values TRUNCATED toward zero (split_hi).  quad_core.hpp's portable form rounds to nearest and
// steps one ulp back when that rounded away (5 instructions per value); here the conversion itself rounds toward zero: MODE.FP_ROUND's
// single-precision field is set for nine v_cvt_f32_f64 at a time inside ONE asm statement (the compiler cannot move an fp32 operation of
// its own between the two s_setreg), 1 instruction per value + 4 scalar ones per 18.  Bit-identical to split_hi (tools/rtz_check.hip: 2^20
// values incl. zeros, exactly representable ones and float denormals, on the device).
// (measured, tools/rtz_check on the GPU box: of 2^20 values 510 269 truncate differently from round-to-nearest; MODE[1:0] = 3 reproduces
//  split_hi on every one of them, MODE[3:2] -- the double / half field -- has no effect on this conversion, and conversions after the
//  restore round to nearest again: profiles/r04_rtz_check.json)
#ifndef GAQ_RTZ_HEADS
#define GAQ_RTZ_HEADS 1
#endif
__device__ __forceinline__ void heads9_rtz(const double* v, float* h) {
  asm volatile(
      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
      "v_cvt_f32_f64 %0, %9\n\tv_cvt_f32_f64 %1, %10\n\tv_cvt_f32_f64 %2, %11\n\tv_cvt_f32_f64 %3, %12\n\tv_cvt_f32_f64 %4, %13\n\t"
      "v_cvt_f32_f64 %5, %14\n\tv_cvt_f32_f64 %6, %15\n\tv_cvt_f32_f64 %7, %16\n\tv_cvt_f32_f64 %8, %17\n\t"
      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
      : "=&v"(h[0]), "=&v"(h[1]), "=&v"(h[2]), "=&v"(h[3]), "=&v"(h[4]), "=&v"(h[5]), "=&v"(h[6]), "=&v"(h[7]), "=&v"(h[8])
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]));
}
// RTZ: the kernels whose observation IS the heads (one wave per SIMD at small batches: 1.6-2.2 % of a 65 536- / 131 072-env step; neutral at
// 2^20).  The packed-observation kernels keep the portable form: the asm statements move their register allocation the wrong way
// (sensor noise <1044>: 66.8 -> 70.8 us with RTZ; profiles/r04_rtz_heads_ab.txt).
template <bool RTZ>
__device__ __forceinline__ void heads18(const double* v, float* h) {
  if constexpr (RTZ && GAQ_RTZ_HEADS) {
    heads9_rtz(v, h); heads9_rtz(v + 9, h + 9);
  } else {
#pragma unroll
    for (int k = 0; k < 18; ++k) h[k] = gaq::split_hi(v[k]);
  }
}

using gaq::split_decode; using gaq::split_hi; using gaq::split_lo; using gaq::split_decode32; using gaq::split_lo32;

// value k (0..17) of env i out of / into the split representation, by state-encoding mode (alias_mode() below):
// 1 = 16-bit residual rows [n][18] int16, 2 = fp32 rows are the whole state, 3 = mixed rows [n][11] words (kLoMix)
__host__ __device__ __forceinline__ double lo_decode(int mode, const float* hi, const void* lo, int64_t i, int k) {
  const float h = hi[i * 18 + k];
  if (mode == 2) return (double)h;
  if (mode == 3) {
    const uint32_t* row = reinterpret_cast<const uint32_t*>(lo) + i * kMixRowWords;
    if (k >= 15) return split_decode32(h, row[8 + (k - 15)]);
    return split_decode(h, row[k >> 1] >> ((k & 1) * 16));
  }
  return split_decode(h, (uint32_t)reinterpret_cast<const uint16_t*>(lo)[i * 18 + k]);
}
__host__ __device__ __forceinline__ void lo_encode(int mode, void* lo, int64_t i, int k, double v) {
  if (mode == 3) {
    uint32_t* row = reinterpret_cast<uint32_t*>(lo) + i * kMixRowWords;
    if (k >= 15) row[8 + (k - 15)] = split_lo32(v);
    else reinterpret_cast<uint16_t*>(row)[k] = (uint16_t)split_lo(v);
  } else if (mode == 1) {
    reinterpret_cast<uint16_t*>(lo)[i * 18 + k] = (uint16_t)split_lo(v);
  }
}

// a tile's [64][18] fp32 rows (4608 B = 4.5 KiB): the same 16-B/lane pieces, bounded by `nbytes` so that the
// half-used 5th piece and the rows of padding envs are dropped by the buffer range check.
template <int PIECES, int AUX = GAQ_LD_AUX>
__device__ __forceinline__ void dma_in_rows(const void* g, char* l, uint32_t lane, uint32_t nbytes) {
  auto r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g), 0, (int)nbytes, 0x00020000);
#pragma unroll
  for (int k = 0; k < PIECES; ++k)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(l + k * 1024), 16, lane * 16u, k * 1024, 0, AUX);
}
template <int PIECES, int TILE_BYTES, int AUX = GAQ_ST_AUX>
__device__ __forceinline__ void copy_out_rows(void* g, const char* l, uint32_t lane, uint32_t nbytes) {
  auto r = __builtin_amdgcn_make_buffer_rsrc(g, 0, (int)nbytes, 0x00020000);
#pragma unroll
  for (int k = 0; k < PIECES; ++k) {
    const uint32_t off = k * 1024 + lane * 16u;
    if (off < (uint32_t)TILE_BYTES) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(l + off);
      __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, AUX);
    }
  }
}

template <uint32_t F>
__device__ __forceinline__ void stage_in(const DevPtrs& p, const StepCfg& cfg, int64_t tile, char* buf, uint32_t lane) {
  const TileImage im = tile_image<F>(cfg);
  if constexpr ((F & gaq::F_ALIAS) != 0) {
    const int64_t first = tile * kTile;
    const uint32_t live = (uint32_t)((p.n - first) < kTile ? (p.n - first) : kTile);
    dma_in_rows<5, kLdAux<F>>(p.obs_in + first * 18, buf, lane, live * kRowBytes);     // hi: the caller's observation rows
    if constexpr ((F & gaq::F_FP32) == 0)
      {
        if constexpr (kLoMix<F>) dma_in_rows<3, kLdAux<F>>(reinterpret_cast<const uint32_t*>(p.lo) + first * kMixRowWords, buf + kRowsLds, lane, kMixRowsBytes);
        else dma_in_rows<3, kLdAux<F>>(reinterpret_cast<const int16_t*>(p.lo) + first * 18, buf + kRowsLds, lane, kLoRowsBytes);  // residual rows
      }
  } else {
    dma_in<9, kLdAux<F>>(p.core + tile * (kCorePlanes * kTile), buf, lane);
  }
  if (gaq::has_lag<F>(cfg)) {
    dma_in<2, kLdAux<F>>(p.lag + tile * (kLagPlanes * kTile), buf + im.lag, lane);
    dma_in<1, kLdAux<F>>(p.cmds + tile * (4 * kTile), buf + im.cmds, lane);
  }
  if (gaq::noise_mode<F>(cfg) != gaq::NOISE_OFF) dma_in<1, kLdAux<F>>(p.ou + tile * (4 * kTile), buf + im.ou, lane);
  if (gaq::has_act_prev<F>(cfg)) dma_in<1, kLdAux<F>>(p.actp + tile * (4 * kTile), buf + im.actp, lane);
  if (gaq::has_env_goal<F>(cfg)) dma_in<1, kLdAux<F>>(p.goal + tile * (4 * kTile), buf + im.goal, lane);
  if (gaq::has_gyro_bias<F>(cfg)) dma_in<1, kLdAux<F>>(p.gyro + tile * (4 * kTile), buf + im.gyro, lane);
}

// each lane reads its own env out of the LDS image (stride-1 across lanes: conflict-free)
template <uint32_t F>
__device__ __forceinline__ void read_image(const StepCfg& cfg, const char* buf, uint32_t lane, EnvState<Real<F>>& s) {
  using T = Real<F>;
  const TileImage im = tile_image<F>(cfg);
  if constexpr ((F & gaq::F_FP32) != 0) {
    // fp32 mode: the 18 observation words are the state itself
    const float2* h = reinterpret_cast<const float2*>(buf + lane * kRowBytes);
    float v[18];
#pragma unroll
    for (int k = 0; k < 9; ++k) { const float2 a = h[k]; v[2 * k] = a.x; v[2 * k + 1] = a.y; }
#pragma unroll
    for (int j = 0; j < 3; ++j) { s.pos[j] = v[j] + (float)cfg.goal_default[j]; s.vel[j] = v[3 + j]; s.omega[j] = v[15 + j]; }
#pragma unroll
    for (int j = 0; j < 9; ++j) s.rot[j] = v[6 + j];
  } else if constexpr ((F & gaq::F_ALIAS) != 0) {
    // row-major rows, 72-B stride: 9 x ds_read_b64 per row block, conflict-free (18 l mod 64 hits every even bank once)
    const float2* h = reinterpret_cast<const float2*>(buf + lane * kRowBytes);
    double gl[3] = {cfg.goal_default[0], cfg.goal_default[1], cfg.goal_default[2]};    // the heads hold pos - goal
    if constexpr ((F & (gaq::F_SWARM | gaq::F_ENVX)) != 0) {                           // ... the agent's own formation goal / this env's goal
      if (gaq::has_env_goal<F>(cfg)) {
        const float* g = reinterpret_cast<const float*>(buf + im.goal) + lane;
#pragma unroll
        for (int j = 0; j < 3; ++j) gl[j] = (double)g[j * kTile];
      }
    }
    double v[18];
    if constexpr (kLoMix<F>) {
      // mixed rows, 11-word stride (odd: conflict-free): words 0-7 = sixteen int16 (15 used), words 8-10 = omega's 32 bits
      const uint32_t* q = reinterpret_cast<const uint32_t*>(buf + kRowsLds + lane * kMixRowBytes);
      float hv[18];
#pragma unroll
      for (int k = 0; k < 9; ++k) { const float2 a = h[k]; hv[2 * k] = a.x; hv[2 * k + 1] = a.y; }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t w = q[k];
        v[2 * k] = split_decode(hv[2 * k], w);
        if (2 * k + 1 < 15) v[2 * k + 1] = split_decode(hv[2 * k + 1], w >> 16);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) v[15 + j] = split_decode32(hv[15 + j], q[8 + j]);
    } else {
      const uint32_t* q = reinterpret_cast<const uint32_t*>(buf + kRowsLds + lane * kLoRowBytes);   // 9-word stride: conflict-free
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const float2 a = h[k];
        const uint32_t w = q[k];
        v[2 * k] = split_decode(a.x, w);
        v[2 * k + 1] = split_decode(a.y, w >> 16);
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) { s.pos[j] = v[j] + gl[j]; s.vel[j] = v[3 + j]; s.omega[j] = v[15 + j]; }
#pragma unroll
    for (int j = 0; j < 9; ++j) s.rot[j] = v[6 + j];
  } else {
    const double* c = reinterpret_cast<const double*>(buf) + lane;
#pragma unroll
    for (int j = 0; j < 3; ++j) s.pos[j] = c[(0 + j) * kTile];
#pragma unroll
    for (int j = 0; j < 3; ++j) s.vel[j] = c[(3 + j) * kTile];
#pragma unroll
    for (int j = 0; j < 9; ++j) s.rot[j] = c[(6 + j) * kTile];
#pragma unroll
    for (int j = 0; j < 3; ++j) s.omega[j] = c[(15 + j) * kTile];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { s.rot_damp[j] = T(0); s.cmds_damp[j] = 0.0f; s.ou[j] = 0.0f; s.act_prev[j] = 0.0f; }
#pragma unroll
  for (int j = 0; j < 3; ++j) { s.goal[j] = T(cfg.goal_default[j]); s.gyro_bias[j] = 0.0f; }
  if (gaq::has_gyro_bias<F>(cfg)) {
    const float* g = reinterpret_cast<const float*>(buf + im.gyro) + lane;
#pragma unroll
    for (int j = 0; j < 3; ++j) s.gyro_bias[j] = g[j * kTile];
  }
  if (gaq::has_lag<F>(cfg)) {
    const double* l = reinterpret_cast<const double*>(buf + im.lag) + lane;
    const float* m = reinterpret_cast<const float*>(buf + im.cmds) + lane;
#pragma unroll
    for (int j = 0; j < 4; ++j) { s.rot_damp[j] = T(l[j * kTile]); s.cmds_damp[j] = m[j * kTile]; }
  }
  if (gaq::noise_mode<F>(cfg) != gaq::NOISE_OFF) {
    const float* o = reinterpret_cast<const float*>(buf + im.ou) + lane;
#pragma unroll
    for (int j = 0; j < 4; ++j) s.ou[j] = o[j * kTile];
  }
  if (gaq::has_act_prev<F>(cfg)) {
    const float* a = reinterpret_cast<const float*>(buf + im.actp) + lane;
#pragma unroll
    for (int j = 0; j < 4; ++j) s.act_prev[j] = a[j * kTile];
  }
  if (gaq::has_env_goal<F>(cfg)) {
    const float* g = reinterpret_cast<const float*>(buf + im.goal) + lane;
#pragma unroll
    for (int j = 0; j < 3; ++j) s.goal[j] = T(g[j * kTile]);
  }
}

template <uint32_t F>
__device__ __forceinline__ void write_image(const StepCfg& cfg, char* buf, uint32_t lane, const EnvState<Real<F>>& s) {
  const TileImage im = tile_image<F>(cfg);
  if constexpr ((F & gaq::F_FP32) != 0) {
    float v[18];
#pragma unroll
    for (int j = 0; j < 3; ++j) { v[j] = s.pos[j] - (float)cfg.goal_default[j]; v[3 + j] = s.vel[j]; v[15 + j] = s.omega[j]; }
#pragma unroll
    for (int j = 0; j < 9; ++j) v[6 + j] = s.rot[j];
    float2* h = reinterpret_cast<float2*>(buf + lane * kRowBytes);
#pragma unroll
    for (int k = 0; k < 9; ++k) h[k] = make_float2(v[2 * k], v[2 * k + 1]);
  } else if constexpr ((F & gaq::F_ALIAS) != 0) {
    double v[18];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if constexpr ((F & (gaq::F_SWARM | gaq::F_ENVX)) != 0) v[j] = s.pos[j] - s.goal[j]; else v[j] = s.pos[j] - cfg.goal_default[j];
      v[3 + j] = s.vel[j]; v[15 + j] = s.omega[j];
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) v[6 + j] = s.rot[j];
    float2* h = reinterpret_cast<float2*>(buf + lane * kRowBytes);
    float hv[18];
    heads18<gaq::kHeadsAreObs<F>>(v, hv);
    if constexpr (kLoMix<F>) {
      uint32_t* q = reinterpret_cast<uint32_t*>(buf + kRowsLds + lane * kMixRowBytes);
#pragma unroll
      for (int k = 0; k < 9; ++k) h[k] = make_float2(hv[2 * k], hv[2 * k + 1]);    // the observation words
#pragma unroll
      for (int k = 0; k < 8; ++k) q[k] = split_lo(v[2 * k]) | ((2 * k + 1 < 15) ? (split_lo(v[2 * k + 1]) << 16) : 0u);
#pragma unroll
      for (int j = 0; j < 3; ++j) q[8 + j] = split_lo32(v[15 + j]);
    } else {
      uint32_t* q = reinterpret_cast<uint32_t*>(buf + kRowsLds + lane * kLoRowBytes);
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        h[k] = make_float2(hv[2 * k], hv[2 * k + 1]);    // the observation words
        q[k] = split_lo(v[2 * k]) | (split_lo(v[2 * k + 1]) << 16);
      }
    }
  } else {
    double* c = reinterpret_cast<double*>(buf) + lane;
#pragma unroll
    for (int j = 0; j < 3; ++j) c[(0 + j) * kTile] = s.pos[j];
#pragma unroll
    for (int j = 0; j < 3; ++j) c[(3 + j) * kTile] = s.vel[j];
#pragma unroll
    for (int j = 0; j < 9; ++j) c[(6 + j) * kTile] = s.rot[j];
#pragma unroll
    for (int j = 0; j < 3; ++j) c[(15 + j) * kTile] = s.omega[j];
  }
  if (gaq::has_lag<F>(cfg)) {
    double* l = reinterpret_cast<double*>(buf + im.lag) + lane;
    float* m = reinterpret_cast<float*>(buf + im.cmds) + lane;
#pragma unroll
    for (int j = 0; j < 4; ++j) { l[j * kTile] = (double)s.rot_damp[j]; m[j * kTile] = s.cmds_damp[j]; }
  }
  if (gaq::noise_mode<F>(cfg) != gaq::NOISE_OFF) {
    float* o = reinterpret_cast<float*>(buf + im.ou) + lane;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j * kTile] = s.ou[j];
  }
  if (gaq::has_act_prev<F>(cfg)) {
    float* a = reinterpret_cast<float*>(buf + im.actp) + lane;
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j * kTile] = s.act_prev[j];
  }
  if (gaq::has_env_goal<F>(cfg)) {
    float* g = reinterpret_cast<float*>(buf + im.goal) + lane;
#pragma unroll
    for (int j = 0; j < 3; ++j) g[j * kTile] = (float)s.goal[j];
  }
  if (gaq::has_gyro_bias<F>(cfg)) {
    float* g = reinterpret_cast<float*>(buf + im.gyro) + lane;
#pragma unroll
    for (int j = 0; j < 3; ++j) g[j * kTile] = s.gyro_bias[j];
    g[3 * kTile] = 0.0f;
  }
}

template <uint32_t F>
__device__ __forceinline__ void stage_out(const DevPtrs& p, const StepCfg& cfg, int64_t tile, const char* buf, uint32_t lane,
                                          float* obs) {
  const TileImage im = tile_image<F>(cfg);
  if constexpr ((F & gaq::F_ALIAS) != 0) {
    const int64_t first = tile * kTile;
    const uint32_t live = (uint32_t)((p.n - first) < kTile ? (p.n - first) : kTile);
    copy_out_rows<5, kRowsBytes, kStAux<F>>(obs + first * 18, buf, lane, live * kRowBytes);        // hi rows ARE the observation
    if constexpr ((F & gaq::F_PACK) == 0)
      if (p.obs_copy) copy_out_rows<5, kRowsBytes, kCopyAux<F>>(p.obs_copy + first * 18, buf, lane, live * kRowBytes);   // shadow mode: + the caller's copy
    if constexpr ((F & gaq::F_FP32) == 0)
      {
        if constexpr (kLoMix<F>) copy_out_rows<3, kMixRowsBytes, kStAux<F>>(reinterpret_cast<uint32_t*>(p.lo) + first * kMixRowWords, buf + kRowsLds, lane, kMixRowsBytes);
        else copy_out_rows<3, kLoRowsBytes, kStAux<F>>(reinterpret_cast<int16_t*>(p.lo) + first * 18, buf + kRowsLds, lane, kLoRowsBytes);
      }
  } else {
    copy_out<9, kStAux<F>>(p.core + tile * (kCorePlanes * kTile), buf, lane);
  }
  if (gaq::has_lag<F>(cfg)) {
    copy_out<2, kStAux<F>>(p.lag + tile * (kLagPlanes * kTile), buf + im.lag, lane);
    copy_out<1, kStAux<F>>(p.cmds + tile * (4 * kTile), buf + im.cmds, lane);
  }
  if (gaq::noise_mode<F>(cfg) != gaq::NOISE_OFF) copy_out<1, kStAux<F>>(p.ou + tile * (4 * kTile), buf + im.ou, lane);
  if (gaq::has_act_prev<F>(cfg)) copy_out<1, kStAux<F>>(p.actp + tile * (4 * kTile), buf + im.actp, lane);
  // the goal plane is written back only where the kernel can change it (resample_goal, excite); swarm formation goals are static
  if (gaq::has_env_goal<F>(cfg) && (cfg.resample_goal || cfg.excite))
    copy_out<1, kStAux<F>>(p.goal + tile * (4 * kTile), buf + im.goal, lane);
  if (gaq::has_gyro_bias<F>(cfg)) copy_out<1, kStAux<F>>(p.gyro + tile * (4 * kTile), buf + im.gyro, lane);
}

template <typename T>
__device__ __forceinline__ void convert_model(const Model<double>& a, Model<T>& m) {
  m.mass = T(a.mass); m.inv_mass = T(a.inv_mass);
#pragma unroll
  for (int j = 0; j < 3; ++j) { m.inertia[j] = T(a.inertia[j]); m.inv_inertia[j] = T(a.inv_inertia[j]); }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    m.thrust_max[j] = T(a.thrust_max[j]); m.torque_max[j] = T(a.torque_max[j]);
    m.prop_x[j] = T(a.prop_x[j]); m.prop_y[j] = T(a.prop_y[j]); m.prop_z[j] = T(a.prop_z[j]);
  }
  m.tau_up = T(a.tau_up); m.tau_down = T(a.tau_down); m.linearity = T(a.linearity); m.arm = T(a.arm);
  m.vel_damp = T(a.vel_damp); m.damp_omega_q = T(a.damp_omega_q); m.c_drag = T(a.c_drag); m.c_roll = T(a.c_roll);
  m.ou_sigma = a.ou_sigma; m.jinv = a.jinv;
}

// per-env model parameters: read-only tile-major planes, one 8-byte buffer load per plane and lane
template <uint32_t F, bool MODEL_IN_VGPRS = false>
__device__ __forceinline__ void load_model(const DevPtrs& p, const StepCfg& cfg, int64_t tile, uint32_t lane,
                                           const Model<double>& um, Model<Real<F>>& m) {
  using T = Real<F>;
  if constexpr ((F & gaq::F_PER_ENV) == 0) {
    if constexpr ((F & gaq::F_FP32) != 0) convert_model(um, m); else m = um;
    // (GAQ_VG_EXTRA: A/B knob -- feature bits whose instantiations also keep the model in VGPRs.  Tried on the F_PACK and F_MELL step kernels
    //  at N = 2^20: 168 -> 206 VGPRs = 3 -> 2 waves/SIMD costs more than the spill traffic it removes, sensor noise 72 -> 77 us, Mellinger
    //  69 -> 71; only the T-step rollout loop, VALU-bound and already at 2 waves, wins -- profiles/r03_vg_extra_ab.txt)
#ifndef GAQ_VG_EXTRA
#define GAQ_VG_EXTRA 0u
#endif
    if constexpr (((F & gaq::F_NT) != 0 && (F & gaq::F_PREDRAW) == 0 && (F & gaq::F_FP32) == 0) || (MODEL_IN_VGPRS && (F & gaq::F_FP32) == 0) ||
                  ((F & (GAQ_VG_EXTRA)) != 0 && (F & (gaq::F_FP32 | gaq::F_GENERIC)) == 0)) {
      // ONE wave per SIMD (the size rule's F_NT-without-F_PREDRAW instantiations; 512 VGPRs are free): the uniform model lives in VECTOR
      // registers.  As kernel arguments its ~35 doubles want 70 of the 106 SGPRs for the whole sub-step loop; hipcc spills the overflow
      // into VGPR lanes and the hot loop is then ~20 % v_readlane / v_writelane / s_nop -- with a single wave per SIMD straight on the
      // critical path: 7.7 -> 7.15 us per step at N = 65 536, 7.6 -> 6.8 at 32 768 (same box; profiles/r03_model_in_vgprs_ab.txt).  At
      // two waves per SIMD the other wave fills those slots and the 50 extra VGPRs only cost (8.65 -> 8.9 us at 131 072): not there.
      // An empty asm with a "+v" constraint is all it takes; the arithmetic and its results are unchanged
      // (test_size_specific_kernel_instantiations_are_bit_identical).
#define GAQ_VG(x) asm volatile("" : "+v"(x))
      GAQ_VG(m.inv_mass); GAQ_VG(m.linearity); GAQ_VG(m.vel_damp); GAQ_VG(m.damp_omega_q); GAQ_VG(m.arm);
#pragma unroll
      for (int j = 0; j < 3; ++j) { GAQ_VG(m.inertia[j]); GAQ_VG(m.inv_inertia[j]); }
#pragma unroll
      for (int j = 0; j < 4; ++j) { GAQ_VG(m.thrust_max[j]); GAQ_VG(m.torque_max[j]); GAQ_VG(m.prop_x[j]); GAQ_VG(m.prop_y[j]); }
      if constexpr ((F & gaq::F_LAG) != 0) { GAQ_VG(m.tau_up); GAQ_VG(m.tau_down); }
#undef GAQ_VG
    }
    return;
  }
  auto r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p.par + tile * (kPar * kTile)), 0, kParBytes, 0x00020000);
  const uint32_t o8 = lane * 8u;
  auto ld = [&](int plane) { return T(__builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, o8, plane * (kTile * 8), 0))); };
  m.inv_mass = ld(PP_INV_MASS);
  if (cfg.compact_params && (F & gaq::F_FP32) == 0) {
    // 20 planes instead of 30 (160 B instead of 240 B per env): the reciprocal inertia, torque_max and the rotor positions
    // are rebuilt with the very operations that made them on the host (one correctly rounded op each -> the same bits)
    auto ldd = [&](int plane) { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, o8, plane * (kTile * 8), 0)); };
    const double t2t = ldd(PP_T2T), mx = ldd(PP_MX), my = ldd(PP_MY), cx = ldd(PP_COMX), cy = ldd(PP_COMY);
    const double sx[4] = {1.0, -1.0, -1.0, 1.0}, sy[4] = {-1.0, -1.0, 1.0, 1.0};     // inertia.py:238-239
#pragma unroll
    for (int j = 0; j < 3; ++j) { const double in = ldd(PP_INERTIA + j); m.inertia[j] = T(in); m.inv_inertia[j] = T(1.0 / in); }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double th = ldd(PP_THRUST_MAX + j);
      m.thrust_max[j] = T(th); m.torque_max[j] = T(t2t * th);
      m.prop_x[j] = T(sx[j] * mx - cx); m.prop_y[j] = T(sy[j] * my - cy);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 3; ++j) { m.inertia[j] = ld(PP_INERTIA + j); m.inv_inertia[j] = ld(PP_INV_INERTIA + j); }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      m.thrust_max[j] = ld(PP_THRUST_MAX + j); m.torque_max[j] = ld(PP_TORQUE_MAX + j);
      m.prop_x[j] = ld(PP_PROP_X + j); m.prop_y[j] = ld(PP_PROP_Y + j);
    }
  }
  m.linearity = ld(PP_LINEARITY); m.arm = ld(PP_ARM);
  m.vel_damp = T(0); m.damp_omega_q = T(0);
  if (!cfg.zero_damp) { m.vel_damp = ld(PP_VEL_DAMP); m.damp_omega_q = ld(PP_DAMP_Q); }
  m.tau_up = T(1); m.tau_down = T(1);
  if (gaq::has_lag<F>(cfg)) { m.tau_up = ld(PP_TAU_UP); m.tau_down = ld(PP_TAU_DOWN); }
  m.ou_sigma = 0.0f;
  // the OU sigma is consumed as fp32: its plane holds 64 floats (the first 256 B of the 512-B slot), 4 B per lane
  if (gaq::noise_mode<F>(cfg) != gaq::NOISE_OFF)
    m.ou_sigma = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane * 4u, PP_OU_SIGMA * (kTile * 8), 0));
  m.mass = T(0); m.c_drag = T(0); m.c_roll = T(0);
  m.jinv = p.jinv ? p.jinv + (tile * kTile + lane) * 16 : nullptr;
#pragma unroll
  for (int j = 0; j < 4; ++j) m.prop_z[j] = T(0);
  if (((F & gaq::F_GENERIC) != 0) && cfg.drag) {
    m.mass = ld(PP_MASS); m.c_drag = ld(PP_C_DRAG); m.c_roll = ld(PP_C_ROLL);
#pragma unroll
    for (int j = 0; j < 4; ++j) m.prop_z[j] = ld(PP_PROP_Z + j);
  }
}

// the wave's [64, D] observation rows sit row-major in LDS; write them to obs[tile*64 .. , :] as 16-byte
// pieces.  `obs` is bounded by a buffer resource of exactly the live rows' bytes: rows of padding envs fall
// outside and are dropped by the hardware range check.
__device__ __forceinline__ void flush_obs(float* obs, int64_t n, int D, int64_t tile, const char* rows, uint32_t lane) {
  const int64_t first = tile * kTile;
  const int64_t live = (n - first) < kTile ? (n - first) : kTile;
  const uint32_t bytes = (uint32_t)live * D * 4u;
  auto r = __builtin_amdgcn_make_buffer_rsrc(obs + first * D, 0, (int)bytes, 0x00020000);
  const int pieces = (kTile * D * 4 + 1023) / 1024;
  for (int k = 0; k < pieces; ++k) {
    const uint32_t off = k * 1024 + lane * 16u;
    if (off < (uint32_t)(kTile * D * 4)) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(rows + off);
      __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, GAQ_ST_AUX);
    }
  }
}

// swarm neighbour exchange: agent (a + j) mod A of the same world sits (a + j) mod A lanes into the world's lane group
struct WaveSwarm {
  uint32_t lane; int agents;
  __device__ __forceinline__ void neighbour(int j, const float* me, float* o) const {
    const int src = (int)((lane & ~(uint32_t)(agents - 1)) | ((lane + (uint32_t)j) & (uint32_t)(agents - 1)));
#pragma unroll
    for (int k = 0; k < 6; ++k) o[k] = __shfl(me[k], src);
  }
  __device__ __forceinline__ bool any(bool b) const { return __any(b) != 0; }
};

__device__ __forceinline__ void wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wave_lds_fence() {
  // LDS operations of one wave execute in order; this only stops the compiler from reordering them
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Graph-safe mode (gaq_set_graph_safe): the step index that keys the noise / reset streams lives in device memory, so that a captured
// launch draws new numbers on every replay -- and the step launch advances it ITSELF: no second launch, no wave waits for anything.
// The counter is the SUM of kCtrSlots words (spread over the memory channels) = step_index << ctr_shift; every wave of a step launch adds 1 to one
// of them (the launch's first wave adds the rest up to 2^ctr_shift) with a NON-RETURNING device-scope atomic, issued only after its
// own read has returned.  A wave that reads while the launch is in flight sees (step << shift) + (check-ins that have landed), and that
// is fewer than 2^shift because its own is still missing -- the shift drops them: every wave of the launch gets the same index,
// whenever it is scheduled, and the words are only ever read together with the kernel boundary between launches.
// (Round 2 found the last wave with one RETURNING atomic per wave on ONE address at the END of the launch: 1024 serialised round trips on
// the critical path, 18.6 instead of 8.6 us per step at N = 65 536; the shipped fallback was a one-thread bump_kernel after every step: a
// second graph node and its dispatch gap.  Here the atomics leave at the start, spread over 64 words, and nothing depends on them.)
// 64 words 4352 B apart: not a round multiple of the memory channels' interleave, so the words -- and the atomics on them -- spread over the
// channels.  (16 words on 16 CONSECUTIVE 128-byte lines were one channel's work: ~5 ns per atomic, serialised -- invisible at 1024 waves,
// 10 us of a 8.6-us step at 2048 waves; profiles/r03_ctr_slots_ab.txt.)
constexpr int kCtrSlots = 64;               // counter words: one per lane of the wave that reads them ...
constexpr int kCtrStride = 544;             // ... this many uint64_t apart
__device__ __forceinline__ uint64_t step_counter_read(const DevPtrs& p, uint32_t lane) {
  static_assert(kCtrSlots == kTile, "one counter word per lane");
  // A PLAIN load (served by this XCD's L2 after its first miss of the launch): what it may return is any value the word has held since the
  // launch began -- the kernel boundary orders it after everything earlier -- and that is all the scheme needs.  (A device-scope atomic load
  // goes to memory every time: 2048 waves x 64 words on the same 64 lines cost 10 us of a 8.6-us step; profiles/r03_ctr_slots_ab.txt.)
  return p.step_ctr[lane * kCtrStride];                                   // one load instruction per wave
}
__device__ __forceinline__ uint64_t step_counter_sum(uint64_t c) {
#pragma unroll
  for (int o = 1; o < kCtrSlots; o <<= 1) c += __shfl_xor(c, o);     // every lane ends up with the sum
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)c), hi = __builtin_amdgcn_readfirstlane((uint32_t)(c >> 32));
  return ((uint64_t)hi << 32) | lo;
}
// the step launch: sum, check in, return this launch's step index
__device__ __forceinline__ uint64_t step_counter_checkin(const DevPtrs& p, uint64_t raw, uint32_t lane) {
  const uint64_t c = step_counter_sum(raw);
  const uint32_t gwave = __builtin_amdgcn_readfirstlane(blockIdx.x * (uint32_t)(kBlock / kTile) + (threadIdx.x >> 6));
  if (lane == 0) {
    uint32_t inc = gwave == 0 ? p.ctr_inc0 : 1u;
    uint32_t dep = (uint32_t)c;
    asm volatile("" : "+s"(inc) : "s"(dep));                 // the add is issued after the read has RETURNED (it must not overtake it)
    (void)__hip_atomic_fetch_add(p.step_ctr + (gwave % (uint32_t)kCtrSlots) * kCtrStride, (uint64_t)inc, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
  }
  return c >> p.ctr_shift;
}
// launches that do not advance the counter themselves (reset / observe; the fused rollout, followed by bump_kernel)
__device__ __forceinline__ uint64_t step_counter_peek(const DevPtrs& p, uint32_t lane) {
  return step_counter_sum(step_counter_read(p, lane)) >> p.ctr_shift;
}

// Multi-GPU return path fused into the step launch (gaq_set_packed_rows_dev): the wave's [64][D + 2] rows [obs | reward | (float) done]
// are assembled in the LDS buffer (free by now) and leave as 16-byte pieces like every other row array -- bit for bit what
// pack_rows_kernel makes of the step's outputs, without the extra launch between the step and the collective and without re-reading
// obs / reward / done from HBM.  `w` = this lane's D observation words (registers), W = D + 2 words per row.
template <int AUX>
__device__ __forceinline__ void flush_packed_rows(float* rows_out, int64_t n, int W, int64_t tile, char* lds, uint32_t lane) {
  const int64_t first = tile * kTile;
  const int64_t live = (n - first) < kTile ? (n - first) : kTile;
  const uint32_t bytes = (uint32_t)live * W * 4u;
  auto r = __builtin_amdgcn_make_buffer_rsrc(rows_out + first * W, 0, (int)bytes, 0x00020000);
  const int pieces = (kTile * W * 4 + 1023) / 1024;
  for (int k = 0; k < pieces; ++k) {
    const uint32_t off = k * 1024 + lane * 16u;
    if (off < (uint32_t)(kTile * W * 4)) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(lds + off);
      __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, AUX);
    }
  }
}

// info-dict extras of the last step (gaq_config.aux_outputs): plain 4-byte stores, 68 B per env
__device__ __forceinline__ void store_aux(const DevPtrs& p, int64_t i, const gaq::StepOut& out) {
  if (!p.aux) return;
  float* ax = p.aux + i * gaq::AUX_WORDS;
#pragma unroll
  for (int j = 0; j < 3; ++j) { ax[gaq::AUX_ACC + j] = out.acc_meter[j]; ax[gaq::AUX_OMEGA_DOT + j] = out.omega_dot[j]; ax[gaq::AUX_TORQUE + j] = out.torque[j]; }
#pragma unroll
  for (int j = 0; j < 4; ++j) { ax[gaq::AUX_CTRL + j] = out.ctrl[j]; ax[gaq::AUX_CMDS + j] = out.cmds[j]; }
}

// quad_core.hpp kModelMem: where the uniform model lies in the kernel-argument segment (the by-value arguments are laid out like the fields
// of a struct); read through this pointer -- constant address space, wave-uniform address -- the model's fields are s_load instructions
struct KernArgsMirror { DevPtrs p; StepCfg cfg; Model<double> um; };
static_assert(offsetof(KernArgsMirror, um) == sizeof(DevPtrs) + sizeof(StepCfg) && sizeof(DevPtrs) % 8 == 0 && sizeof(StepCfg) % 8 == 0,
              "kernel-argument layout of step_kernel's first three arguments");
typedef __attribute__((address_space(4))) const char karg_char;
__device__ __forceinline__ karg_char* kernarg_model_ptr() {
#if defined(__HIP_DEVICE_COMPILE__)
  return (karg_char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(KernArgsMirror, um);
#else
  return nullptr;
#endif
}

// ---- the fused step kernel: controller + step1 x sim_steps + crash + reward + done (+ reset) + obs ----
// (the uniform CrazyFlie kernel <22> sits 2 VGPRs above the 3-waves/SIMD line; forcing it there -- 2 spilled VGPRs -- changes
//  nothing: 72.86 vs 72.94 us at N = 2^20, profiles/r02_v4: it runs at the copy ceiling like the per-env kernel)
// Occupancy floors (none at present).  The kernel arguments (DevPtrs + StepCfg + Model: ~1 KB) do not fit the 106 SGPRs; hipcc spills the
// overflow into VGPR lanes (64 per VGPR: 2-3 VGPRs per kernel, `SGPRs Spill` of -Rpass-analysis=kernel-resource-usage), and several
// kernels sit right on an occupancy line because of it (<4> and <1044> at 168 VGPRs = 3 waves/SIMD): tools/kernel_resources.py after
// every change to this file, the table is profiles/r03_kernel_resources.txt.
// (round 3 tried the floor again on the VALU-bound neighbours of that line -- <1046> CrazyFlie with sensor noise, 180 VGPRs: 3 waves/SIMD
//  with 12 spilled VGPRs runs 105 us instead of 95, profiles/r03_w3_ab.txt -- so the allocator's own choice stands everywhere.)
template <uint32_t F> constexpr int kStepMinWaves = 1;
// (round 4: a floor pays where the kernel is 1 VGPR over the line and VALU / latency-bound -- these two, 4 VGPRs spilled: 82.3 -> 74.9 us and
//  92.2 -> 87.1; it costs where 7-10 are spilled or the kernel is bandwidth-bound anyway -- <1042> 83.6 -> 88.8, <197650> 85.7 -> 87.3, <214034>
//  88.2 -> 89.1, the swarm kernel <33812> (169 VGPRs, 4 spilled) 113.5 -> 115.1: profiles/r04_mellinger_auxp_rates.txt)
#ifndef GAQ_NO_FLOORS      // (A/B builds without the floors)
template <> inline constexpr int kStepMinWaves<214036u> = 3;      // one VGPR above the line (169): 4 spilled, profiles/r04_mellinger_auxp_rates.txt
template <> inline constexpr int kStepMinWaves<82962u> = 3;
#endif
template <uint32_t F>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(kStepMinWaves<F>))) void step_kernel(DevPtrs p, StepCfg cfg, Model<double> um,
                                                       const float* __restrict__ actions, float* obs,
                                                       float* __restrict__ reward, uint8_t* __restrict__ done,
                                                       int lds_per_wave) {
  constexpr bool G = (F & gaq::F_GENERIC) != 0;
  constexpr bool A = (F & gaq::F_ALIAS) != 0;
  static_assert(!(G && A), "obs/state aliasing exists in the specialised kernels only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform by construction
  const uint32_t lane = threadIdx.x & 63u;
  const int64_t tile = (int64_t)blockIdx.x * (kBlock / kTile) + wave;
  if constexpr ((F & gaq::F_CTR) != 0) {
    if (tile >= p.ntiles) {                                                // (every wave of the launch checks in)
      (void)step_counter_checkin(p, step_counter_read(p, lane), lane);
      return;
    }
  } else {
    if (p.step_ctr) cfg.step_index = *p.step_ctr >> p.ctr_shift;          // graph-safe mode, advanced by bump_kernel: one scalar load (the
    if (tile >= p.ntiles) return;                                          // host keeps the whole count in the first word for these kernels)
  }
  char* buf = smem + wave * lds_per_wave;
  const int64_t i = tile * kTile + lane;
  const bool live = i < p.n;
  const int D = cfg.obs_dim;

  stage_in<F>(p, cfg, tile, buf, lane);                                    // asynchronous LDS-DMA
  // everything that does not need the image is issued under the DMA's latency
  using T = Real<F>;
  Model<T> m;
  if constexpr (!gaq::kModelMem<F>) load_model<F>(p, cfg, tile, lane, um, m);
  const Model<T>* mp = &m;
  if constexpr (gaq::kModelMem<F>) mp = (const Model<T>*)kernarg_model_ptr();
  float4 a4;
  uint32_t cw;
  {
    auto ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(actions), 0, (int)(p.n * 16), 0x00020000);
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ra, (uint32_t)i * 16u, 0, (F & gaq::F_NT) ? 2 : GAQ_ACT_AUX);
    a4 = __builtin_bit_cast(float4, v);
    auto rc = __builtin_amdgcn_make_buffer_rsrc(p.ctr, 0, (int)(p.ntiles * kTile * 4), 0x00020000);
    cw = __builtin_amdgcn_raw_buffer_load_b32(rc, (uint32_t)i * 4u, 0, (F & gaq::F_NT) ? 2 : GAQ_ACT_AUX);
  }
  // F_CTR (graph-safe mode, small batches): read + check in AFTER the state loads have been issued
  if constexpr ((F & gaq::F_CTR) != 0) cfg.step_index = step_counter_checkin(p, step_counter_read(p, lane), lane);
  float pre0[4] = {0.0f, 0.0f, 0.0f, 0.0f}, pre1[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if constexpr ((F & gaq::F_PREDRAW) != 0) {
    // small batches: every wave of the launch sits in this wait at the same time and nothing else hides it, so the
    // thrust-noise draws of the two sub-steps (2 Philox blocks + Box-Muller, a third of the arithmetic) go here
    const gaq::Philox r0(cfg.seed, cfg.env_offset + (uint64_t)i, cfg.step_index, gaq::RNG_OU0);
    gaq::normals4(r0, pre0);
    if (cfg.sim_steps > 1) { const gaq::Philox r1(cfg.seed, cfg.env_offset + (uint64_t)i, cfg.step_index, gaq::RNG_OU0 + 1u); gaq::normals4(r1, pre1); }
  }
  if constexpr ((F & gaq::F_PREDRAW) != 0) {
    // pin BOTH sub-steps' normals before the wait: volatile asm statements keep their order, and left alone the scheduler sinks the
    // second Philox block (not needed until the second sub-step) below the wait it was meant to hide under
    asm volatile("" :: "v"(pre0[0]), "v"(pre0[1]), "v"(pre0[2]), "v"(pre0[3]), "v"(pre1[0]), "v"(pre1[1]), "v"(pre1[2]), "v"(pre1[3]));
  }
  wait_dma();
  EnvState<T> s;
  read_image<F>(cfg, buf, lane, s);
  s.tick = cw & 0xFFFFu;
  s.svd_ctr = cw >> 16;
  wave_lds_fence();                                                        // image consumed: buffer is free

  const float act[4] = {a4.x, a4.y, a4.z, a4.w};
  gaq::StepOut out;
  out.reward = 0.0f; out.done = 0; out.crashed = 0;
  constexpr int kObWords = (F & gaq::F_AUXP) ? 28 : 26;                    // (+ t2w, t2t: the F_AUXP instantiations)
  float ob[kObWords];                                                      // specialised kernels: obs stays in VGPRs:
#pragma unroll
  for (int k = 0; k < kObWords; ++k) ob[k] = 0.0f;                         // 18 words + fixed slots for h, acc[3], act[4]
  char* rows = buf;                                                        // obs rows take over the consumed image's LDS
  float* term_row = p.term_obs ? p.term_obs + i * D : nullptr;
#ifndef GAQ_PACK_ROWS_LDS
#define GAQ_PACK_ROWS_LDS 0      // A/B knob: the F_PACK kernels pack their observation rows straight into LDS too (instead of 18-28 VGPRs held
#endif                           // from pack_obs to the end of the kernel)
  constexpr bool kObsRowsInLds = G || (F & gaq::F_SWARM) != 0 || (F & gaq::F_AUXP) != 0 ||      // the observation rows are packed straight into the LDS buffer
                                 (GAQ_PACK_ROWS_LDS && (F & gaq::F_PACK) != 0 && (F & gaq::F_ALIAS) != 0);
  // (F_AUXP: measured -- 114-118 -> 99-101 us per step with the aux row at N = 2^20, profiles/r04_auxp_ab.txt; the other F_PACK kernels gain or
  //  lose 2 % either way and keep their rows in registers)
  constexpr bool kAuxRowsInLds = gaq::kAuxIsRow<F>;          // ... and the info dict's 17-word aux rows behind them, flushed as 16-byte pieces too
  const int aux_off = (kTile * D * 4 + 15) & ~15;
  if (live && !ablated(cfg, 1)) {
    if constexpr ((F & gaq::F_SWARM) != 0) {
      // split state, observation rows (self block + neighbour terms by wave shuffles) packed into LDS like in the generic kernel
      float* row = reinterpret_cast<float*>(rows) + lane * D;
      gaq::env_step<T, F>(s, *mp, cfg, act, cfg.env_offset + (uint64_t)i, [&](int, int) { return 0.0f; }, out,
                          [&](int k, float v, int) { row[k] = v; }, term_row, WaveSwarm{lane, cfg.swarm.agents});
    } else if constexpr (G) {
      const float* nz = p.noise_in;
      const int64_t n = p.n;
      float* row = reinterpret_cast<float*>(rows) + lane * D;
      if constexpr (kAuxRowsInLds) out.aux_row = reinterpret_cast<float*>(rows + aux_off) + lane * gaq::AUX_WORDS;
      const float* sz = p.sense_in;
      gaq::env_step<T, F>(s, *mp, cfg, act, cfg.env_offset + (uint64_t)i,
                               [&](int k, int c) { return nz[((int64_t)k * 4 + c) * n + i]; }, out,
                               [&](int k, float v, int) { row[k] = v; }, term_row, WaveSwarm{lane, cfg.swarm.agents},
                               [&](int c, int slot, int j) { return sz ? sz[((int64_t)(c * 12 + slot) * 3 + j) * n + i] : 0.0f; });
      if constexpr (gaq::kAux<F> && !kAuxRowsInLds) store_aux(p, i, out);
    } else if constexpr (gaq::kHeadsAreObs<F>) {
      // the observation is the fp32 head of the new state: written by write_image, nothing to pack
      gaq::env_step<T, F>(s, *mp, cfg, act, cfg.env_offset + (uint64_t)i, [&](int k, int c) { return k == 0 ? pre0[c] : pre1[c]; }, out,
                          [&](int, float, int) {}, term_row);
    } else if constexpr (kObsRowsInLds) {   // (GAQ_PACK_ROWS_LDS: split state, observation rows packed straight into the LDS buffer)
      float* row = reinterpret_cast<float*>(rows) + lane * D;
      // (F_AUXP: env_step stores the aux values straight into this lane's 17-word row -- odd stride: conflict-free -- behind the
      //  observation rows; both leave as 16-byte pieces below)
      if constexpr (kAuxRowsInLds) out.aux_row = reinterpret_cast<float*>(rows + aux_off) + lane * gaq::AUX_WORDS;
      gaq::env_step<T, F>(s, *mp, cfg, act, cfg.env_offset + (uint64_t)i, [&](int, int) { return 0.0f; }, out,
                          [&](int k, float v, int) { row[k] = v; }, term_row);
      if constexpr (gaq::kAux<F> && !kAuxRowsInLds) store_aux(p, i, out);
    } else {   // plain layout, or split state with an explicitly packed observation (F_PACK)
      gaq::env_step<T, F>(s, *mp, cfg, act, cfg.env_offset + (uint64_t)i, [&](int, int) { return 0.0f; }, out,
                          [&](int k, float v, int slot) { if (slot < 0) ob[k] = v; else ob[18 + slot] = v; }, term_row);
      if constexpr (gaq::kAux<F>) store_aux(p, i, out);      // (F_AUXP: the info dict's aux row beside the split state)
    }
  }
  // dynamics_randomize_every on the device (quadrotor.py:1063-1066 per env): an env that finished its episode and is due takes
  // the parameter planes staged for it (derived ahead of time by the refill pass, off the critical path) -- "a new
  // QuadrotorDynamics": since_last_svd = 0 (:104), a fresh OUNoise (:198).  The planes are copied at the very end of the wave.
  bool promote = false;
  if constexpr ((F & gaq::F_RZ) != 0) {
    if (p.rz_every > 0 && live && out.done && !ablated(cfg, 4)) {
      auto rt = __builtin_amdgcn_make_buffer_rsrc(p.traj, 0, (int)(p.ntiles * (3 * kTile * 4)), 0x00020000);
      const int third = (int)(p.ntiles * (kTile * 4));                       // traj | rcount | rz_flag
      bool due = true;                                                       // every episode: the episode count is not needed
      if (p.rz_every > 1) {                                                  // (only finished lanes pay this one round trip)
        const uint32_t tr = __builtin_amdgcn_raw_buffer_load_b32(rt, (uint32_t)i * 4u, 0, 0) + 1u;
        __builtin_amdgcn_raw_buffer_store_b32(tr, rt, (uint32_t)i * 4u, 0, 0);
        due = ((tr + 1u) % (uint32_t)p.rz_every) == 0u;
      }
      if (due) {   // the rest waits for nothing: non-returning atomics
        promote = true;
        (void)__builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, rt, (uint32_t)i * 4u, third, 0);
        (void)__builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, rt, (uint32_t)i * 4u, 2 * third, 0);
        s.svd_ctr = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) s.ou[j] = 0.0f;
      }
    }
  }
  if constexpr (kObsRowsInLds) {   // observation rows (row-major in LDS) -> HBM, before the new image overwrites them
    wave_lds_fence();
    if constexpr (A) flush_obs(p.obs_copy, p.n, D, tile, rows, lane);      // (split state: `obs` is where the library keeps the heads)
    else flush_obs(obs, p.n, D, tile, rows, lane);
    if constexpr (kAuxRowsInLds) { if (p.aux) flush_obs(p.aux, p.n, gaq::AUX_WORDS, tile, rows + aux_off, lane); }
    wave_lds_fence();
  }
  // new state -> LDS image -> HBM
  write_image<F>(cfg, buf, lane, s);
  wave_lds_fence();
  stage_out<F>(p, cfg, tile, buf, lane, obs);
  {
    auto rc = __builtin_amdgcn_make_buffer_rsrc(p.ctr, 0, (int)(p.ntiles * kTile * 4), 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32((s.tick & 0xFFFFu) | (s.svd_ctr << 16), rc, (uint32_t)i * 4u, 0, kStAux<F>);
  }
  if (live) {
    reward[i] = out.reward;
    done[i] = out.done;
    if (!isfinite(out.reward)) atomicAdd(p.nan_count, 1u);
  }
  // observation rows -> LDS (row-major) -> HBM   (alias mode: already written by stage_out; F_PACK: the caller's tensor is
  // p.obs_copy -- `obs` is where the library keeps the state heads)
  if constexpr (!gaq::kHeadsAreObs<F> && !kObsRowsInLds) {
    float* obs_rows_out = obs;
    if constexpr (A) obs_rows_out = p.obs_copy;
    wave_lds_fence();                                                      // image reads of stage_out are done
    float* row = reinterpret_cast<float*>(rows) + lane * D;
    if (D == 18) {
#pragma unroll
      for (int k = 0; k < 18; k += 2) *reinterpret_cast<float2*>(row + k) = make_float2(ob[k], ob[k + 1]);
    } else {   // the 19 ... 25-word variants: rows are only 4-byte aligned; the appended words close up in the row
      int k = 18;
      bool quat = false;
      if constexpr ((F & gaq::F_AUXP) != 0) quat = (cfg.obs_flags & gaq::OBS_QUAT) != 0;    // 13-word base block [pos vel quat omega]
      if (quat) {
#pragma unroll
        for (int j = 0; j < 13; ++j) row[j] = ob[j];
        k = 13;
      } else {
#pragma unroll
        for (int j = 0; j < 18; ++j) row[j] = ob[j];
      }
      if (cfg.obs_flags & gaq::OBS_APPEND_H) row[k++] = ob[18];
      if (cfg.obs_flags & gaq::OBS_APPEND_ACC) { row[k] = ob[19]; row[k + 1] = ob[20]; row[k + 2] = ob[21]; k += 3; }
      if (cfg.obs_flags & gaq::OBS_APPEND_ACT) { row[k] = ob[22]; row[k + 1] = ob[23]; row[k + 2] = ob[24]; row[k + 3] = ob[25]; k += 4; }
      if constexpr ((F & gaq::F_AUXP) != 0) {
        if (cfg.obs_flags & gaq::OBS_APPEND_T2W) row[k++] = ob[26];
        if (cfg.obs_flags & gaq::OBS_APPEND_T2T) row[k++] = ob[27];
      }
    }
    wave_lds_fence();
    flush_obs(obs_rows_out, p.n, D, tile, rows, lane);
  }

  // F_ROWS (multi-GPU return path): the packed [obs | reward | done] rows leave with this launch.  The observation words are the heads
  // still sitting in the image; every lane picks its own up, then the rows are laid out 20 words apart in the same buffer.
  if constexpr ((F & gaq::F_ROWS) != 0) {
    static_assert(gaq::kHeadsAreObs<F> && (F & gaq::F_FP32) == 0, "fused packed rows: alias kernels whose observation is the heads");
    float hv[18];
    const float2* h = reinterpret_cast<const float2*>(buf + lane * kRowBytes);
#pragma unroll
    for (int k = 0; k < 9; ++k) { const float2 a = h[k]; hv[2 * k] = a.x; hv[2 * k + 1] = a.y; }
    wave_lds_fence();                                                      // every read of the image has been issued (LDS runs a wave's operations in order)
    float* row = reinterpret_cast<float*>(buf) + lane * 20;                // 80-byte rows: 8-byte aligned
#pragma unroll
    for (int k = 0; k < 18; k += 2) *reinterpret_cast<float2*>(row + k) = make_float2(hv[k], hv[k + 1]);
    *reinterpret_cast<float2*>(row + 18) = make_float2(out.reward, (float)out.done);
    wave_lds_fence();
    flush_packed_rows<kStAux<F>>(p.rows_out, p.n, 20, tile, buf, lane);
  }

  if constexpr ((F & gaq::F_RZ) != 0) {
    unsigned long long pm = __ballot(promote);
    if (pm && !ablated(cfg, 2)) {   // staged planes -> current planes of the promoted lanes
      // The staged draw of env i is ONE ROW of 45 doubles (par_next[i][45], behind the planes in the same allocation): lane k moves
      // word k of the promoted env's row into plane k -- one coalesced 360-byte load and one scattered store per env instead of
      // 2 x 45 memory instructions (the step kernels are bound by the rate of memory instructions and by the number of cache lines
      // they touch, not only by bytes).
      double* cur = const_cast<double*>(p.par) + tile * (int64_t)(kPar * kTile);
      const double* rows = p.par + p.ntiles * (int64_t)(kPar * kTile) + kParNextSkew + tile * (int64_t)(kPar * kTile);
      // With the compact parameter path and no damping planes (every shipped model and its perturbations) the step kernels read 18 of the
      // 45 planes; those -- plus torque_max[0], which the reset kernel's t2t observation reads -- are all a promotion has to move: 19
      // scattered stores per promoted env instead of 45 (the others are re-derived from the env's resample count when somebody asks for
      // them: gaq_get_params).  101.8 -> 99.0 us per step (plain: 90.2) with every episode of 2^20 staggered envs re-randomised.
      // (the fp32 forms do not take the compact path -- load_model reads all 30 planes there -- so their promotions move everything:
      //  found by tests/test_gpu_kernel_coverage.py, the first test to fly <2097> ... <2103> through an episode end)
      const bool hot_only = cfg.compact_params != 0 && cfg.zero_damp != 0 && (F & gaq::F_FP32) == 0;
      const int pl = hot_only ? (int)lane + ((int)lane < 4 ? 1 : (int)lane < 9 ? 4 : (int)lane < 13 ? 19 : 23) : (int)lane;
      static_assert(PP_INV_MASS == 1 && PP_INERTIA == 2 && PP_THRUST_MAX == 8 && PP_TORQUE_MAX == 12 && PP_TAU_UP == 28 && PP_TAU_DOWN == 29 &&
                    PP_LINEARITY == 30 && PP_ARM == 31 && PP_OU_SIGMA == 36 && PP_T2T == 37 && PP_COMY == 41, "hot planes of a promotion");
      const bool mine = hot_only ? (int)lane < kHotPlanes : (int)lane < kPar;
      while (pm) {
        const int L = __ffsll((long long)pm) - 1;
        pm &= pm - 1ull;
        if (mine) {
          const double v = rows[L * kPar + pl];
          if (pl == PP_OU_SIGMA) reinterpret_cast<float*>(cur + PP_OU_SIGMA * kTile)[L] = (float)v;   // 64 floats in half a slot
          else cur[pl * kTile + L] = v;
        }
      }
    }
  }
  if (p.done_list) {   // wavefront compaction of the done env indices (host-side episode bookkeeping)
    const bool is_done = live && out.done;
    uint32_t* cnt = p.done_count + (cfg.step_index & 1);
    if (i == 0) p.done_count[(cfg.step_index + 1) & 1] = 0;                // next step's counter
    const unsigned long long mask = __ballot(is_done);
    if (mask) {
      const int leader = __ffsll((long long)mask) - 1;
      uint32_t base = 0;
      if ((int)lane == leader) base = atomicAdd(cnt, (uint32_t)__popcll(mask));
      base = __shfl(base, leader);
      if (is_done) p.done_list[base + __popcll(mask & ((1ull << lane) - 1ull))] = (uint32_t)i;
    }
  }
}

// ---- fused T-step rollout: open-loop action sequences [T,N,4], state kept in registers --------
// One launch advances every env by T steps: the split state is read once, each step only loads its action tile
// (prefetched one step ahead), writes its observation rows / reward / done for slot t, and the residual rows and
// noise / motor state go back to HBM once at the end.  Traffic per env-step drops from 277 B to ~93 B + 1/T of the
// rest; at N = 65 536 (one tile per SIMD, where a single-step launch is pure latency) this removes the per-step
// load -> store round trip.  Results are those of T gaq_step_dev calls (same arithmetic, same RNG keys).
// Registers: the T-step loop is VALU-bound (~1160 instructions per wave and step) and -- with the uniform model in SGPRs -- spilled 204 SGPRs
// into VGPR lanes: a fifth of that loop was v_readlane / v_writelane / s_nop.  Round 2 asked the allocator for 3 waves/SIMD (168 VGPRs, 6 of
// them spilled to scratch), worth 5 % over the 2 it would pick.  Round 3 keeps the uniform model in VGPRs instead (load_model<F, true>: 234
// VGPRs, 2 waves/SIMD, nothing spilled to scratch): 2^20 envs 2430 -> 2350 us per 64 steps, 262 144 envs 652 -> 594, 65 536 envs 245 -> 190
// (1.7e10 -> 2.2e10 env-steps/s), uniform CrazyFlie 2810 -> 2670 (one box, interleaved; profiles/r03_rollout_model_in_vgprs_ab.txt).
template <uint32_t F> constexpr int kRollMinWaves = 1;
template <uint32_t F>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(kRollMinWaves<F>))) void rollout_kernel(DevPtrs p, StepCfg cfg, Model<double> um, int T,
                                                          const float* __restrict__ actions, float* obs,
                                                          float* __restrict__ reward, uint8_t* __restrict__ done,
                                                          int lds_per_wave) {
  static_assert((F & gaq::F_ALIAS) != 0 && (F & gaq::F_GENERIC) == 0, "fused rollout: alias layout only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63u;
  if (p.step_ctr) cfg.step_index = step_counter_peek(p, lane);            // (advanced by bump_kernel after the launch: T steps at once)
  const int64_t tile = (int64_t)blockIdx.x * (kBlock / kTile) + wave;
  if (tile >= p.ntiles) return;
  char* buf = smem + wave * lds_per_wave;
  const int64_t i = tile * kTile + lane;
  const bool live = i < p.n;
  const int64_t first = tile * kTile;
  const uint32_t nlive = (uint32_t)((p.n - first) < kTile ? (p.n - first) : kTile);
  const TileImage im = tile_image<F>(cfg);

  stage_in<F>(p, cfg, tile, buf, lane);
  using RT = Real<F>;
  Model<RT> m;
  load_model<F, true>(p, cfg, tile, lane, um, m);         // (uniform model in VGPRs: see kRollMinWaves)
  auto rc = __builtin_amdgcn_make_buffer_rsrc(p.ctr, 0, (int)(p.ntiles * kTile * 4), 0x00020000);
  const uint32_t cw = __builtin_amdgcn_raw_buffer_load_b32(rc, (uint32_t)i * 4u, 0, 0);
  auto load_action = [&](int t) {
    auto ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(actions + ((int64_t)t * p.n + first) * 4), 0,
                                                (int)(nlive * 16u), 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b128(ra, lane * 16u, 0, 0);
  };
  u32x4 a_next = load_action(0);
  wait_dma();
  EnvState<RT> s;
  read_image<F>(cfg, buf, lane, s);
  s.tick = cw & 0xFFFFu;
  s.svd_ctr = cw >> 16;
  wave_lds_fence();
  StepCfg c = cfg;
  for (int t = 0; t < T; ++t) {
    const float4 a4 = __builtin_bit_cast(float4, a_next);
    if (t + 1 < T) a_next = load_action(t + 1);                            // in flight during this step's compute
    const float act[4] = {a4.x, a4.y, a4.z, a4.w};
    gaq::StepOut out;
    out.reward = 0.0f; out.done = 0; out.crashed = 0;
    float* term_row = p.term_obs ? p.term_obs + i * 18 : nullptr;
    if (live)
      gaq::env_step<RT, F>(s, m, c, act, c.env_offset + (uint64_t)i, [&](int, int) { return 0.0f; }, out,
                          [&](int, float, int) {}, term_row);
    // observation rows of slot t = heads of the new state
    {
      RT v[18];
#pragma unroll
      for (int j = 0; j < 3; ++j) { v[j] = s.pos[j] - RT(cfg.goal_default[j]); v[3 + j] = s.vel[j]; v[15 + j] = s.omega[j]; }
#pragma unroll
      for (int j = 0; j < 9; ++j) v[6 + j] = s.rot[j];
      float2* h = reinterpret_cast<float2*>(buf + lane * kRowBytes);
      if constexpr ((F & gaq::F_FP32) != 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) h[k] = make_float2((float)v[2 * k], (float)v[2 * k + 1]);
        // the state the next step continues from is exactly what the caller sees (fp32 mode has no hidden bits)
#pragma unroll
        for (int j = 0; j < 3; ++j) s.pos[j] = RT((float)v[j]) + RT(cfg.goal_default[j]);
      } else {
        double vd[18];
#pragma unroll
        for (int k = 0; k < 18; ++k) vd[k] = (double)v[k];
        float hv[18];
        heads18<true>(vd, hv);
#pragma unroll
        for (int k = 0; k < 9; ++k) h[k] = make_float2(hv[2 * k], hv[2 * k + 1]);
      }
    }
    wave_lds_fence();
    const int64_t slot = (int64_t)t * p.n;
    copy_out_rows<5, kRowsBytes>(obs + (slot + first) * 18, buf, lane, nlive * kRowBytes);
    if (live) {
      reward[slot + i] = out.reward;
      done[slot + i] = out.done;
      if (!isfinite(out.reward)) atomicAdd(p.nan_count, 1u);
    }
    wave_lds_fence();                                                      // rows read out before the next step refills them
    c.step_index += 1;
  }
  // final state -> image -> HBM (hi rows again: they are also the state head the next launch reads from slot T-1)
  write_image<F>(cfg, buf, lane, s);
  wave_lds_fence();
  if (p.hi_final) copy_out_rows<5, kRowsBytes>(p.hi_final + first * 18, buf, lane, nlive * kRowBytes);   // shadow mode: the library's heads
  if constexpr ((F & gaq::F_FP32) == 0) {
    if constexpr (kLoMix<F>) copy_out_rows<3, kMixRowsBytes>(reinterpret_cast<uint32_t*>(p.lo) + first * kMixRowWords, buf + im.lo, lane, kMixRowsBytes);
    else copy_out_rows<3, kLoRowsBytes>(reinterpret_cast<int16_t*>(p.lo) + first * 18, buf + im.lo, lane, kLoRowsBytes);
  }
  if (gaq::has_lag<F>(cfg)) {
    copy_out<2>(p.lag + tile * (kLagPlanes * kTile), buf + im.lag, lane);
    copy_out<1>(p.cmds + tile * (4 * kTile), buf + im.cmds, lane);
  }
  if (gaq::noise_mode<F>(cfg) != gaq::NOISE_OFF) copy_out<1>(p.ou + tile * (4 * kTile), buf + im.ou, lane);
  __builtin_amdgcn_raw_buffer_store_b32((s.tick & 0xFFFFu) | (s.svd_ctr << 16), rc, (uint32_t)i * 4u, 0, 0);
}

}  // namespace gaqk

// ---- every instantiation that exists, in eight parts of similar compile time (the generic ones are the heavy ones) ----------
// step_kernel<F>: F = gaq::Feature mask (quad_core.hpp)
// (twins: + 4096 = F_ROWS of 20 / 22 / 23 in their four size forms; + 8192 = F_CTR of their six non-temporal small-batch forms (at most two waves per SIMD: every wave of a
//  launch reads the counter's 64 words -- 8 KB of L2 traffic per wave, nothing at 2048 waves, 107 instead of 48 us per step at 16 384);
//  + 16384 = F_MELL: the Mellinger controller on the plain (0 .. 7) and split (16 .. 23) layouts and with a packed observation (+ 1024),
//  uniform model or -- odd masks -- per-env models (one inverse jacobian per env; + 2048 = F_RZ: re-randomised on the device, gaq.hip jinv_kernel);
//  33808 .. 33814 = F_SWARM | F_PACK | F_ALIAS | lag | noise: the swarm layer on the split state;
//  66576 .. 66583 = F_AUXP | F_PACK | F_ALIAS | per-env | lag | noise: the info dict's aux row / the quaternion, t2w, t2t observations on the split
//  state (68625 .. 68631: + F_RZ, per-env models with per-episode re-randomisation);
//  197648 .. 197655 = F_ENVX | F_AUXP | F_PACK | F_ALIAS | per-env | lag | noise: per-env goals (resample_goal, excite) on the split state
//  (199697 .. 199703: + F_RZ);
//  459792 .. 459798: + F_BIAS, the gyro-bias random walk (with or without per-env goals), uniform model;
//  82960 .. 82966 = F_MELL | F_AUXP ..., 214032 .. 214038 = F_MELL | F_ENVX | F_AUXP ...: the same for the Mellinger controller, uniform model)
#define GAQ_STEP_PART0(X) X(8u) X(1u) X(3u) X(16u) X(48u) X(2049u) X(2065u) X(3089u) X(2097u) X(4116u) X(8468u) X(16384u) X(33808u) X(197648u) X(459792u) X(197649u) X(199697u) X(16385u) X(16401u) X(18433u) X(18449u)
#define GAQ_STEP_PART1(X) X(9u) X(0u) X(2u) X(17u) X(49u) X(2051u) X(2067u) X(3091u) X(4244u) X(4372u) X(8470u) X(16386u) X(33810u) X(17424u) X(197650u) X(459794u) X(197651u) X(199699u) X(16387u) X(16403u) X(18435u) X(18451u)
#define GAQ_STEP_PART2(X) X(72u) X(2057u) X(4u) X(18u) X(50u) X(2053u) X(2069u) X(3093u) X(4500u) X(4118u) X(8471u) X(16388u) X(33812u) X(66578u) X(66583u) X(68625u) X(16389u) X(16405u) X(18437u) X(18453u)
#define GAQ_STEP_PART3(X) X(73u) X(2121u) X(5u) X(19u) X(51u) X(2055u) X(2071u) X(3095u) X(4246u) X(4374u) X(8596u) X(16390u) X(33814u) X(17426u) X(197652u) X(459796u) X(197653u) X(199701u) X(16391u) X(16407u) X(18439u) X(18455u)
#define GAQ_STEP_PART4(X) X(520u) X(6u) X(20u) X(52u) X(1040u) X(1041u) X(148u) X(276u) X(2099u) X(4502u) X(16400u) X(16402u) X(66576u) X(66581u) X(68627u) X(17425u) X(82960u) X(214032u) X(19473u)
#define GAQ_STEP_PART5(X) X(521u) X(7u) X(21u) X(53u) X(1042u) X(1043u) X(150u) X(278u) X(2101u) X(4119u) X(8598u) X(584u) X(17428u) X(197654u) X(459798u) X(197655u) X(199703u) X(17427u) X(82962u) X(214034u) X(19475u)
#define GAQ_STEP_PART6(X) X(2569u) X(22u) X(54u) X(1044u) X(1045u) X(151u) X(279u) X(404u) X(2103u) X(4247u) X(16404u) X(66580u) X(66577u) X(68631u) X(17429u) X(82964u) X(214036u) X(19477u)
#define GAQ_STEP_PART7(X) X(23u) X(55u) X(1046u) X(1047u) X(406u) X(407u) X(4375u) X(4503u) X(8599u) X(16406u) X(17430u) X(66582u) X(66579u) X(68629u) X(17431u) X(82966u) X(214038u) X(19479u)
#define GAQ_STEP_ALL(X) GAQ_STEP_PART0(X) GAQ_STEP_PART1(X) GAQ_STEP_PART2(X) GAQ_STEP_PART3(X) GAQ_STEP_PART4(X) GAQ_STEP_PART5(X) \
                        GAQ_STEP_PART6(X) GAQ_STEP_PART7(X)
// rollout_kernel<F>: the alias kernels (16 ... 23) and their fp32 forms (48 ... 55)
#define GAQ_ROLL_PART0(X) X(16u) X(48u)
#define GAQ_ROLL_PART1(X) X(17u) X(49u)
#define GAQ_ROLL_PART2(X) X(18u) X(50u)
#define GAQ_ROLL_PART3(X) X(19u) X(51u)
#define GAQ_ROLL_PART4(X) X(20u) X(52u)
#define GAQ_ROLL_PART5(X) X(21u) X(53u)
#define GAQ_ROLL_PART6(X) X(22u) X(54u)
#define GAQ_ROLL_PART7(X) X(23u) X(55u)
#define GAQ_ROLL_ALL(X) GAQ_ROLL_PART0(X) GAQ_ROLL_PART1(X) GAQ_ROLL_PART2(X) GAQ_ROLL_PART3(X) GAQ_ROLL_PART4(X) GAQ_ROLL_PART5(X) \
                        GAQ_ROLL_PART6(X) GAQ_ROLL_PART7(X)
#define GAQ_STEP_SIG(FEAT) \
  void gaqk::step_kernel<(FEAT)>(gaqk::DevPtrs, gaq::StepCfg, gaq::Model<double>, const float*, float*, float*, uint8_t*, int);
#define GAQ_ROLL_SIG(FEAT) \
  void gaqk::rollout_kernel<(FEAT)>(gaqk::DevPtrs, gaq::StepCfg, gaq::Model<double>, int, const float*, float*, float*, uint8_t*, int);
