// Issue cost of the vector instructions the step kernels are made of (fp64 arithmetic, the 32-bit integer multiplies of Philox, the
// transcendentals of Box-Muller, v_readlane), relative to v_fma_f32, at 1 / 2 / 3 waves per SIMD: which of them a VALU-bound kernel
// (sensor noise, Mellinger) should be counting.  Each kernel runs 8 independent chains of ONE instruction, 64 per loop trip.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_rates tools/valu_rates.hip && /tmp/valu_rates > profiles/rNN_valu_rates.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define REP64(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S)

// 32-bit destination, three 32-bit sources
#define K32(NAME, INS)                                                                                  \
  __global__ void NAME(uint32_t* out, int iters) {                                                      \
    uint32_t a[8], b = threadIdx.x * 2654435761u + 12345u, c = threadIdx.x | 3u;                        \
    for (int j = 0; j < 8; ++j) a[j] = threadIdx.x + j;                                                 \
    for (int i = 0; i < iters; ++i) {                                                                   \
      REP64(INS)                                                                                        \
    }                                                                                                   \
    uint32_t s = 0;                                                                                     \
    for (int j = 0; j < 8; ++j) s ^= a[j];                                                              \
    if (s == 0x12345u) out[threadIdx.x] = s;                                                            \
  }
#define K64(NAME, INS)                                                                                  \
  __global__ void NAME(uint32_t* out, int iters) {                                                      \
    double a[8], b = 1.0 + 1e-9 * threadIdx.x, c = 1e-12 * threadIdx.x;                                 \
    for (int j = 0; j < 8; ++j) a[j] = 1.0 + threadIdx.x + j;                                           \
    for (int i = 0; i < iters; ++i) {                                                                   \
      REP64(INS)                                                                                        \
    }                                                                                                   \
    double s = 0;                                                                                       \
    for (int j = 0; j < 8; ++j) s += a[j];                                                              \
    if (s == 0.12345) out[threadIdx.x] = 1;                                                             \
  }
#define KF32(NAME, INS)                                                                                 \
  __global__ void NAME(uint32_t* out, int iters) {                                                      \
    float a[8], b = 1.0f + 1e-6f * threadIdx.x, c = 1e-7f * threadIdx.x;                                \
    for (int j = 0; j < 8; ++j) a[j] = 1.0f + threadIdx.x + j;                                          \
    for (int i = 0; i < iters; ++i) {                                                                   \
      REP64(INS)                                                                                        \
    }                                                                                                   \
    float s = 0;                                                                                        \
    for (int j = 0; j < 8; ++j) s += a[j];                                                              \
    if (s == 0.12345f) out[threadIdx.x] = 1;                                                            \
  }

#define I_FMA32(j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
#define I_MUL32(j) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[j]) : "v"(b));
#define I_FMA64(j) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
#define I_MUL64(j) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[j]) : "v"(b));
#define I_ADD64(j) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[j]) : "v"(b));
#define I_RCP64(j) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[j]));
#define I_RSQ64(j) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[j]));
#define I_SQRT64(j) asm volatile("v_sqrt_f64 %0, %0" : "+v"(a[j]));
#define I_MAX64(j) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[j]) : "v"(b));
#define I_MULLO(j) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[j]) : "v"(b));
#define I_MULHI(j) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[j]) : "v"(b));
#define I_MUL24(j) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[j]) : "v"(b));
#define I_XOR(j) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[j]) : "v"(b));
#define I_ADDU(j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j]) : "v"(b));
#define I_LOG(j) asm volatile("v_log_f32 %0, %0" : "+v"(a[j]));
#define I_SIN(j) asm volatile("v_sin_f32 %0, %0" : "+v"(a[j]));
#define I_SQRT32(j) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[j]));
#define I_RCP32(j) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[j]));
#define I_PKFMA(j) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[j]) : "v"(b));
#define I_CVT3264(j) asm volatile("v_cvt_f32_f64 %0, %1" : "+v"(a[j]) : "v"(bd));

K32(k_xor, I_XOR) K32(k_addu, I_ADDU) K32(k_mullo, I_MULLO) K32(k_mulhi, I_MULHI) K32(k_mul24, I_MUL24)
KF32(k_fma32, I_FMA32) KF32(k_mul32, I_MUL32) KF32(k_log, I_LOG) KF32(k_sin, I_SIN) KF32(k_sqrt32, I_SQRT32) KF32(k_rcp32, I_RCP32)
K64(k_fma64, I_FMA64) K64(k_mul64, I_MUL64) K64(k_add64, I_ADD64) K64(k_rcp64, I_RCP64) K64(k_rsq64, I_RSQ64) K64(k_sqrt64, I_SQRT64)
K64(k_max64, I_MAX64) K64(k_pkfma, I_PKFMA)

// v_mad_u64_u32: the 32 x 32 -> 64 multiply (both halves of a Philox product in one instruction)
__global__ void k_mad64(uint32_t* out, int iters) {
  uint64_t a[8]; uint32_t b = threadIdx.x * 2654435761u + 12345u;
  for (int j = 0; j < 8; ++j) a[j] = threadIdx.x + j;
  for (int i = 0; i < iters; ++i) {
#define I_MAD64(j) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(a[j]) : "v"(b), "v"((uint32_t)(a[j])) : "vcc");
    REP64(I_MAD64)
  }
  uint64_t s = 0;
  for (int j = 0; j < 8; ++j) s ^= a[j];
  if (s == 0x12345u) out[threadIdx.x] = (uint32_t)s;
}
__global__ void k_cvt(uint32_t* out, int iters) {
  float a[8]; double bd = 1.0 + 1e-9 * threadIdx.x;
  for (int j = 0; j < 8; ++j) a[j] = 1.0f + threadIdx.x + j;
  for (int i = 0; i < iters; ++i) { REP64(I_CVT3264) asm volatile("" : "+v"(bd)); }
  float s = 0;
  for (int j = 0; j < 8; ++j) s += a[j];
  if (s == 0.12345f) out[threadIdx.x] = 1;
}
__global__ void k_readlane(uint32_t* out, int iters) {
  uint32_t a[8];
  for (int j = 0; j < 8; ++j) a[j] = threadIdx.x + j;
  for (int i = 0; i < iters; ++i) {
#define I_RL(j) { uint32_t sg; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sg) : "v"(a[j])); asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[j]) : "s"(sg)); }
    REP64(I_RL)
  }
  uint32_t s = 0;
  for (int j = 0; j < 8; ++j) s ^= a[j];
  if (s == 0x12345u) out[threadIdx.x] = s;
}

// the same streams with lanes 32..63 switched off (EXEC = 0x00000000ffffffff): does a half-empty wave issue in half the passes?
#define KHALF(NAME, TYPE, INIT, INS)                                                                    \
  __global__ void NAME(uint32_t* out, int iters) {                                                      \
    if (threadIdx.x & 32) return;                                                                       \
    TYPE a[8], b = INIT, c = INIT;                                                                      \
    for (int j = 0; j < 8; ++j) a[j] = (TYPE)(1 + threadIdx.x + j);                                     \
    for (int i = 0; i < iters; ++i) {                                                                   \
      REP64(INS)                                                                                        \
    }                                                                                                   \
    TYPE s = 0;                                                                                         \
    for (int j = 0; j < 8; ++j) s += a[j];                                                              \
    if (s == (TYPE)12345) out[threadIdx.x] = 1;                                                         \
  }
KHALF(h_fma32, float, 1.0f + 1e-6f * threadIdx.x, I_FMA32) KHALF(h_fma64, double, 1.0 + 1e-9 * threadIdx.x, I_FMA64)
KHALF(h_mulhi, uint32_t, threadIdx.x * 2654435761u + 12345u, I_MULHI) KHALF(h_log, float, 1.0f + 1e-6f * threadIdx.x, I_LOG)
KHALF(h_rcp64, double, 1.0 + 1e-9 * threadIdx.x, I_RCP64)

struct Case { const char* name; void (*fn)(uint32_t*, int); int per_trip; const char* note; };

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  uint32_t* out;
  CK(hipMalloc(&out, 4096));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 4000;
  std::vector<Case> cases = {
    {"v_fma_f32", k_fma32, 64, ""}, {"v_mul_f32", k_mul32, 64, ""}, {"v_pk_fma_f32", k_pkfma, 64, "two fp32 FMAs"},
    {"v_xor_b32", k_xor, 64, ""}, {"v_add_u32", k_addu, 64, ""},
    {"v_mul_u32_u24", k_mul24, 64, ""}, {"v_mul_lo_u32", k_mullo, 64, "Philox"}, {"v_mul_hi_u32", k_mulhi, 64, "Philox"},
    {"v_mad_u64_u32", k_mad64, 64, "32x32->64 in one"},
    {"v_fma_f64", k_fma64, 64, ""}, {"v_mul_f64", k_mul64, 64, ""}, {"v_add_f64", k_add64, 64, ""}, {"v_max_f64", k_max64, 64, ""},
    {"v_rcp_f64", k_rcp64, 64, ""}, {"v_rsq_f64", k_rsq64, 64, ""}, {"v_sqrt_f64", k_sqrt64, 64, ""}, {"v_cvt_f32_f64", k_cvt, 64, ""},
    {"v_log_f32", k_log, 64, "Box-Muller"}, {"v_sin_f32", k_sin, 64, "Box-Muller"}, {"v_sqrt_f32", k_sqrt32, 64, ""}, {"v_rcp_f32", k_rcp32, 64, ""},
    {"v_readlane+v_xor(sgpr)", k_readlane, 64, "pair; subtract one v_xor"},
    {"v_fma_f32, 32 lanes on", h_fma32, 64, "EXEC = low half"}, {"v_fma_f64, 32 lanes on", h_fma64, 64, "EXEC = low half"},
    {"v_mul_hi_u32, 32 lanes on", h_mulhi, 64, "EXEC = low half"}, {"v_log_f32, 32 lanes on", h_log, 64, "EXEC = low half"},
    {"v_rcp_f64, 32 lanes on", h_rcp64, 64, "EXEC = low half"},
  };
  printf("SIMD cycles per wave64 instruction (wall time x SIMDs / instructions issued, scaled so that v_fma_f32 at ONE wave per SIMD = 4 cycles,\n"
         "MI355X_MICROARCH.md's figure), with 1 / 2 / 3 waves per SIMD all running the same stream; %d CUs\n", cus);
  printf("%-26s %10s %10s %10s   %s\n", "instruction", "1 wave", "2 waves", "3 waves", "");
  double base = 0;
  for (auto& c : cases) {
    double t[3];
    for (int w = 1; w <= 3; ++w) {
      dim3 grid(cus * w), block(256);        // 4 waves per block = one per SIMD; w blocks per CU
      c.fn<<<grid, block>>>(out, 10);
      CK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0));
        c.fn<<<grid, block>>>(out, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      t[w - 1] = best * 1e6 / (double(iters) * c.per_trip);     // ns per instruction of ONE wave, with w waves on the SIMD
    }
    if (base == 0) base = t[0];
    printf("%-26s %10.2f %10.2f %10.2f   %s\n", c.name, 4 * t[0] / base, 4 * t[1] / base / 2, 4 * t[2] / base / 3, c.note);
  }
  printf("(ns per v_fma_f32 at one wave per SIMD: %.3f)\n", base);
  return 0;
}
